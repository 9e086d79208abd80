"""CPU-only checks of the oracle: pose algebra against numpy/scipy, per-kernel properties, the whole pipeline on
the synthetic stream, and serial == OpenMP bit for bit.  (The oracle is test infrastructure; these tests keep the
checker honest — the HIP path is compared with it in the -m gpu tests.)"""
import numpy as np
import pytest
import scipy.linalg

from tests import util


# ------------------------------------------------------------------ pose algebra (commons.h:365-412 / TooN)
def test_inverse_and_matmul(orc):
    rng = np.random.default_rng(0)
    for _ in range(20):
        R = scipy.linalg.expm(np.cross(np.eye(3), rng.normal(size=3) * 0.7))
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.normal(size=3) * 2
        T = T.astype(np.float32)
        inv = orc.inverse(T)
        assert np.abs(inv.astype(np.float64) - np.linalg.inv(T.astype(np.float64))).max() < 2e-6
        assert np.abs(orc.matmul(T, inv) - np.eye(4)).max() < 2e-6
        B = rng.normal(size=(4, 4)).astype(np.float32)
        assert np.abs(orc.matmul(T, B) - T.astype(np.float64) @ B.astype(np.float64)).max() < 1e-5
    # singular input: no check in the reference (TooN gaussian_elimination) -> inf/NaN, not an exception
    assert not np.isfinite(orc.inverse(np.zeros((4, 4), np.float32))).all()
    # exact cases
    assert np.array_equal(orc.inverse(np.eye(4, dtype=np.float32)), np.eye(4, dtype=np.float32))


def test_se3_exp(orc):
    rng = np.random.default_rng(1)
    for scale in (1e-6, 5e-4, 2e-3, 0.3, 2.0):      # crosses the 1e-8 and 1e-6 theta^2 branches of TooN's exp
        mu = rng.normal(size=6) * scale
        v, w = mu[:3], mu[3:]
        twist = np.zeros((4, 4)); twist[:3, :3] = np.cross(np.eye(3), w) * -1; twist[:3, 3] = v
        twist[:3, :3] = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        ref = scipy.linalg.expm(twist)
        got = orc.se3_exp(mu)
        assert np.abs(got - ref).max() < 3e-7 * max(1.0, np.abs(ref).max())
    assert np.array_equal(orc.se3_exp(np.zeros(6)), np.eye(4, dtype=np.float32))


def _vals(C, b):
    v = np.zeros(27, np.float32)
    v[:6] = b
    v[6:] = C[np.triu_indices(6)]
    return v


def test_solve6_is_the_truncated_pseudo_inverse(orc):
    rng = np.random.default_rng(2)
    for _ in range(20):
        J = rng.normal(size=(200, 6)) * np.array([1, 1, 1, 3, 3, 3])
        C = (J.T @ J).astype(np.float32); b = (J.T @ rng.normal(size=200) * 0.01).astype(np.float32)
        x = orc.solve6(_vals(C, b))
        ref = np.linalg.solve(C.astype(np.float64), b.astype(np.float64))
        assert np.abs(x - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # rank-deficient: singular values below max/1e6 are dropped (GR_SVD::backsub(b, 1e6), commons.h:403)
    J = rng.normal(size=(50, 6)); J[:, 5] = J[:, 4]
    C = (J.T @ J).astype(np.float32); b = (J.T @ rng.normal(size=50)).astype(np.float32)
    x = orc.solve6(_vals(C, b))
    ref = np.linalg.pinv(C.astype(np.float64), rcond=1e-6, hermitian=True) @ b.astype(np.float64)
    assert np.abs(x - ref).max() < 1e-6 * max(1.0, np.abs(ref).max())
    assert np.array_equal(orc.solve6(np.zeros(27, np.float32)), np.zeros(6))      # frames 0-3: x = 0


def test_update_and_check_pose(orc):
    pose = np.eye(4, dtype=np.float32); pose[:3, 3] = (1, 2, 3)
    red = np.zeros((8, 32), np.float32)
    new, conv = orc.update_pose(pose, red)
    assert conv and np.array_equal(new, pose)
    old = np.eye(4, dtype=np.float32)
    for e2, n, ok in [(1e-4 * 5000, 5000.0, True), (1.0 * 5000, 5000.0, False), (1e-4 * 100, 100.0, False), (0.0, 0.0, False)]:
        r = np.zeros(32, np.float32); r[0] = e2; r[28] = n
        p, good = orc.check_pose(pose, old, r, 160, 120)
        assert good == ok and np.array_equal(p, pose if ok else old)


# ------------------------------------------------------------------ kernels
def test_preprocessing_kernels(orc):
    s = util.small_stream(1)
    d = s.depth_mm[0]
    for ratio in (1, 2, 4):
        out = orc.mm2meters(d, d.shape[1] // ratio, d.shape[0] // ratio)
        assert np.array_equal(out, (d[::ratio, ::ratio].astype(np.float32) / np.float32(1000.0)))
    with pytest.raises(ValueError):
        orc.mm2meters(d, 150, 120)
    const = np.full((20, 30), 1.5, np.float32)
    assert np.allclose(orc.bilateral_filter(const), 1.5, atol=1e-6)
    holes = const.copy(); holes[5, 5] = 0; holes[0, 0] = 0
    f = orc.bilateral_filter(holes)
    assert f[5, 5] == 0 and f[0, 0] == 0 and np.allclose(f[holes > 0], 1.5, atol=1e-6)
    hs = orc.half_sample(const)
    assert hs.shape == (10, 15) and np.array_equal(hs, np.full((10, 15), 1.5, np.float32))
    edge = const.copy(); edge[:, 15:] = 3.0          # robust mean ignores samples more than 0.3 m from the centre
    assert set(np.unique(orc.half_sample(edge))) == {np.float32(1.5), np.float32(3.0)}


def test_vertex_normal_of_a_plane(orc):
    k = util.synth.camera_for(160)
    invK = orc.inverse_camera_matrix(k)
    depth = np.full((120, 160), 2.0, np.float32)
    v = orc.depth2vertex(depth, invK)
    u, w = np.meshgrid(np.arange(160, dtype=np.float32), np.arange(120, dtype=np.float32))
    assert np.allclose(v[..., 0], 2.0 * (u - k[2]) / k[0], atol=1e-5) and np.all(v[..., 2] == 2.0)
    n = orc.vertex2normal(v)
    assert np.allclose(n, (0, 0, -1), atol=1e-6)      # fronto-parallel plane, normal towards the camera
    depth[60, 80] = 0
    n = orc.vertex2normal(orc.depth2vertex(depth, invK), out=np.full((120, 160, 3), 7.0, np.float32))
    assert n[60, 81, 0] == -2 and n[60, 81, 1] == 7.0 and n[59, 80, 0] == -2      # only .x is written


def test_track_and_reduce_identity(orc):
    k = util.synth.camera_for(160)
    s = util.small_stream(1)
    depth = orc.mm2meters(s.depth_mm[0], 160, 120)
    v = orc.depth2vertex(depth, orc.inverse_camera_matrix(k))
    n = orc.vertex2normal(v)
    pose = np.eye(4, dtype=np.float32)
    view = orc.camera_matrix(k)                          # K * inverse(identity)
    td = orc.track(v, n, v, n, pose, view)
    ok = td["result"] == 1
    assert ok.mean() > 0.9 and np.all(td["error"][ok] == 0)       # every pixel lands on itself
    assert set(np.unique(td["result"])) <= {1, -1, -2, -3, -4, -5}
    red = orc.reduce(td, 160, 120)
    assert red[0, 28] == ok.sum() and red[0, 0] == 0
    J = td["J"][ok].astype(np.float64)
    assert np.allclose(red[0, 7], np.sum(J[:, 0] ** 2), rtol=1e-4)
    # result codes on a coarser level keep the full-resolution row stride (cpp:512)
    td2 = orc.track(v[::2, ::2].copy(), n[::2, ::2].copy(), v, n, pose, view)
    assert td2.shape == (120, 160) and np.all(td2["result"][60:, :] == 0) and np.all(td2["result"][:60, 80:] == 0)
    # no reference yet (frames 0-3): nothing associates, nothing crashes
    td3 = orc.track(v, n, np.zeros_like(v), np.zeros_like(n), pose, np.zeros((4, 4), np.float32), have_reference=False)
    assert not np.any(td3["result"] == 1)


def test_integrate_raycast_round_trip(orc):
    """Fuse one synthetic view, ray-cast it back from the same pose: the surface comes back within a voxel."""
    s = util.small_stream(1)
    res = (96, 96, 96)
    vol = orc.init_volume(res)
    assert np.all(vol[..., 0] == 32766) and np.all(vol[..., 1] == 0)
    depth = orc.mm2meters(s.depth_mm[0], 160, 120)
    pose = s.poses[0].astype(np.float32)
    K = orc.camera_matrix(s.k)
    updated = 0
    for _ in range(3):
        updated += orc.integrate(vol, util.VOL_DIM, depth, orc.inverse(pose), K)
    assert updated > 0 and np.all(vol[..., 1] <= 3) and (vol[..., 1] == 3).sum() * 3 == updated
    v, n, distinct = orc.raycast(vol, util.VOL_DIM, orc.matmul(pose, orc.inverse_camera_matrix(s.k)), 160, 120, count=True)
    hit = n[..., 0] != -2
    assert hit.mean() > 0.8 and distinct > hit.sum()
    z_cam = (v[..., 2] - pose[2, 3])[hit]               # frame 0 has the identity rotation
    assert np.abs(z_cam - depth[hit]).mean() < 0.5 * 4.8 / 96
    assert np.allclose(np.linalg.norm(n[hit], axis=-1), 1.0, atol=1e-5)
    img = orc.render_volume(vol, util.VOL_DIM, orc.matmul(pose, orc.inverse_camera_matrix(s.k)), 160, 120)
    assert img.shape == (120, 160, 4) and (img[..., :3].sum(-1) > 0).mean() > 0.5 and np.all(img[..., 3] == 0)


def test_renders(orc):
    d = np.array([[0.0, 0.39, 0.4, 2.2, 4.0, 4.5]], np.float32)
    img = orc.render_depth(d)
    assert list(img[0, 0]) == [255, 255, 255, 0] and list(img[0, 1]) == [255, 255, 255, 0] and list(img[0, 5]) == [0, 0, 0, 0]
    assert list(img[0, 4]) == [0, 0, 0, 0]               # d == farPlane -> sextant 6 -> black (commons.h:135-139)
    assert list(img[0, 2]) == [191, 63, 63, 0]           # d == nearPlane -> h = 0 -> (v, m, m) * 255
    td = np.zeros((1, 8), orc.TRACK_DTYPE); td["result"] = [[1, -1, -2, -3, -4, -5, 0, 9]]
    img = orc.render_track(td)
    assert [list(p[:3]) for p in img[0]] == [[128, 128, 128], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0],
                                             [255, 128, 128], [255, 128, 128]]


# ------------------------------------------------------------------ pipeline
def test_pipeline_tracks_the_synthetic_stream(orc):
    s = util.small_stream(14)
    kf, rows = util.run_oracle(orc, s, 14, vol_res=(96, 96, 96), integration_rate=2)
    tracked = [r[1] for r in rows]; integrated = [r[2] for r in rows]
    assert tracked[:4] == [False] * 4 and all(tracked[4:])
    assert integrated[:4] == [True] * 4                                   # cpp:994 forces frames 0-3
    assert integrated[4:] == [f % 2 == 0 for f in range(4, 14)]          # -r 2
    est = kf.get_pose()[:3, 3] - s.init_pos
    gt = s.poses[13][:3, 3] - s.init_pos
    assert np.linalg.norm(est - gt) < 0.02
    assert kf.counters()["icp_iterations"] >= 3 * 4 + 10                  # one pass per level on frames 0-3, real ICP after
    kf.close()


def test_gating_of_tracking_rate(orc):
    s = util.small_stream(8)
    kf, rows = util.run_oracle(orc, s, 8, vol_res=(48, 48, 48), tracking_rate=2, integration_rate=1)
    assert [r[1] for r in rows][5] is False and [r[1] for r in rows][7] is False      # odd frames are not tracked (cpp:927)
    kf.close()


def test_openmp_twin_is_bit_identical(orc):
    """liborc_omp.so parallelises rows / reduction blocks only, so it keeps the serial order (unlike the reference's
    own OpenMP reduction clause): same pose and same TSDF bits."""
    s = util.small_stream(7)
    a, _ = util.run_oracle(orc, s, 7, vol_res=(48, 48, 48))
    b, _ = util.run_oracle(orc, s, 7, vol_res=(48, 48, 48), omp=True)
    assert np.array_equal(a.get_pose().view(np.uint32), b.get_pose().view(np.uint32))
    assert np.array_equal(a.buffer("volume"), b.buffer("volume"))
    assert np.array_equal(a.buffer("vertex").view(np.uint32), b.buffer("vertex").view(np.uint32))
    a.close(); b.close()


def test_dump_volume_layout(orc, tmp_path):
    s = util.small_stream(2)
    kf, _ = util.run_oracle(orc, s, 2, vol_res=(16, 16, 16))
    path = str(tmp_path / "v.bin")
    assert kf.dump_volume(path) == 0
    assert np.array_equal(np.fromfile(path, np.int16), kf.buffer("volume")[..., 0].reshape(-1))   # cpp:1022-1026
    kf.close()


def test_fixed_schedule_emulation_of_the_references_openmp_reduction(orc):
    """oracle mode "reduce emulation": the reference's `#pragma omp parallel for reduction(+:sums0..31)` (cpp:295-296) with T threads
    under gcc's static schedule and a FIXED combine order, computed serially.  Pins: one thread is the serial chain bit for bit;
    two threads equal the real libgomp run bit for bit (with two shares the combine order cannot matter: fp addition commutes);
    more threads re-associate further, deterministically."""
    s = util.small_stream(8)

    def run(emulated, real_mode=False, threads=None, omp=True):
        orc.set_reduce_emulation(emulated, omp=omp)
        if omp:
            orc.set_omp_reduce_mode(real_mode)
            if threads:
                orc.set_omp_threads(threads)
        try:
            kf, _ = util.run_oracle(orc, s, 8, vol_res=(48, 48, 48), omp=omp)
            out = (kf.get_pose().copy(), kf.buffer("reduction").copy())
            kf.close()
            return out
        finally:
            orc.set_reduce_emulation(0, omp=omp)
            if omp:
                orc.set_omp_reduce_mode(False)

    serial = run(0, omp=False)
    for omp in (False, True):
        one = run(1, omp=omp)
        assert np.array_equal(one[0].view(np.uint32), serial[0].view(np.uint32)) and np.array_equal(one[1].view(np.uint32), serial[1].view(np.uint32))
    two, real_two = run(2), run(0, real_mode=True, threads=2)
    assert np.array_equal(two[0].view(np.uint32), real_two[0].view(np.uint32))
    assert np.array_equal(two[1].view(np.uint32), real_two[1].view(np.uint32))
    assert not np.array_equal(two[1].view(np.uint32), serial[1].view(np.uint32))          # ... and it IS a different association
    sixteen, again = run(16), run(16, omp=False)
    assert np.array_equal(sixteen[1].view(np.uint32), again[1].view(np.uint32))           # no dependence on the build or the host's threads
    assert np.abs(sixteen[0] - serial[0]).max() < 1e-5 and np.array_equal(sixteen[1][0, 28:], serial[1][0, 28:])   # counters are exact


def test_squared_length_bound_of_the_icp_distance_test():
    """The HIP association decides `length(diff) > dist_threshold` (cpp:543) on the squared length: sqrtf(d) > t  <=>  d > X with X the
    largest float whose correctly rounded root is <= t (kf_math.h sqrt_gt_bound; on the device k_verify_sqrt_bound holds the device's
    square root to it for all 2^31 non-negative floats).  This is the host-side construction restated in numpy float32: the bound
    exists, sits where the rounded root crosses t, and agrees with the square root on a dense neighbourhood and on the special values."""
    f32 = np.float32

    def bound(t):
        t = f32(t)
        x = f32(t * t)
        for _ in range(8):
            if np.sqrt(x) > t:
                x = np.nextafter(x, f32(-1))
        for _ in range(8):
            if np.sqrt(np.nextafter(x, f32(np.inf))) <= t:
                x = np.nextafter(x, f32(np.inf))
        assert np.sqrt(x) <= t < np.sqrt(np.nextafter(x, f32(np.inf)))
        return x

    rng = np.random.default_rng(7)
    for t in [0.1, 0.0, 1e-20, 0.5, 1.0, 3.0000002, 1234.5678, 1e18] + list(rng.uniform(1e-3, 10.0, 200)):
        x = bound(t)
        t32 = f32(t)
        # every float within 64 ulps of the bound, both sides
        d = x.copy()
        near = [x]
        for _ in range(64):
            d = np.nextafter(d, f32(np.inf)); near.append(d)
        d = x.copy()
        for _ in range(64):
            if d > 0:
                d = np.nextafter(d, f32(-1)); near.append(d)
        near = np.array(near + [0.0, np.inf, np.nan, 1e-45, 3.4e38], dtype=f32)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(np.sqrt(near) > t32, near > x), t
        wide = rng.uniform(0, 4 * float(x) + 1e-30, 2000).astype(f32)
        assert np.array_equal(np.sqrt(wide) > t32, wide > x), t


def test_row_of_a_pixel_by_a_reciprocal_multiplication():
    """The HIP ICP association needs the row of pixel p on a coarse level (the result map keeps the full-resolution row stride,
    cpp:512) and computes it as (unsigned)(((float)p + 0.5f) * RN(1 / inW)) while the level has at most 2^22 pixels
    (kf_icp.hip make_icp_args / icp_track_block), instead of an integer division by a run-time value.  The same float32
    operations in numpy, for EVERY pixel below 2^22 of widths of every kind (powers of two, odd, prime, the pyramid levels of the
    benchmark sizes): equal to p // inW."""
    f32 = np.float32
    p = np.arange(1 << 22, dtype=np.uint32)
    pf = p.astype(f32) + f32(0.5)
    for w in (1, 2, 3, 7, 36, 45, 50, 75, 80, 90, 100, 113, 125, 150, 160, 161, 250, 320, 333, 500, 640, 641, 1000, 1023, 1024, 1280, 1920, 2047, 4093):
        inv = f32(1.0) / f32(w)
        n = min(1 << 22, w * 4096)
        rows = (pf[:n] * inv).astype(np.uint32)
        assert np.array_equal(rows, p[:n] // np.uint32(w)), w
