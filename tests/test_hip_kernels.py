"""Kernel-level parity: every HIP kernel against the CPU oracle on the same seeded inputs.

All calls go through the C ABI (include/kfusion_hip.h, kfh_k_*).  Bars: bit-exact for integer outputs
(TSDF shorts, result codes, colours) and for every float kernel whose arithmetic is +,-,*,/,sqrt only;
the bilateral filter restates glibc's expf and is exact as well; the ICP
reduction re-associates a float sum (relative 2e-5 of the magnitude sum, counters exact).
"""
import numpy as np
import pytest

from tests import util
from tests.util import Dev, f32, m16, pad4, same_bits, u32

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _dev(hip):
    return Dev()


@pytest.fixture()
def dev(_dev):
    _dev.clear()
    yield _dev
    _dev.clear()


@pytest.fixture(scope="module")
def scene(orc):
    """A short oracle run at 160x120 / 64^3 that leaves realistic maps, volume and poses behind."""
    s = util.small_stream(7)
    kf, rows = util.run_oracle(orc, s, 7)
    return {"stream": s, "kf": kf, "rows": rows}


def put_u16(dev, a):
    return dev.put(np.ascontiguousarray(a, np.uint16).view(np.int16))


# ------------------------------------------------------------------ preprocessing
@pytest.mark.parametrize("ratio", [1, 2, 4])
def test_mm2meters(hip, dev, orc, ratio):
    s = util.small_stream(1)
    d = s.depth_mm[0]
    H, W = d.shape[0] // ratio, d.shape[1] // ratio
    ref = orc.mm2meters(d, W, H)
    out = dev.empty((H, W), np.float32)
    assert hip.kfh_k_mm2meters(out.data_ptr(), W, H, put_u16(dev, d).data_ptr(), d.shape[1], d.shape[0], dev.stream) == 0
    assert same_bits(dev.get(out), ref)


def test_mm2meters_invalid_ratio(hip, dev):
    out = dev.empty((7, 7), np.float32)
    src = dev.empty((10, 10), np.int16)
    # 10 % 7 != 0 -> the reference prints "Invalid ratio." and exits (cpp:566-577)
    assert hip.kfh_k_mm2meters(out.data_ptr(), 7, 7, src.data_ptr(), 10, 10, dev.stream) == -1
    assert hip.kfh_k_mm2meters(out.data_ptr(), 20, 20, src.data_ptr(), 10, 10, dev.stream) == -1
    assert hip.kfh_k_mm2meters(out.data_ptr(), 5, 10, src.data_ptr(), 10, 10, dev.stream) == -1


BILATERAL_ULP = 0  # the kernel restates glibc's expf algorithm (expf_glibc): bit-identical to the CPU filter


@pytest.mark.parametrize("shape", [(120, 160), (37, 53), (8, 8), (1, 5)])
def test_bilateral_filter(hip, dev, orc, shape):
    rng = np.random.default_rng(5)
    depth = (rng.uniform(0.4, 4.0, size=shape)).astype(np.float32)
    depth[rng.uniform(size=shape) < 0.1] = 0.0          # holes: centre zero -> 0, zero taps skipped
    depth[:, : shape[1] // 3] += np.float32(0.5)         # a depth edge
    g = orc.gaussian()
    ref = orc.bilateral_filter(depth, g)
    out = dev.empty(shape, np.float32)
    hip.kfh_k_bilateral_filter(out.data_ptr(), dev.put(depth).data_ptr(), shape[1], shape[0], f32(g), 0.1, 2, dev.stream)
    got = dev.get(out)
    assert np.array_equal(got == 0, ref == 0)
    assert util.ulp_diff(got, ref).max() <= BILATERAL_ULP


def test_preprocess_fused(hip, dev, orc):
    s = util.small_stream(1, 320, 240)
    d = s.depth_mm[0]
    for ratio in (1, 2):
        H, W = d.shape[0] // ratio, d.shape[1] // ratio
        fd_ref = orc.mm2meters(d, W, H)
        bl_ref = orc.bilateral_filter(fd_ref)
        fd, bl = dev.empty((H, W), np.float32), dev.empty((H, W), np.float32)
        assert hip.kfh_k_preprocess(fd.data_ptr(), bl.data_ptr(), W, H, put_u16(dev, d).data_ptr(), d.shape[1], d.shape[0],
                                    f32(orc.gaussian()), 0.1, dev.stream) == 0
        assert same_bits(dev.get(fd), fd_ref)
        assert util.ulp_diff(dev.get(bl), bl_ref).max() <= BILATERAL_ULP


@pytest.mark.parametrize("shape", [(120, 160), (60, 80), (6, 10), (7, 9)])
def test_half_sample(hip, dev, orc, shape):
    rng = np.random.default_rng(9)
    depth = rng.uniform(0.4, 4.0, size=shape).astype(np.float32)
    depth[rng.uniform(size=shape) < 0.15] = 0.0
    ref = orc.half_sample(depth)
    out = dev.empty((shape[0] // 2, shape[1] // 2), np.float32)
    hip.kfh_k_half_sample(out.data_ptr(), dev.put(depth).data_ptr(), shape[1], shape[0], float(np.float32(0.1) * 3), 1, dev.stream)
    got = dev.get(out)
    assert same_bits(got, ref)   # includes 0/0 -> NaN where the centre is an isolated hole... (same bits)


# ------------------------------------------------------------------ vertex / normal maps
def _depth_with_holes(shape, seed):
    rng = np.random.default_rng(seed)
    d = rng.uniform(0.5, 3.5, size=shape).astype(np.float32)
    d[rng.uniform(size=shape) < 0.05] = 0
    return d


@pytest.mark.parametrize("shape,level", [((120, 160), 0), ((60, 80), 1), ((33, 47), 2)])
def test_depth2vertex_and_normal(hip, dev, orc, shape, level):
    depth = _depth_with_holes(shape, 11 + level)
    k = np.array(util.synth.camera_for(160), np.float32) / np.float32(1 << level)
    invK = orc.inverse_camera_matrix(k)
    v_ref = orc.depth2vertex(depth, invK)
    stale = np.random.default_rng(3).uniform(-1, 1, size=shape + (3,)).astype(np.float32)
    n_ref = orc.vertex2normal(v_ref, out=stale.copy())
    H, W = shape
    # separate kernels
    v = dev.empty((H, W, 4), np.float32)
    hip.kfh_k_depth2vertex(v.data_ptr(), dev.put(depth).data_ptr(), W, H, m16(invK), dev.stream)
    assert same_bits(dev.get(v)[..., :3], v_ref)
    n = dev.put(pad4(stale))
    hip.kfh_k_vertex2normal(n.data_ptr(), v.data_ptr(), W, H, dev.stream)
    assert same_bits(dev.get(n)[..., :3], n_ref)   # invalid pixels keep stale y/z
    # fused kernel used by the pipeline
    v2, n2 = dev.empty((H, W, 4), np.float32), dev.put(pad4(stale))
    hip.kfh_k_vertex_normal(v2.data_ptr(), n2.data_ptr(), dev.put(depth).data_ptr(), W, H, m16(invK), dev.stream)
    assert same_bits(dev.get(v2)[..., :3], v_ref)
    assert same_bits(dev.get(n2)[..., :3], n_ref)


# ------------------------------------------------------------------ ICP
def _track_inputs(scene, orc, level):
    kf, s = scene["kf"], scene["stream"]
    inV, inN = kf.buffer("inputVertex%d" % level), kf.buffer("inputNormal%d" % level)
    refV, refN = kf.buffer("vertex"), kf.buffer("normal")
    pose = scene["rows"][-1][0]                      # pose before the last frame: a real ICP starting point
    view = orc.matmul(orc.camera_matrix(s.k), orc.inverse(kf.buffer("raycastPose")))
    return inV, inN, refV, refN, pose, view


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("have_reference", [1, 0])
def test_track(hip, dev, orc, scene, level, have_reference):
    inV, inN, refV, refN, pose, view = _track_inputs(scene, orc, level)
    rh, rw = refV.shape[:2]
    ih, iw = inV.shape[:2]
    ref = orc.track(inV, inN, refV, refN, pose, view, have_reference=bool(have_reference))
    out = dev.put(np.zeros((rh, rw), orc.TRACK_DTYPE).view(np.uint8))
    hip.kfh_k_track(out.data_ptr(), dev.put(pad4(inV)).data_ptr(), dev.put(pad4(inN)).data_ptr(), iw, ih,
                    dev.put(pad4(refV)).data_ptr(), dev.put(pad4(refN)).data_ptr(), rw, rh, m16(pose), m16(view), 0.1, 0.8,
                    have_reference, dev.stream)
    got = dev.get(out).view(orc.TRACK_DTYPE).reshape(rh, rw)
    sub_g, sub_r = got[:ih, :iw], ref[:ih, :iw]
    assert np.array_equal(sub_g["result"], sub_r["result"])
    ok = sub_r["result"] == 1
    if have_reference:
        assert ok.sum() > 0.3 * ok.size      # the case is a real one: most pixels associate
    assert same_bits(sub_g["error"][ok], sub_r["error"][ok])
    assert same_bits(sub_g["J"][ok], sub_r["J"][ok])


def _reduce_close(got, ref, mag):
    """Counters exact; float sums within 2e-5 of the sum of magnitudes (re-associated fp32 sum)."""
    assert np.array_equal(got[28:32], ref[28:32])
    assert np.all(np.abs(got[:28].astype(np.float64) - ref[:28]) <= 2e-5 * mag[:28] + 1e-12)


def _reduce_magnitudes(td, w, h):
    sub = td[:h, :w]
    ok = sub["result"] == 1
    e = sub["error"][ok].astype(np.float64); J = sub["J"][ok].astype(np.float64)
    mag = np.zeros(32)
    mag[0] = np.sum(e * e)
    for i in range(6):
        mag[1 + i] = np.sum(np.abs(e * J[:, i]))
    k = 7
    for r in range(6):
        for c in range(r, 6):
            mag[k] = np.sum(np.abs(J[:, r] * J[:, c])); k += 1
    return mag


@pytest.mark.parametrize("level", [0, 2])
def test_reduce_and_fused_track_reduce(hip, dev, orc, scene, level):
    inV, inN, refV, refN, pose, view = _track_inputs(scene, orc, level)
    rh, rw = refV.shape[:2]
    ih, iw = inV.shape[:2]
    td = orc.track(inV, inN, refV, refN, pose, view)
    ref = orc.reduce(td, iw, ih)[0]
    mag = _reduce_magnitudes(td, iw, ih)
    # standalone reduce on the oracle's TrackData
    out = dev.empty((8, 32), np.float32)
    hip.kfh_k_reduce(out.data_ptr(), dev.put(td.view(np.uint8)).data_ptr(), rw, rh, iw, ih, dev.stream)
    _reduce_close(dev.get(out)[0], ref, mag)
    # fused kernel used by the pipeline
    out32, rmap = dev.empty((32,), np.float32), dev.empty((rh, rw), np.int32)
    hip.kfh_k_track_reduce(out32.data_ptr(), rmap.data_ptr(), dev.put(pad4(inV)).data_ptr(), dev.put(pad4(inN)).data_ptr(), iw,
                           ih, dev.put(pad4(refV)).data_ptr(), dev.put(pad4(refN)).data_ptr(), rw, rh, m16(pose), m16(view),
                           0.1, 0.8, 1, dev.stream)
    _reduce_close(dev.get(out32), ref, mag)
    assert np.array_equal(dev.get(rmap)[:ih, :iw], td["result"][:ih, :iw])
    # bitwise reproducible run to run (fixed reduction tree, no atomics)
    out32b = dev.empty((32,), np.float32)
    hip.kfh_k_track_reduce(out32b.data_ptr(), rmap.data_ptr(), dev.put(pad4(inV)).data_ptr(), dev.put(pad4(inN)).data_ptr(), iw,
                           ih, dev.put(pad4(refV)).data_ptr(), dev.put(pad4(refN)).data_ptr(), rw, rh, m16(pose), m16(view),
                           0.1, 0.8, 1, dev.stream)
    assert same_bits(dev.get(out32), dev.get(out32b))


def test_update_and_check_pose(hip, dev, orc, scene):
    inV, inN, refV, refN, pose, view = _track_inputs(scene, orc, 0)
    ih, iw = inV.shape[:2]
    red = orc.reduce(orc.track(inV, inN, refV, refN, pose, view), iw, ih)
    ref_pose, ref_conv = orc.update_pose(pose, red)
    d_pose = dev.put(pose)
    conv = hip.kfh_k_update_pose(d_pose.data_ptr(), dev.put(red[0]).data_ptr(), 1e-5, dev.stream)
    got = dev.get(d_pose)
    assert bool(conv) == ref_conv
    # double-precision solve + exp on both sides; only libm-vs-ocml sin/cos and the Jacobi sweep order can differ
    assert np.abs(got - ref_pose).max() <= 2e-7
    # degenerate system (frames 0-3: J^T J = 0): x = 0, pose unchanged, "converged" 
    zero = np.zeros(32, np.float32)
    d_pose = dev.put(pose)
    assert hip.kfh_k_update_pose(d_pose.data_ptr(), dev.put(zero).data_ptr(), 1e-5, dev.stream) == 1
    assert same_bits(dev.get(d_pose), pose)
    # checkPose: accept, reject on RMS, reject on inlier ratio, 0/0
    for e2, n, expect in [(1e-4 * 5000, 5000.0, 1), (1.0 * 5000, 5000.0, 0), (1e-4 * 100, 100.0, 0), (0.0, 0.0, 0)]:
        r = np.zeros(32, np.float32); r[0] = e2; r[28] = n
        old = np.eye(4, dtype=np.float32)
        ref_p, ref_ok = orc.check_pose(pose, old, r, iw, ih)
        d_pose = dev.put(pose)
        ok = hip.kfh_k_check_pose(d_pose.data_ptr(), dev.put(old).data_ptr(), dev.put(r).data_ptr(), iw, ih, 0.15, dev.stream)
        assert ok == expect == int(ref_ok)
        assert same_bits(dev.get(d_pose), ref_p)


# ------------------------------------------------------------------ volume kernels
def test_init_volume(hip, dev, orc):
    for res in [(16, 16, 16), (20, 12, 7)]:
        ref = orc.init_volume(res)
        v = dev.empty(ref.shape, np.int16)
        hip.kfh_k_init_volume(v.data_ptr(), u32(res), dev.stream)
        assert np.array_equal(dev.get(v), ref)


@pytest.mark.parametrize("res", [(64, 64, 64), (48, 40, 36), (96, 64, 32)])
def test_integrate(hip, dev, orc, scene, res):
    s = scene["stream"]
    K = orc.camera_matrix(s.k)
    vol_ref = orc.init_volume(res)
    d_vol = dev.put(vol_ref.copy())
    total = 0
    for f in range(4):                                  # several frames: running average + weights
        depth = orc.mm2meters(s.depth_mm[f], s.width, s.height)
        inv_track = orc.inverse(s.poses[f].astype(np.float32))
        total += orc.integrate(vol_ref, util.VOL_DIM, depth, inv_track, K)
        hip.kfh_k_integrate(d_vol.data_ptr(), u32(res), f32(util.VOL_DIM), dev.put(depth).data_ptr(), s.width, s.height,
                            m16(inv_track), m16(K), 0.1, 100.0, dev.stream)
        assert np.array_equal(dev.get(d_vol), vol_ref), "TSDF differs after frame %d" % f
    assert total > 0.02 * np.prod(res)                  # the case exercises real updates


@pytest.mark.parametrize("row2", [(0.0, 0.0, 1.0, 0.0), (0.01, -0.02, 0.97, 0.05), (0.0, 0.0, 1.0, 1e-3), (0.0, 0.0, 2.0, 0.0)])
def test_integrate_general_projection_matrix(hip, dev, orc, scene, row2):
    """integrateKernel takes K as a Matrix4 (cpp:628): nothing says its third row is (0, 0, 1, 0).  The kernel carries cam.z
    as a running sum of its own unless that row is exactly the pinhole one (then cam.z IS pos.z, bit for bit); both forms and
    a skewed first row against the CPU loop."""
    s = scene["stream"]
    K = orc.camera_matrix(s.k).copy()
    K[2] = row2
    K[0, 1] = 0.0 if row2 == (0.0, 0.0, 1.0, 0.0) else 2.5
    res = (64, 48, 72)
    vol_ref = orc.init_volume(res)
    d_vol = dev.put(vol_ref.copy())
    total = 0
    for f in range(3):
        depth = orc.mm2meters(s.depth_mm[f], s.width, s.height)
        inv_track = orc.inverse(s.poses[f].astype(np.float32))
        total += orc.integrate(vol_ref, util.VOL_DIM, depth, inv_track, K)
        hip.kfh_k_integrate(d_vol.data_ptr(), u32(res), f32(util.VOL_DIM), dev.put(depth).data_ptr(), s.width, s.height,
                            m16(inv_track), m16(K), 0.1, 100.0, dev.stream)
        assert np.array_equal(dev.get(d_vol), vol_ref), "TSDF differs after frame %d" % f
    assert total > 0.01 * np.prod(res)


def test_integrate_no_depth_is_identity(hip, dev, orc):
    res = (32, 32, 32)
    vol = orc.init_volume(res)
    d_vol = dev.put(vol.copy())
    depth = np.zeros((60, 80), np.float32)
    hip.kfh_k_integrate(d_vol.data_ptr(), u32(res), f32(util.VOL_DIM), dev.put(depth).data_ptr(), 80, 60,
                        m16(np.eye(4, dtype=np.float32)), m16(orc.camera_matrix((120, 120, 40, 30))), 0.1, 100.0, dev.stream)
    assert np.array_equal(dev.get(d_vol), vol)


def test_raycast(hip, dev, orc, scene):
    kf, s = scene["kf"], scene["stream"]
    vol = kf.buffer("volume")
    res = (vol.shape[2], vol.shape[1], vol.shape[0])
    W, H = s.width, s.height
    rng = np.random.default_rng(21)
    for f in (3, 6):
        view = orc.matmul(s.poses[f].astype(np.float32), orc.inverse_camera_matrix(s.k))
        stale = rng.uniform(-1, 1, size=(H, W, 3)).astype(np.float32)
        v_ref, n_ref = orc.raycast(vol, util.VOL_DIM, view, W, H, vertex=np.zeros((H, W, 3), np.float32), normal=stale.copy())
        dv, dn = dev.empty((H, W, 4), np.float32), dev.put(pad4(stale))
        step = float(np.float32(min(util.VOL_DIM)) / np.float32(max(res)))
        hip.kfh_k_raycast(dv.data_ptr(), dn.data_ptr(), W, H, dev.put(vol).data_ptr(), u32(res), f32(util.VOL_DIM), m16(view),
                          0.4, 4.0, step, 0.075, dev.stream)
        assert (v_ref[..., 2] != 0).mean() > 0.5       # most rays hit the model
        assert same_bits(dev.get(dv)[..., :3], v_ref)
        assert same_bits(dev.get(dn)[..., :3], n_ref)


@pytest.mark.parametrize("res,dim", [((64, 48, 80), (4.8, 3.6, 6.0)), ((40, 72, 56), (5.0, 4.8, 5.5)), ((96, 96, 32), (4.8, 4.8, 7.0))])
def test_integrate_and_raycast_anisotropic_volume(hip, dev, orc, scene, res, dim):
    """Non-cubic resolutions and per-axis extents (Volume carries a uint3 and a float3, commons.h:149-160): the voxel size
    differs per axis, the ray march divides by three different extents (each divisor of the reciprocal form is verified
    on the device on first use, otherwise the IEEE division is kept) and the step is min(extent) / max(resolution)."""
    s = scene["stream"]
    K = orc.camera_matrix(s.k)
    vol = orc.init_volume(res)
    d_vol = dev.put(vol.copy())
    W, H = s.width, s.height
    for f in range(3):
        depth = orc.mm2meters(s.depth_mm[f], W, H)
        inv_track = orc.inverse(s.poses[f].astype(np.float32))
        orc.integrate(vol, dim, depth, inv_track, K)
        hip.kfh_k_integrate(d_vol.data_ptr(), u32(res), f32(dim), dev.put(depth).data_ptr(), W, H, m16(inv_track), m16(K), 0.1, 100.0,
                            dev.stream)
    assert np.array_equal(dev.get(d_vol), vol)
    assert (vol[..., 1] > 0).mean() > 0.01
    view = orc.matmul(s.poses[2].astype(np.float32), orc.inverse_camera_matrix(s.k))
    step = float(np.float32(min(dim)) / np.float32(max(res)))
    v_ref, n_ref = orc.raycast(vol, dim, view, W, H, step=step)
    dv, dn = dev.empty((H, W, 4), np.float32), dev.empty((H, W, 4), np.float32)
    hip.kfh_k_raycast(dv.data_ptr(), dn.data_ptr(), W, H, d_vol.data_ptr(), u32(res), f32(dim), m16(view), 0.4, 4.0, step, 0.075,
                      dev.stream)
    assert (n_ref[..., 0] != -2).mean() > 0.3
    assert same_bits(dev.get(dv)[..., :3], v_ref)
    assert same_bits(dev.get(dn)[..., :3], n_ref)


def test_raycast_empty_volume_all_miss(hip, dev, orc):
    res = (32, 32, 32)
    vol = orc.init_volume(res)
    W, H = 80, 60
    view = orc.matmul(np.eye(4, dtype=np.float32), orc.inverse_camera_matrix((120, 120, 40, 30)))
    dv, dn = dev.empty((H, W, 4), np.float32), dev.empty((H, W, 4), np.float32)
    hip.kfh_k_raycast(dv.data_ptr(), dn.data_ptr(), W, H, dev.put(vol).data_ptr(), u32(res), f32(util.VOL_DIM), m16(view), 0.4,
                      4.0, 0.15, 0.075, dev.stream)
    assert np.all(dev.get(dv) == 0)
    n = dev.get(dn)
    assert np.all(n[..., 0] == -2) and np.all(n[..., 1:3] == 0)


# ------------------------------------------------------------------ renders
def test_render_depth(hip, dev, orc):
    rng = np.random.default_rng(2)
    d = rng.uniform(0.0, 5.0, size=(60, 80)).astype(np.float32)
    d[0, :6] = [0.0, 0.4, 4.0, 0.39999998, 4.0000005, 2.2]
    ref = orc.render_depth(d)
    out = dev.empty((60, 80, 4), np.uint8)
    hip.kfh_k_render_depth(out.data_ptr(), dev.put(d).data_ptr(), 80, 60, 0.4, 4.0, dev.stream)
    assert np.array_equal(dev.get(out), ref)


def test_render_track(hip, dev, orc):
    codes = np.array([1, -1, -2, -3, -4, -5, 0, 7, -9], np.int32)
    rmap = np.resize(codes, (6, 12)).astype(np.int32)
    td = np.zeros((6, 12), orc.TRACK_DTYPE); td["result"] = rmap
    ref = orc.render_track(td)
    out = dev.empty((6, 12, 4), np.uint8)
    hip.kfh_k_render_track(out.data_ptr(), dev.put(rmap).data_ptr(), 12, 6, dev.stream)
    assert np.array_equal(dev.get(out), ref)


def test_render_volume(hip, dev, orc, scene):
    kf, s = scene["kf"], scene["stream"]
    vol = kf.buffer("volume")
    res = (vol.shape[2], vol.shape[1], vol.shape[0])
    W, H = s.width, s.height
    view = orc.matmul(s.poses[5].astype(np.float32), orc.inverse_camera_matrix(s.k))
    step = float(np.float32(min(util.VOL_DIM)) / np.float32(max(res)))
    ref = orc.render_volume(vol, util.VOL_DIM, view, W, H, far=8.0, step=step)
    out = dev.empty((H, W, 4), np.uint8)
    hip.kfh_k_render_volume(out.data_ptr(), W, H, dev.put(vol).data_ptr(), u32(res), f32(util.VOL_DIM), m16(view), 0.4, 8.0, step,
                            0.075, f32((1, 1, -1)), f32((0.1, 0.1, 0.1)), dev.stream)
    assert (ref[..., :3].sum(axis=-1) > 0).mean() > 0.5
    assert np.array_equal(dev.get(out), ref)
