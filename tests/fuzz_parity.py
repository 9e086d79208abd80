"""Randomised parity of the two volume kernels against the oracle: random (non-cubic) resolutions, random per-axis volume
extents, three image sizes, camera poses perturbed by up to 5 degrees / 5 cm, three integrations followed by a raycast -
all through the kernel-level C ABI, compared bit for bit.

    python -m tests.fuzz_parity [seed [trials]]        (on a GPU box; 60 trials take ~5 s)

tests/test_hip_fuzz.py runs a dozen trials of it in the GPU suite."""
import sys

import numpy as np

from tests import util
from tests.util import Dev, f32, m16, same_bits, u32


def run(hip, orc, seed=7, trials=12, verbose=False):
    rng = np.random.default_rng(seed)
    dev = Dev()
    bad = []
    for trial in range(trials):
        W, H = [(160, 120), (200, 152), (96, 80)][trial % 3]
        s = util.synth.make_stream(3, W, H, seed=int(rng.integers(1, 10 ** 6)), variant=int(rng.integers(0, 4)))
        res = tuple(int(v) for v in rng.choice([40, 64, 72, 96, 128], 3))
        dim = tuple(float(v) for v in rng.uniform(4.0, 6.0, 3)) if trial % 2 else (4.8, 4.8, 4.8)
        K = orc.camera_matrix(s.k)
        vol = orc.init_volume(res, omp=True)
        d_vol = dev.put(vol.copy())
        pose = None
        for f in range(3):
            depth = orc.mm2meters(s.depth_mm[f], W, H, omp=True)
            pose = s.poses[f].astype(np.float32).copy()
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            ang = rng.uniform(-0.09, 0.09)
            Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
            pose[:3, :3] = (R @ pose[:3, :3]).astype(np.float32)
            pose[:3, 3] += rng.uniform(-0.05, 0.05, 3).astype(np.float32)
            inv_track = orc.inverse(pose)
            orc.integrate(vol, dim, depth, inv_track, K, omp=True)
            hip.kfh_k_integrate(d_vol.data_ptr(), u32(res), f32(dim), dev.put(depth).data_ptr(), W, H, m16(inv_track), m16(K), 0.1,
                                100.0, dev.stream)
        ok_i = np.array_equal(dev.get(d_vol), vol)
        view = orc.matmul(pose, orc.inverse_camera_matrix(s.k))
        step = float(np.float32(min(dim)) / np.float32(max(res)))
        v_ref, n_ref = orc.raycast(vol, dim, view, W, H, step=step, omp=True)
        dv, dn = dev.empty((H, W, 4), np.float32), dev.empty((H, W, 4), np.float32)
        hip.kfh_k_raycast(dv.data_ptr(), dn.data_ptr(), W, H, d_vol.data_ptr(), u32(res), f32(dim), m16(view), 0.4, 4.0, step, 0.075,
                          dev.stream)
        ok_r = same_bits(dev.get(dv)[..., :3], v_ref) and same_bits(dev.get(dn)[..., :3], n_ref)
        updated, hits = float((vol[..., 1] > 0).mean()), float((n_ref[..., 0] != -2).mean())
        if verbose:
            print("trial %3d  %dx%d  res %s  dim %s  updated %.3f  hits %.2f  integrate %s  raycast %s" % (
                trial, W, H, res, tuple(round(d, 3) for d in dim), updated, hits, ok_i, ok_r))
        if not (ok_i and ok_r):
            bad.append((trial, res, dim, ok_i, ok_r))
        dev.clear()
    return bad


if __name__ == "__main__":
    import torch  # noqa: F401  (before the backend: one HIP runtime per process, see slambench_amd/_capi.py)
    from oracle import orc as _orc
    from slambench_amd import _capi
    _orc.build()
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    failures = run(_capi.lib(), _orc, seed, trials, verbose=True)
    print("trials %d, mismatches %d" % (trials, len(failures)))
    sys.exit(1 if failures else 0)
