"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes as C

import numpy as np

from slambench_amd import synth

VOL_DIM = (4.8, 4.8, 4.8)
MU = 0.1
ICP_THRESHOLD = 1e-5


def pad4(a3):
    """(H, W, 3) float32 -> (H, W, 4) with w = 0 (backend-internal vertex/normal layout)."""
    out = np.zeros(a3.shape[:2] + (4,), np.float32)
    out[..., :3] = a3
    return out


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same_bits(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def ulp_diff(a, b):
    """Distance in units in the last place between two float32 arrays (same sign assumed)."""
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def small_stream(n=8, width=160, height=120, seed=1234):
    return synth.make_stream(n, width, height, seed)


def run_oracle(orc, stream, n_frames, vol_res=(64, 64, 64), pyramid=(10, 5, 4), ratio=1, integration_rate=1,
               tracking_rate=1, per_frame=None, omp=False):
    """Run the CPU twin over the first n_frames; returns (kfusion, rows) with rows[f] = (pose_before, tracked, integrated)."""
    W, H = stream.width // ratio, stream.height // ratio
    k = [v / ratio for v in stream.k]
    kf = orc.OracleKfusion((W, H), vol_res, VOL_DIM, stream.init_pos, pyramid, omp=omp)
    rows = []
    for f in range(n_frames):
        pose = kf.get_pose()
        kf.preprocessing(stream.depth_mm[f])
        tr = kf.tracking(k, ICP_THRESHOLD, tracking_rate, f)
        it = kf.integration(k, integration_rate, MU, f)
        kf.raycasting(k, MU, f)
        rows.append((pose, tr, it))
        if per_frame:
            per_frame(f, kf)
    return kf, rows


class Dev:
    """Device buffers for kernel-level calls, backed by torch (plumbing only)."""

    def __init__(self):
        import torch
        self.torch = torch
        self.stream = torch.cuda.current_stream().cuda_stream
        self._keep = []   # a tensor whose only use is `.data_ptr()` in a call expression would be freed (and its
                          # block handed to the next allocation) before the kernel runs: hold every buffer until clear()

    def clear(self):
        self.torch.cuda.synchronize()
        self._keep = []

    def put(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a)).cuda()
        self._keep.append(t)
        return t

    def empty(self, shape, dtype):
        t = {np.float32: self.torch.float32, np.int32: self.torch.int32, np.int16: self.torch.int16,
             np.uint8: self.torch.uint8}[dtype]
        out = self.torch.zeros(shape, dtype=t, device="cuda")
        self._keep.append(out)
        return out

    def get(self, t):
        self.torch.cuda.synchronize()
        return t.cpu().numpy()


def f32(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


def u32(v):
    return (C.c_uint32 * len(v))(*[int(x) for x in v])


def m16(m):
    return f32(np.asarray(m, np.float32).reshape(-1))
