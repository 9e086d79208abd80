"""Kernel microbenchmark on a realistic state: runs the pipeline for 40 frames, then launches the integrate and raycast
kernels repeatedly through the kernel-level C ABI on that state (volume, depth, pose).  Meant to be run under
`rocprofv3 --kernel-trace --stats` (the kernel-level entry points synchronise, so host timing is meaningless)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from slambench_amd import kfusion, synth, _capi
res = int(os.environ.get("KB_RES", "256")); reps = int(os.environ.get("KB_REPS", "20"))
s = synth.cached_stream(48)
k = list(s.k)
kf = kfusion.Kfusion((640, 480), (res,) * 3, (4.8,) * 3, s.init_pos)
d = torch.from_numpy(s.depth_mm.view(np.int16)).cuda()
for f in range(40):
    kf.compute_frame(d.data_ptr() + f * 640 * 480 * 2, (640, 480), k, 1, 1, 1e-5, 0.1, f)
kf.synchronise()
pose = kf.get_pose()
L = kf.L
vol = torch.from_numpy(kf.buffer("volume").copy()).cuda()
depth = torch.from_numpy(kf.buffer("floatDepth").copy()).cuda()
inv = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
K = np.array([[k[0], 0, k[2], 0], [0, k[1], k[3], 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
view = (pose.astype(np.float64) @ np.linalg.inv(K.astype(np.float64))).astype(np.float32)
f32 = lambda a: (C.c_float * a.size)(*a.reshape(-1))
size = (C.c_uint32 * 3)(res, res, res); dim = (C.c_float * 3)(4.8, 4.8, 4.8)
vtx = torch.zeros((480, 640, 4), device="cuda"); nrm = torch.zeros((480, 640, 4), device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    L.kfh_k_integrate(vol.data_ptr(), size, dim, depth.data_ptr(), 640, 480, f32(inv), f32(K), 0.1, 100.0, st)
for _ in range(reps):
    L.kfh_k_raycast(vtx.data_ptr(), nrm.data_ptr(), 640, 480, vol.data_ptr(), size, dim, f32(view), 0.4, 4.0, 4.8 / res, 0.075, st)
torch.cuda.synchronize()
print("done", res)
