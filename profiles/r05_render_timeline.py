"""Queued frames with the three renders and no synchronisation in between (bench.py's fps_total loop), to be run under
rocprofv3 --kernel-trace: where does the read-back's copy kernel run?  profiles/r05_render_timeline.sh"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from slambench_amd import kfusion, synth, _capi
W, H, N = 640, 480, int(os.environ.get('FRAMES', '120'))
s = synth.make_stream(N, W, H)
k = list(s.k)
kf = kfusion.Kfusion((W, H), (256,) * 3, (4.8,) * 3, s.init_pos)
L, h = kf.L, kf.h
u8 = C.POINTER(C.c_uint8)
img = [np.zeros((H, W, 4), np.uint8) for _ in range(3)]
dev = torch.from_numpy(s.depth_mm.astype(np.int16)).cuda()      # the depth frames resident in HBM, as in bench.py's loop
t0 = None
for f in range(N):
    if f == 20:
        kf.synchronise(); t0 = time.perf_counter()
    kf.compute_frame(dev[f].data_ptr(), (W, H), k, 2, 1, 1e-5, 0.1, f)
    L.kfh_render_depth(h, img[0].ctypes.data_as(u8), W, H)
    L.kfh_render_track(h, img[1].ctypes.data_as(u8), W, H)
    L.kfh_render_volume(h, img[2].ctypes.data_as(u8), W, H, f, 1, _capi.f32(k), 0.75 * 0.1)
t_host = time.perf_counter() - t0
kf.synchronise()
t_all = time.perf_counter() - t0
print("frames/s with renders: %.1f  (%.1f us per frame; the host had queued them after %.1f us per frame)" % ((N - 20) / t_all, 1e6 * t_all / (N - 20), 1e6 * t_host / (N - 20)))
kf.close()
