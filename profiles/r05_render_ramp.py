"""How the rendered queued loop (bench.py's fps_total) comes up to speed in the first process on an idle box: frames/s per chunk of
100 frames, for ~3 s.  The first such leg of a box measured 2300-2900 frames/s, later processes 4000-4500."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from slambench_amd import kfusion, synth, _capi
W, H, N = 640, 480, 100
s = synth.make_stream(N, W, H)
k = list(s.k)
dev = torch.from_numpy(s.depth_mm.astype(np.int16)).cuda()
u8 = C.POINTER(C.c_uint8)
img = [np.zeros((H, W, 4), np.uint8) for _ in range(3)]
for renders in (True, False, True):
    kf = kfusion.Kfusion((W, H), (256,) * 3, (4.8,) * 3, s.init_pos)
    L, h = kf.L, kf.h
    rates = []
    t_begin = time.perf_counter()
    for chunk in range(60 if renders else 10):
        kf.synchronise(); t0 = time.perf_counter()
        for f in range(N):
            kf.compute_frame(dev[f].data_ptr(), (W, H), k, 2, 1, 1e-5, 0.1, chunk * N + f)
            if renders:
                L.kfh_render_depth(h, img[0].ctypes.data_as(u8), W, H)
                L.kfh_render_track(h, img[1].ctypes.data_as(u8), W, H)
                L.kfh_render_volume(h, img[2].ctypes.data_as(u8), W, H, f, 4, _capi.f32(k), 0.75 * 0.1)
        kf.synchronise()
        rates.append(N / (time.perf_counter() - t0))
    print("renders %d, %.2f s: frames/s per 100 frames: %s" % (renders, time.perf_counter() - t_begin, " ".join("%.0f" % r for r in rates)))
    kf.close()
