"""Host stalls while queueing frames: which calls of a queued loop block the calling thread for more than 0.5 ms, and when
(frame index, seconds since the process began)?  Legs with and without renders, contexts created and destroyed as bench.py does."""
import ctypes as C, gc, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from slambench_amd import kfusion, synth, _capi
W, H, N = 640, 480, 420
s = synth.make_stream(N, W, H)
k = list(s.k)
dev = torch.from_numpy(s.depth_mm.astype(np.int16)).cuda()
u8 = C.POINTER(C.c_uint8)
t_proc = time.perf_counter()
if os.environ.get('NO_GC') == '1':
    gc.collect(); gc.disable()
gc.callbacks.append(lambda phase, info: print('    [gc %s generation %d at t=%.3f s]' % (phase, info['generation'], time.perf_counter() - t_proc)) if info['generation'] == 2 else None)
frames_total = 0
def leg(renders, n):
    global frames_total
    img = [np.zeros((H, W, 4), np.uint8) for _ in range(3)]
    kf = kfusion.Kfusion((W, H), (256,) * 3, (4.8,) * 3, s.init_pos)
    L, h = kf.L, kf.h
    stalls = []
    t_all = time.perf_counter()
    for f in range(n):
        t0 = time.perf_counter()
        kf.compute_frame(dev[f].data_ptr(), (W, H), k, 2, 1, 1e-5, 0.1, f)
        t1 = time.perf_counter()
        if renders:
            L.kfh_render_depth(h, img[0].ctypes.data_as(u8), W, H)
            L.kfh_render_track(h, img[1].ctypes.data_as(u8), W, H)
            L.kfh_render_volume(h, img[2].ctypes.data_as(u8), W, H, f, 4, _capi.f32(k), 0.75 * 0.1)
        t2 = time.perf_counter()
        if t2 - t0 > 0.5e-3:
            stalls.append("frame %d (%d of the process, t=%.3f s): compute_frame %.2f ms, renders %.2f ms" % (f, frames_total + f, t0 - t_proc, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    t_host = time.perf_counter() - t_all
    kf.synchronise(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all
    frames_total += n
    kf.close()
    print("renders %d: %d frames, %.0f frames/s (host done after %.1f of %.1f ms)%s" % (renders, n, n / t_all, 1e3 * t_host, 1e3 * t_all, "".join("\n    " + x for x in stalls)))
for rep in range(4):
    leg(False, 68)
    leg(True, 220)
