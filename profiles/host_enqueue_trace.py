"""Host-side duration of every kfh_compute_frame call of a queued run (device-resident input): where does the host block?
    python profiles/host_enqueue_trace.py [frames]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from slambench_amd import kfusion, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
s = synth.cached_stream(n)
k = list(s.k)
d = torch.from_numpy(s.depth_mm.view(np.int16)).cuda()
for rep in range(2):
    kf = kfusion.Kfusion((640, 480), (256,) * 3, (4.8,) * 3, s.init_pos)
    torch.cuda.synchronize()
    t = []
    t0 = time.perf_counter()
    for f in range(n):
        a = time.perf_counter()
        kf.compute_frame(d.data_ptr() + f * 640 * 480 * 2, (640, 480), k, 2, 1, 1e-5, 0.1, f)
        t.append(time.perf_counter() - a)
    enq = time.perf_counter() - t0
    kf.synchronise()
    tot = time.perf_counter() - t0
    t = np.array(t) * 1e6
    print("run %d: %d frames, enqueue %.1f us/frame (median call %.1f us), total %.1f us/frame; calls over 150 us: %s" %
          (rep, n, enq / n * 1e6, np.median(t), tot / n * 1e6, [(i, round(v)) for i, v in enumerate(t) if v > 150]))
    kf.close()
