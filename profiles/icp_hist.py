"""Histogram of executed ICP iterations per frame and level (cfg2 stream).  usage: python profiles/icp_hist.py [frames]"""
import sys, collections
import numpy as np, torch
sys.path.insert(0, ".")
from slambench_amd import kfusion, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 320
W, H = 640, 480
stream = synth.cached_stream(n, W, H, seed=1234, variant=0)
d = torch.from_numpy(stream.depth_mm.view(np.int16)).cuda()
kf = kfusion.Kfusion((W, H), (256,) * 3, (4.8,) * 3, stream.init_pos, (10, 5, 4), device=0)
prev = [0, 0, 0]
hist = [collections.Counter() for _ in range(3)]
for f in range(n):
    kf.compute_frame(d.data_ptr() + f * W * H * 2, (W, H), list(stream.k), 2, 1, 1e-5, 0.1, f)
    cur = kf.icp_level_iterations()[:3]
    for l in range(3):
        hist[l][cur[l] - prev[l]] += 1
    prev = cur
for l in range(3):
    print("level", l, sorted(hist[l].items()))
