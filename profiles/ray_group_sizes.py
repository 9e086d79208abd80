"""k_raycast time by image size for the current KFH_RAY_GROUP (event pairs around the raycast phase, queued frames, 256^3):
    for q in 1 2 4; do KFH_RAY_GROUP=$q python profiles/ray_group_sizes.py; done"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from slambench_amd import kfusion, synth

for (W, H) in [(320, 240), (480, 360), (640, 480), (800, 600), (960, 720), (1280, 960)]:
    n = 40
    s = synth.make_stream(n, W, H, seed=7)
    k = list(s.k)
    kf = kfusion.Kfusion((W, H), (256, 256, 256), (4.8,) * 3, s.init_pos, (10, 5, 4))
    d = torch.from_numpy(np.ascontiguousarray(s.depth_mm).view(np.int16)).cuda()
    fb = W * H * 2
    for f in range(12):
        kf.compute_frame(d.data_ptr() + f * fb, (W, H), k, 2, 1, 1e-5, 0.1, f)
    kf.synchronise()
    kf.set_profiling(["raycast"], every=1)
    for f in range(12, n):
        kf.compute_frame(d.data_ptr() + f * fb, (W, H), k, 2, 1, 1e-5, 0.1, f)
    kf.synchronise()
    prof = kf.read_profile()
    ms, cnt = prof["raycast"]
    print("KFH_RAY_GROUP=%s  %4dx%-4d  %8d rays  k_raycast %.1f us (events, %d launches)" % (os.environ.get("KFH_RAY_GROUP", "auto"), W, H, W * H, 1e3 * ms / max(cnt, 1), cnt))
    kf.close()
