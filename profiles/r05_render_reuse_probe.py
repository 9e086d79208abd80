"""k_render_volume with and without the reuse of the frame's raycast (KFH_RENDER_REUSE), event-timed; how many rays the tracking raycast missed"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from slambench_amd import kfusion, synth
W, H = 640, 480
s = synth.make_stream(40, W, H)
k = list(s.k)
for reuse in ("1", "0", "1", "0"):
    os.environ["KFH_RENDER_REUSE"] = reuse
    kf = kfusion.Kfusion((W, H), (256,) * 3, (4.8,) * 3, s.init_pos)
    for f in range(12):
        kf.compute_frame(s.depth_mm[f], (W, H), k, 2, 1, 1e-5, 0.1, f)
    kf.synchronise()
    kf.set_profiling(["render_volume", "raycast"], every=1)
    for f in range(12, 40):
        kf.compute_frame(s.depth_mm[f], (W, H), k, 2, 1, 1e-5, 0.1, f)
        img = kf.render_volume(f, 1, k, 0.75 * 0.1)
    kf.synchronise()
    prof = kf.read_profile()
    miss = float((kf.buffer("vertex")[..., :3] == 0).all(axis=-1).mean())
    print("KFH_RENDER_REUSE=%s  k_render_volume %.1f us (%d launches)  k_raycast %.1f us   rays the tracking raycast missed: %.1f %%" % (
        reuse, 1e3 * prof["render_volume"][0] / max(1, prof["render_volume"][1]), prof["render_volume"][1],
        1e3 * prof["raycast"][0] / max(1, prof["raycast"][1]), 100 * miss))
    kf.close()
