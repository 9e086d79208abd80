"""Steady frame time (queued frames, device-resident input, no event records) for ICP schedule variants given as KEY=VAL[,KEY=VAL] arguments.
usage: python profiles/r03_steps_sweep.py [config] "KFH_ICP_GENERIC=0" "KFH_ICP_STEPS=19" ..."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from slambench_amd import kfusion
import types
cfg_name = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].startswith("cfg") else "cfg2"
variants = [a for a in sys.argv[1:] if not a.startswith("cfg")] or ["A=1"]
args = types.SimpleNamespace(raw=None, gt=None, traj=2, config=cfg_name)
cfg = bench.CONFIGS[cfg_name]
W, H = cfg[0], cfg[1]
wl = bench.load_workload(args, 0, 330, W, H)
d_depth = torch.from_numpy(wl.depth_mm.view(np.int16)).cuda()
KEYS = ("KFH_ICP_GENERIC", "KFH_ICP_STEPS", "KFH_ICP_TAIL_HEAD")
for rep in range(2):
    for v in variants:
        for k in KEYS:
            os.environ.pop(k, None)
        for kv in v.split(","):
            k, val = kv.split("=")
            os.environ[k] = val
        leg = bench.Leg(kfusion, wl, d_depth, cfg, 0, False)
        for f in range(40): leg.step(f)
        leg.sync()
        t0 = time.perf_counter()
        for f in range(40, 320): leg.step(f)
        leg.sync()
        t2 = time.perf_counter()
        its = leg.kf.icp_iterations() / 320.0
        print("%-40s %.1f us/frame (%.0f fps)  its/frame %.2f" % (v, (t2 - t0) / 280 * 1e6, 280 / (t2 - t0), its))
        leg.close()
