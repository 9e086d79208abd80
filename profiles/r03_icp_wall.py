"""Free-running wall-clock stamps (100 MHz) of the launches of one frame's ICP loop (KFH_ICP_WALL): python profiles/r03_icp_wall.py <out.txt> [KEY=VAL ...]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
out = sys.argv[1]
for kv in sys.argv[2:]:
    k, v = kv.split("="); os.environ[k] = v
os.environ["KFH_ICP_WALL"] = out
import numpy as np, torch, types
import bench
from slambench_amd import kfusion
args = types.SimpleNamespace(raw=None, gt=None, traj=2, config="cfg2")
cfg = bench.CONFIGS["cfg2"]
wl = bench.load_workload(args, 0, 330, cfg[0], cfg[1])
d_depth = torch.from_numpy(wl.depth_mm.view(np.int16)).cuda()
leg = bench.Leg(kfusion, wl, d_depth, cfg, 0, False)
for f in range(200): leg.step(f)
leg.sync()
leg.close()
print(open(out).read())
