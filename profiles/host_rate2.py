import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from slambench_amd import kfusion, synth
import types
args = types.SimpleNamespace(raw=None, gt=None, traj=2, config="cfg2")
cfg = bench.CONFIGS["cfg2"]
W, H = cfg[0], cfg[1]
wl = bench.load_workload(args, 0, 330, W, H)
d_depth = torch.from_numpy(wl.depth_mm.view(np.int16)).cuda()
for host in (False, True, False, True):
    leg = bench.Leg(kfusion, wl, d_depth, cfg, 0, False, host_input=host)
    for f in range(20): leg.step(f)
    leg.sync()
    t0 = time.perf_counter()
    for f in range(20, 320): leg.step(f)
    t1 = time.perf_counter()
    leg.sync()
    t2 = time.perf_counter()
    print("host_input=%s  enqueue %.1f us/frame  total %.1f us/frame (%.0f fps)" % (host, (t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6, 300 / (t2 - t0)))
    leg.close()
