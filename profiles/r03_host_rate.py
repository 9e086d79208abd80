"""Host-side cost of queueing a frame (bursts of 6 frames into an idle queue, so that the host never waits for the device) and the
steady rate, for the slot schedule (KFH_ICP_GENERIC=0) and the generic step launches.  usage: python profiles/r03_host_rate.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from slambench_amd import kfusion
import types
args = types.SimpleNamespace(raw=None, gt=None, traj=2, config="cfg2")
cfg = bench.CONFIGS["cfg2"]
W, H = cfg[0], cfg[1]
wl = bench.load_workload(args, 0, 330, W, H)
d_depth = torch.from_numpy(wl.depth_mm.view(np.int16)).cuda()
for mode in ("0", "1", "0", "1"):
    os.environ["KFH_ICP_GENERIC"] = mode
    leg = bench.Leg(kfusion, wl, d_depth, cfg, 0, False)
    for f in range(20): leg.step(f)
    leg.sync()
    burst = []
    f = 20
    for rep in range(20):
        t0 = time.perf_counter()
        for i in range(6):
            leg.step(f); f += 1
        t1 = time.perf_counter()
        leg.sync()
        burst.append((t1 - t0) / 6 * 1e6)
    t0 = time.perf_counter()
    for g in range(f, f + 160): leg.step(g)
    t1 = time.perf_counter()
    leg.sync()
    t2 = time.perf_counter()
    print("generic=%s  host enqueue (idle queue) %.1f us/frame (min %.1f)   steady: enqueue %.1f total %.1f us/frame (%.0f fps)" % (
        mode, float(np.median(burst)), min(burst), (t1 - t0) / 160 * 1e6, (t2 - t0) / 160 * 1e6, 160 / (t2 - t0)))
    leg.close()
