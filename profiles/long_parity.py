import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import util
from oracle import orc
from slambench_amd import kfusion
orc.build()
N = 80
W, H = 640, 480
s = util.synth.make_stream(N, W, H)
k = list(s.k)
res = (256, 256, 256)
g = kfusion.Kfusion((W, H), res, util.VOL_DIM, s.init_pos)
o = orc.OracleKfusion((W, H), res, util.VOL_DIM, s.init_pos, omp=True)
worst = 0.0
flags_equal = True
for f in range(N):
    g.compute_frame(s.depth_mm[f], (W, H), k, 2, 1, util.ICP_THRESHOLD, util.MU, f)
    o.preprocessing(s.depth_mm[f]); to = o.tracking(k, util.ICP_THRESHOLD, 1, f); io = o.integration(k, 2, util.MU, f); o.raycasting(k, util.MU, f)
    d = float(np.abs(g.get_pose() - o.get_pose()).max())
    worst = max(worst, d)
    flags_equal &= (bool(g.get_tracked()) == bool(to)) and (bool(g.get_integrated()) == bool(io))
vg, vo = g.buffer("volume"), o.buffer("volume")
diff = np.abs(vg[..., 0].astype(np.float32) - vo[..., 0].astype(np.float32)) / 32766.0
print("frames %d  max |pose - oracle| %.3g  flags equal %s  voxels off by > 1e-4: %.3g  weights differing: %.3g  timeouts %d" %
      (N, worst, flags_equal, float(np.mean(diff > 1e-4)), float(np.mean(vg[..., 1] != vo[..., 1])), g.barrier_timeouts()))
