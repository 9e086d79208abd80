import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from slambench_amd import kfusion, synth
from tests import util
s = synth.cached_stream(400)
W, H = s.width, s.height
k = list(s.k)
mk = lambda: kfusion.Kfusion((W, H), (256, 256, 256), (4.8,) * 3, s.init_pos, (10, 5, 4))
ctx = [mk() for _ in range(3)]
d_all = torch.from_numpy(np.ascontiguousarray(s.depth_mm).view(np.int16)).cuda()
fb = W * H * 2
t0 = time.perf_counter()
scratch = np.zeros_like(s.depth_mm[0])
for rep in range(4):
    for f in range(400):
        g = rep * 400 + f
        ctx[0].compute_frame(d_all.data_ptr() + f * fb, (W, H), k, 2, 1, 1e-5, 0.1, g)     # device-resident
        scratch[...] = s.depth_mm[f]
        ctx[1].compute_frame(scratch, (W, H), k, 2, 1, 1e-5, 0.1, g)                       # host, buffer reused
        ctx[2].compute_frame(s.depth_mm[f], (W, H), k, 2, 1, 1e-5, 0.1, g)                 # host, distinct buffers
        scratch[...] = 0
        if g % 97 == 0:
            ctx[1].get_pose()          # a getter in the middle of the queue
for c in ctx: c.synchronise()
t1 = time.perf_counter()
p = [c.get_pose() for c in ctx]
v = [c.buffer("volume") for c in ctx]
print("1600 frames x 3 contexts in %.2f s; poses equal: %s %s; volumes equal: %s %s; errors %s" % (
    t1 - t0, np.array_equal(p[0], p[1]), np.array_equal(p[0], p[2]), np.array_equal(v[0], v[1]), np.array_equal(v[0], v[2]), [c.get_error() for c in ctx]))
for c in ctx: c.close()
