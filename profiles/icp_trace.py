"""Cycle stamps of block 0 / thread 0 of one k_icp_iter launch (KFH_ICP_TRACE=<file> [KFH_ICP_TRACE_IT=<launch index>]); a build with
-DKFH_SOLVE_STAMPS=1 adds the stamps inside the solve / pose update (words 8-15)."""
import sys
import numpy as np
w = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64)
a = w[:7]
names = ["state hand-over", "partial sums", "solve + pose update", "publish state", "association", "block reduction"]
for i, n in enumerate(names):
    print("%-22s %7d cycles" % (n, a[i + 1] - a[i]))
print("%-22s %7d cycles" % ("total", a[6] - a[0]))
if len(w) >= 16 and w[8] and w[15]:
    inner = ["entry -> LDLT done", "M = L^-1", "certificate + x", "SE3 exp", "pose product", "norm", "pose to LDS"]
    print("inside 'solve + pose update' (stamp 2 -> entry %d cycles):" % (w[8] - w[2]))
    for i, n in enumerate(inner):
        print("  %-20s %7d cycles" % (n, w[9 + i] - w[8 + i]))
    print("  %-20s %7d cycles" % ("exit -> stamp 3", w[3] - w[15]))
