"""Cycle stamps of block 0 / thread 0 of one k_icp_iter launch (KFH_ICP_TRACE=<file> [KFH_ICP_TRACE_IT=<launch index>])."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64)[:7].astype(np.int64)
names = ["state hand-over", "partial sums", "solve + pose update", "publish state", "association", "block reduction"]
for i, n in enumerate(names):
    print("%-22s %7d cycles" % (n, a[i + 1] - a[i]))
print("%-22s %7d cycles" % ("total", a[6] - a[0]))
