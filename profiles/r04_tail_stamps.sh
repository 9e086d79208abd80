#!/bin/bash
# round 4: cycle stamps of k_icp_tail's usual path (block 0 / thread 0, 200th frame): entry -> state read -> records -> control ops
out=$1; shift
: > $out
for flags in "$@"; do
  KFH_EXTRA_HIPCC_FLAGS="$flags" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags" >> $out; exit 1; }
  echo "== flags [$flags]" >> $out
  KFH_ICP_TAIL_TRACE=/tmp/tail_trace.bin python bench.py --steps 260 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 --extra-configs none --no-measure-traffic --no-preheat > /tmp/st.json 2> /tmp/st.err || { tail -3 /tmp/st.err >> $out; exit 1; }
  python - >> $out <<'PY'
import numpy as np
w = np.fromfile("/tmp/tail_trace.bin", dtype=np.uint64).astype(np.int64)
print("enter (state read, barriers) %6d | records %6d | close (control ops) %6d | total %6d cycles" % (w[1] - w[0], w[2] - w[1], w[3] - w[2], w[3] - w[0]))
PY
  rm -f /tmp/tail_trace.bin
done
cat $out
