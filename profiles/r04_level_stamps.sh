#!/bin/bash
# round 4: cycle stamps of block 0 / thread 0 of a whole-level launch (k_icp_level and its side-job variants), 200th frame
# usage: profiles/r04_level_stamps.sh <out.txt> [first iteration of the level: 0 = coarsest, 4 = next at -y 10,5,4]...
out=$1; shift
: > $out
for it in "$@"; do
  echo "== level launch that starts at iteration $it" >> $out
  KFH_ICP_LEVEL_TRACE=/tmp/lv_trace.bin KFH_ICP_TRACE_IT=$it python bench.py --steps 260 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 --extra-configs none --no-measure-traffic --no-preheat > /tmp/st.json 2> /tmp/st.err || { tail -3 /tmp/st.err >> $out; exit 1; }
  python - >> $out <<'PY'
import numpy as np
w = np.fromfile("/tmp/lv_trace.bin", dtype=np.uint64).astype(np.int64)
print("entry (stamp 0 -> state in LDS + record): %d cycles" % (w[7] - w[0]))
j = 0
while 10 + 3 * j < len(w) and w[8 + 3 * j]:
    prev = w[7] if j == 0 else w[10 + 3 * (j - 1)]
    print("iteration %d: association + reduction %6d | barrier %6d | rows + solve %6d" % (j, w[8 + 3 * j] - prev, w[9 + 3 * j] - w[8 + 3 * j], w[10 + 3 * j] - w[9 + 3 * j]))
    j += 1
PY
  rm -f /tmp/lv_trace.bin
done
cat $out
