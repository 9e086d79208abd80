#!/bin/bash
# round 4: where the ~4.2 k cycles of "solve + pose update" go (thread 0 of block 0, launch KFH_ICP_TRACE_IT of the 200th frame);
# builds with -DKFH_SOLVE_STAMPS=1.   usage: profiles/r04_solve_stamps.sh <out.txt> "<extra flags>" ...
out=$1; shift
: > $out
for flags in "$@"; do
  KFH_EXTRA_HIPCC_FLAGS="-DKFH_SOLVE_STAMPS=1 $flags" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags" >> $out; exit 1; }
  for it in 10 2; do
    echo "== flags [$flags] launch $it" >> $out
    KFH_ICP_TRACE=/tmp/icp_trace.bin KFH_ICP_TRACE_IT=$it python bench.py --steps 260 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 --extra-configs none --no-measure-traffic --no-preheat > /tmp/st.json 2> /tmp/st.err || { tail -3 /tmp/st.err >> $out; exit 1; }
    python profiles/icp_trace.py /tmp/icp_trace.bin >> $out
    rm -f /tmp/icp_trace.bin
  done
done
cat $out
