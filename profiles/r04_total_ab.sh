#!/bin/bash
# round 4: fps_total (queued loop + the three renders and their read-back) of the working tree against the sources parked in
# .abtest/old_csrc (round 3's), same box, alternating.   usage: profiles/r04_total_ab.sh <out.txt>
out=$1
run() { python bench.py --steps 300 --no-cpu-baseline --no-host-buffers --extra-configs none --no-measure-traffic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 value %.1f host_input %s fps_total %s' % (d['value'], d['value_host_input'], d['fps_total']))
"; }
: > $out
mkdir -p .abtest/new_csrc && cp slambench_amd/csrc/kf_kernels.hip slambench_amd/csrc/kf_kernels.h slambench_amd/csrc/kf_math.h slambench_amd/csrc/kf_pipeline.hip .abtest/new_csrc/
for rep in 1 2; do
  cp .abtest/new_csrc/* slambench_amd/csrc/ && python -m slambench_amd.build --force > /dev/null 2>&1 || exit 1
  run new >> $out
  cp .abtest/old_csrc/* slambench_amd/csrc/ && python -m slambench_amd.build --force > /dev/null 2>&1 || exit 1
  run old >> $out
done
cp .abtest/new_csrc/* slambench_amd/csrc/
cat $out
