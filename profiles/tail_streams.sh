#!/bin/bash
# k_icp_tail's grid barrier with several contexts sharing the device (every queued iteration forced to run)
for spg in 1 3 6; do
  for thr in 1e-5 0; do
    KFH_DEBUG_ICP_THRESHOLD=$thr timeout -k 10 200 python bench.py --steps 200 --no-cpu-baseline --no-total --no-host-buffers --extra-configs none --no-measure-traffic --profile-every 0 --streams-per-gpu $spg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $spg threshold $thr: fps %.1f  its %.2f ate %.9f' % (d['value'], d['config']['icp_iterations_per_frame'], d['ate_m']['mean']))
" || exit 1
  done
done
