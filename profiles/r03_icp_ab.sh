#!/bin/bash
# A/B of the ICP launch schedule: slot launches + k_icp_tail (rounds 1-2, KFH_ICP_GENERIC=0) against generic step launches (default).
# usage: bash profiles/r03_icp_ab.sh [config ...] > gpurun_out/r03_icp_ab.txt
for cfg in ${@:-cfg2}; do
  for rep in 1 2; do
    for mode in 0 1; do
      KFH_ICP_GENERIC=$mode python bench.py --config $cfg --steps 300 --warmup 20 --no-cpu-baseline --no-host-buffers --no-total --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg generic=$mode rep=$rep fps %.1f ms %.4f phase_us %s its/frame %s ate %.6f' % (d['value'], d['ms_per_step'], d['phase_us'], d['config']['icp_iterations_per_frame'], d['ate_m']['mean']))
"
    done
  done
done
