#!/bin/bash
# cycle stamps of block 0 in the launches of iterations 2, 3, 5 and 10 of a frame (no side jobs)
for it in 2 3 5 10; do
  KFH_PIPELINE=0 KFH_ICP_TRACE=gpurun_out/icp_tr_$it.bin KFH_ICP_TRACE_IT=$it python3 bench.py --no-measure-traffic --extra-configs none --config cfg2 --steps 260 --warmup 20 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 --no-preheat > /dev/null 2>&1
  echo "== iteration $it"; python3 profiles/icp_trace.py gpurun_out/icp_tr_$it.bin; rm -f gpurun_out/icp_tr_$it.bin
done
