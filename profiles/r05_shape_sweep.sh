#!/bin/bash
# re-check of the tile / batch shapes earlier rounds chose, on this round's kernels (kernel durations from rocprofv3, kstat.sh)
out=$1
{
echo "== k_integrate voxels per batch (run-time, KFH_INTEGRATE_BATCH)"
for c in cfg2 cfg3; do bash profiles/kstat.sh $c "KFH_INTEGRATE_BATCH=4" "KFH_INTEGRATE_BATCH=8" "KFH_INTEGRATE_BATCH=16" | grep fps | sed "s/^/$c /"; done
for flags in "default" "-DKFH_RAY_BLOCK_WAVES=1" "-DKFH_RAY_BLOCK_WAVES=4" "-DKFH_INT_TX=8 -DKFH_INT_BX=16 -DKFH_INT_BY=16" "-DKFH_INT_TX=8 -DKFH_INT_BX=32 -DKFH_INT_BY=8" "-DKFH_INT_TX=32 -DKFH_INT_BX=64 -DKFH_INT_BY=4" "-DKFH_INT_TX=16 -DKFH_INT_BX=64 -DKFH_INT_BY=4" "-DKFH_INT_TX=16 -DKFH_INT_BX=16 -DKFH_INT_BY=16" "default"; do
  f="$flags"; [ "$flags" = "default" ] && f=""
  KFH_EXTRA_HIPCC_FLAGS="$f" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== build flags: [$flags]"
  for c in cfg2 cfg3 cfg4; do bash profiles/kstat.sh $c | head -1 | sed "s/^A=1 */$c /"; done
done
} > $out 2>&1
python -m slambench_amd.build --force > /dev/null 2>&1
cat $out
