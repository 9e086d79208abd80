#!/bin/bash
# round 5: brick size of the empty-space skipping, 8^3 (default) against 4^3 and 16^3 - rebuild per flag, parity tests once, kernel times
out=$1
{
for flags in "default" "-DKFH_BRICK_SHIFT=2" "-DKFH_BRICK_SHIFT=4" "default" "-DKFH_BRICK_SHIFT=2"; do
  f="$flags"; [ "$f" = "default" ] && f=""
  KFH_EXTRA_HIPCC_FLAGS="$f" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== build flags: [$f]"
  python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py -m gpu -x -q -k "lockstep or variants or skip or config2" 2>&1 | tail -1
  for c in cfg2 cfg3 cfg4; do bash profiles/kstat.sh $c | head -1 | sed "s/^A=1 */$c /"; done
done
} > $out 2>&1
python -m slambench_amd.build --force > /dev/null 2>&1
cat $out
