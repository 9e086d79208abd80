#!/bin/bash
# round 5: k_integrate A/B on one box - rebuild per flag set, kernel durations from rocprofv3 (kstat.sh) at cfg2 / cfg3 / cfg4
# usage: profiles/r05_ab_int.sh <out.txt> "<flags A>" "<flags B>" ...
out=$1; shift
{
for flags in "$@"; do
  [ "$flags" = "default" ] && flags=""
  KFH_EXTRA_HIPCC_FLAGS="$flags" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== build flags: [$flags]"
  for c in ${AB_CONFIGS:-cfg2 cfg3 cfg4}; do bash profiles/kstat.sh $c | head -1 | sed "s/^A=1 */$c /"; done
done
} > $out 2>&1
python -m slambench_amd.build --force > /dev/null 2>&1
cat $out
