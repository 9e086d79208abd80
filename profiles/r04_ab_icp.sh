#!/bin/bash
# round 4: same-box A/B of the ICP instruction diets (compile-time flags), kernel times from rocprofv3 (profiles/kstat.sh)
# usage: profiles/r04_ab_icp.sh <out.txt> "<flags>" ...
out=$1; shift
: > $out
for flags in "$@"; do
  for cfg in cfg2 cfg4; do
    bash profiles/ab_flags.sh "$flags" $cfg >> $out 2>&1 || exit 1
  done
done
cat $out
