#!/bin/bash
# round 4: same-box A/B of k_integrate / k_raycast compile-time variants: profiles/r04_ab_int.sh <out.txt> "<flags>" ...
out=$1; shift
: > $out
for flags in "$@"; do
  for cfg in cfg2 cfg3 cfg4; do
    bash profiles/ab_flags.sh "$flags" $cfg 2>&1 | grep -v "^    " >> $out || exit 1
  done
done
cat $out
