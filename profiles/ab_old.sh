#!/bin/bash
# usage: profiles/ab_old.sh <config> ["ENV=VAL ..."]...  - kernel times of the working tree, then of the sources parked in .abtest/old_csrc
# (git show <commit>:slambench_amd/csrc/<file> > .abtest/old_csrc/<file> for the four files first; .abtest/ is git-ignored but travels with gpurun)
bash profiles/kstat.sh "$@"
cp .abtest/old_csrc/* slambench_amd/csrc/ && python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "old build failed"; exit 1; }
echo "== old sources"
bash profiles/kstat.sh "$@"
