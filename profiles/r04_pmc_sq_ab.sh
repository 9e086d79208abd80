#!/bin/bash
# round 4: SQ instruction counters per kernel (max over launches) of the working tree against round 3's sources (.abtest/old_csrc)
# usage: profiles/r04_pmc_sq_ab.sh <out.txt>
out=$1
mkdir -p .abtest/new_csrc && cp slambench_amd/csrc/kf_kernels.hip slambench_amd/csrc/kf_kernels.h slambench_amd/csrc/kf_math.h slambench_amd/csrc/kf_pipeline.hip .abtest/new_csrc/
{
  echo "== round 4 (working tree)"
  bash profiles/pmc_sq.sh /tmp/sq_new | grep "kernel (max\|k_icp\|k_integrate\|k_raycast"
  cp .abtest/old_csrc/* slambench_amd/csrc/ && python -m slambench_amd.build --force > /dev/null 2>&1 || exit 1
  echo "== round 3 sources"
  bash profiles/pmc_sq.sh /tmp/sq_old | grep "kernel (max\|k_icp\|k_integrate\|k_raycast"
} > $out 2>&1
cp .abtest/new_csrc/* slambench_amd/csrc/
cat $out
