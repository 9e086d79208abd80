#!/bin/bash
# usage: profiles/ab_build.sh "<hipcc -D flags>" <config> ["ENV=VAL ..."]...   - rebuilds the library on the GPU box with extra
# compile-time knobs (KFH_EXTRA_HIPCC_FLAGS), runs the short bench of <config> once per environment setting, restores nothing
# (the box is thrown away after the call).
flags=$1; cfg=$2; shift 2
KFH_EXTRA_HIPCC_FLAGS="$flags" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
echo "== build flags: [$flags]"
if [ $# -eq 0 ]; then set -- "A=1"; fi
bash profiles/sweep_cfg.sh $cfg "$@"
