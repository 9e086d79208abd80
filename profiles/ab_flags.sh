#!/bin/bash
# usage: profiles/ab_flags.sh "<hipcc -D flags>" <config> ["ENV=VAL ..."]...  - rebuild with compile-time knobs, then kstat.sh
flags=$1; shift
KFH_EXTRA_HIPCC_FLAGS="$flags" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
echo "== build flags: [$flags]"
bash profiles/kstat.sh "$@"
