#!/bin/bash
# usage: profiles/r05_int_stamps.sh <config> <out.txt>  - per-batch stamps of one k_integrate launch (build with -DKFH_INT_STAMPS=1, then the default build again)
cfg=$1; out=$2
KFH_EXTRA_HIPCC_FLAGS="-DKFH_INT_STAMPS=1 $STAMP_EXTRA" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
steps=230; [ "$cfg" != "cfg3" ] && steps=430
KFH_INTEGRATE_TRACE=/tmp/int_stamps_$cfg.bin python3 bench.py --no-measure-traffic --no-preheat --extra-configs none --config $cfg --steps $steps --warmup 10 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 > /dev/null 2>&1
python profiles/int_stamps.py /tmp/int_stamps_$cfg.bin > $out 2>&1
python -m slambench_amd.build --force > /dev/null 2>&1
cat $out
