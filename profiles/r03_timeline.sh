#!/bin/bash
# per-frame kernel timeline (rocprofv3 --kernel-trace) of one configuration: profiles/r03_timeline.sh <config> <out.txt> ["ENV=VAL ..."]
cfg=$1; out=$2; envs=${3:-A=1}
export TMPDIR=/tmp
d=/tmp/tl_$$
env $envs rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --config $cfg --steps 120 --warmup 20 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 --extra-configs none --no-measure-traffic > $d.json 2> $d.err
{ echo "== $cfg $envs"; python3 profiles/timeline.py $d 60; python3 profiles/timeline.py $d 61; } > $out
rm -rf $d $d.json $d.err
