#!/bin/bash
# per-wave lifetimes of the 200th k_integrate launch (KFH_INTEGRATE_TRACE, profiles/int_trace.py)
# usage: profiles/r05_int_trace.sh <config> <out.txt>
cfg=$1; out=$2
steps=230; [ "$cfg" != "cfg3" ] && steps=430
KFH_INTEGRATE_TRACE=/tmp/int_trace_$$.bin python3 bench.py --no-measure-traffic --no-preheat --extra-configs none --config $cfg --steps $steps --warmup 10 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 > /dev/null 2>/tmp/int_trace_$$.err
python3 profiles/int_trace.py /tmp/int_trace_$$.bin > $out 2>&1 || cat /tmp/int_trace_$$.err >> $out
rm -f /tmp/int_trace_$$.bin /tmp/int_trace_$$.err
cat $out
