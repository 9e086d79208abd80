#!/bin/bash
# usage: profiles/sweep_cfg.sh <config> "<ENV=VAL ...>" ...   — short bench of another configuration, one line per setting
cfg=$1; shift
for envs in "$@"; do
  env $envs python bench.py --config $cfg --steps 100 --no-cpu-baseline --no-total --no-host-buffers 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-44s fps %.1f ms %.4f  phases %s' % ('$envs', d['value'], d['ms_per_step'], d['phase_us']))
"
done
