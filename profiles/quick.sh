#!/bin/bash
# usage: profiles/quick.sh "<ENV=VAL ...>" ...   — 300-frame bench without event records, one line per setting
for envs in "$@"; do
  env $envs python bench.py --steps 300 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-60s fps %.1f  ms %.4f' % ('$envs', d['value'], d['ms_per_step']))
"
done
