#!/bin/bash
# usage: profiles/sweep.sh "<ENV=VAL ...>" ...   — runs the short bench once per environment setting
for envs in "$@"; do
  echo "== $envs"
  env $envs python bench.py --steps 100 --no-cpu-baseline --no-total --no-host-buffers 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; o=d['roofline_other']
ks={r['kernel']:r}; ks.update({v['kernel']:v for v in o.values()})
print('fps %.1f ms %.4f | ' % (d['value'], d['ms_per_step']) + ' | '.join('%s %.1fus %.0fGB/s' % (k, v['avg_us'], v['achieved']) for k,v in sorted(ks.items())), '| icp %.1fus' % d.get('phase_us',{}).get('icp',0), '| ate %.4f' % d['ate_m']['mean'])
"
done
