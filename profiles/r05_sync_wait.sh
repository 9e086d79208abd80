#!/bin/bash
# how the host waits at a synchronisation (hipSetDeviceFlags: spin / yield / blocking; unset = the runtime's default): the driver's
# 20-step shape of bench.py and the drop-in driver (five synchronisations per frame)
for w in default spin block yield default spin; do
  if [ $w = default ]; then unset KFH_SYNC_WAIT; else export KFH_SYNC_WAIT=$w; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-total --no-measure-traffic --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); h=d['fps_host_buffers']; print('KFH_SYNC_WAIT=$w: value (20 steps) %.1f   drop-in driver: computation %.1f total %.1f frames/s' % (d['value'], h['computation'], h['total']))
"
done
