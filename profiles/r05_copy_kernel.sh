#!/bin/bash
# fps_total (queued frames + the three renders and their read-back) with the read-back by the copy engine / by a copy kernel of n blocks
for v in 0 64 256 1024 0 256; do
  KFH_RENDER_COPY_KERNEL=$v python bench.py --steps 200 --no-cpu-baseline --no-host-buffers --no-measure-traffic --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('KFH_RENDER_COPY_KERNEL=$v value', d['value'], 'fps_total', d['fps_total'], d['roofline_other']['render_readback']['readback_and_queue_breaks_us_per_frame'])
"
done
