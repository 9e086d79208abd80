#!/bin/bash
# fps_total (queued frames + the three renders and their read-back): one copy kernel per image right behind its render kernel
# (KFH_RENDER_COPY_BATCH=0) against ONE copy kernel per frame for all images, issued by the next kfh_compute_frame (1)
for v in 0 0 1 0 1 0 1; do
  KFH_RENDER_COPY_BATCH=$v python bench.py --steps 200 --no-cpu-baseline --no-host-buffers --no-measure-traffic --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('KFH_RENDER_COPY_BATCH=$v: value', d['value'], 'fps_total', d['fps_total'], d['roofline_other']['render_readback']['readback_and_queue_breaks_us_per_frame'])
"
done
