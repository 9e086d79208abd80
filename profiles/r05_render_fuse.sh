#!/bin/bash
# fps_total: renderDepth and renderTrack of a queued frame as two launches (KFH_RENDER_FUSE=0) or one (1)
for v in 0 0 1 0 1 0 1; do
  KFH_RENDER_FUSE=$v python bench.py --steps 200 --no-cpu-baseline --no-host-buffers --no-measure-traffic --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('KFH_RENDER_FUSE=$v: value', d['value'], 'fps_total', d['fps_total'], d['roofline_other']['render_readback']['frame_us_with_renders'])
"
done
