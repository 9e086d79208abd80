#!/bin/bash
# fps_total (queued frames + the three renders and their read-back): the frame closed by its first render (KFH_RENDER_DEFER=0) against
# the renders carried out by the call that closes the frame (1: the next frame's preprocessing rides on the ICP launches as without renders)
for v in 0 0 1 0 1 0 1; do
  KFH_RENDER_DEFER=$v python bench.py --steps 200 --no-cpu-baseline --no-host-buffers --no-measure-traffic --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('KFH_RENDER_DEFER=$v: value', d['value'], 'fps_total', d['fps_total'], d['roofline_other']['render_readback'])
"
done
