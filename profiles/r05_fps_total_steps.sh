for st in 100 200 300 400 300; do
python bench.py --steps $st --no-cpu-baseline --no-host-buffers --no-measure-traffic --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('steps $st value', d['value'], 'fps_total', d['fps_total'], d['roofline_other']['render_readback']['readback_and_queue_breaks_us_per_frame'])
"
done
