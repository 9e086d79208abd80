#!/bin/bash
# the timed region of bench.py by step count: T = a + b * steps; a is what a 20-step run (the driver's shape) pays once
for st in 20 20 40 80 160 320 20; do
python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-total --no-host-buffers --no-measure-traffic --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('steps $st: value %.1f  region %.1f us  (%.2f us per step)' % (d['value'], d['ms_per_step'] * 1e3 * $st, d['ms_per_step'] * 1e3))
"
done
