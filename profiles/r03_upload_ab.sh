#!/bin/bash
# A/B of who waits for the upload of a queued host frame (KFH_UPLOAD_WAIT=device|host), in the driver's 20-step shape and at 300 steps.
# usage: bash profiles/r03_upload_ab.sh > gpurun_out/r03_upload_ab.txt
for rep in 1 2; do
  for mode in device host; do
    for shape in "20 5" "300 20"; do
      set -- $shape
      KFH_UPLOAD_WAIT=$mode python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-host-buffers --extra-configs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode steps=$1 rep=$rep device-resident %.1f host-input %.1f ratio %.4f fps_total %.1f' % (d['value'], d['value_host_input'], d['value_host_input']/d['value'], d['fps_total']))
"
    done
  done
done
