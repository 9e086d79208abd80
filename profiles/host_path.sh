#!/bin/bash
# usage: profiles/host_path.sh ["ENV=VAL ..."]...  - the drop-in driver on a 120-frame .raw of the bench stream: mean of each timing column
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from slambench_amd import synth
s = synth.cached_stream(120)
synth.write_raw("/tmp/host_path.raw", s)
PY
if [ $# -eq 0 ]; then set -- "A=1"; fi
for envs in "$@"; do
  env $envs slambench_amd/kfusion-benchmark-hip -i /tmp/host_path.raw -s 4.8 -p 0.34,0.5,0.24 -z 4 -c 1 -r 2 -k 481.2,480,320,240 -o /tmp/host_path.log 2> /dev/null
  python3 - "$envs" <<'PY'
import sys
sys.path.insert(0, ".")
from slambench_amd import checkpos
c = checkpos.parse_log(open("/tmp/host_path.log").read())["columns"]
m = lambda k: 1e6 * sum(c[k][10:]) / len(c[k][10:])
print("%-28s preprocessing %6.1f  tracking %6.1f  integration %6.1f  raycasting %6.1f  rendering %6.1f | computation %6.1f us  total %6.1f us" %
      (sys.argv[1], m("preprocessing"), m("tracking"), m("integration"), m("raycasting"), m("rendering"), m("computation"), m("total")))
PY
done
