#!/bin/bash
# End-to-end run of the drop-in driver: synthetic .raw + freiburg GT -> kfusion-benchmark-hip -> ATE / timing summary.
# Usage: profiles/e2e_driver.sh <out-dir> <frames> [driver flags...]
set -e
out=$1; n=$2; shift 2
mkdir -p "$out"
python3 - "$out" "$n" <<'PY'
import sys
sys.path.insert(0, ".")
from slambench_amd import synth
out, n = sys.argv[1], int(sys.argv[2])
s = synth.cached_stream(n)
synth.write_raw(out + "/synth.raw", s)
synth.write_freiburg(out + "/synth.gt.freiburg", s)
PY
env -u KERNEL_TIMINGS slambench_amd/kfusion-benchmark-hip -i "$out/synth.raw" -s 4.8 -p 0.34,0.5,0.24 -z 4 -c 1 -r 2 -k 481.2,480,320,240 -o "$out/benchmark.log" "$@" 2> "$out/driver.err"
python3 -m slambench_amd.checkpos "$out/benchmark.log" "$out/synth.gt.freiburg" | tee "$out/resume.txt"
rm -f "$out/synth.raw"
