#!/bin/bash
# usage: profiles/kbench.sh <out-dir> "<ENV=VAL ...>" ...   — kernel microbenchmarks under rocprofv3 --stats
out=$1; shift
export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  echo "== $envs"
  env $envs timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/k$i" -- python3 profiles/kbench_volume.py > "$out.k$i.log" 2>&1
  python3 - "$out/k$i" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
import collections
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    if "integrate" in k or "raycast" in k:
        tail = v[len(v) // 2:]          # second half: the kernel-level repetitions on the settled state
        print("   %-26s n=%3d  last-half avg %8.2f us  min %8.2f" % (k[:26], len(v), sum(tail) / len(tail), min(tail)))
PY
done
