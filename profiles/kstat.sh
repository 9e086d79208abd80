#!/bin/bash
# usage: profiles/kstat.sh <config> ["ENV=VAL ..."]...   - kernel durations from rocprofv3 --kernel-trace (no HIP events in the run)
cfg=$1; shift
export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "A=1"; fi
i=0
for envs in "$@"; do
  i=$((i+1))
  out=/tmp/kstat_$$_$i
  env $envs rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-measure-traffic --extra-configs none --config $cfg --steps 120 --warmup 20 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 > $out.json 2> $out.err
  python3 - $out "$envs" $out.json <<'PY'
import csv, glob, json, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kf::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
try:
    fps = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])["value"]
except Exception:
    fps = float("nan")
def steady(v):
    v = v[len(v) // 3:]
    return sum(v) / max(len(v), 1)
parts = []
for k in sorted(d):
    if any(s in k for s in ("integrate", "raycast")):
        parts.append("%s %.1f" % (k.split("<")[0], steady(d[k])))
icp = sum(sum(v[len(v) // 3:]) for k, v in d.items() if "icp" in k)
frames = len(d.get([k for k in d if "raycast" in k][0], [])) if any("raycast" in k for k in d) else 1
frames = max(1, frames - frames // 3)
print("%-44s fps %7.1f | %s | icp loop %.1f us/frame" % (sys.argv[2], fps, " | ".join(parts), icp / frames))
# per ICP kernel: steady average (rocprofv3's duration of a launch in a dependent chain includes its launch boundary)
def med(v):
    v = sorted(v[len(v) // 3:])
    return v[len(v) // 2] if v else float("nan")
print("    " + "  ".join("%s %.2f (x%.1f)" % (k.split("<")[0].replace("k_icp_", ""), steady(v), (len(v) - len(v) // 3) / frames) for k, v in sorted(d.items()) if "icp" in k))
if "KSTAT_MEDIANS" in __import__("os").environ:
    print("    medians: " + "  ".join("%s %.2f" % (k.replace("kf::", "").replace("k_icp_", "").split("(")[0], med(v)) for k, v in sorted(d.items()) if "icp" in k))
PY
  rm -rf $out $out.json $out.err
done
