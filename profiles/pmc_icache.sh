#!/bin/bash
# instruction-cache behaviour per kernel (one PMC pass).  Usage: profiles/pmc_icache.sh <out-dir>
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$out" -- python3 bench.py --no-measure-traffic --extra-configs none --steps 40 --warmup 10 --no-cpu-baseline --no-total > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
python3 - "$out" <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_BUSY_CYCLES"]
print("%-28s" % "kernel (mean over launches)" + "".join("%18s" % n for n in names))
for k, v in sorted(agg.items()):
    if "kf::" in k:
        print("%-28s" % k[:28] + "".join("%18.0f" % (sum(v.get(n, [0])) / max(1, len(v.get(n, [0])))) for n in names))
PY
