#!/bin/bash
# SQ counters per kernel (one pass, 8 SQ slots).  Usage: profiles/pmc_sq.sh <out-dir> [env assignments / bench args via BENCH_ARGS]
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d "$out" -- python3 bench.py --no-measure-traffic --extra-configs none --steps 40 --warmup 10 --no-cpu-baseline --no-total --no-host-buffers $BENCH_ARGS > "$out/bench.json" 2> "$out/bench.err"
python3 - "$out" <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"]
print("%-28s" % "kernel (max over launches)" + "".join("%16s" % n[3:] for n in names))
for k, v in sorted(agg.items()):
    if "kf::" in k:
        print("%-28s" % k[:28] + "".join("%16.0f" % max(v.get(n, [0])) for n in names))
PY
