#!/bin/bash
# memory-path counters per kernel, several passes (each block has few slots).  Usage: profiles/pmc_mem.sh <out-dir>
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
i=0
# at most two counters per TA / TCP block per pass: four made rocprofiler abort ("exceeds the capabilities of the hardware")
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/p$i" -- python3 bench.py --no-measure-traffic --extra-configs none --steps 30 --warmup 10 --no-cpu-baseline --no-total --no-host-buffers $BENCH_ARGS > "$out/p$i.json" 2> "$out/p$i.err" || { echo "pass $i failed"; tail -3 "$out/p$i.err" | cut -c1-200; }
  echo "pass $i done"
done
python3 - "$out" <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if any(s in k for s in ("raycast", "integrate", "icp_iter", "bilateral")):
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print("    %-36s max %16.0f  mean %16.0f  n=%d" % (c, max(v), sum(v) / len(v), len(v)))
PY
