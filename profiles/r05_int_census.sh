#!/bin/bash
# round 5: per-stage instruction census of k_integrate.  A build with -DKFH_INT_PROBES=1 cuts the kernel short at a stage
# (results wrong from launch KFH_INT_PROBE_FROM on; the state up to there is real): 0 full kernel, 8 no whole-chunk reject, 1 every batch
# rejected after its end-voxel test (chunk reject + z0 replay + march + batch tests), 2 stop after the projection pass, 4 stop after the
# depth gathers + classification, 16 stop after the voxel loads (before the update arithmetic).  SQ counters of the last 10 launches.
# usage: profiles/r05_int_census.sh <config> <out.txt> [extra hipcc flags]
cfg=$1; out=$2; extra=$3
export TMPDIR=/tmp
KFH_EXTRA_HIPCC_FLAGS="-DKFH_INT_PROBES=1 $extra" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
{
echo "== k_integrate census, $cfg, flags [-DKFH_INT_PROBES=1 $extra]: mean of the last 10 launches"
printf "%-8s %10s %12s %12s %12s %14s %14s %14s %14s\n" stage us INSTS_VALU INSTS_SALU VMEM_RD WAVE_CYCLES ACTIVE_INST WAIT_INST WAIT_ANY
for stage in 0 8 1 2 4 16; do
  d=/tmp/census_$$_$stage
  KFH_INT_PROBE_STAGE=$stage KFH_INT_PROBE_FROM=30 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $d -- python3 bench.py --no-measure-traffic --no-preheat --extra-configs none --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 > $d.json 2> $d.err
  python3 - $d $stage <<'PY'
import csv, collections, glob, sys
d, stage = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
t = {}
for r in csv.DictReader(open(f)):
    if "k_integrate" not in r["Kernel_Name"]:
        continue
    per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    t[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if "End_Timestamp" in r and r.get("End_Timestamp") else float("nan")
ids = sorted(per)[-10:]
def mean(name):
    return sum(per[i][name] for i in ids) / max(1, len(ids))
kt = glob.glob(d + "/*/*_kernel_trace.csv")
us = float("nan")
if kt:
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if "k_integrate" in r["Kernel_Name"]]
    us = sum(v[-10:]) / max(1, len(v[-10:]))
print("%-8s %10.1f %12.0f %12.0f %12.0f %14.0f %14.0f %14.0f %14.0f" % (stage, us, mean("SQ_INSTS_VALU"), mean("SQ_INSTS_SALU"), mean("SQ_INSTS_VMEM_RD"), mean("SQ_WAVE_CYCLES"), mean("SQ_ACTIVE_INST_ANY"), mean("SQ_WAIT_INST_ANY"), mean("SQ_WAIT_ANY")))
PY
  rm -rf $d $d.json $d.err
done
} > $out 2>&1
python -m slambench_amd.build --force > /dev/null 2>&1
cat $out
