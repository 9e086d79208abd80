#!/bin/bash
# round 5: instruction census of k_raycast.  -DKFH_RAY_PROBES=1 build, KFH_RAY_PROBE_STAGE: 0 whole kernel, 1 every ray ends
# after its set-up (view, direction, three divisions, slab test), 2 no surface gradient at the hit, 4 every sample treated as lying in an
# empty brick (no trilinear sample: what the march costs without one).  Results are wrong with a probe on; SQ counters, mean over the
# last 20 launches.   usage: profiles/r05_ray_census.sh <config> <out.txt>
cfg=$1; out=$2
export TMPDIR=/tmp
KFH_EXTRA_HIPCC_FLAGS="-DKFH_RAY_PROBES=1" python -m slambench_amd.build --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
{
echo "== k_raycast census, $cfg: mean of the last 20 launches"
printf "%-8s %10s %12s %12s %12s %14s %14s %14s %14s\n" stage us INSTS_VALU INSTS_SALU VMEM_RD WAVE_CYCLES ACTIVE_INST WAIT_INST WAIT_ANY
for stage in 0 1 2 4; do
  d=/tmp/rcensus_$$_$stage
  KFH_RAY_PROBE_STAGE=$stage rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $d -- python3 bench.py --no-measure-traffic --no-preheat --extra-configs none --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-total --no-host-buffers --profile-every 0 > $d.json 2> $d.err
  python3 - $d $stage <<'PY'
import csv, collections, glob, sys
d, stage = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "k_raycast" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(per)[-20:]
mean = lambda name: sum(per[i][name] for i in ids) / max(1, len(ids))
kt = glob.glob(d + "/*/*_kernel_trace.csv")
us = float("nan")
if kt:
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if "k_raycast" in r["Kernel_Name"]]
    us = sum(v[-20:]) / max(1, len(v[-20:]))
print("%-8s %10.1f %12.0f %12.0f %12.0f %14.0f %14.0f %14.0f %14.0f" % (stage, us, mean("SQ_INSTS_VALU"), mean("SQ_INSTS_SALU"), mean("SQ_INSTS_VMEM_RD"), mean("SQ_WAVE_CYCLES"), mean("SQ_ACTIVE_INST_ANY"), mean("SQ_WAIT_INST_ANY"), mean("SQ_WAIT_ANY")))
PY
  rm -rf $d $d.json $d.err
done
} > $out 2>&1
python -m slambench_amd.build --force > /dev/null 2>&1
cat $out
