"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run: python profiles/prof_summary.py <dir>
Columns as in the *_kernel_stats.csv, plus the average over the second half of each kernel's launches (steady state: the
first frames of a run integrate into an empty volume and raycast nothing, bench.py times frames 20..319 only)."""
import collections, csv, glob, sys
d = sys.argv[1]
stats = list(csv.DictReader(open(glob.glob(d + '/*/*_kernel_stats.csv')[0])))
trace = glob.glob(d + '/*/*_kernel_trace.csv')
late = {}
if trace:
    per = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r['Start_Timestamp'])):
        per[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, v in per.items():
        h = v[len(v) // 2:]
        late[k] = sum(h) / len(h) / 1e3
for r in stats[:14]:
    print("%-52s calls=%6s total_ms=%9.3f avg_us=%8.2f min_us=%7.2f max_us=%8.2f pct=%5s steady_avg_us=%8.2f" % (
        r['Name'][:52], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
        float(r['MaxNs']) / 1e3, r['Percentage'], late.get(r['Name'], float('nan'))))
