import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-52s calls=%6s total_ms=%9.3f avg_us=%8.2f min_us=%7.2f max_us=%8.2f pct=%5s" % (r['Name'][:52], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
