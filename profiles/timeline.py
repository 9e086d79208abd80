"""Per-frame kernel timeline from a rocprofv3 --kernel-trace CSV: python profiles/timeline.py <dir> [frame-index]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kf::", "") for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith("k_raycast")]
fi = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
a, b = marks[fi] + 1, marks[fi + 1] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for i in range(a, b):
    r = rows[i]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-26s start %8.2f  dur %7.2f  gap %6.2f  grid %7s wg %5s" % (names[i][:26], (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3,
                                                                      r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
    prev_end = en
print("frame total %.2f us" % ((int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3))
