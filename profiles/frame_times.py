"""Per-frame device time from a rocprofv3 --kernel-trace CSV (frames delimited by k_raycast launches; frames 0-2 have none and are
lumped into the first): python profiles/frame_times.py <dir> [first [last]]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kf::", "") for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith("k_raycast")]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else min(len(marks) - 1, 30)
for fi in range(lo, hi):
    a, b = marks[fi] + 1, marks[fi + 1] + 1
    t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b - 1]["End_Timestamp"])
    busy = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(a, b))
    icp = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(a, b) if "icp" in names[i])
    integ = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(a, b) if "integrate" in names[i])
    ray = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(a, b) if "raycast" in names[i])
    print("frame %3d (raycast #%d)  span %7.1f us  kernels %7.1f  icp %6.1f (%2d launches)  integrate %6.1f  raycast %6.1f" %
          (fi + 4, fi + 1, (t1 - t0) / 1e3, busy / 1e3, icp / 1e3, sum(1 for i in range(a, b) if "icp" in names[i]), integ / 1e3, ray / 1e3))
