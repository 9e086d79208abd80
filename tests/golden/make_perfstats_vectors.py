#!/usr/bin/env python3
"""Regenerates tests/golden/perfstats_vectors.json from the REFERENCE's own PerfStats (kfusion/thirdparty/perfstats.h).

`make -C oracle ref` compiles tests/cpp/perfstats_probe.cpp against that header where it lies, with the reference's -O3
(CMakeLists.txt:7-8), into oracle/_ref/perfstats_probe; its output on the seeded sample tables below - the columns of a benchmark
log, typed as mainQt.cpp:70-87 types them - is the fixture: the closing "# <name> Mean: ... Standard deviation: ... min: ... max: ..."
block in both forms the reference prints it (log to stdout / log to a file).  Only data is committed."""
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
COLUMNS = [("frame", "FRAME"), ("acquisition", "TIME"), ("preprocessing", "TIME"), ("tracking", "TIME"), ("integration", "TIME"),
           ("raycasting", "TIME"), ("rendering", "TIME"), ("computation", "TIME"), ("total", "TIME"), ("X", "DISTANCE"), ("Y", "DISTANCE"),
           ("Z", "DISTANCE"), ("tracked", "INT"), ("integrated", "INT")]


def table(seed, n):
    """n rows shaped like a benchmark log (benchmark.cpp:160-170): times in seconds, positions in metres, flags."""
    rng = np.random.default_rng(seed)
    rows = []
    for f in range(n):
        t = rng.uniform(1e-5, 4e-3, size=6)
        pos = rng.normal(0.0, 0.3, size=3)
        rows.append([f] + [round(float(v), 6) for v in t] + [round(float(t[1:5].sum()), 6), round(float(t.sum()), 6)]
                    + [round(float(v), 6) for v in pos] + [int(f >= 4), int(f < 4 or f % 2 == 0)])
    return rows


CASES = {"twelve_frames": table(7, 12), "one_frame": table(8, 1), "ninety_frames": table(9, 90)}


def run(probe, rows, to_file):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "in.txt")
        with open(src, "w") as f:
            f.write(" ".join("%s:%s" % c for c in COLUMNS) + "\n")
            for r in rows:
                f.write(" ".join(repr(v) for v in r) + "\n")
        out = os.path.join(d, "log.txt")
        stdout = subprocess.check_output([probe, src, out if to_file else "-"]).decode()
        return [l for l in stdout.split("\n") if l.startswith("#") and "Mean:" in l]


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    probe = os.path.join(ROOT, "oracle", "_ref", "perfstats_probe")
    data = {"_source": "oracle/_ref/perfstats_probe = tests/cpp/perfstats_probe.cpp + the reference's kfusion/thirdparty/perfstats.h (g++ -O3 -std=gnu++11)",
            "columns": [c[0] for c in COLUMNS], "cases": {}}
    for name, rows in CASES.items():
        data["cases"][name] = {"rows": rows, "summary_log_on_stdout": run(probe, rows, False), "summary_log_in_file": run(probe, rows, True)}
    with open(os.path.join(HERE, "perfstats_vectors.json"), "w") as f:
        json.dump(data, f, indent=0)
    print({k: len(v["rows"]) for k, v in data["cases"].items()})
    print("\n".join(data["cases"]["twelve_frames"]["summary_log_in_file"][:3]))
    print("\n".join(data["cases"]["twelve_frames"]["summary_log_on_stdout"][:3]))
    print("\n".join(data["cases"]["one_frame"]["summary_log_on_stdout"][:3]))
