"""Absolute-trajectory-error and timing summary of a kfusion benchmark log.

Python-3 restatement of the reference's harness ``kfusion/thirdparty/checkPos.py`` (Python 2,
not runnable in this image): same log and ground-truth parsing (checkPos.py:13-19,46-84), same
padding of dropped frames with the last valid row (53-57), same alignment
``est = (X + gt0.x, -(Y + gt0.y), Z + gt0.z)`` with Y negated (line 104), per-frame Euclidean
error (105-106) and min/max/mean/total per column (144-155).

Row *n* of the log holds the pose *before* frame *n* is processed (benchmark.cpp:127-131).
"""
from __future__ import annotations

import math
import re
import sys

KFUSION_LOG_REGEX = r"([0-9]+[\s]*)\t" + 8 * r"([0-9.]+)\t" + 3 * r"([-0-9.]+)\t" + r"([01])\s+([01])"
NUIM_LOG_REGEX = r"([0-9]+)" + 7 * r"\s+([-0-9e.]+)\s*"


def parse_log(text: str):
    lines = text.split("\n")
    headers = lines[0].split("\t")
    if len(headers) == 15 and headers[14] == "":
        del headers[14]
    if len(headers) != 14:
        raise ValueError("Wrong KFusion log  file. Expected 14 columns but found %d" % len(headers))
    headers = [h.strip() for h in headers]
    full = {h: [] for h in headers}
    traj, dropped_total, valid, last_frame, untracked = [], 0, 0, -1, -4
    last_valid = None
    for line in lines[1:]:
        m = re.match(KFUSION_LOG_REGEX, line)
        if not m:
            break
        dropped = int(m.group(1)) - last_frame - 1
        if dropped > 0:
            dropped_total += dropped
            traj.extend([last_valid] * dropped)
        traj.append((m.group(10), m.group(11), m.group(12), m.group(13), 1))
        last_valid = (m.group(10), m.group(11), m.group(12), m.group(13), 0)
        if int(m.group(13)) == 0:
            untracked += 1
        valid += 1
        for idx, h in enumerate(headers):
            full[h].append(float(m.group(idx + 1)))
        last_frame = int(m.group(1))
    return {"columns": full, "traj": traj, "dropped": dropped_total, "valid": valid, "untracked": untracked}


def parse_freiburg(text: str):
    out = []
    for line in text.split("\n"):
        m = re.match(NUIM_LOG_REGEX, line)
        if not m:
            break
        out.append((float(m.group(2)), float(m.group(3)), float(m.group(4))))
    return out


def ate(traj_xyz, gt_xyz):
    """traj_xyz: iterable of logged (X, Y, Z); gt_xyz: ground-truth rows.  Returns the per-frame ATE list."""
    n = min(len(traj_xyz), len(gt_xyz))
    first = gt_xyz[0]
    errs = []
    for p in range(n):
        x, y, z = (float(v) for v in traj_xyz[p][:3])
        e = (x + first[0], -(y + first[1]), z + first[2])
        d = [abs(e[i] - gt_xyz[p][i]) for i in range(3)]
        errs.append(math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
    return errs


def stats(v):
    return {"min": min(v), "max": max(v), "mean": sum(v) / len(v), "total": sum(v)}


def summarise(log_text: str, gt_text: str):
    log = parse_log(log_text)
    gt = parse_freiburg(gt_text)
    errs = ate(log["traj"], gt)
    cols = dict(log["columns"])
    cols["ATE"] = errs
    # ATE_wrt_kfusion leaves out the frames the reader dropped (their rows repeat the last valid position with valid = 0);
    # it is reported only when frames were dropped (checkPos.py:99,127-128,147-148)
    if log["dropped"] > 0:
        cols["ATE_wrt_kfusion"] = [e for e, t in zip(errs, log["traj"]) if int(t[4]) == 1]
    report = {}
    for name in sorted(cols):
        if any(t in name for t in ("X", "Y", "Z", "frame", "tracked", "integrated")):
            continue
        report[name] = stats(cols[name])
    report["_meta"] = {"valid": log["valid"], "dropped": log["dropped"], "untracked": log["untracked"],
                       "positions": min(len(log["traj"]), len(gt))}
    return report


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) != 3:
        print("I need two parameters, the benchmark log file and the original scene camera position file.")
        return 1
    with open(argv[1]) as f:
        log_text = f.read()
    with open(argv[2]) as f:
        gt_text = f.read()
    rep = summarise(log_text, gt_text)
    meta = rep.pop("_meta")
    print("KFusion valid frames %d,  dropped frames: %d" % (meta["valid"], meta["dropped"]))
    print("Working position is : %d" % meta["positions"])
    print("Untracked frames: %d" % meta["untracked"])
    print("Runtimes are in seconds and the absolute trajectory error (ATE) is in meters.")
    for name in sorted(rep):
        s = rep[name]
        print("%20.20s\tMin : %6.6f\tMax : %0.6f\tMean : %0.6f\tTotal : %0.8f" % (name, s["min"], s["max"], s["mean"], s["total"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
