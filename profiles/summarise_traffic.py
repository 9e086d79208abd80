#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of profiles/collect_traffic.sh.

Units and corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, i.e. reports exactly half the bytes of a wide coalesced read, so reads are doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Narrow accesses (the 8-byte gathers of the ray march,
4-byte voxel accesses of integrate) are not calibrated by the guide: the doubled figure is an upper bound for them
and the raw figure a lower bound; both are written out."""
import collections
import csv
import glob
import json
import sys


def per_kernel(dirname, counter):
    files = glob.glob(dirname + "/*/*_counter_collection.csv")
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc


def main(out, frames=None):
    """frames: frames of the profiled run (steps + warm-up of its one context), written into the summary: bench.py divides the ICP
    loop's bytes by it (without it by the launches of k_raycast, which is a few per cent high)"""
    fetch = per_kernel(out + "/fetch", "FETCH_SIZE")
    write = per_kernel(out + "/write", "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if "kf::" not in k:
            continue
        f = fetch.get(k, [0.0]); w = write.get(k, [0.0])
        # skipped ICP launches (level already converged) move nothing: average the launches that did work
        fa = [v for v in f if v > 0] or [0.0]
        wa = [v for v in w if v > 0] or [0.0]
        f_kib, w_kib = sum(fa) / len(fa), sum(wa) / len(wa)
        res[k.replace("void ", "")] = {
            "launches": len(f), "working_launches": len([v for v in f if v > 0]),
            "fetch_kib_raw": round(f_kib, 1), "write_kib": round(w_kib, 1),
            "hbm_bytes_per_launch_raw": round((f_kib + w_kib) * 1024),
            "hbm_bytes_per_launch": round((2 * f_kib + w_kib) * 1024),
            # over ALL launches of the run (the ICP loop is many launches of several kernels: bench.py adds these up
            # and divides by the frames of the run, i.e. the launches of k_raycast)
            "hbm_bytes_total": round((2 * sum(f) + sum(w)) * 1024),
        }
    if frames:
        res["frames"] = int(frames)
    json.dump(res, open(out + "/traffic.json", "w"), indent=1)
    for k, v in res.items():
        if not isinstance(v, dict):
            continue
        print("%-34s launches %5d  fetch %10.1f KiB (x2 corrected %10.1f)  write %10.1f KiB  -> %12d B/launch" %
              (k[:34], v["launches"], v["fetch_kib_raw"], 2 * v["fetch_kib_raw"], v["write_kib"], v["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
