#!/usr/bin/env python3
"""Regenerates tests/golden/host_vectors.json from the REFERENCE's own code, compiled where it lies (build container only).

`make -C oracle ref` builds, without stand-ins of any kind,
  oracle/_ref/host_probe   tests/cpp/host_probe.cpp against /root/reference/kfusion/include/{default_parameters,interface}.h
  oracle/_ref/scene2raw    /root/reference/kfusion/thirdparty/scene2raw.cpp + lodepng.cpp
This script runs them on the seeded inputs of host_inputs.py and stores what they produced: stdout / stderr / exit code of
every Configuration case, and for the readers and scene2raw the SHA-256 of the output plus sampled values.  The reference
cannot travel; this file of expected outputs can.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import host_inputs  # noqa: E402

READ_CASES = [("in.raw", "u16"), ("in.raw", "f32"), ("in.raw", "rgb"), ("tail.raw", "u16"), ("vga.raw", "f32"),
              ("scene", "u16"), ("scene", "f32"), ("scene", "rgb"), ("scene.raw", "u16")]


def sample_stride(n):       # primes: samples walk through every column and row
    return 9973 if n > 200000 else 97


def run_config(probe, args, cwd):
    args = ["in.raw" if a == "@raw" else a for a in args]
    r = subprocess.run([probe, "config"] + args, cwd=cwd, capture_output=True, timeout=60)
    out = r.stdout.decode()
    if "---values---\n" in out:                       # drop the wall-clock line of print_values
        head, tail = out.split("---values---\n", 1)
        out = head + "---values---\n" + "\n".join(tail.split("\n")[1:])
    return {"args": args, "returncode": r.returncode, "stdout": out, "stderr": r.stderr.decode()}


def digest(path, dtype):
    data = open(path, "rb").read()
    a = np.frombuffer(data, dtype=dtype)
    return {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(), "samples": a[::sample_stride(a.size)].astype(np.uint32 if dtype != "<u1" else np.uint8).tolist()[:400]}


def run_read(probe, name, what, cwd):
    out = os.path.join(cwd, "out.bin")
    r = subprocess.run([probe, "read", name, what, out], cwd=cwd, capture_output=True, timeout=300)
    meta = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
    d = digest(out, {"u16": "<u2", "f32": "<u4", "rgb": "<u1"}[what])
    said_end = "End of file" in r.stdout.decode()
    return {"path": name, "what": what, "returncode": r.returncode, "meta": meta, "announced_end": said_end, "garbage": "garbage" in r.stdout.decode(), **d}


def main():
    ref = os.path.join(ROOT, "oracle", "_ref")
    probe, s2r = os.path.join(ref, "host_probe"), os.path.join(ref, "scene2raw")
    if not (os.path.exists(probe) and os.path.exists(s2r)):
        sys.exit("run `make -C oracle ref` first (needs /root/reference)")
    with tempfile.TemporaryDirectory() as tmp:
        host_inputs.make_inputs(tmp)
        vec = {"generator": "tests/golden/make_host_vectors.py", "config": [run_config(probe, a, tmp) for a in host_inputs.CONFIG_CASES]}
        r = subprocess.run([s2r, "scene", "scene.raw"], cwd=tmp, capture_output=True, timeout=600)
        vec["scene2raw"] = {"returncode": r.returncode, **digest(os.path.join(tmp, "scene.raw"), "<u1")}
        bad = subprocess.run([s2r, "scene"], cwd=tmp, capture_output=True, timeout=60)
        vec["scene2raw_usage"] = {"returncode": bad.returncode, "stdout": bad.stdout.decode()}
        vec["read"] = [run_read(probe, n, w, tmp) for n, w in READ_CASES]
    with open(os.path.join(HERE, "host_vectors.json"), "w") as f:
        json.dump(vec, f, indent=0, sort_keys=True)
    print("wrote host_vectors.json: %d config cases, %d reader cases" % (len(vec["config"]), len(vec["read"])))


if __name__ == "__main__":
    main()
