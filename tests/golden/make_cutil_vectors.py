#!/usr/bin/env python3
"""Regenerates tests/golden/cutil_math_vectors.json from the REFERENCE's own helper headers.

`make -C oracle ref` compiles oracle/ref_probe.cpp against /root/reference/kfusion/thirdparty/
{vector_types.h,cutil_math.h} and kfusion/include/constant_parameters.h where they lie (the only part of the
hot path that builds in this image without TooN) into oracle/_ref/cutil_probe; its stdout is the fixture.
Only data is committed: the numeric results of the reference's functions on fixed inputs.
"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "cutil_probe")]).decode()
    data = json.loads(out)
    data["_source"] = "oracle/_ref/cutil_probe = oracle/ref_probe.cpp + the reference's kfusion/thirdparty/cutil_math.h, vector_types.h, include/constant_parameters.h (g++ -O3 -std=gnu++11)"
    with open(os.path.join(HERE, "cutil_math_vectors.json"), "w") as f:
        json.dump(data, f, separators=(",", ":"))
    print("wrote cutil_math_vectors.json:", {k: (len(v) if isinstance(v, list) else "...") for k, v in data.items()})
