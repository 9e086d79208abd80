"""In-tree build of the HIP backend: ``python -m slambench_amd.build``.

Compiles every HIP source for gfx950 with hipcc into ``slambench_amd/libkfusion_hip.so`` (the C ABI
of ``include/kfusion_hip.h``) and the ``kfusion-benchmark-hip`` driver.  Flags that matter for
parity with the CPU reference: no FMA contraction, correctly rounded fp32
divide/sqrt, no fast-math.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkfusion_hip.so")
BENCH_BIN = os.path.join(HERE, "kfusion-benchmark-hip")

HIP_SOURCES = ["kf_kernels.hip", "kf_pipeline.hip"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wno-unused-result", "-Wno-unused-value"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP backend cannot be built")
    return exe


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("kf_math.h", "kf_kernels.h")] + [os.path.join(ROOT, "include", "kfusion_hip.h")]
    if not force and _newer(LIB, deps):
        return LIB
    extra = os.environ.get("KFH_EXTRA_HIPCC_FLAGS", "").split()     # experiments only (e.g. -DKFH_ICP_WAVES_PER_SIMD=8)
    cmd = [hipcc()] + HIP_FLAGS + extra + ["-shared", "-I", os.path.join(ROOT, "include")] + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_benchmark(force: bool = False, verbose: bool = False) -> str | None:
    src = os.path.join(CSRC, "benchmark.cpp")
    if not os.path.exists(src):
        return None
    deps = [src, LIB, os.path.join(ROOT, "include", "kfusion", "kernels.h"), os.path.join(ROOT, "include", "kfusion", "default_parameters.h"),
            os.path.join(ROOT, "include", "kfusion", "interface.h")]
    deps = [d for d in deps if os.path.exists(d)]
    if not force and _newer(BENCH_BIN, deps):
        return BENCH_BIN
    cmd = ["g++", "-O3", "-std=gnu++11", "-I", os.path.join(ROOT, "include"), src, "-o", BENCH_BIN,
           "-L", HERE, "-lkfusion_hip", "-Wl,-rpath,$ORIGIN", "-lrt", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return BENCH_BIN


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_library(force, verbose)
    build_benchmark(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
