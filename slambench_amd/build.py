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

HIP_SOURCES = ["kf_preprocess.hip", "kf_icp.hip", "kf_integrate.hip", "kf_raycast.hip", "kf_render.hip", "kf_launch.hip", "kf_pipeline.hip"]
HIP_HEADERS = ["kf_math.h", "kf_kernels.h", "kf_dev.h"]
OBJ_DIR = os.path.join(HERE, "csrc", "build")
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


def _compile(args):
    cmd, verbose = args
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """One object per translation unit (compiled in parallel, only the stale ones), then one link: a change to one kernel file
    recompiles that file."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(ROOT, "include", "kfusion_hip.h")]
    extra = os.environ.get("KFH_EXTRA_HIPCC_FLAGS", "").split()     # experiments only (e.g. -DKFH_ICP_WAVES_PER_SIMD=8)
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    flags_text = " ".join(HIP_FLAGS + extra)
    if not os.path.exists(stamp) or open(stamp).read() != flags_text:
        force = True            # other flags than the objects were built with
    objs, jobs = [], []
    for src in srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + hdrs):
            jobs.append(([hipcc()] + HIP_FLAGS + extra + ["-c", "-I", os.path.join(ROOT, "include"), src, "-o", obj], verbose))
    if not jobs and _newer(LIB, objs):
        return LIB
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(_compile, jobs))
    open(stamp, "w").write(flags_text)
    _compile(([hipcc()] + HIP_FLAGS + ["-shared"] + objs + ["-o", LIB], verbose))
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
