"""Per-kernel timing summary of a ``KERNEL_TIMINGS`` log (python-3 twin of kfusion/thirdparty/checkKernels.py).

Input: lines ``<kernelName> <nanoseconds> [size]`` as the backends print to stderr when ``KERNEL_TIMINGS`` is set
(cpp/kernels.cpp:31-35).  Names are normalised with the reference's table (checkKernels.py:16-32) so CPP / OpenCL /
CUDA / HIP logs line up; for each kernel: count, min, max, mean, total (ns).
"""
from __future__ import annotations

import re
import sys

KERNEL_CONSISTENCY = [
    ["mm2meters", "mm2metersKernel"],
    ["bilateral_filter", "bilateralFilterKernel"],
    ["halfSampleRobust", "halfSampleRobustImageKernel"],
    ["depth2vertex", "depth2vertexKernel"],
    ["vertex2normal", "vertex2normalKernel"],
    ["track", "trackKernel"],
    ["reduce", "reduceKernel"],
    ["integrate", "integrateKernel"],
    ["raycast", "raycastKernel"],
    ["renderDepth", "renderDepthKernel"],
    ["renderLight", "renderLightKernel"],
    ["renderTrack", "renderTrackKernel"],
    ["renderVolume", "renderVolumeKernel"],
    ["ResetVolume", "initVolumeKernel"],
    ["updatePose", "updatePoseKernel"],
]
LINE = re.compile(r"([^ ]+)\s([0-9.]+)")


def translate(name: str) -> str:
    for variations in KERNEL_CONSISTENCY:
        if name in variations:
            return variations[0]
    return name


def summarise(text: str) -> dict:
    data = {}
    for line in text.split("\n"):
        m = LINE.match(line)
        if m:
            data.setdefault(translate(m.group(1)), []).append(float(m.group(2)))
    return {k: {"count": len(v), "min": min(v), "max": max(v), "mean": sum(v) / len(v), "total": sum(v)} for k, v in data.items()}


def main(argv=None) -> int:
    argv = sys.argv if argv is None else argv
    if len(argv) != 2:
        print("I need only one parameter, the kernel log file.")
        return 1
    print("\nKernel-level statistics. Times are in nanoseconds.")
    rep = summarise(sys.stdin.read() if argv[1] == "-" else open(argv[1]).read())
    for name in sorted(rep):
        s = rep[name]
        print("%20.20s\tCount : %d\tMin   : %d\tMax   : %d\tMean  : %f\tTotal : %d" % (name, s["count"], s["min"], s["max"], s["mean"], s["total"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
