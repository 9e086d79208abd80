"""rocprofv3 kernel trace -> ``<kernelName> <nanoseconds>`` lines, the input format of checkkernels (the role
thirdparty/nvprof2log.py plays for nvprof output).

Reads the ``*_kernel_trace.csv`` that ``rocprofv3 --kernel-trace --output-format csv`` writes (columns Kernel_Name,
Start_Timestamp, End_Timestamp in ns) and prints one line per launch under the reference's kernel names, so that a HIP
profile lines up with KERNEL_TIMINGS logs of the CPP / OpenMP / OpenCL / CUDA backends:

    rocprofv3 --kernel-trace --output-format csv -d out -- slambench_amd/kfusion-benchmark-hip ...
    python -m slambench_amd.rocprof2log out/*/*_kernel_trace.csv | python -m slambench_amd.checkkernels -

Fused launches are reported under the reference kernel that dominates them (the same convention as KERNEL_TIMINGS in
include/kfusion_hip.h): mm2meters + bilateral as bilateralFilterKernel, the fused pyramid as pyramidKernel, every launch of
the ICP loop as trackKernel.
"""
from __future__ import annotations

import csv
import re
import sys

NAMES = [
    (r"k_bilateral", "bilateralFilterKernel"),
    (r"k_mm2meters", "mm2metersKernel"),
    (r"k_half_sample", "halfSampleRobustImageKernel"),
    (r"k_depth2vertex", "depth2vertexKernel"),
    (r"k_vertex2normal|k_vertex_normal", "vertex2normalKernel"),
    (r"k_pyramid", "pyramidKernel"),
    (r"k_icp_iter|k_icp_tail|k_icp_final|k_track_reduce|k_control", "trackKernel"),
    (r"k_reduce_trackdata|k_solve", "reduceKernel"),
    (r"k_update_pose|k_check_pose", "updatePoseKernel"),
    (r"k_integrate", "integrateKernel"),
    (r"k_raycast", "raycastKernel"),
    (r"k_render_depth", "renderDepthKernel"),
    (r"k_render_track", "renderTrackKernel"),
    (r"k_render_volume", "renderVolumeKernel"),
    (r"k_init_volume", "initVolumeKernel"),
]


def translate(kernel_name: str):
    short = kernel_name.split("(")[0]
    for pat, ref in NAMES:
        if re.search(r"\b(?:%s)" % pat, short):
            return ref
    return None


def convert(rows):
    """rows: dicts with Kernel_Name / Start_Timestamp / End_Timestamp -> list of (reference name, ns) in start order."""
    out = []
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        name = translate(r["Kernel_Name"])
        if name is not None:
            out.append((name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        print("usage: python -m slambench_amd.rocprof2log <kernel_trace.csv> [...]", file=sys.stderr)
        return 1
    for path in argv:
        with open(path) as f:
            for name, ns in convert(csv.DictReader(f)):
                print("%s %d" % (name, ns))
    return 0


if __name__ == "__main__":
    sys.exit(main())
