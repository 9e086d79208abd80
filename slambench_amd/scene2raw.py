"""ICL-NUIM scene directory -> SLAMBench ``.raw`` stream (python-3 twin of kfusion/thirdparty/scene2raw.cpp).

``scene_00_%04d.depth`` holds 640x480 ray lengths in metres as text; the reference turns them into z-depth in
millimetres with the ICL-NUIM intrinsics (scene2raw.cpp:76-113: double arithmetic, truncation to ushort) and writes, per
frame, ``uint32 W,H | depth u16 | uint32 W,H | rgb u8x3`` (scene2raw.cpp:170-176).  The colour image is copied from
``scene_00_%04d.png`` when a PNG decoder (PIL) is importable and left black otherwise — KinectFusion never reads it.

    python -m slambench_amd.scene2raw living_room_traj2_loop living_room_traj2_loop.raw
"""
from __future__ import annotations

import os
import struct
import sys

import numpy as np

WIDTH, HEIGHT = 640, 480
U0, V0, FX, FY = 319.50, 239.50, 481.20, -480.00   # scene2raw.cpp:24-33 (float constants)


def _ray_to_z_scale() -> np.ndarray:
    u, v = np.meshgrid(np.arange(WIDTH, dtype=np.float64), np.arange(HEIGHT, dtype=np.float64))
    a = (u - np.float64(np.float32(U0))) / np.float64(np.float32(FX))
    b = (v - np.float64(np.float32(V0))) / np.float64(np.float32(FY))
    return np.sqrt(a * a + b * b + 1)


def read_depth_file(path: str) -> np.ndarray:
    """One ``.depth`` text file -> (480, 640) uint16 z-depth in millimetres (scene2raw.cpp:76-113)."""
    with open(path) as f:
        vals = np.array(f.read().split(), dtype=np.float64)
    buf = np.zeros(WIDTH * HEIGHT, np.float64)
    n = min(vals.size, buf.size)
    buf[:n] = vals[:n] * 1000
    z = buf.reshape(HEIGHT, WIDTH) / _ray_to_z_scale()
    return z.astype(np.uint16)          # C conversion double -> ushort truncates


def read_rgb(path: str) -> np.ndarray:
    try:
        from PIL import Image   # optional
        return np.asarray(Image.open(path).convert("RGB"), np.uint8)
    except Exception:
        return np.zeros((HEIGHT, WIDTH, 3), np.uint8)


def convert(scene_dir: str, out_path: str) -> int:
    hdr = struct.pack("<II", WIDTH, HEIGHT)
    n = 0
    with open(out_path, "wb") as out:
        while True:
            depth_file = os.path.join(scene_dir, "scene_00_%04d.depth" % n)
            if not os.path.exists(depth_file):
                break
            d = read_depth_file(depth_file)
            rgb = read_rgb(os.path.join(scene_dir, "scene_00_%04d.png" % n))
            out.write(hdr); out.write(d.astype("<u2").tobytes())
            out.write(hdr); out.write(np.ascontiguousarray(rgb[:HEIGHT, :WIDTH]).tobytes())
            n += 1
    return n


def main(argv=None) -> int:
    argv = sys.argv if argv is None else argv
    if len(argv) != 3:
        print("Bad entries... I just need the scene directory and the output file")
        return 1
    print("Read %d frames" % convert(argv[1], argv[2]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
