"""ICL-NUIM scene directory -> SLAMBench ``.raw`` stream (python-3 twin of kfusion/thirdparty/scene2raw.cpp).

``scene_00_%04d.depth`` holds 640x480 ray lengths in metres as text; the reference turns them into z-depth in
millimetres with the ICL-NUIM intrinsics (scene2raw.cpp:76-113) and writes, per frame,
``uint32 W,H | depth u16 | uint32 W,H | rgb u8x3`` (scene2raw.cpp:170-176).  Arithmetic as the reference compiles it:
``(u - _u0) / _focal_x`` is a FLOAT division (both constants are ``static const float``, scene2raw.cpp:33-36) widened to double
afterwards (:97-98); the squares, the sum, the square root, ``d * 1000`` and the final division are double; the store truncates
to ushort.  Held byte for byte to the reference's compiled tool by tests/golden/host_vectors.json.  The colour image is decoded
from ``scene_00_%04d.png`` (built-in decoder for 8-bit non-interlaced PNGs, PIL for anything else); when a PNG cannot be read
the previous frame's colours are written again, as the reference does (scene2raw.cpp:155-163) — KinectFusion never reads them.

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
    u, v = np.meshgrid(np.arange(WIDTH, dtype=np.float32), np.arange(HEIGHT, dtype=np.float32))
    a = ((u - np.float32(U0)) / np.float32(FX)).astype(np.float64)      # float subtraction and division, then widened
    b = ((v - np.float32(V0)) / np.float32(FY)).astype(np.float64)
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


def _decode_png(data: bytes) -> np.ndarray:
    """8-bit, non-interlaced PNG (grey, RGB, palette, with or without alpha) -> (H, W, 3) uint8."""
    import zlib
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, palette, hdr = 8, [], None, None
    while pos + 8 <= len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    if depth != 8 or interlace:
        raise ValueError("unsupported PNG layout")
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * channels)
    out = np.zeros((h, w * channels), np.uint8)
    zero = np.zeros(w * channels, np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = out[y - 1].astype(np.int32) if y else zero
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + up) & 255
        else:                                   # filters 1, 3, 4 depend on the pixel to the left: walk the row
            cur = np.zeros(w * channels, np.int32)
            for i in range(w * channels):
                a = int(cur[i - channels]) if i >= channels else 0
                b = int(up[i])
                c = int(up[i - channels]) if i >= channels else 0
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (int(line[i]) + pred) & 255
        out[y] = cur
    px = out.reshape(h, w, channels)
    if ctype == 3:
        return palette[px[..., 0]]
    if ctype in (0, 4):
        return np.repeat(px[..., :1], 3, axis=2)
    return np.ascontiguousarray(px[..., :3])


def read_rgb(path: str):
    """(H, W, 3) uint8, or None when the file cannot be decoded."""
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    try:
        return _decode_png(data)
    except Exception:
        pass
    try:
        from PIL import Image   # optional
        return np.asarray(Image.open(path).convert("RGB"), np.uint8)
    except Exception:
        return None


def convert(scene_dir: str, out_path: str) -> int:
    hdr = struct.pack("<II", WIDTH, HEIGHT)
    n = 0
    last_rgb = np.zeros((HEIGHT, WIDTH, 3), np.uint8)
    with open(out_path, "wb") as out:
        while True:
            depth_file = os.path.join(scene_dir, "scene_00_%04d.depth" % n)
            if not os.path.exists(depth_file):
                break
            d = read_depth_file(depth_file)
            rgb = read_rgb(os.path.join(scene_dir, "scene_00_%04d.png" % n))
            if rgb is not None and rgb.shape[0] * rgb.shape[1] >= WIDTH * HEIGHT:
                # the reference copies the first 640*480 decoded pixels, whatever the row length (scene2raw.cpp:159-161)
                last_rgb = np.ascontiguousarray(rgb.reshape(-1, 3)[:WIDTH * HEIGHT].reshape(HEIGHT, WIDTH, 3))
            out.write(hdr); out.write(d.astype("<u2").tobytes())
            out.write(hdr); out.write(last_rgb.tobytes())
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
