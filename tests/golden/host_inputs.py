"""Seeded inputs and the case list shared by make_host_vectors.py (which runs the REFERENCE builds on them, in the build
container) and tests/test_host_headers.py (which runs this repo's headers and python twins on the very same bytes).
Nothing here comes from the reference: it is input data only."""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

W, H = 640, 480
SCENE_FRAMES = 3

# Configuration cases: argv after the program name.  "@raw" is replaced by an existing file name (relative to the cwd).
CONFIG_CASES = [
    [],
    # the four parameter lines of the reference Makefile (Makefile:3-6) as benchmark arguments
    "-s 5.0 -p 0.34,0.5,0.24 -z 4 -c 2 -r 2 -k 481.2,480,320,240".split(),
    "-s 5.0 -p 0.485,0.5,0.55 -z 4 -c 2 -r 2 -k 481.2,480,320,240".split(),
    "-s 4.8 -p 0.34,0.5,0.24 -z 4 -c 2 -r 2 -k 481.2,480,320,240".split(),
    "-s 5.0 -p 0.2685,0.5,0.4 -z 4 -c 2 -r 2 -k 481.2,480,320,240".split(),
    # BASELINE configs 0..3
    "-i @raw -c 2 -v 128,128,128 -r 1".split(),
    "-i @raw -c 1 -v 256,256,256 -y 10,5,4".split(),
    "-i @raw -v 512,512,512 -r 1 -z 1".split(),
    "-i @raw -v 512,512,512 -y 10,5,4,4".split(),
    "--compute-size-ratio 4 --dump-volume vol.bin --fps 30 --camera 500,501,300,200 --icp-threshold 1e-6 --log-file x.log --mu 0.2 "
    "--init-pose 0.1,0.2,0.3 --no-gui --integration-rate 3 --volume-size 1,2,3 --tracking-rate 2 --volume-resolution 64,32,16 "
    "--pyramid-levels 8,4 --rendering-rate 7".split(),
    ["-v", "64"], ["-v", "64,32"], ["-v", "64,32,16,8"], ["-s", "1.5"], ["-s", "1.5,2.5"], ["-p", "0.25"], ["-p", "0.25,0.75"],
    ["-k", "500"], ["-k", "500,400"], ["-k", "500,400,300"], ["-k", "500,400,300,200,100"],
    ["-y", "3"], ["-y", "10,5,4,4"], ["-y", "2.9,1.2"], ["-y", ""], ["-t", "2.7"], ["-z", "3.9"], ["-l", "0.5e-3", "-m", "0.075"],
    ["-c", "8"], ["-q", "-q"], ["-d", "a", "-d", "b"], ["-v", "abc"], ["-s", "1e-3"], ["-f", "0"],
    # error exits (default_parameters.h:270-283, 341-390)
    ["-c", "3"], ["-c", "0"], ["-r", "0"], ["-r", "-2"], ["-f", "-1"], ["-s", "-1"], ["-s", "1,0,1"], ["-s", "0"], ["-v", "0"],
    ["-v", "8,0,8"], ["-i", "no/such/file"], ["-c", "3", "-r", "0", "-f", "-5"], ["-c", "5", "-v", "0"],
    ["--bogus"], ["-x"], ["-c"], ["-b"], ["-q", "stray", "-c", "2"],
]


def write_png_rgb(path: str, rgb: np.ndarray) -> None:
    """Minimal 8-bit RGB PNG (filter 0 on every row)."""
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6))
                + chunk(b"IEND", b""))


def make_scene_dir(directory: str, frames: int = SCENE_FRAMES, seed: int = 20261004) -> None:
    """ICL-NUIM style scene directory: scene_00_%04d.depth (ray lengths in metres as text) + scene_00_%04d.png."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    for i in range(frames):
        # ramps with steps, in the ICL-NUIM range (integer arithmetic and IEEE basic operations only: the same text on every host)
        rays = 0.6 + ((u * 7 + v * 13 + i * 101) % 1000) / 1000.0 + 1.7 * (((u // 80) + (v // 60)) % 3) + rng.uniform(0, 0.01, size=(H, W))
        rays[rng.uniform(size=(H, W)) < 0.01] = 0.0
        fmt = "%.6f" if i != 1 else "%.9g"
        with open(os.path.join(directory, "scene_00_%04d.depth" % i), "w") as f:
            f.write(" ".join(fmt % x for x in rays.reshape(-1)))
            f.write("\n" if i == 2 else " ")
        write_png_rgb(os.path.join(directory, "scene_00_%04d.png" % i), rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8))


def make_raw_file(path: str, frames: int = 4, width: int = 64, height: int = 48, seed: int = 77, garbage: int = 0) -> None:
    """A `.raw` stream (interface.h:244-283 wire format) of random depth and colour, optionally with trailing garbage."""
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for _ in range(frames):
            f.write(struct.pack("<II", width, height))
            f.write(rng.integers(0, 6000, size=width * height, dtype=np.uint16).astype("<u2").tobytes())
            f.write(struct.pack("<II", width, height))
            f.write(rng.integers(0, 256, size=width * height * 3, dtype=np.uint8).tobytes())
        f.write(bytes(range(garbage)))


def make_inputs(root: str) -> None:
    make_scene_dir(os.path.join(root, "scene"))
    make_raw_file(os.path.join(root, "in.raw"))
    make_raw_file(os.path.join(root, "tail.raw"), frames=2, seed=78, garbage=100)
    make_raw_file(os.path.join(root, "vga.raw"), frames=1, width=W, height=H, seed=79)
