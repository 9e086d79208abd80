"""Seeded synthetic RGB-D streams in the SLAMBench ``.raw`` wire format.

No ICL-NUIM data exists offline, so every configuration of BASELINE.json
runs on an analytic stand-in: the inside of an axis-aligned room with three spheres, seen by
a pinhole camera with the ICL-NUIM intrinsics the reference Makefile passes with ``-k``
(``481.2,480,320,240`` at 640x480; scaled with the image), z-depth in millimetres with
N(0, 1 mm) noise, and a smooth trajectory (about 4 mm and 0.2 degrees per frame).

Wire format per frame (reference ``kfusion/include/interface.h:244-283`` reads it,
``kfusion/thirdparty/scene2raw.cpp:170-176`` writes it)::

    uint32 W, H | W*H uint16 depth (mm) | uint32 W, H | W*H*3 uint8 rgb

Ground truth is written in the ``.gt.freiburg`` layout ``idx tx ty tz qx qy qz qw`` with the
ICL-NUIM sign convention that ``checkPos.py:104`` assumes (Y negated).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

# Room interior (metres) inside the 4.8 m cube that "-s 4.8" spans, and three spheres.
ROOM_LO = np.array([0.4, 0.9, 0.4])
ROOM_HI = np.array([4.4, 3.9, 4.4])
SPHERES = (
    (np.array([1.2, 2.9, 3.2]), 0.50),
    (np.array([2.6, 2.2, 3.6]), 0.60),
    (np.array([2.0, 3.3, 2.6]), 0.35),
)
VOLUME_SIZE = 4.8                      # -s 4.8          (reference Makefile:5, trajectory 2)
INIT_POS_FACTOR = (0.34, 0.5, 0.24)    # -p 0.34,0.5,0.24
BASE_K = (481.2, 480.0, 320.0, 240.0)  # -k 481.2,480,320,240 at 640x480


@dataclass
class Stream:
    width: int
    height: int
    k: tuple            # fx, fy, cx, cy for this resolution
    depth_mm: np.ndarray  # (n, H, W) uint16
    poses: np.ndarray     # (n, 4, 4) float64 camera-to-world
    init_pos: np.ndarray  # (3,) world position of frame 0


def camera_for(width: int) -> tuple:
    s = width / 640.0
    return tuple(v * s for v in BASE_K)


def _rot(yaw: float, pitch: float) -> np.ndarray:
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return ry @ rx


def trajectory(n: int, variant: int = 0) -> np.ndarray:
    """Camera-to-world poses; frame 0 is the identity rotation at the -p start point."""
    p0 = np.array(INIT_POS_FACTOR) * VOLUME_SIZE
    ph = 0.37 * variant
    poses = np.zeros((n, 4, 4))
    for i in range(n):
        a = 2 * np.pi * i
        yaw = 0.25 * (np.sin(a / 400 + ph) - np.sin(ph))
        pitch = 0.10 * (np.sin(a / 300 + ph) - np.sin(ph))
        d = np.array([0.25 * (np.sin(a / 400 + ph) - np.sin(ph)),
                      0.10 * (np.sin(a / 250 + ph) - np.sin(ph)),
                      0.20 * (1 - np.cos(a / 500))])
        poses[i, :3, :3] = _rot(yaw, pitch)
        poses[i, :3, 3] = p0 + d
        poses[i, 3, 3] = 1
    return poses


def render_depth(pose: np.ndarray, width: int, height: int, k: tuple) -> np.ndarray:
    """Exact z-depth (metres, float64) of the room + spheres along each pixel ray."""
    fx, fy, cx, cy = k
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dc = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)   # camera ray, z == 1
    d = dc @ pose[:3, :3].T
    o = pose[:3, 3]
    t = np.full(u.shape, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        for axis in range(3):
            for bound in (ROOM_LO[axis], ROOM_HI[axis]):
                tt = (bound - o[axis]) / d[..., axis]
                tt = np.where(tt > 1e-6, tt, np.inf)
                t = np.minimum(t, tt)
        for c, r in SPHERES:
            oc = o - c
            a = np.sum(d * d, axis=-1)
            b = 2 * (d @ oc)
            cc = oc @ oc - r * r
            disc = b * b - 4 * a * cc
            ts = (-b - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2 * a)
            ts = np.where((disc > 0) & (ts > 1e-6), ts, np.inf)
            t = np.minimum(t, ts)
    return t  # dc.z == 1, so the ray parameter is the camera-space z


def make_stream(n_frames: int, width: int = 640, height: int = 480, seed: int = 1234, variant: int = 0,
                noise_mm: float = 1.0) -> Stream:
    k = camera_for(width)
    poses = trajectory(n_frames, variant)
    rng = np.random.default_rng(seed + 7919 * variant)
    depth = np.zeros((n_frames, height, width), dtype=np.uint16)
    for i in range(n_frames):
        z = render_depth(poses[i], width, height, k) * 1000.0
        z = z + rng.normal(0.0, noise_mm, size=z.shape)
        z = np.where(np.isfinite(z), z, 0.0)
        depth[i] = np.clip(np.rint(z), 0, 65535).astype(np.uint16)
    return Stream(width, height, k, depth, poses, poses[0, :3, 3].copy())


def write_raw(path: str, stream: Stream) -> None:
    hdr = struct.pack("<II", stream.width, stream.height)
    with open(path, "wb") as f:
        for i in range(stream.depth_mm.shape[0]):
            d = stream.depth_mm[i]
            f.write(hdr)
            f.write(d.astype("<u2").tobytes())
            f.write(hdr)
            g = np.clip(d.astype(np.float64) / 4000.0 * 255.0, 0, 255).astype(np.uint8)
            f.write(np.repeat(g[..., None], 3, axis=-1).tobytes())


def _quat(R: np.ndarray) -> tuple:
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    if w < 1e-9:
        return (1.0, 0.0, 0.0, 0.0)
    return ((R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w)


def ground_truth_rows(stream: Stream) -> np.ndarray:
    """(n, 3) positions in the frame checkPos.py compares in: (X, -Y, Z) displacement from frame 0."""
    d = stream.poses[:, :3, 3] - stream.init_pos
    return np.stack([d[:, 0], -d[:, 1], d[:, 2]], axis=-1)


def write_freiburg(path: str, stream: Stream) -> None:
    gt = ground_truth_rows(stream)
    with open(path, "w") as f:
        for i in range(gt.shape[0]):
            qx, qy, qz, qw = _quat(stream.poses[i, :3, :3])
            f.write("%d %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % (i, gt[i, 0], gt[i, 1], gt[i, 2], qx, qy, qz, qw))


def cached_stream(n_frames: int, width: int = 640, height: int = 480, seed: int = 1234, variant: int = 0,
                  cache_dir: str | None = None) -> Stream:
    """make_stream with an .npz cache (generation costs ~40 ms per 640x480 frame)."""
    cache_dir = cache_dir or os.environ.get("KFUSION_SYNTH_CACHE", "/tmp/kfusion_synth")
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, "s%d_%dx%d_n%d_v%d.npz" % (seed, width, height, n_frames, variant))
    if os.path.exists(path):
        try:
            z = np.load(path)
            return Stream(width, height, camera_for(width), z["depth"], z["poses"], z["poses"][0, :3, 3].copy())
        except Exception:
            try:
                os.remove(path)          # unreadable cache entry; another rank may have removed it already
            except OSError:
                pass
    s = make_stream(n_frames, width, height, seed, variant)
    tmp = path + ".tmp%d.npz" % os.getpid()
    np.savez(tmp, depth=s.depth_mm, poses=s.poses)
    os.replace(tmp, path)
    return s


def write_scene_dir(directory: str, stream: Stream, seed: int = 4321) -> None:
    """The stream as an ICL-NUIM scene directory (what `SceneDepthReader`, interface.h:99-197, and `scene2raw` read): per frame
    ``scene_00_%04d.depth`` = 640x480 RAY LENGTHS in metres as text - the z-depth of the stream divided back through the ICL-NUIM
    conversion ``z = ray / sqrt(((u - 319.5) / 481.2)^2 + ((v - 239.5) / -480)^2 + 1)`` (interface.h:181-193) - and a grey
    ``scene_00_%04d.png`` (KinectFusion never reads the colours).  The stream's integer millimetres get a seeded fractional part
    first: ICL-NUIM's ray lengths are not quantised, and whole millimetres would put every pixel on the truncation boundary of
    the readers' float -> ushort conversion."""
    import zlib
    if (stream.width, stream.height) != (640, 480):
        raise ValueError("ICL-NUIM scene files are 640x480")
    os.makedirs(directory, exist_ok=True)
    u, v = np.meshgrid(np.arange(640, dtype=np.float64), np.arange(480, dtype=np.float64))
    scale = np.sqrt(((u - 319.5) / 481.2) ** 2 + ((v - 239.5) / -480.0) ** 2 + 1.0)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    rng = np.random.default_rng(seed)
    for i in range(stream.depth_mm.shape[0]):
        mm = stream.depth_mm[i].astype(np.float64)
        mm = np.where(mm > 0, mm + rng.uniform(0.0, 1.0, size=mm.shape), 0.0)
        rays = mm / 1000.0 * scale
        with open(os.path.join(directory, "scene_00_%04d.depth" % i), "w") as f:
            f.write(" ".join("%.6f" % x for x in rays.reshape(-1)))
            f.write(" ")
        g = np.clip(stream.depth_mm[i].astype(np.float64) / 4000.0 * 255.0, 0, 255).astype(np.uint8)
        rgb = np.repeat(g[..., None], 3, axis=-1)
        raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(480))
        with open(os.path.join(directory, "scene_00_%04d.png" % i), "wb") as f:
            f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 640, 480, 8, 2, 0, 0, 0))
                    + chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))
