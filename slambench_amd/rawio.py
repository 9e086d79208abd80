"""Reader for SLAMBench ``.raw`` depth streams (python twin of RawDepthReader, interface.h:199-318)."""
from __future__ import annotations

import struct

import numpy as np


class RawDepthReader:
    """Iterates the uint16 millimetre depth frames of a ``.raw`` file; ``get_k`` mirrors interface.h:313-315."""

    def __init__(self, path: str):
        self.f = open(path, "rb")
        hdr = self.f.read(8)
        if len(hdr) != 8:
            raise ValueError("Invalid Raw file.")
        self.width, self.height = struct.unpack("<II", hdr)
        self.file_bytes = self.f.seek(0, 2)
        self.frame = -1
        self.frame_bytes = 16 + self.width * self.height * 2 + self.width * self.height * 3

    def get_input_size(self):
        return (self.width, self.height)

    def get_k(self):
        return (531.15, 531.15, 640 / 2, 480 / 2)

    def restart(self):
        self.frame = -1

    def read_next_depth_frame(self):
        """Returns an (H, W) uint16 array, or None at end of file."""
        self.frame += 1
        self.f.seek(self.frame_bytes * self.frame)
        hdr = self.f.read(8)
        if len(hdr) != 8:
            return None
        w, h = struct.unpack("<II", hdr)
        if w * h * 2 > self.file_bytes - self.f.tell():      # trailing garbage reads as a header: a short read ends the stream
            return None
        raw = self.f.read(w * h * 2)
        return np.frombuffer(raw, dtype="<u2").reshape(h, w).copy()

    def read_next_depth_frame_float(self):
        """Metres as float32: ``(float) mm / 1000.0f`` (interface.h:299-307)."""
        d = self.read_next_depth_frame()
        return None if d is None else d.astype(np.float32) / np.float32(1000.0)

    def __iter__(self):
        while True:
            d = self.read_next_depth_frame()
            if d is None:
                return
            yield d

    def close(self):
        self.f.close()


class SceneDepthReader:
    """ICL-NUIM scene directory reader (python twin of SceneDepthReader, interface.h:99-197): float32 arithmetic,
    ray length -> z-depth, then metres -> uint16 millimetres by truncation (interface.h:139-149)."""

    WIDTH, HEIGHT = 640, 480

    def __init__(self, directory: str):
        import os
        if not os.path.isdir(directory):
            raise ValueError("No such directory " + directory)
        self.dir = directory
        self.frame = -1
        u, v = np.meshgrid(np.arange(self.WIDTH, dtype=np.float32), np.arange(self.HEIGHT, dtype=np.float32))
        a = (u - np.float32(319.50)) / np.float32(481.20)
        b = (v - np.float32(239.50)) / np.float32(-480.00)
        self.scale = np.sqrt(a * a + b * b + np.float32(1)).astype(np.float32)

    def get_input_size(self):
        return (self.WIDTH, self.HEIGHT)

    def get_k(self):
        return (481.20, 480.00, 319.50, 239.50)

    def read_next_depth_frame(self):
        """uint16 millimetres: ``z * 1000.0f`` truncated (interface.h:139-149)."""
        z = self.read_next_depth_frame_float()
        return None if z is None else (z * np.float32(1000.0)).astype(np.uint16)

    def read_next_depth_frame_float(self):
        """z-depth in metres, float32 arithmetic throughout (interface.h:151-195)."""
        import os
        self.frame += 1
        path = os.path.join(self.dir, "scene_00_%04d.depth" % self.frame)
        if not os.path.exists(path):
            return None
        with open(path) as f:
            vals = np.array(f.read().split(), dtype=np.float32)
        buf = np.zeros(self.WIDTH * self.HEIGHT, np.float32)
        n = min(vals.size, buf.size)
        buf[:n] = vals[:n]
        return buf.reshape(self.HEIGHT, self.WIDTH) / self.scale
