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
        raw = self.f.read(w * h * 2)
        if len(raw) != w * h * 2:
            return None
        return np.frombuffer(raw, dtype="<u2").reshape(h, w).copy()

    def __iter__(self):
        while True:
            d = self.read_next_depth_frame()
            if d is None:
                return
            yield d

    def close(self):
        self.f.close()
