"""ICL-NUIM living-room trajectories: the per-trajectory parameter lines of the reference Makefile (Makefile:3-6)
and the file names its data-set rules produce (Makefile:33-44), so real data runs the moment it is present.

    0 = -s 5.0 -p 0.34,0.5,0.24    -z 4 -c 2 -r 2 -k 481.2,480,320,240
    1 = -s 5.0 -p 0.485,0.5,0.55   -z 4 -c 2 -r 2 -k 481.2,480,320,240
    2 = -s 4.8 -p 0.34,0.5,0.24    -z 4 -c 2 -r 2 -k 481.2,480,320,240
    3 = -s 5.0 -p 0.2685,0.5,0.4   -z 4 -c 2 -r 2 -k 481.2,480,320,240
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Trajectory:
    index: int
    volume_size: float          # -s (metres, cubic)
    init_pose: tuple            # -p (factors of the volume size)
    rendering_rate: int = 4     # -z
    compute_size_ratio: int = 2  # -c
    integration_rate: int = 2   # -r
    camera: tuple = (481.2, 480.0, 320.0, 240.0)   # -k (fx, fy, cx, cy at the input resolution)

    def cli(self) -> list:
        """The argument list the reference Makefile passes (`$($(*F))`, Makefile:53,59,65,71)."""
        return ["-s", "%.1f" % self.volume_size, "-p", ",".join("%g" % v for v in self.init_pose), "-z", str(self.rendering_rate),
                "-c", str(self.compute_size_ratio), "-r", str(self.integration_rate), "-k", ",".join("%g" % v for v in self.camera)]

    def init_position(self) -> np.ndarray:
        """benchmark.cpp:86 — `initial_pos_factor * volume_size`, formed in float32."""
        return (np.array(self.init_pose, np.float32) * np.float32(self.volume_size)).astype(np.float32)

    def raw_name(self) -> str:
        return "living_room_traj%d_loop.raw" % self.index

    def scene_dir_name(self) -> str:
        return "living_room_traj%d_loop" % self.index

    def gt_name(self) -> str:
        return "livingRoom%d.gt.freiburg" % self.index


TRAJECTORIES = {
    0: Trajectory(0, 5.0, (0.34, 0.5, 0.24)),
    1: Trajectory(1, 5.0, (0.485, 0.5, 0.55)),
    2: Trajectory(2, 4.8, (0.34, 0.5, 0.24)),
    3: Trajectory(3, 5.0, (0.2685, 0.5, 0.4)),
}


def guess_index(path: str, default: int = 2) -> int:
    """Trajectory number from a file name such as living_room_traj2_loop.raw or livingRoom2.gt.freiburg."""
    m = re.search(r"(?:traj|livingRoom)(\d)", os.path.basename(path))
    return int(m.group(1)) if m and int(m.group(1)) in TRAJECTORIES else default


def find_dataset(directory: str | None, index: int):
    """(raw_or_scene_path, gt_path or None) for trajectory `index` under `directory` (default $KFH_ICL_NUIM_DIR); None when
    the data is not there.  A `.raw` wins over the scene directory it was converted from."""
    directory = directory or os.environ.get("KFH_ICL_NUIM_DIR")
    if not directory or not os.path.isdir(directory):
        return None
    t = TRAJECTORIES[index]
    gt = os.path.join(directory, t.gt_name())
    gt = gt if os.path.isfile(gt) else None
    for cand in (os.path.join(directory, t.raw_name()), os.path.join(directory, t.scene_dir_name())):
        if os.path.exists(cand):
            return cand, gt
    return None
