"""SURVEY §8(f1) on the GPU: ICL-NUIM ingestion through the HIP driver.  benchmark.cpp:74-81 picks `SceneDepthReader` for a
directory and `RawDepthReader` for a file; interface.h:157-195 turns the scene files' ray lengths into z-depth in float
arithmetic, scene2raw.cpp:76-113 does it in double before writing the `.raw` - so the two inputs differ by 1 mm in a few dozen
pixels per frame BY THE REFERENCE'S OWN CODE (both conversions are held byte for byte to the reference's compiled reader and tool
by tests/test_host_headers.py).  Each path is therefore held to its own twin: the driver's log and `-d` dump must equal, bit for
bit, the python binding fed with the python twin of that reader, and follow the CPU oracle fed with the same frames."""
import os
import subprocess

import numpy as np
import pytest

from slambench_amd import build, checkpos, kfusion, rawio, scene2raw, synth, trajectories
from tests import util

pytestmark = pytest.mark.gpu

N = 10
# reference Makefile:5 (living_room_traj2): -s 4.8 -p 0.34,0.5,0.24 -z 4 -c 2 -r 2 -k 481.2,480,320,240
TRAJ2 = ["-s", "4.8", "-p", "0.34,0.5,0.24", "-z", "4", "-c", "2", "-r", "2", "-k", "481.2,480,320,240"]


def _drive(input_path, tmp_path, tag):
    log, vol = str(tmp_path / (tag + ".log")), str(tmp_path / (tag + ".vol"))
    r = subprocess.run([build.BENCH_BIN, "-i", input_path, "-o", log, "-d", vol, "-v", "128"] + TRAJ2, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return checkpos.parse_log(open(log).read()), np.fromfile(vol, np.int16), r.stderr


def _replay(frames, make, n):
    """the driver's frame loop (benchmark.cpp:125-158) on a backend object: rows of (position before the frame, tracked, integrated)"""
    t = trajectories.TRAJECTORIES[2]
    init = np.array(t.init_pose, np.float32) * np.float32(t.volume_size)        # float3 * float (benchmark.cpp:86)
    k = [float(np.float32(v) / np.float32(2)) for v in np.array(t.camera, np.float32)]
    g = make((320, 240), (128,) * 3, util.VOL_DIM, init)
    rows = []
    for f in range(n):
        pos = g.get_pose()[:3, 3] - init
        g.preprocessing(frames[f])
        tr = g.tracking(k, 1e-5, 1, f); it = g.integration(k, 2, 0.1, f); g.raycasting(k, 0.1, f)
        rows.append((pos, int(tr), int(it)))
    vol = g.buffer("volume")[..., 0].reshape(-1).copy()
    g.close()
    return rows, vol


def _check(parsed, dump, frames, orc):
    cols = parsed["columns"]
    assert len(parsed["traj"]) == N
    ours, vol = _replay(frames, kfusion.Kfusion, N)
    for f in range(N):
        # the same library on the same frames: the log's six decimals and both flags agree exactly, the dump bit for bit
        assert np.allclose(ours[f][0], (cols["X"][f], cols["Y"][f], cols["Z"][f]), atol=5.1e-7), f
        assert (ours[f][1], ours[f][2]) == (int(cols["tracked"][f]), int(cols["integrated"][f])), f
    assert np.array_equal(dump, vol)
    ref, ref_vol = _replay(frames, orc.OracleKfusion, N)
    for f in range(N):
        assert np.allclose(ref[f][0], ours[f][0], atol=1e-4), f
        assert ref[f][1:] == ours[f][1:], f
    off = np.abs(vol.astype(np.int32) - ref_vol.astype(np.int32)) > 3           # 1e-4 of the TSDF range is 3.3 of 32766
    assert off.mean() <= 1e-4, off.mean()
    assert [r[1] for r in ours] == [0, 0, 0, 0] + [1] * (N - 4)                  # the stream tracks from frame 4 on
    assert [r[2] for r in ours] == [1, 1, 1, 1] + [1 if f % 2 == 0 else 0 for f in range(4, N)]   # -r 2 (cpp:994)


def test_scene_directory_and_its_scene2raw_file_through_the_hip_driver(hip, orc, tmp_path):
    if not os.path.exists(build.BENCH_BIN):
        build.build_all()
    s = synth.make_stream(N, 640, 480)
    scene = str(tmp_path / "living_room_traj2_loop")
    synth.write_scene_dir(scene, s)
    raw = str(tmp_path / "living_room_traj2_loop.raw")
    assert scene2raw.convert(scene, raw) == N

    # -i <directory>: SceneDepthReader (benchmark.cpp:79-81)
    parsed_dir, dump_dir, err = _drive(scene, tmp_path, "dir")
    rd = rawio.SceneDepthReader(scene)
    scene_frames = [rd.read_next_depth_frame() for _ in range(N)]
    _check(parsed_dir, dump_dir, scene_frames, orc)
    assert "input Size is = 640,480" in err and "scene_00_%04d.depth" % N in err      # benchmark.cpp:88; the reader ends at the first missing file

    # -i <file>: the same scene converted by scene2raw (RawDepthReader, benchmark.cpp:74-78)
    parsed_raw, dump_raw, _ = _drive(raw, tmp_path, "raw")
    raw_frames = list(rawio.RawDepthReader(raw))
    assert len(raw_frames) == N
    _check(parsed_raw, dump_raw, raw_frames, orc)

    # the two ingestion paths see the same scene: identical verdicts, positions within a tenth of a millimetre; their depth
    # frames differ exactly as the reference's float reader and double tool differ (<= 1 mm, a handful of pixels)
    diff = [np.abs(a.astype(np.int32) - b.astype(np.int32)) for a, b in zip(scene_frames, raw_frames)]
    assert max(d.max() for d in diff) <= 1 and 0 < sum(int((d > 0).sum()) for d in diff) < 200 * N
    for name in ("tracked", "integrated"):
        assert parsed_dir["columns"][name] == parsed_raw["columns"][name]
    for name in ("X", "Y", "Z"):
        assert np.allclose(parsed_dir["columns"][name], parsed_raw["columns"][name], atol=1e-4)
    # and both follow the ground truth of the synthetic trajectory like the .raw written directly
    gt = str(tmp_path / "gt.freiburg")
    synth.write_freiburg(gt, s)
    rep = checkpos.summarise(open(str(tmp_path / "dir.log")).read(), open(gt).read())
    assert rep["ATE"]["max"] < 0.03
