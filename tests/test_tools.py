"""Host-side tooling: synthetic stream generator, .raw wire format, freiburg ground truth, ATE tool, CLI parsing."""
import os
import subprocess

import numpy as np

from slambench_amd import checkpos, rawio, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_is_seeded_and_plausible():
    a = synth.make_stream(3, 80, 60)
    b = synth.make_stream(3, 80, 60)
    c = synth.make_stream(3, 80, 60, variant=1)
    assert np.array_equal(a.depth_mm, b.depth_mm) and not np.array_equal(a.depth_mm[1:], c.depth_mm[1:])
    assert a.depth_mm.dtype == np.uint16 and a.depth_mm.min() > 400 and a.depth_mm.max() < 4000   # inside [nearPlane, farPlane]
    assert np.allclose(a.poses[0], np.eye(4) + np.pad(a.init_pos[:, None], ((0, 1), (3, 0))))      # identity rotation at -p
    step = np.linalg.norm(a.poses[1][:3, 3] - a.poses[0][:3, 3])
    assert 0.002 < step < 0.008
    assert synth.camera_for(1280) == tuple(2 * v for v in synth.BASE_K)


def test_raw_round_trip(tmp_path):
    s = synth.make_stream(3, 80, 60)
    path = str(tmp_path / "s.raw")
    synth.write_raw(path, s)
    assert os.path.getsize(path) == 3 * (16 + 80 * 60 * 2 + 80 * 60 * 3)        # interface.h:244-250
    r = rawio.RawDepthReader(path)
    assert r.get_input_size() == (80, 60)
    frames = list(r)
    assert len(frames) == 3 and all(np.array_equal(f, s.depth_mm[i]) for i, f in enumerate(frames))
    r.close()


def _fake_log(positions, tracked):
    head = "frame\tacquisition\tpreprocessing\ttracking\tintegration\traycasting\trendering\tcomputation\ttotal    \tX          \tY          \tZ         \ttracked   \tintegrated"
    lines = [head]
    for i, (p, t) in enumerate(zip(positions, tracked)):
        lines.append("%d\t0.000100\t0.000200\t0.000300\t0.000400\t0.000500\t0.000600\t0.001400\t0.002100\t%.6f\t%.6f\t%.6f\t%d        \t1" % (i, p[0], p[1], p[2], t))
    return "\n".join(lines) + "\n"


def test_checkpos_matches_the_reference_conventions(tmp_path):
    s = synth.make_stream(6, 40, 30)
    gt_path = str(tmp_path / "gt.freiburg")
    synth.write_freiburg(gt_path, s)
    gt = checkpos.parse_freiburg(open(gt_path).read())
    assert len(gt) == 6 and gt[0] == (0.0, 0.0, 0.0)
    disp = s.poses[:, :3, 3] - s.init_pos
    # a perfect tracker logs the displacement; Y is negated inside the tool (checkPos.py:104)
    rep = checkpos.summarise(_fake_log(disp, [0, 0, 0, 0, 1, 1]), open(gt_path).read())
    assert rep["ATE"]["max"] < 1e-6 and rep["_meta"]["untracked"] == 0 and rep["_meta"]["valid"] == 6
    assert abs(rep["computation"]["mean"] - 0.0014) < 1e-9 and abs(rep["total"]["total"] - 6 * 0.0021) < 1e-9
    # a 1 cm offset in X shows up as exactly that
    off = disp + np.array([0.01, 0, 0])
    rep = checkpos.summarise(_fake_log(off, [1] * 6), open(gt_path).read())
    assert abs(rep["ATE"]["mean"] - 0.01) < 1e-6 and rep["_meta"]["untracked"] == -4   # the reference starts its count at -4 (checkPos.py:34)
    # dropped frames are padded with the last valid row (checkPos.py:53-57)
    text = _fake_log(disp, [1] * 6).split("\n")
    del text[3]
    log = checkpos.parse_log("\n".join(text))
    assert log["dropped"] == 1 and len(log["traj"]) == 6


def test_benchmark_cli_rejects_bad_flags():
    """Argument validation happens before any GPU call, so the driver's exit codes can be checked on CPU
    (default_parameters.h:270-283, 341-390, 412-417)."""
    from slambench_amd import build
    exe = build.build_benchmark() or build.BENCH_BIN
    if not os.path.exists(exe):
        build.build_all()
    run = lambda *a: subprocess.run([exe] + list(a), capture_output=True, text=True, timeout=60)
    r = run("-c", "3")
    assert r.returncode == 1 and "must be 1, 2 ,4 or 8" in r.stderr
    r = run("-r", "0")
    assert r.returncode == 1 and "--integration-rate" in r.stderr
    r = run("-i", "/nonexistent/file.raw")
    assert r.returncode == 1 and "does not exist" in r.stderr
    r = run("-s", "0")
    assert r.returncode == 1
    r = run("-x")
    assert r.returncode == 0 and "Unknown option" in r.stderr       # the reference exits 0 here
    r = run()
    assert r.returncode == 1 and "No input found." in r.stderr
