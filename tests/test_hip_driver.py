"""kfusion-benchmark-hip end to end on the GPU box: reference command line in, reference log format out."""
import os
import subprocess

import numpy as np
import pytest

from slambench_amd import build, checkkernels, checkpos, synth
from tests import util

pytestmark = pytest.mark.gpu


def test_driver_log_dump_and_kernel_timings(hip, tmp_path):
    exe = build.BENCH_BIN
    if not os.path.exists(exe):
        build.build_all()
    n = 12
    s = synth.make_stream(n, 160, 120)
    raw, gt, log, vol = (str(tmp_path / f) for f in ("s.raw", "s.gt.freiburg", "bench.log", "vol.bin"))
    synth.write_raw(raw, s)
    synth.write_freiburg(gt, s)
    k = ",".join("%g" % v for v in s.k)
    cmd = [exe, "-i", raw, "-s", "4.8", "-p", "0.34,0.5,0.24", "-z", "4", "-c", "1", "-r", "1", "-k", k, "-v", "64", "-o", log, "-d", vol]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, KERNEL_TIMINGS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(log).read()
    lines = text.strip().split("\n")
    assert lines[0].split("\t")[0] == "frame" and len(lines[0].split("\t")) in (14, 15) and len(lines) == n + 1
    rep = checkpos.summarise(text, open(gt).read())
    assert rep["_meta"]["valid"] == n and rep["_meta"]["untracked"] == 0          # 4 untracked frames, counted from -4
    assert rep["ATE"]["max"] < 0.03 and rep["computation"]["mean"] > 0
    parsed = checkpos.parse_log(text)
    assert [int(t[3]) for t in parsed["traj"]] == [0, 0, 0, 0] + [1] * (n - 4)
    # the driver and the python binding sit on the same library: same trajectory to the last printed digit
    from slambench_amd import kfusion
    # the driver forms the initial position in float32: initial_pos_factor * volume_size (benchmark.cpp:86)
    init = np.array([0.34, 0.5, 0.24], np.float32) * np.float32(4.8)
    g = kfusion.Kfusion((160, 120), (64,) * 3, util.VOL_DIM, init)
    for f in range(n):
        x, y, z = (float(v) for v in parsed["traj"][f][:3])
        assert np.allclose(g.get_position(), (x, y, z), atol=5.1e-7)            # 6 decimals in the log
        g.preprocessing(s.depth_mm[f]); g.tracking(list(s.k), 1e-5, 1, f); g.integration(list(s.k), 1, 0.1, f); g.raycasting(list(s.k), 0.1, f)
    assert os.path.getsize(vol) == 64 ** 3 * 2
    assert np.array_equal(np.fromfile(vol, np.int16), g.buffer("volume")[..., 0].reshape(-1))
    g.close()
    # KERNEL_TIMINGS lines (cpp/kernels.cpp:33-35 format) feed the checkKernels twin
    rep = checkkernels.summarise(r.stderr)
    for name in ("bilateral_filter", "track", "integrate", "raycast", "renderDepth", "renderTrack", "renderVolume", "pyramidKernel"):
        assert name in rep and rep[name]["count"] >= 3 and rep[name]["mean"] > 0, name
    assert rep["bilateral_filter"]["count"] == n and rep["renderVolume"]["count"] == 3      # -z 4 on 12 frames
    assert "input Size is = 160,120" in r.stderr


REFDRIVER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "kfusion-benchmark-refdriver")


def _run_driver(exe, raw, log, vol, k, extra):
    cmd = [exe, "-i", raw, "-s", "4.8", "-p", "0.34,0.5,0.24", "-z", "4", "-k", k, "-o", log, "-d", vol] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return checkpos.parse_log(open(log).read())


def test_baseline_config0_c2_v128_r1_against_the_oracle_and_the_reference_driver(hip, orc, tmp_path):
    """BASELINE configs[0]: `-c 2 -v 128,128,128 -r 1` on a 640x480 stream (the README's `-c 2` shape, README.md:194) through
    kfusion-benchmark-hip, held to the CPU oracle driven the same way; and, when the build container left it in oracle/_ref,
    the reference's OWN benchmark.cpp (unmodified, compiled against include/kfusion, linked with libkfusion_hip.so) gives the
    same log and the same volume dump byte for byte."""
    n = 10
    s = synth.make_stream(n, 640, 480)
    raw = str(tmp_path / "vga.raw")
    synth.write_raw(raw, s)
    k = ",".join("%g" % v for v in s.k)
    extra = ["-c", "2", "-v", "128,128,128", "-r", "1"]
    ours = _run_driver(build.BENCH_BIN, raw, str(tmp_path / "a.log"), str(tmp_path / "a.vol"), k, extra)
    cols = ours["columns"]
    assert len(ours["traj"]) == n
    # oracle: same frame loop as benchmark.cpp:125-158 (float32 init pose, camera / ratio)
    init = np.array([0.34, 0.5, 0.24], np.float32) * np.float32(4.8)
    o = orc.OracleKfusion((320, 240), (128,) * 3, util.VOL_DIM, init)
    kk = [float(np.float32(v) / np.float32(2)) for v in np.array([float(x) for x in k.split(",")], np.float32)]
    for f in range(n):
        pos = o.get_pose()[:3, 3] - init
        tracked, integrated = cols["tracked"][f], cols["integrated"][f]
        assert np.allclose(pos, (cols["X"][f], cols["Y"][f], cols["Z"][f]), atol=1e-4), f
        o.preprocessing(s.depth_mm[f])
        to = o.tracking(kk, 1e-5, 1, f); io = o.integration(kk, 1, 0.1, f); o.raycasting(kk, 0.1, f)
        assert (int(tracked), int(integrated)) == (int(to), int(io)), f
    vol = np.fromfile(str(tmp_path / "a.vol"), np.int16)
    ref = o.buffer("volume")[..., 0].reshape(-1)
    assert vol.size == 128 ** 3
    off = np.abs(vol.astype(np.int32) - ref.astype(np.int32)) > 3          # 1e-4 of the TSDF range is 3.3 of 32766
    assert off.mean() <= 1e-4, off.mean()
    o.close()
    if os.path.exists(REFDRIVER):
        theirs = _run_driver(REFDRIVER, raw, str(tmp_path / "b.log"), str(tmp_path / "b.vol"), k, extra)
        for name in ("frame", "X", "Y", "Z", "tracked", "integrated"):
            assert theirs["columns"][name] == cols[name], name
        assert open(str(tmp_path / "a.vol"), "rb").read() == open(str(tmp_path / "b.vol"), "rb").read()
        # its `computation` column is a real device time: every tock() waited for the device (synchroniseDevices -> kfh_synchronise_all)
        assert min(theirs["columns"]["computation"]) > 2e-5
