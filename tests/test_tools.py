"""Host-side tooling: synthetic stream generator, .raw wire format, freiburg ground truth, ATE tool, CLI parsing."""
import os
import subprocess

import numpy as np
import pytest

from slambench_amd import checkkernels, checkpos, perfstats, rawio, rocprof2log, synth

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
    # ... and then ATE_wrt_kfusion scores the frames that were processed only (checkPos.py:99,127-128,147-148);
    # without dropped frames the row is not reported
    rep = checkpos.summarise("\n".join(text), open(gt_path).read())
    padded = np.linalg.norm(disp[2] - disp[1])             # the dropped frame 2 repeats frame 1's position
    assert abs(rep["ATE"]["max"] - padded) < 1e-6 and abs(rep["ATE"]["total"] - padded) < 5e-6
    assert rep["ATE_wrt_kfusion"]["max"] < 1e-6 and rep["_meta"]["dropped"] == 1
    assert "ATE_wrt_kfusion" not in checkpos.summarise(_fake_log(disp, [1] * 6), open(gt_path).read())


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


def test_perfstats_summary_block_equals_the_references_compiled_perfstats():
    """SURVEY §8 f4.  tests/golden/perfstats_vectors.json is what the reference's thirdparty/perfstats.h, compiled where it lies with the
    reference's -O3 (oracle/_ref/perfstats_probe, tests/golden/make_perfstats_vectors.py), prints as the closing block of
    PerfStats::print_all_data for seeded tables of benchmark-log columns: in the form it takes when the log went to stdout (fixed
    notation, left-aligned names) and when it went to a file (default notation, right-aligned).  The python twin must produce the same
    characters - including the reference's divisor N + 1 under the "standard deviation"."""
    import json
    import os
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "perfstats_vectors.json")))
    assert set(vec["cases"]) == {"twelve_frames", "one_frame", "ninety_frames"}
    for name, case in vec["cases"].items():
        cols = {n: [float(r[i]) for r in case["rows"]] for i, n in enumerate(vec["columns"])}
        st = perfstats.summarise_columns(cols)
        assert perfstats.render(st, log_on_stdout=True).rstrip("\n").split("\n") == case["summary_log_on_stdout"], name
        assert perfstats.render(st, log_on_stdout=False).rstrip("\n").split("\n") == case["summary_log_in_file"], name
    # from a log as kfusion-benchmark-* writes it: every column of the header, frame included, in the header's order
    pos = np.zeros((4, 3))
    text = _fake_log(pos, [1, 1, 1, 1]).split("\n")
    rows = [r.split("\t") for r in text[1:5]]
    for i, r in enumerate(rows):
        r[7] = "%.6f" % (0.001 * (i + 1))            # computation column: 1, 2, 3, 4 ms
    text = "\n".join([text[0]] + ["\t".join(r) for r in rows]) + "\n"
    st = perfstats.summary(text)
    assert list(st)[:2] == ["frame", "acquisition"] and st["computation"]["n"] == 4
    assert abs(st["computation"]["mean"] - 0.0025) < 1e-12
    assert abs(st["computation"]["sd"] - np.sqrt(np.sum((np.array([1e-3, 2e-3, 3e-3, 4e-3]) - 0.0025) ** 2) / 5)) < 1e-15      # N + 1 = 5
    out = perfstats.render(st)
    assert out.count("\n") == 14 and "#   computation\tMean:\t0.0025\tStandard deviation:\t0.001\tmin:\t0.001\tmax:\t0.004" in out


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "perfstats_probe")),
                    reason="reference build absent (GPU box)")
def test_perfstats_golden_file_is_what_the_reference_build_prints_today():
    """Build container only: the committed fixture is not stale."""
    import json
    import sys
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, golden)
    import make_perfstats_vectors as mk
    vec = json.load(open(os.path.join(golden, "perfstats_vectors.json")))
    probe = os.path.join(os.path.dirname(golden), "..", "oracle", "_ref", "perfstats_probe")
    for name, case in vec["cases"].items():
        assert mk.run(probe, case["rows"], False) == case["summary_log_on_stdout"], name
        assert mk.run(probe, case["rows"], True) == case["summary_log_in_file"], name


def test_rocprof_trace_to_kernel_log():
    """rocprofv3 kernel-trace rows come out as `<reference kernel name> <ns>` lines that checkkernels aggregates (the
    nvprof2log.py -> checkKernels.py chain of the reference)."""
    rows = [
        {"Kernel_Name": "void kf::k_raycast<4, 1>(HIP_vector_type<float, 4u>*, ...)", "Start_Timestamp": "3000", "End_Timestamp": "53000"},
        {"Kernel_Name": "kf::k_icp_iter(kf::IcpIterArgs)", "Start_Timestamp": "1000", "End_Timestamp": "2500"},
        {"Kernel_Name": "kf::k_icp_iter_bilateral(kf::IcpIterArgs, kf::SideBilateral)", "Start_Timestamp": "2500", "End_Timestamp": "3000"},
        {"Kernel_Name": "void kf::k_integrate<8>(HIP_vector_type<short, 2u>*, ...)", "Start_Timestamp": "500", "End_Timestamp": "900"},
        {"Kernel_Name": "__amd_rocclr_fillBufferAligned", "Start_Timestamp": "0", "End_Timestamp": "100"},
        {"Kernel_Name": "void kf::k_bilateral<true>(float*, float*, ...)", "Start_Timestamp": "100", "End_Timestamp": "400"},
    ]
    lines = rocprof2log.convert(rows)
    assert lines == [("bilateralFilterKernel", 300), ("integrateKernel", 400), ("trackKernel", 1500), ("trackKernel", 500), ("raycastKernel", 50000)]
    rep = checkkernels.summarise("\n".join("%s %d" % l for l in lines))
    assert rep["track"]["count"] == 2 and rep["track"]["total"] == 2000 and rep["raycast"]["mean"] == 50000 and rep["bilateral_filter"]["count"] == 1


def test_trajectory_table_and_real_data_hook(tmp_path, monkeypatch):
    """Makefile:3-6 parameter lines, the Makefile's file names, and bench.py's workload loader on a `.raw` + freiburg pair."""
    import argparse
    import sys
    from slambench_amd import trajectories
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    import host_inputs
    assert [" ".join(trajectories.TRAJECTORIES[i].cli()) for i in range(4)] == [" ".join(c) for c in host_inputs.CONFIG_CASES[1:5]]
    t2 = trajectories.TRAJECTORIES[2]
    assert np.array_equal(t2.init_position(), np.array([0.34, 0.5, 0.24], np.float32) * np.float32(4.8))
    assert trajectories.guess_index("/data/living_room_traj3_loop.raw") == 3 and trajectories.guess_index("livingRoom1.gt.freiburg") == 1
    assert trajectories.guess_index("whatever.raw") == 2 and trajectories.find_dataset(str(tmp_path), 2) is None
    s = synth.make_stream(7, 640, 480)
    synth.write_raw(str(tmp_path / t2.raw_name()), s)
    synth.write_freiburg(str(tmp_path / t2.gt_name()), s)
    assert trajectories.find_dataset(str(tmp_path), 2) == (str(tmp_path / t2.raw_name()), str(tmp_path / t2.gt_name()))
    monkeypatch.setenv("KFH_ICL_NUIM_DIR", str(tmp_path))
    assert trajectories.find_dataset(None, 2)[0].endswith("living_room_traj2_loop.raw") and trajectories.find_dataset(None, 0) is None
    import bench
    args = argparse.Namespace(raw=None, gt=None, traj=None, config="cfg2")
    wl = bench.load_workload(args, 0, 7, 640, 480)            # found through KFH_ICL_NUIM_DIR
    assert wl.data == "icl-nuim living_room_traj2" and wl.volume_size == 4.8 and wl.k == (481.2, 480.0, 320.0, 240.0)
    assert np.array_equal(wl.depth_mm, s.depth_mm) and len(wl.gt_rows) == 7 and wl.source.endswith(".raw")
    with pytest.raises(SystemExit):
        bench.load_workload(args, 0, 8, 640, 480)             # more frames than the file holds
    monkeypatch.delenv("KFH_ICL_NUIM_DIR")
    monkeypatch.setenv("KFUSION_SYNTH_CACHE", str(tmp_path / "cache"))
    wl = bench.load_workload(args, 0, 3, 160, 120)
    assert wl.data == "synthetic" and wl.depth_mm.shape == (3, 120, 160) and wl.source is None
    wl.write_raw(str(tmp_path / "w.raw"), 2)
    assert os.path.getsize(str(tmp_path / "w.raw")) == 2 * (16 + 160 * 120 * 5)


def test_cpu_sibling_drivers_run_baseline_config0(tmp_path):
    """BASELINE configs[0] (`-c 2 -v 128,128,128 -r 1`, "CPU ref, plumbing") on the CPU siblings of kfusion-benchmark-hip: the same
    driver source over the CPU restatement (oracle/orc_kfh_shim.cpp), serial and OpenMP - as the reference builds
    kfusion-benchmark-{cpp,openmp} from one benchmark.cpp.  Same log format, same positions from both (the OpenMP build is
    bit-identical by default), and the python binding of the restatement agrees to the printed digit."""
    from oracle import orc
    orc.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 7
    s = synth.make_stream(n, 320, 240)
    raw = str(tmp_path / "qvga.raw")
    synth.write_raw(raw, s)
    k = ",".join("%g" % v for v in s.k)
    logs = {}
    for name in ("cpp", "openmp"):
        exe = os.path.join(root, "oracle", "kfusion-benchmark-" + name)
        log, vol = str(tmp_path / (name + ".log")), str(tmp_path / (name + ".vol"))
        r = subprocess.run([exe, "-i", raw, "-s", "4.8", "-p", "0.34,0.5,0.24", "-z", "4", "-c", "2", "-v", "128,128,128", "-r", "1", "-k", k,
                            "-o", log, "-d", vol], capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="4"))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "input Size is = 320,240" in r.stderr
        logs[name] = checkpos.parse_log(open(log).read())["columns"]
        assert len(logs[name]["frame"]) == n and os.path.getsize(vol) == 128 ** 3 * 2
    for col in ("X", "Y", "Z", "tracked", "integrated"):
        assert logs["cpp"][col] == logs["openmp"][col], col
    assert logs["cpp"]["tracked"] == [0.0] * 4 + [1.0] * (n - 4) and min(logs["cpp"]["computation"]) > 0
    assert open(str(tmp_path / "cpp.vol"), "rb").read() == open(str(tmp_path / "openmp.vol"), "rb").read()
    init = np.array([0.34, 0.5, 0.24], np.float32) * np.float32(4.8)
    o = orc.OracleKfusion((160, 120), (128,) * 3, (4.8,) * 3, init)
    kk = [float(np.float32(v) / np.float32(2)) for v in np.array(s.k, np.float32)]
    for f in range(n):
        assert np.allclose(o.get_position(), (logs["cpp"]["X"][f], logs["cpp"]["Y"][f], logs["cpp"]["Z"][f]), atol=5.1e-7), f
        o.preprocessing(s.depth_mm[f]); o.tracking(kk, 1e-5, 1, f); o.integration(kk, 1, 0.1, f); o.raycasting(kk, 0.1, f)
    assert np.array_equal(np.fromfile(str(tmp_path / "cpp.vol"), np.int16), o.buffer("volume")[..., 0].reshape(-1))
    o.close()


def test_bench_timed_run_reports_a_failing_rank(capsys):
    """bench.py's timed region on fakes (no GPU): a leg that cannot be set up ends the run through the collective check instead
    of leaving other ranks in a barrier; a leg that fails inside the timed region hands the error back; a good leg is stepped
    warm-up + steps times and synchronised on both sides."""
    import types
    import bench
    from slambench_amd import multi
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(synchronize=lambda: None))

    class Leg:
        with_renders = False

        def __init__(self, fail_at=None):
            self.steps, self.syncs, self.fail_at = [], 0, fail_at
            self.kf = types.SimpleNamespace(set_profiling=lambda *a, **k: None)

        def step(self, f):
            if f == self.fail_at:
                raise RuntimeError("boom at %d" % f)
            self.steps.append(f)

        def sync(self):
            self.syncs += 1

    def broken():
        raise RuntimeError("no device")

    S = bench.Session(fake_torch, None, 1, 0, 0, "cpu", False)
    with pytest.raises(SystemExit):
        bench.timed_run(S, broken, multi, 2, 3, 1)
    assert "failed before the timed region" in capsys.readouterr().err
    elapsed, leg, err = bench.timed_run(S, lambda: Leg(), multi, 2, 3, 1)
    assert err is None and leg.steps == [0, 1, 2, 3, 4] and leg.syncs == 2 and elapsed >= 0
    elapsed, leg, err = bench.timed_run(S, lambda: Leg(fail_at=3), multi, 2, 3, 1)
    assert isinstance(err, RuntimeError) and leg.steps == [0, 1, 2]
    assert bench.icp_launches((10, 5, 4)) == 12 and bench.icp_launches((10, 5, 4, 4)) == 16
    assert bench.icp_launches((10, 5, 4), heads=(4, 3)) == 13 and bench.icp_launches((10, 5, 4, 4), heads=(4, 3)) == 17
    assert bench.icp_launches((10, 5, 4), heads=(4,)) == 14 and bench.icp_launches((10, 5, 4, 4), heads=(4,)) == 18


def test_bench_traffic_falls_back_to_the_committed_summary(monkeypatch):
    """roofline.traffic: measured by child runs under rocprofv3 when that works, else the committed profiles/traffic_<config>.json -
    and `traffic_source` says which of the two it was."""
    import bench
    monkeypatch.setattr(bench, "measure_traffic", lambda config, timeout_s=90: None)
    traffic, source = bench.traffic_of("cfg2", True)
    assert set(traffic) == {"integrate", "raycast", "icp"} and all(v > 1e6 for v in traffic.values())
    assert "not measured in this run" in source and "profiles/traffic_cfg2.json" in source
    monkeypatch.setattr(bench, "measure_traffic", lambda config, timeout_s=90: ({"integrate": 1, "raycast": 2, "icp": 3}, "measured in this run: test"))
    assert bench.traffic_of("cfg2", True) == ({"integrate": 1, "raycast": 2, "icp": 3}, "measured in this run: test")
    assert "not measured" in bench.traffic_of("cfg2", False)[1]           # --no-measure-traffic / N > 1
    # the ICP loop's figure: bytes of every k_icp* / k_control launch of the run over its frames (= launches of k_raycast)
    sample = {"kf::k_raycast<4, 1>": {"launches": 10, "hbm_bytes_per_launch": 200, "hbm_bytes_total": 2000},
              "kf::k_integrate<8, true>": {"launches": 5, "hbm_bytes_per_launch": 300, "hbm_bytes_total": 1500},
              "kf::k_icp_iter": {"launches": 90, "hbm_bytes_per_launch": 50, "hbm_bytes_total": 4000},
              "kf::k_icp_tail": {"launches": 10, "hbm_bytes_per_launch": 10, "hbm_bytes_total": 100}}
    assert bench._traffic_from(sample) == {"raycast": 200, "integrate": 300, "icp": 410}
    # with the frames of the profiled run known (steps + warmup of its one context) they are the divisor: frames 0-2 cast no rays
    assert bench._traffic_from(sample, frames=13)["icp"] == round(4100 / 13)
    assert bench._traffic_from(dict(sample, frames=20))["icp"] == 205


def test_bench_never_profiles_a_child_from_a_profiled_process(monkeypatch):
    """Under rocprofv3 (its preload / tool variables inherited) bench.py must not start profiled children of its own: nested
    profiling, and every launcher hop of the child would load the preloaded library - which initialises the GPU - and exec the next
    program.  measure_traffic() then returns None without spawning anything and the committed summary is used."""
    import subprocess
    import bench
    spawned = []
    monkeypatch.setattr(subprocess, "run", lambda *a, **kw: spawned.append(a) or (_ for _ in ()).throw(AssertionError("spawned a child")))
    monkeypatch.setattr(bench, "_TRAFFIC_MEASURABLE", [True])
    clean = {k: v for k, v in __import__("os").environ.items() if not bench.profiler_in_environment({k: v})}
    assert not bench.profiler_in_environment(clean)
    for var, val in (("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"), ("ROCPROF_COUNTERS", "pmc: FETCH_SIZE"),
                     ("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk.so"), ("ROCPROFILER_LIBRARY_CTOR", "1")):
        assert bench.profiler_in_environment(dict(clean, **{var: val})), var
        monkeypatch.setenv(var, val)
        assert bench.measure_traffic("cfg2") is None
        traffic, source = bench.traffic_of("cfg2", True)
        assert "not measured in this run" in source and traffic
        monkeypatch.delenv(var)
    assert not spawned
    assert bench._TRAFFIC_MEASURABLE == [True]          # refusing is not a failed pass: later configurations are not disabled by it
    assert not bench.profiler_in_environment(dict(clean, LD_PRELOAD="/usr/lib/libfoo.so"))
