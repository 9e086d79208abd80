"""Host side of the drop-in layer without a GPU, held to the REFERENCE's own compiled code.

tests/golden/host_vectors.json is what the reference's `Configuration` (default_parameters.h:53-420), `RawDepthReader` /
`SceneDepthReader` (interface.h:99-318) and `scene2raw` (thirdparty/scene2raw.cpp + lodepng.cpp) produced, compiled where they
lie without stand-ins (oracle/Makefile target `ref`, tests/golden/make_host_vectors.py) on the seeded inputs of
tests/golden/host_inputs.py.  Here the same probe source (tests/cpp/host_probe.cpp) is compiled against THIS repo's
include/kfusion headers and the python twins are run on the same bytes: every output must be identical.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from slambench_amd import checkkernels, rawio, scene2raw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLDEN)
import host_inputs  # noqa: E402
import make_host_vectors as mk  # noqa: E402

with open(os.path.join(GOLDEN, "host_vectors.json")) as _f:
    VEC = json.load(_f)


@pytest.fixture(scope="module")
def work(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("host"))
    host_inputs.make_inputs(d)
    return d


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("probe") / "host_probe")
    subprocess.check_call(["g++", "-O1", "-std=gnu++11", "-Wall", "-I", os.path.join(ROOT, "include", "kfusion"),
                           os.path.join(ROOT, "tests", "cpp", "host_probe.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("case", VEC["config"], ids=lambda c: " ".join(c["args"]) or "defaults")
def test_configuration_equals_the_reference_build(probe, work, case):
    """Fields (float bits), print_values text, stderr chatter and exit code of every CLI case, error exits included."""
    got = mk.run_config(probe, case["args"], work)
    assert got["returncode"] == case["returncode"]
    assert got["stdout"] == case["stdout"]
    assert got["stderr"] == case["stderr"]


@pytest.mark.parametrize("case", [c for c in VEC["read"] if c["path"] != "scene.raw"], ids=lambda c: c["path"] + ":" + c["what"])
def test_cpp_readers_equal_the_reference_build(probe, work, case):
    got = mk.run_read(probe, case["path"], case["what"], work)
    assert got["meta"] == case["meta"]
    assert (got["announced_end"], got["garbage"]) == (case["announced_end"], case["garbage"])
    assert got["samples"] == case["samples"]
    assert (got["bytes"], got["sha256"]) == (case["bytes"], case["sha256"])


def _case(path, what):
    return [c for c in VEC["read"] if c["path"] == path and c["what"] == what][0]


def test_scene2raw_py_is_byte_identical_to_the_reference_tool(probe, work):
    raw = os.path.join(work, "scene.raw")
    assert scene2raw.convert(os.path.join(work, "scene"), raw) == host_inputs.SCENE_FRAMES
    got = mk.digest(raw, "<u1")
    want = VEC["scene2raw"]
    assert got["bytes"] == want["bytes"] == host_inputs.SCENE_FRAMES * (16 + 640 * 480 * 5)
    assert got["samples"] == want["samples"]
    assert got["sha256"] == want["sha256"]
    # and the C++ reader of this repo reads that file like the reference's reader reads the reference tool's file
    c = _case("scene.raw", "u16")
    got = mk.run_read(probe, "scene.raw", "u16", work)
    assert (got["meta"], got["sha256"]) == (c["meta"], c["sha256"])
    assert scene2raw.main(["scene2raw", "only-one-argument"]) == VEC["scene2raw_usage"]["returncode"] == 1


def test_python_readers_equal_the_reference_build(work):
    for name in ("in.raw", "tail.raw"):
        c = _case(name, "u16")
        frames = list(rawio.RawDepthReader(os.path.join(work, name)))
        assert len(frames) == c["meta"]["frames"]
        assert hashlib.sha256(b"".join(f.astype("<u2").tobytes() for f in frames)).hexdigest() == c["sha256"]
    c = _case("vga.raw", "f32")
    rd = rawio.RawDepthReader(os.path.join(work, "vga.raw"))
    assert hashlib.sha256(rd.read_next_depth_frame_float().astype("<f4").tobytes()).hexdigest() == c["sha256"]
    assert np.array_equal(np.array(rd.get_k(), np.float32).view(np.uint32), c["meta"]["k"])
    for what in ("u16", "f32"):
        c = _case("scene", what)
        py = rawio.SceneDepthReader(os.path.join(work, "scene"))
        assert np.array_equal(np.array(py.get_k(), np.float32).view(np.uint32), c["meta"]["k"])
        frames = []
        while True:
            f = py.read_next_depth_frame() if what == "u16" else py.read_next_depth_frame_float()
            if f is None:
                break
            frames.append(f)
        assert len(frames) == c["meta"]["frames"]
        assert hashlib.sha256(b"".join(f.tobytes() for f in frames)).hexdigest() == c["sha256"]


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "host_probe")), reason="reference build absent (GPU box)")
def test_golden_file_is_what_the_reference_build_prints_today(work):
    """Build container only: the committed fixture is not stale."""
    ref = os.path.join(ROOT, "oracle", "_ref", "host_probe")
    for case in VEC["config"]:
        assert mk.run_config(ref, case["args"], work) == case
    for case in VEC["read"]:
        if case["path"] != "scene.raw":
            assert mk.run_read(ref, case["path"], case["what"], work) == case


def test_checkkernels_summary():
    log = "mm2metersKernel 1200 307200\nbilateralFilterKernel 9000 307200\nbilateral_filter 11000 307200\ntrackKernel 500 1\nnot a line\n"
    rep = checkkernels.summarise(log)
    assert rep["bilateral_filter"] == {"count": 2, "min": 9000.0, "max": 11000.0, "mean": 10000.0, "total": 20000.0}
    assert rep["mm2meters"]["count"] == 1 and "track" in rep
