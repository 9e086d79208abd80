"""bench.py with two REAL ranks on the one GPU of the test box (rehearsal switches: collectives over gloo on the host, both ranks on
cuda:0 - RCCL refuses two ranks on one device): the N > 1 path end to end with the frames computed, through the self-spawning entry
(`python bench.py --gpus 2`) - what the 8-GPU scaling run does with one device per rank and RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_compute_their_own_streams_and_aggregate(hip):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KFH_BENCH_BACKEND="gloo", KFH_BENCH_ONE_DEVICE="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "5", "--no-cpu-baseline",
           "--extra-configs", "cfg3", "--extra-steps", "6"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["scaling"] == "weak"
    assert j["config"]["frames_per_gpu"] == 12 and j["value"] > 0
    # whole-job aggregate: frames of both ranks over the slower rank's time
    assert abs(j["value"] - 2 * 12 / j["timed_s"]) <= 0.01 * j["value"]
    assert j["ate_m"]["untracked_all_ranks"] == 2 * 4 and j["ate_m"]["max"] < 0.03          # both trajectories track from frame 4 on
    assert j["cpu_baseline"] is None and j["fps_host_buffers"] is None                       # rank 0 at N = 1 only
    assert j["value_host_input"] and j["fps_total"]
    c3 = j["configs"]["cfg3"]
    assert c3["n_gpus"] == 2 and c3["value"] > 0 and c3["roofline"]["integrate"]["frac"] > 0


def test_nccl_process_group_of_one_rank(hip):
    """VERDICT r4 #6c: the `nccl` branch of bench.py - init_process_group("nccl", device_id=...), the all-reduces on CUDA tensors,
    the barriers, destroy_process_group - executed once on the one-GPU box (a group of one rank, KFH_BENCH_FORCE_GROUP=1) before an
    8-GPU node meets it; RCCL is what carries the start / stop barriers and the statistics there."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KFH_BENCH_FORCE_GROUP="1", KFH_BENCH_BACKEND="nccl", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "5", "--no-cpu-baseline",
           "--no-host-buffers", "--no-total", "--no-measure-traffic", "--extra-configs", "none"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["ranks_seen"] == 1 and j["value"] > 0
    assert j["placement"]["collective_backend"] == "nccl" and j["placement"]["trajectories"] == [0]
    assert "cpu affinity" in out.stderr
