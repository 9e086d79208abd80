"""bench.py's own main() under N ranks: launched plainly with --gpus N it starts the N ranks itself as a child
torch.distributed.run, inside a rank it refuses a world size that differs from --gpus, and the line it prints proves how many ranks
the collective library saw.  --dry-collectives keeps everything but the frames (no GPU here): rendezvous, barriers, all-reduces, JSON."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", **extra)
    return env


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_plain_launch_spawns_the_ranks_itself():
    # `python bench.py --gpus 2`: no torch.distributed environment -> the parent starts two ranks as a child process
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-collectives"],
                         capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    j = _json_line(out.stdout)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["dry_run"] is True and j["value"] is None
    assert j["frames_all_ranks"] == 2 * 3                      # whole-job aggregate: frames summed over the ranks
    assert set(j["configs"]) == {"cfg3", "cfg4"}               # the compact legs ran their collectives on both ranks too
    assert "starting 2 ranks" in out.stderr


def test_under_torch_distributed_run_as_the_driver_launches_it():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-collectives", "--extra-configs", "none"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    j = _json_line(out.stdout)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["configs"] == {}


def test_world_size_must_equal_gpus():
    # two ranks but --gpus 4: every rank exits non-zero and nothing is printed (never a line with the wrong n_gpus)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "4", "--dry-collectives"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "WORLD_SIZE=2" in out.stderr
    # a single process that inherited WORLD_SIZE=1 but is asked for 2 GPUs refuses as well
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-collectives"], capture_output=True, text=True, timeout=120,
                         env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert out.returncode == 2 and "WORLD_SIZE=1" in out.stderr and not out.stdout.strip()


def test_more_gpus_than_the_node_has_fails_loudly():
    # no GPU in the CPU container: the real (non-dry) path must refuse instead of measuring something else
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, env=_env(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""), cwd=ROOT)
    assert out.returncode == 2, out.stdout[-500:] + out.stderr[-2000:]
    assert "needs 1 visible GPU(s)" in out.stderr and not out.stdout.strip()
