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


def test_eight_ranks_dry_as_the_scaling_run_will_launch_them():
    """VERDICT r4 #6b: the N = 8 shape of BASELINE.json configs[4] - eight ranks through the real main() (gloo, no frames): the world
    size, the ranks the collective saw, and the trajectory each rank runs (4 distinct ones, reused x2)."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-collectives", "--extra-configs", "cfg3"],
                         capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    j = _json_line(out.stdout)
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["scaling"] == "weak"
    assert j["placement"]["trajectories"] == [0, 1, 2, 3, 0, 1, 2, 3]
    assert j["frames_all_ranks"] == 8 * 2 and j["configs"]["cfg3"]["frames"] == 8 * 40
    assert "independent" in j["placement"]["scaling_note"]


def test_numa_pinning_reads_the_topology(tmp_path, monkeypatch):
    """slambench_amd/affinity.py on a fake sysfs: two sockets, four GPUs in KFD order behind two CPU nodes, a visibility filter, a host
    without NUMA information - and on this machine's real sysfs, whatever it holds, it must return a record and never raise."""
    from slambench_amd import affinity
    kfd, pci, node = tmp_path / "kfd", tmp_path / "pci", tmp_path / "node"
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    sockets = [allowed[:half], allowed[half:] or allowed[:half]]
    for i, cpus in enumerate(sockets):
        (node / ("node%d" % i)).mkdir(parents=True)
        (node / ("node%d" % i) / "cpulist").write_text(affinity.format_cpulist(cpus) + "\n")
    props = ["simd_count 0\nlocation_id 0\ndomain 0\n"] * 2
    bdfs = []
    for g in range(4):
        bus = 0x10 + 0x20 * g
        props.append("simd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        bdf = "0000:%02x:00.0" % bus
        bdfs.append(bdf)
        (pci / bdf).mkdir(parents=True)
        (pci / bdf / "numa_node").write_text("%d\n" % (g // 2))
    for i, text in enumerate(props):
        (kfd / str(i)).mkdir(parents=True)
        (kfd / str(i) / "properties").write_text(text)
    roots = dict(kfd_root=str(kfd), pci_root=str(pci))
    assert affinity.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and affinity.format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    assert [affinity.gpu_numa_node(r, {}, **roots)[0] for r in range(4)] == [0, 0, 1, 1]
    assert affinity.gpu_numa_node(0, {"HIP_VISIBLE_DEVICES": "3,0"}, **roots) == (1, bdfs[3])
    assert affinity.gpu_numa_node(5, {}, **roots)[0] is None and affinity.gpu_numa_node(0, {"HIP_VISIBLE_DEVICES": "GPU-abc"}, **roots)[0] is None
    (pci / bdfs[0] / "numa_node").write_text("-1\n")
    assert affinity.gpu_numa_node(0, {}, **roots)[0] is None
    try:
        rec = affinity.pin_to_gpu_numa_node(3, {}, node_root=str(node), **roots)
        assert rec["pinned"] and rec["numa_node"] == 1 and affinity.parse_cpulist(rec["cpus"]) == sorted(set(sockets[1]))
        assert sorted(os.sched_getaffinity(0)) == sorted(set(sockets[1]))
    finally:
        os.sched_setaffinity(0, allowed)
    rec = affinity.pin_to_gpu_numa_node(0)            # the real sysfs of this machine: a record, never an exception
    assert set(rec) == {"pinned", "cpus", "numa_node", "why"}
    os.sched_setaffinity(0, allowed)
