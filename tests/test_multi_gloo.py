"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Each rank runs its own stream through
the CPU twin of the pipeline (the GPU kernels are not needed to test the sharding and the aggregation) and the
job statistics are all-reduced exactly as bench.py does on RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from slambench_amd import multi, synth, checkpos
    from oracle import orc
    world, rank, local_rank = multi.env_rank()
    dist.init_process_group("gloo")
    assert dist.get_world_size() == world == 2
    n = 6
    s = synth.make_stream(n, 80, 60, variant=multi.stream_variant(rank))
    kf = orc.OracleKfusion((80, 60), (32, 32, 32), (4.8,) * 3, s.init_pos, (4, 3))
    rows = []
    dist.barrier()
    t0 = time.perf_counter()
    for f in range(n):
        rows.append(tuple(kf.get_position()))
        kf.preprocessing(s.depth_mm[f]); kf.tracking(list(s.k), 1e-5, 1, f); kf.integration(list(s.k), 1, 0.1, f); kf.raycasting(list(s.k), 0.1, f)
    sec = time.perf_counter() - t0
    dist.barrier()
    errs = checkpos.ate(rows, [tuple(g) for g in synth.ground_truth_rows(s)])
    mine = multi.RankStats(n, sec, float(sum(errs)), float(max(errs)), 4)
    job = multi.aggregate(mine, dist)
    # every rank sees the same aggregate
    gathered = [None, None]
    dist.all_gather_object(gathered, (job.frames, round(job.seconds, 9), job.untracked, sec, s.poses[3][0, 3]))
    assert gathered[0][:3] == gathered[1][:3]
    assert job.frames == 2 * n and job.world == 2 and job.untracked == 8
    assert abs(job.seconds - max(gathered[0][3], gathered[1][3])) < 1e-12       # MAX over ranks
    assert abs(job.fps - job.frames / job.seconds) < 1e-9
    assert gathered[0][4] != gathered[1][4]                                     # the two ranks ran different streams
    # a rank-local failure reaches every rank through the collectives instead of leaving the others in a barrier
    assert multi.all_ok(True, dist) is True
    assert multi.all_ok(rank != 1, dist) is False
    job = multi.aggregate(multi.RankStats(n, sec, failed=1 if rank == 1 else 0), dist)
    assert job.failed == 1
    if rank == 0:
        print(json.dumps({"fps": job.fps, "frames": job.frames, "ate_max": job.ate_max}))
    dist.destroy_process_group()
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert '"frames": 12' in out.stdout


def test_single_process_aggregate():
    from slambench_amd import multi
    job = multi.aggregate(multi.RankStats(10, 2.0, 0.5, 0.2, 4))
    assert (job.world, job.frames, job.fps, job.ate_mean, job.ate_max, job.untracked) == (1, 10, 5.0, 0.05, 0.2, 4)
    assert [multi.stream_variant(r) for r in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    assert job.failed == 0 and multi.aggregate(multi.RankStats(10, 2.0, failed=1)).failed == 1 and multi.all_ok(False) is False
