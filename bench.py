#!/usr/bin/env python3
"""Throughput bench of the KinectFusion hot path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W            (N > 1: under torch.distributed.run - or plainly: bench.py then
                                                             starts the N ranks itself as a child torch.distributed.run)
    python bench.py --raw living_room_traj2_loop.raw --gt livingRoom2.gt.freiburg      (real ICL-NUIM data, when present;
                                                             or KFH_ICL_NUIM_DIR=<dir holding the Makefile's file names>)

One *step* is one frame through preprocessing -> tracking (multi-level ICP) -> integration -> raycasting —
the reference's `computation` column (kfusion/src/benchmark.cpp:164), the figure its README quotes — on the
configuration BASELINE.json's metric is quoted on: 640x480 depth, `-c 1 -v 256 -y 10,5,4` and the reference
Makefile's trajectory-2 flags `-s 4.8 -p 0.34,0.5,0.24 -k 481.2,480,320,240 -r 2`.  Depth frames are a seeded
synthetic stream (no dataset offline) already resident in HBM when the timed region starts, and frames are QUEUED
(kfh_compute_frame, no host round trip per frame).  With N GPUs every rank runs its own independent stream
(frames shard by stream; no data-path collective) and RCCL is used only for the barriers and one all-reduce.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  n_gpus / ranks_seen         the world size of the process group and an all-reduce of ones over it (RCCL saw that many ranks)
  configs                     compact legs of BASELINE configs[2] (cfg3) and configs[3] (cfg4) of the same run: value, ms_per_step,
                              value_host_input, roofline of integrate / raycast / the ICP loop with traffic and traffic_source
  roofline*.traffic           HBM bytes per launch from rocprofv3 --pmc passes over child runs of this script (N = 1; else the committed file)
  roofline / roofline_other   icp loop, k_integrate, k_raycast: algorithmic bytes / HIP-event time measured in this run
  cpu_baseline                the OpenMP oracle timed on the host cores on a bounded prefix of the same stream
  fps_total                   the same queued loop plus the three renders and their read-back
  fps_queued_host_input       the same queued loop with every frame handed over as pageable host memory (H2D inside the loop)
  fps_host_buffers            what a reference-style caller gets (benchmark.cpp:125-158): kfusion-benchmark-hip itself on a
                              `.raw` of the same frames - pageable host depth, four phase calls, a device sync at every tock()
  ate_m                       ATE of the estimated trajectory, scored as checkPos.py scores a log
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import subprocess
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)

CONFIGS = {
    # name: (width, height, volume, pyramid, integration_rate, rendering_rate)
    "cfg2": (640, 480, 256, (10, 5, 4), 2, 4),        # BASELINE.json configs[1]: default KFusion
    "cfg3": (640, 480, 512, (10, 5, 4), 1, 1),        # configs[2]: HBM-bound integrate+raycast stress
    "cfg4": (1280, 960, 512, (10, 5, 4, 4), 2, 4),    # configs[3]: 4-level ICP
}
MU, ICP_THRESHOLD, TRACKING_RATE = 0.1, float(os.environ.get("KFH_DEBUG_ICP_THRESHOLD", "1e-5")), 1
V_FRAMES = 2            # frames after the cpu_baseline prefix on which the oracle counts V (distinct voxels a raycast samples)


def icp_launches(pyramid, heads=(3, 3)):
    """launches of a frame's ICP loop for the FIRST context of a device (what a context really queues is kfh_icp_launches_per_frame,
    which the roofline entry uses; this twin of it is what the CPU tests check the schedule against): per level the first `head` iterations one each (3 for the first context of a device, whose
    shared launch is 64 workgroups wide; 4 on level 0 for further contexts), the rest (if >= 2) share one (kf_pipeline.hip
    tail_length); the level-0 tail also closes the loop, otherwise k_icp_final does"""
    n, closed = 0, False
    for level, its in enumerate(pyramid):
        head = heads[min(level, len(heads) - 1)]
        tail = its - head if its - head >= 2 else 0
        n += its - tail + (1 if tail else 0)
        closed = closed or (tail > 0 and level == 0)
    return n + (0 if closed else 1)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--raw", default=None, help="real data: a SLAMBench .raw file or an ICL-NUIM scene directory (default: "
                    "$KFH_ICL_NUIM_DIR/living_room_traj<N>_loop.raw when present, else the synthetic stream)")
    ap.add_argument("--gt", default=None, help="ground truth of --raw in freiburg format (livingRoom<N>.gt.freiburg)")
    ap.add_argument("--traj", type=int, default=None, help="ICL-NUIM trajectory 0-3: picks the Makefile's -s/-p/-k line (default: from the file name, else 2)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-total", action="store_true", help="skip the secondary loops (renders; frames handed over as host memory)")
    ap.add_argument("--host-input", action="store_true", help="profiling aid: the timed loop itself takes its frames from pageable host memory (the line then says so)")
    ap.add_argument("--no-host-buffers", action="store_true", help="skip the kfusion-benchmark-hip leg (fps_host_buffers)")
    ap.add_argument("--profile-every", type=int, default=None,
                    help="an HIP-event pair around ONE of icp/integrate/raycast (rotating) on every n-th timed frame; default "
                         "steps/10 clamped to 1..8; 0 = none")
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="independent RGB-D streams processed concurrently on each GPU (throughput mode; the headline is 1)")
    ap.add_argument("--extra-configs", default=None,
                    help="comma list of configurations measured as compact legs after the headline and reported under \"configs\" "
                         "(default: cfg3,cfg4 when the headline is cfg2, else none; \"none\" = none)")
    ap.add_argument("--extra-steps", type=int, default=40, help="timed frames of a compact leg (warm-up: 8)")
    ap.add_argument("--no-measure-traffic", action="store_true",
                    help="roofline.traffic from the committed profiles/traffic_<config>.json instead of two child runs under rocprofv3 --pmc (N = 1 only)")
    ap.add_argument("--no-preheat", action="store_true",
                    help="skip the throw-away leg that brings the GPU to its working clocks (the profiled child runs of the traffic measurement: "
                         "one context, so that every launch of the trace belongs to the 'steps + warmup' frames)")
    ap.add_argument("--dry-collectives", action="store_true",
                    help="rehearsal of the N-rank plumbing without a GPU: rendezvous, barriers, all-reduces and the JSON line are real, "
                         "the frames are not run (value is null, \"dry_run\": true); used by tests/test_bench_ranks.py")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` launched plainly: start the N ranks as a CHILD `torch.distributed.run` and relay its output and exit
    code.  Called before anything in this process has touched torch or the GPU (a process that has must never exec another)."""
    # --standalone: torchrun starts its own rendezvous store on a free port of 127.0.0.1 and hands it to the ranks - no port is
    # picked here (a bind(0)-close-reuse sequence can lose the port to another process in between)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           os.path.abspath(__file__)] + list(argv)
    print("bench.py: --gpus %d without a torch.distributed environment: starting %d ranks: %s" % (args.gpus, args.gpus, " ".join(cmd)), file=sys.stderr)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def load_workload(args, rank, n_frames, W, H, real_ok=True):
    """The depth stream of this rank and the scene parameters that go with it.  Returns a namespace with depth_mm (n, H, W)
    uint16, k, init_pos, volume_size, gt_rows (or None), data label, and the trajectory record when the data is real."""
    from slambench_amd import checkpos, multi, rawio, synth, trajectories
    raw, gt = (args.raw, args.gt) if real_ok else (None, None)
    if raw is None and real_ok:
        idx = args.traj if args.traj is not None else (2 if multi.env_rank()[0] == 1 else rank % multi.N_TRAJECTORIES)
        found = trajectories.find_dataset(None, idx)
        if found:
            raw, gt = found[0], gt or found[1]
    if raw is None:
        s = synth.cached_stream(n_frames, W, H, seed=1234, variant=multi.stream_variant(rank))
        return types.SimpleNamespace(depth_mm=s.depth_mm, k=tuple(s.k), init_pos=np.asarray(s.init_pos, np.float32), volume_size=4.8,
                                     init_factors=(0.34, 0.5, 0.24), gt_rows=[tuple(g) for g in synth.ground_truth_rows(s)], data="synthetic",
                                     source=None, width=W, height=H, write_raw=lambda path, n: synth.write_raw(path, types.SimpleNamespace(
                                         width=W, height=H, depth_mm=s.depth_mm[:n])))
    idx = args.traj if args.traj is not None else trajectories.guess_index(raw)
    t = trajectories.TRAJECTORIES[idx]
    reader = rawio.RawDepthReader(raw) if os.path.isfile(raw) else rawio.SceneDepthReader(raw)
    if tuple(reader.get_input_size()) != (W, H):
        raise SystemExit("--raw holds %dx%d frames, this configuration needs %dx%d" % (*reader.get_input_size(), W, H))
    frames = []
    while len(frames) < n_frames:
        d = reader.read_next_depth_frame()
        if d is None:
            break
        frames.append(d)
    if len(frames) < n_frames:
        raise SystemExit("%s holds %d frames, --warmup + --steps (+%d) need %d" % (raw, len(frames), V_FRAMES, n_frames))
    gt_rows = checkpos.parse_freiburg(open(gt).read()) if gt else None
    return types.SimpleNamespace(depth_mm=np.stack(frames), k=t.camera, init_pos=t.init_position(), volume_size=t.volume_size,
                                 init_factors=t.init_pose, gt_rows=gt_rows, data="icl-nuim living_room_traj%d" % idx, source=raw if os.path.isfile(raw) else None,
                                 width=W, height=H, write_raw=None)


class Leg:
    """One context (plus optional extra ones) stepping through the stream with kfh_compute_frame."""

    def __init__(self, kfusion, wl, d_depth, cfg, device, with_renders, extra_streams=0, host_input=False):
        W, H, vres, pyramid, irate, zrate = cfg
        self.cfg, self.wl, self.with_renders, self.host_input = cfg, wl, with_renders, host_input
        mk = lambda: kfusion.Kfusion((W, H), (vres,) * 3, (wl.volume_size,) * 3, wl.init_pos, pyramid, device=device)
        self.kf = mk()
        self.others = [mk() for _ in range(extra_streams)]
        self.base, self.frame_bytes = d_depth.data_ptr(), W * H * 2
        self.renders = [np.zeros((H, W, 4), np.uint8) for _ in range(3)] if with_renders else None

    def step(self, f):
        import ctypes as C
        W, H, _, _, irate, zrate = self.cfg
        k = list(self.wl.k)
        for ctx in [self.kf] + self.others:
            # host_input: the frame is handed over as pageable host memory, like Kfusion::computeFrame's inputDepth (PCIe inside the loop)
            src = self.wl.depth_mm[f] if self.host_input else self.base + f * self.frame_bytes
            ctx.compute_frame(src, (W, H), k, irate, TRACKING_RATE, ICP_THRESHOLD, MU, f)
        if self.with_renders:   # benchmark.cpp:153-156
            L, h = self.kf.L, self.kf.h
            u8 = C.POINTER(C.c_uint8)
            L.kfh_render_depth(h, self.renders[0].ctypes.data_as(u8), W, H)
            L.kfh_render_track(h, self.renders[1].ctypes.data_as(u8), W, H)
            L.kfh_render_volume(h, self.renders[2].ctypes.data_as(u8), W, H, f, zrate, (C.c_float * 4)(*k), 0.75 * MU)

    def sync(self):
        for ctx in [self.kf] + self.others:
            ctx.synchronise()

    def close(self):
        for ctx in [self.kf] + self.others:
            ctx.close()


class DryLeg:
    """--dry-collectives: the frames are not run; everything around them (collectives, timing brackets, JSON) is."""
    kf, with_renders = None, True

    def step(self, f):
        pass

    def sync(self):
        pass

    def close(self):
        pass


class Session:
    """What every leg of a run shares: the process group (or None), this rank, the device the collectives run on."""

    def __init__(self, torch, dist, world, rank, local_rank, coll_device, dry):
        self.torch, self.dist, self.world, self.rank, self.local_rank, self.coll_device, self.dry = torch, dist, world, rank, local_rank, coll_device, dry

    def device_sync(self):
        if not self.dry:
            self.torch.cuda.synchronize()


def timed_run(S, make_leg, multi, n_warm, n_steps, prof_every):
    """Warm up on frames [0, n_warm), time frames [n_warm, n_warm + n_steps) between barrier + synchronize pairs.
    A rank that fails still takes part in every collective, so the job ends with the reason instead of a watchdog kill."""
    leg, err = None, None
    try:
        leg = make_leg()
        for f in range(n_warm):
            leg.step(f)
        leg.sync()
        if prof_every > 0 and not leg.with_renders:
            leg.kf.set_profiling(["icp", "integrate", "raycast"], every=prof_every, rotate=True)
        elif prof_every < 0 and leg.with_renders and leg.kf is not None:      # the render-profile leg: an event pair around every render launch
            leg.kf.set_profiling(["render_depth", "render_track", "render_volume"], every=1)
    except Exception as e:       # noqa: BLE001
        err = e
    if not multi.all_ok(err is None, S.dist, S.coll_device):
        if err is not None:
            print("bench.py: rank %d failed before the timed region: %r" % (S.rank, err), file=sys.stderr)
        raise SystemExit(1)
    # (the collector stays out of the timed region, as timeit keeps it: one full collection of a process that has torch loaded is a
    # 40-45 ms pause of the thread that queues the frames - found inside a render call of the first rendered leg of a process, where
    # it turned 4500 frames/s into 2400, profiles/r05_host_stalls.txt - and when it comes depends on what the imports allocated)
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    S.device_sync()
    if S.dist is not None:
        S.dist.barrier()
    t0 = time.perf_counter()
    try:
        for f in range(n_warm, n_warm + n_steps):
            leg.step(f)
        leg.sync()
    except Exception as e:       # noqa: BLE001
        err = e
    S.device_sync()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    if S.dist is not None:
        S.dist.barrier()
    return elapsed, leg, err


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # cgroup CPU quota of the box (a one-GPU share is 16 CPUs)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per))))
    except Exception:
        pass
    return cores


def cpu_baseline(wl, cfg, budget_s, max_frames, min_frames=8):
    """OpenMP oracle on the host cores over a bounded prefix of the same stream (computation column only); then V of
    SURVEY §8(d) - the distinct voxels one raycast samples - counted by the oracle on the V_FRAMES frames that follow the prefix."""
    from oracle import orc
    orc.build()
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    orc.set_omp_threads(cores)      # (the environment variable is read once, when libgomp is first loaded - torch has loaded it long before this point)
    W, H, vres, pyramid, irate, _ = cfg
    k = list(wl.k)
    o = orc.OracleKfusion((W, H), (vres,) * 3, (wl.volume_size,) * 3, wl.init_pos, pyramid, omp=True)

    def frame(f):
        o.preprocessing(wl.depth_mm[f]); o.tracking(k, ICP_THRESHOLD, TRACKING_RATE, f)
        o.integration(k, irate, MU, f); o.raycasting(k, MU, f)

    n, spent = 0, 0.0
    timed_max = max(1, min(max_frames, wl.depth_mm.shape[0]) - V_FRAMES)
    while n < timed_max and (spent < budget_s or n < min_frames):
        t = time.perf_counter()
        frame(n)
        spent += time.perf_counter() - t
        n += 1
    o.set_counting(True)
    vs = []
    for f in range(n, min(n + V_FRAMES, wl.depth_mm.shape[0])):
        frame(f)
        if f > 2:                                        # cpp:977: no raycast before frame 3
            vs.append(o.counters()["raycast_distinct_voxels"])
    o.close()
    return {"value": round(n / spent, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "first %d frames of the same stream, OpenMP oracle (liborc_omp.so), %d threads, %.1f s" % (n, cores, spent)}, \
        (float(np.mean(vs)) if vs else None)


def host_buffers_leg(wl, cfg, n_frames):
    """kfusion-benchmark-hip (the drop-in driver, reference call sequence benchmark.cpp:125-158) on a .raw of the same frames."""
    from slambench_amd import build, checkpos
    W, H, vres, pyramid, irate, zrate = cfg
    exe = build.BENCH_BIN
    if not os.path.exists(exe):
        return {"error": "kfusion-benchmark-hip is not built"}
    tmp = None
    try:
        if wl.source:
            raw = wl.source
        else:
            n = int(max(16, min(n_frames, 60, (128 << 20) // (W * H * 5 + 16))))
            n = min(n, wl.depth_mm.shape[0])
            tmp = tempfile.mkdtemp(prefix="kfh_bench_")
            raw = os.path.join(tmp, "stream.raw")
            wl.write_raw(raw, n)
        log = os.path.join(tmp or tempfile.gettempdir(), "kfh_bench_%d.log" % os.getpid())
        cmd = [exe, "-i", raw, "-o", log, "-s", "%g" % wl.volume_size, "-p", ",".join("%g" % v for v in wl.init_factors), "-z", str(zrate),
               "-c", "1", "-r", str(irate), "-k", ",".join("%g" % v for v in wl.k), "-v", str(vres), "-y", ",".join(map(str, pyramid)),
               "-m", "%g" % MU, "-l", "%g" % ICP_THRESHOLD]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": "driver exit %d: %s" % (r.returncode, r.stderr[-300:])}
        cols = checkpos.parse_log(open(log).read())["columns"]
        n = len(cols["computation"])
        skip = min(10, n // 3)                          # first frames: lazy initialisation, frames 0-3 are not tracked
        comp, tot = cols["computation"][skip:], cols["total"][skip:]
        return {"computation": round(len(comp) / sum(comp), 2), "total": round(len(tot) / sum(tot), 2), "unit": "frames/s", "frames": len(comp),
                "computation_ms": round(1e3 * sum(comp) / len(comp), 4),
                "what": "kfusion-benchmark-hip on a .raw of the same stream: pageable host depth through preprocessing(), four phase calls, "
                        "synchroniseDevices() at every tock() (benchmark.cpp:26-39,125-158); PCIe-inclusive, never the headline value"}
    except Exception as e:       # noqa: BLE001
        return {"error": repr(e)}
    finally:
        if tmp:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched plainly with --gpus N: this process has made no torch / GPU call; the ranks run in a child process
        raise SystemExit(spawn_ranks(args, argv))

    # each rank next to its GPU: confined to the cores of the GPU's NUMA node BEFORE torch or HIP is loaded, so that every thread
    # and every pinned buffer of the rank is made there (slambench_amd/affinity.py; no numactl wrapper)
    from slambench_amd import affinity, multi
    world, rank, local_rank = multi.env_rank()
    one_device = os.environ.get("KFH_BENCH_ONE_DEVICE") == "1"     # rehearsal on a one-GPU box: the ranks side by side on cuda:0, collectives on the host
    pin = {"pinned": False, "cpus": None, "numa_node": None, "why": "dry run"} if args.dry_collectives else \
        affinity.pin_to_gpu_numa_node(0 if one_device else local_rank)
    print("bench.py: rank %d: cpu affinity %s (%s%s)" % (rank, pin["cpus"], "pinned to NUMA node %s, " % pin["numa_node"] if pin["pinned"] else "not pinned: ", pin["why"]), file=sys.stderr)

    import torch

    if world != args.gpus:
        # never measure fewer GPUs than were asked for (nor more): the record would carry the wrong n_gpus
        print("bench.py: --gpus %d but the torch.distributed environment holds WORLD_SIZE=%d: start it as `python bench.py --gpus %d` "
              "or `python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d`" % (args.gpus, world, args.gpus, args.gpus, args.gpus), file=sys.stderr)
        raise SystemExit(2)
    dry = args.dry_collectives
    backend = "gloo" if dry else os.environ.get("KFH_BENCH_BACKEND", "nccl")
    if not dry:
        have = torch.cuda.device_count()
        need = 1 if one_device else world
        if have < need:
            print("bench.py: rank %d: --gpus %d needs %d visible GPU(s), this node shows %d" % (rank, args.gpus, need, have), file=sys.stderr)
            raise SystemExit(2)
    dist = None
    # KFH_BENCH_FORCE_GROUP=1: a process group even for one rank (rehearsal: init_process_group("nccl", device_id=...), the CUDA-tensor
    # all-reduces and destroy_process_group run once on a one-GPU box before an 8-GPU node sees them, tests/test_hip_bench_ranks.py)
    force_group = world == 1 and os.environ.get("KFH_BENCH_FORCE_GROUP") == "1"
    if force_group:
        os.environ.setdefault("MASTER_PORT", "29555"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    if world > 1 or force_group:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            local_rank = 0
        if not dry:
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
    elif not dry:
        torch.cuda.set_device(0)
        local_rank = 0
    coll_device = "cuda" if (backend == "nccl" and not dry) else "cpu"
    S = Session(torch, dist, world, rank, local_rank, coll_device, dry)
    S.pin, S.backend = pin, (backend if dist is not None else None)
    try:
        run(S, args, multi)
    finally:
        if dist is not None:
            dist.destroy_process_group()


def _traffic_from(tj, frames=None):
    """{integrate, raycast, icp: HBM bytes per launch} from a summary of profiles/summarise_traffic.py.  `frames`: frames of the profiled
    run (steps + warmup of its ONE context); None for a summary that does not say - then the launches of k_raycast stand in, which is a
    few per cent high (frames 0-2 of a context cast no rays)."""
    traffic = {}
    for kname, v in tj.items():
        if not isinstance(v, dict):
            continue
        for short in ("integrate", "raycast"):
            if ("k_" + short) in kname:
                traffic[short] = v["hbm_bytes_per_launch"]
    # the ICP loop of a frame = all k_icp* launches (+ k_control when it is a launch of its own)
    kernels = {k: v for k, v in tj.items() if isinstance(v, dict)}
    frames_t = frames or tj.get("frames") or sum(v["launches"] for kname, v in kernels.items() if "k_raycast" in kname)
    loop_t = sum(v.get("hbm_bytes_total", 0) for kname, v in kernels.items() if "k_icp" in kname or "k_control" in kname)
    if frames_t and loop_t:
        traffic["icp"] = round(loop_t / frames_t)
    return traffic


def profiler_in_environment(environ=None):
    """True when this process itself runs under rocprofv3 (its preload / tool variables are inherited): a profiled child started from
    here would be nested profiling, and its launcher hops (env -> python3 -> the program) would each load the profiler's preloaded
    library - which initialises the GPU - and then exec the next program."""
    environ = os.environ if environ is None else environ
    if any(k in environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")) or any(k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in environ):
        return True
    return any(("rocprof" in v) or ("rocprofiler" in v) for v in (environ.get("LD_PRELOAD", ""),))


TRAFFIC_STEPS, TRAFFIC_WARMUP = 40, 10


_TRAFFIC_MEASURABLE = [True]      # cleared by the first pass that fails: the remaining configurations then use the committed summaries


def measure_traffic(config, timeout_s=90):
    """HBM bytes per launch measured NOW: rocprofv3 cannot attach to this process, so two child runs of this script (40 frames of the
    same configuration) are profiled, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes as MI355X_MICROARCH.md prescribes
    (profiles/collect_traffic.sh does the same by hand), and summarised with its units and gfx950 corrections
    (profiles/summarise_traffic.py).  Returns (traffic, source) or None if rocprofv3 is missing or a pass fails."""
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not _TRAFFIC_MEASURABLE[0] or not os.path.exists(exe):
        return None
    if profiler_in_environment():
        print("bench.py: running under a profiler: roofline.traffic comes from the committed summary (no nested rocprofv3 runs)", file=sys.stderr)
        return None
    _TRAFFIC_MEASURABLE[0] = False      # (until this configuration's two passes have gone through)
    tmp = tempfile.mkdtemp(prefix="kfh_traffic_")
    try:
        # (no profiler variable can be in this environment - checked above -, so only the rank variables are taken out)
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env["TMPDIR"] = "/tmp"
        for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            # rocprofv3 is a python script: started with this interpreter (no `env` hop), and the program itself after `--`: the
            # profiler's preloaded library initialises the GPU before the child starts.  The child runs ONE context (--no-preheat) and
            # no profiled children of its own (--no-measure-traffic).
            cmd = [sys.executable, exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, sub), "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--config", config, "--steps", str(TRAFFIC_STEPS), "--warmup", str(TRAFFIC_WARMUP), "--no-cpu-baseline",
                   "--no-total", "--no-host-buffers", "--extra-configs", "none", "--no-measure-traffic", "--no-preheat"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd="/tmp")
            if r.returncode != 0:
                print("bench.py: traffic pass %s of %s failed (rc %d): %s" % (counter, config, r.returncode, r.stderr[-300:]), file=sys.stderr)
                return None
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import summarise_traffic
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            summarise_traffic.main(tmp)
        traffic = _traffic_from(json.load(open(os.path.join(tmp, "traffic.json"))), frames=TRAFFIC_STEPS + TRAFFIC_WARMUP)
        if not traffic:
            return None
        _TRAFFIC_MEASURABLE[0] = True
        return traffic, ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate child runs of bench.py --config %s --steps 40 --warmup 10, one context, "
                         "KiB units, FETCH doubled for gfx950 as profiles/summarise_traffic.py documents)" % config)
    except Exception as e:       # noqa: BLE001
        print("bench.py: traffic measurement of %s failed: %r" % (config, e), file=sys.stderr)
        return None
    finally:
        import shutil as _sh
        _sh.rmtree(tmp, ignore_errors=True)


def traffic_of(config, measure):
    """HBM bytes per launch from the PMC counters: measured by child runs under rocprofv3 when asked (the default at N = 1), else - or if
    that fails - read from the committed summary of this configuration; `traffic_source` in the line says which."""
    if measure:
        got = measure_traffic(config)
        if got:
            return got
    traffic, source = {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % config)
    if os.path.exists(tpath):
        traffic = _traffic_from(json.load(open(tpath)))
        source = "profiles/traffic_%s.json: committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, profiles/collect_traffic.sh) of the same command, not measured in this run" % config
    return traffic, source


def rooflines(cfg, config_name, prof, prof_every, updated, integrated, icp_by_level, n_frames, v_distinct, measure=False, icp_launch_count=None):
    """roofline entries of k_integrate, k_raycast and the ICP loop: algorithmic bytes / HIP-event time of this run"""
    W, H, vres, pyramid, irate, zrate = cfg
    traffic, source = traffic_of(config_name, measure)
    P = W * H
    roof = {}
    for name in ("integrate", "raycast"):
        ms, n = prof[name]
        if not n:
            continue
        avg_s = ms / n * 1e-3
        if name == "integrate":
            u_per_launch = updated / max(1, int(integrated.sum()))
            alg = 8.0 * u_per_launch + 4.0 * P                                   # 8*U + 4*P  
            detail = {"U_voxels_updated_per_launch": round(u_per_launch)}
        else:
            if v_distinct is None:                                                # no oracle count in this run: say so, report no fraction
                roof[name] = {"kernel": "k_raycast", "bound": "hbm", "avg_us": round(avg_s * 1e6, 2), "launches": n, "achieved": None,
                              "frac": None, "note": "V (distinct voxels sampled) is counted by the CPU oracle: run without --no-cpu-baseline"}
                continue
            alg = 24.0 * P + 4.0 * v_distinct                                     # 24*P + 4*V
            detail = {"V_distinct_voxels_sampled": round(v_distinct)}
        gbs = alg / avg_s / 1e9
        roof[name] = {"kernel": "k_" + name, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic.get(name), "traffic_source": source if traffic.get(name) else None,
                      "avg_us": round(avg_s * 1e6, 2), "launches": n,
                      "sampled_every_n_frames": prof_every, "algorithmic_bytes_per_launch": round(alg), **detail}
    # The ICP loop of a frame (k_icp_iter* launches, k_icp_tail for the late iterations of a level, k_icp_final): 48 + 4 bytes per associated pixel
    # for every EXECUTED iteration (below), plus - on the queued-frame path - the
    # next frame's bilateral filter (2P in, 8P out) and pyramid (4P in, 36 bytes per pixel of every level out) that ride
    # on it.  One "launch" of this entry is one frame's loop.
    ms, n = prof["icp"]
    if n:
        levels = len(pyramid)
        px = [(W >> l) * (H >> l) for l in range(levels)]
        # SURVEY §8(d), fused track + reduce: 48 bytes per associated pixel (two float4 input maps, a 32-byte gather from the
        # reference maps) + 4 for the result map the launch keeps for renderTrack, per EXECUTED iteration
        alg_icp = 52.0 * sum(icp_by_level[l] * px[l] for l in range(levels)) / max(1, n_frames)
        alg_pre = 10.0 * P + 4.0 * P + 36.0 * sum(px)
        alg = alg_icp + alg_pre
        avg_s = ms / n * 1e-3
        gbs = alg / avg_s / 1e9
        roof["icp"] = {"kernel": "k_icp_iter* + k_icp_tail (one frame's ICP loop: %d queued iterations in %d launches)" % (sum(pyramid), icp_launch_count or icp_launches(pyramid)), "bound": "hbm",
                       "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                       "traffic": traffic.get("icp"), "traffic_source": source if traffic.get("icp") else None, "avg_us": round(avg_s * 1e6, 2), "launches": n,
                       "sampled_every_n_frames": prof_every, "algorithmic_bytes_per_launch": round(alg),
                       "of_which_next_frame_preprocessing_bytes": round(alg_pre),
                       "executed_iterations_per_frame_by_level": [round(v / max(1, n_frames), 2) for v in icp_by_level]}
    return roof


def workload_string(name, cfg, wl):
    W, H, vres, pyramid, irate, zrate = cfg
    return ("%s: %dx%d depth, -c 1 -v %d -y %s -r %d -z %d -s %g -p %s -k %s -m 0.1, one stream per GPU"
            % (name, W, H, vres, ",".join(map(str, pyramid)), irate, zrate, wl.volume_size,
               ",".join("%g" % v for v in wl.init_factors), ",".join("%g" % v for v in wl.k)))


def measure(S, args, multi, config_name, steps, warmup, full):
    """One configuration: the queued loop with device-resident input (timed with sampled HIP-event pairs for the rooflines), the same
    loop fed from pageable host memory, and - `full` only - the loop with the three renders, the drop-in driver and the CPU baseline.
    Every rank runs every leg (the collectives must match); rank 0 returns the result dictionary, the others None."""
    from slambench_amd import checkpos
    torch, dist = S.torch, S.dist
    cfg = CONFIGS[config_name]
    if full and os.environ.get("KFH_DEBUG_PYRAMID"):    # experiments only (cost of the queued-but-empty ICP launches)
        cfg = cfg[:3] + (tuple(int(v) for v in os.environ["KFH_DEBUG_PYRAMID"].split(",")),) + cfg[4:]
    W, H, vres, pyramid, irate, zrate = cfg
    n_frames = warmup + steps
    prof_every = args.profile_every if args.profile_every is not None else max(1, min(8, steps // 10))
    while prof_every > 1 and math.gcd(prof_every, irate) != 1:     # integrate runs on frames = 0 (mod -r): the sampled frames must meet them
        prof_every += 1
    spg = max(1, args.streams_per_gpu) if full else 1

    if S.dry:
        mk = lambda *a, **kw: DryLeg()
        wl = d_depth = kfusion = None
    else:
        from slambench_amd import kfusion
        # one independent RGB-D stream per GPU: 4 distinct trajectories, reused x2 on 8 GPUs (BASELINE configs[4])
        # (the CPU baseline wants ~15 s of work whatever --steps is: a synthetic stream is generated long enough for it; the GPU
        # legs use its first warmup + steps frames)
        cpu_frames = 150 if (full and S.world == 1 and not args.no_cpu_baseline and args.raw is None) else 0
        wl = load_workload(args, S.rank, max(n_frames, cpu_frames) + V_FRAMES, W, H, real_ok=(W, H) == CONFIGS[args.config][:2])
        d_depth = torch.from_numpy(wl.depth_mm.view(np.int16)).cuda()
        mk = lambda renders, extra=0, host_input=False: Leg(kfusion, wl, d_depth, cfg, S.local_rank, renders, extra, host_input)

    def secondary(make, what):
        """a leg whose only product is its whole-job rate; a failure is reported (stderr and the line), never silent"""
        e, leg, err = timed_run(S, make, multi, warmup, steps, 0)
        if leg is not None:
            leg.close()
        if err is not None:
            print("bench.py: rank %d: the %s leg of %s failed: %r" % (S.rank, what, config_name, err), file=sys.stderr)
        j = multi.aggregate(multi.RankStats(steps, e, failed=0 if err is None else 1), dist, device=S.coll_device)
        return (None, "%d rank(s) failed%s" % (j.failed, ": %r" % (err,) if err is not None else "")) if j.failed else (j.fps, None)

    # A throw-away leg first, so that the first measured leg does not start from an idle-clocked GPU.  Not timed, not reported.
    # (Rounds 3-5 blamed it for a first rendered leg that came out 15-40 % low - fps_total 2400-3200 against 4000+ in later
    # processes; that was the Python collector's first full collection, which timed_run now keeps out of the timed region.)
    if full and not S.dry and not args.no_preheat:
        _, pre, perr = timed_run(S, lambda: mk(False), multi, min(warmup, 8), min(60, steps), 0)
        if pre is not None:
            pre.close()
        if perr is not None:
            print("bench.py: rank %d: the pre-heating leg failed: %r" % (S.rank, perr), file=sys.stderr)
    # The secondary legs run FIRST: each is a measurement of its own and leaves the device at its working clocks for the headline
    # leg, which otherwise starts from an idle-clocked GPU and, at the 25 frames of a `--steps 20 --warmup 5` run, ends before the
    # clocks have settled.
    fps_total = fps_total_error = None
    if full and not args.no_total:      # the same loop plus the three renders and their read-back
        fps_total, fps_total_error = secondary(lambda: mk(True), "render")
    render_prof = None
    if full and not args.no_total:      # a short leg of its own with an event pair around every render kernel (it would cost the leg above ~5 %)
        _, rleg, rerr = timed_run(S, lambda: mk(True), multi, min(warmup, 8), min(40, steps), -1)
        if rleg is not None:
            if rerr is None and not S.dry:
                render_prof = rleg.kf.read_profile()
            rleg.close()
        if rerr is not None:
            print("bench.py: rank %d: the render-profile leg of %s failed: %r" % (S.rank, config_name, rerr), file=sys.stderr)
    fps_host_input = fps_host_input_error = None
    if not (full and args.no_total):    # the queued loop fed from pageable host memory (PCIe inside the loop; never `value`)
        fps_host_input, fps_host_input_error = secondary(lambda: mk(False, host_input=True), "host-input")

    elapsed, leg, err = timed_run(S, lambda: mk(False, spg - 1, host_input=full and args.host_input), multi, warmup, steps, prof_every)
    failed, errs = 0, [0.0]
    prof = updated = icp_iters = icp_by_level = fallbacks = tracked = integrated = icp_launch_count = None
    try:
        if err is not None:
            raise err
        if not S.dry:
            kf = leg.kf
            prof = kf.read_profile()
            updated = kf.integrate_updated()
            icp_iters = kf.icp_iterations()
            icp_by_level = kf.icp_level_iterations()
            icp_launch_count = kf.icp_launches_per_frame()      # as THIS context queues them (its shared-launch width decides)
            if kf.barrier_timeouts():
                raise RuntimeError("k_icp_tail: %d workgroups timed out at the grid barrier - the run is invalid" % kf.barrier_timeouts())
            fallbacks = kf.solve_fallbacks()
            poses, tracked, integrated = kf.frame_log(n_frames)
            leg.close()
            # ATE exactly as checkPos.py scores a log: row n holds the position before frame n (benchmark.cpp:127-131)
            if wl.gt_rows is not None:
                pos_after = poses[:, :3, 3] - wl.init_pos.astype(np.float32)
                rows = np.vstack([np.zeros((1, 3), np.float32), pos_after[:-1]])
                errs = checkpos.ate([tuple(r) for r in rows], wl.gt_rows)
    except Exception as e:       # noqa: BLE001
        failed = 1
        print("bench.py: rank %d failed (%s): %r" % (S.rank, config_name, e), file=sys.stderr)

    # whole-job aggregate: frames summed over ranks / max time over ranks (one small all-reduce on RCCL)
    job = multi.aggregate(multi.RankStats(steps * spg, elapsed, float(np.sum(errs)), float(np.max(errs)),
                                          0 if tracked is None else int((tracked == 0).sum()), failed), dist, device=S.coll_device)
    if job.failed:
        raise SystemExit("bench.py: %d rank(s) failed - no result" % job.failed)
    t_max, total_frames = job.seconds, job.frames
    if S.rank != 0:
        return None
    if S.dry:
        return {"value": None, "ms_per_step": None, "timed_s": round(t_max, 6), "frames": total_frames, "dry_run": True}

    ate = None if wl.gt_rows is None else {"mean": float(np.mean(errs)), "max": job.ate_max, "frames_rank0": len(errs),
                                           "untracked_all_ranks": job.untracked}
    cpu, v_distinct = (None, None)
    if full and not args.no_cpu_baseline and S.world == 1:
        cpu, v_distinct = cpu_baseline(wl, cfg, args.cpu_seconds, wl.depth_mm.shape[0])
    elif not args.no_cpu_baseline:
        # no CPU baseline here (rank 0 at N = 1, headline configuration only), but raycast's algorithmic bytes still need V: the oracle
        # runs just far enough to have a surface to cast at (6 frames) and counts on the two that follow
        _, v_distinct = cpu_baseline(wl, cfg, 0.0, min(n_frames + V_FRAMES, 6 + V_FRAMES), min_frames=6)
    host = None
    d_depth = None                      # (the legs are over: the child processes below get the device to themselves)
    torch.cuda.empty_cache()
    if full and not args.no_host_buffers and S.world == 1:
        host = host_buffers_leg(wl, cfg, n_frames)
    roof = rooflines(cfg, config_name, prof, prof_every, updated, integrated, icp_by_level, n_frames, v_distinct,
                     measure=S.world == 1 and not args.no_measure_traffic, icp_launch_count=icp_launch_count)
    if render_prof:
        # renderDepth 8 P (float depth in, uchar4 out), renderTrack 8 P (the int32 result map in, uchar4 out), renderVolume
        # 4 P + 4 V' (V' = distinct voxels its march samples: taken as the tracking raycast's V - the same rays, a farther far plane).
        # Their read-back is the expensive part of a rendered frame: 4 P bytes per image over PCIe, in the compute stream.
        P = W * H
        alg = {"render_depth": 8.0 * P, "render_track": 8.0 * P, "render_volume": 4.0 * P + 4.0 * (v_distinct or 0.0)}
        for name, bytes_ in alg.items():
            ms, n = render_prof[name]
            if n:
                gbs = bytes_ / (ms / n * 1e-3) / 1e9
                roof[name] = {"kernel": "k_" + name, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": None, "avg_us": round(ms / n * 1e3, 2), "launches": n,
                              "algorithmic_bytes_per_launch": round(bytes_)}
        if fps_total:
            launched = {k: render_prof[k][1] for k in alg}
            frames_r = max(1, launched["render_depth"])
            kern_us = sum(render_prof[k][0] for k in alg) / frames_r * 1e3
            images = sum(launched.values()) / frames_r
            roof["render_readback"] = {"what": "the rendered images of a queued frame into pinned staging: one copy kernel per frame in the compute stream, 16-byte stores over PCIe",
                                       "bytes_per_frame": round(images * 4.0 * P), "images_per_frame": round(images, 2),
                                       "render_kernels_us_per_frame": round(kern_us, 2),
                                       "frame_us_with_renders": round(1e6 / fps_total, 2), "frame_us_without": round(t_max / (steps * spg) * 1e6, 2),
                                       "readback_and_queue_breaks_us_per_frame": round(1e6 / fps_total - t_max / (steps * spg) * 1e6 - kern_us, 2)}
    res = {
        "workload": workload_string(config_name, cfg, wl),
        "value": round(total_frames / t_max, 2), "ms_per_step": round(t_max / (steps * spg) * 1e3, 4), "steps": steps, "warmup": warmup,
        "timed_s": round(t_max, 6),
        "value_host_input": None if fps_host_input is None else round(fps_host_input, 2),
        "phase_us": {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in prof.items() if v[1]},
        "ate_m": ate, "data": wl.data,
        "icp_iterations_per_frame": round(icp_iters / n_frames, 2), "icp_solve_fallbacks": fallbacks,
        "roof": roof, "irate": irate, "prof": prof, "spg": spg,
        "fps_total": None if fps_total is None else round(fps_total, 2), "fps_total_error": fps_total_error,
        "fps_host_input_error": fps_host_input_error, "fps_host_buffers": host, "cpu_baseline": cpu,
    }
    return res


def run(S, args, multi):
    seen = multi.ranks_seen(S.dist, S.coll_device)
    if seen != args.gpus:
        raise SystemExit("bench.py: the all-reduce over the process group counted %d ranks, --gpus says %d" % (seen, args.gpus))
    # who ran what and where, by rank: trajectory index of the rank's stream (4 distinct trajectories, reused x2 on 8 GPUs:
    # BASELINE.json configs[4]) and the host cores the rank was confined to
    pin = getattr(S, "pin", None) or {"pinned": False, "cpus": None, "numa_node": None, "why": None}
    from slambench_amd import affinity as _aff
    rows = multi.gather_rows([multi.stream_variant(S.rank), 1.0 if pin["pinned"] else 0.0, -1.0 if pin["numa_node"] is None else pin["numa_node"],
                              len(_aff.parse_cpulist(pin["cpus"])) if pin["cpus"] else 0], S.dist, S.coll_device)
    placement = {"trajectories": [int(r[0]) for r in rows], "ranks_pinned": int(sum(r[1] for r in rows)),
                 "numa_node_by_rank": [None if r[2] < 0 else int(r[2]) for r in rows], "cpus_by_rank": [int(r[3]) for r in rows],
                 "rank0_cpu_affinity": pin["cpus"], "rank0_note": pin["why"], "collective_backend": getattr(S, "backend", None),
                 "scaling_note": "weak scaling of independent RGB-D streams: one stream and one TSDF volume per GPU, no per-frame exchange"}
    head = measure(S, args, multi, args.config, args.steps, args.warmup, True)
    extra_names = args.extra_configs
    if extra_names is None:
        extra_names = "cfg3,cfg4" if (args.config == "cfg2" and args.streams_per_gpu == 1 and not args.host_input) else ""
    extra = {}
    for name in [n for n in extra_names.split(",") if n and n != "none"]:
        if name not in CONFIGS:
            raise SystemExit("bench.py: --extra-configs: unknown configuration %r" % name)
        r = measure(S, args, multi, name, max(1, args.extra_steps), 8, False)
        if r is not None:
            extra[name] = r
    if S.rank != 0:
        return

    def compact(r):      # what a leg under "configs" carries
        if r.get("dry_run"):
            return r
        return {"workload": r["workload"], "value": r["value"], "unit": "frames/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                "warmup": r["warmup"], "timed_s": r["timed_s"], "value_host_input": r["value_host_input"], "value_host_input_error": r["fps_host_input_error"],
                "n_gpus": S.world, "data": r["data"], "phase_us": r["phase_us"], "ate_m": r["ate_m"],
                "icp_iterations_per_frame": r["icp_iterations_per_frame"], "roofline": r["roof"]}

    W, H, vres = CONFIGS[args.config][:3]
    if S.dry:
        out = {"metric": "DRY RUN (--dry-collectives): rendezvous, barriers and all-reduces only, no frame was computed", "value": None,
               "unit": "frames/s", "n_gpus": S.world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none", "dry_run": True,
               "config": {"workload": args.config}, "timed_s": head["timed_s"], "frames_all_ranks": head["frames"], "placement": placement,
               "configs": {k: compact(v) for k, v in extra.items()}}
        print(json.dumps(out))
        return
    roof, prof, irate = head["roof"], head["prof"], head["irate"]
    frame_kernels = [k for k in ("icp", "integrate", "raycast") if k in roof]      # (the render entries are never the headline's roofline)
    dominant = max(frame_kernels, key=lambda k: prof[k][0] / max(prof[k][1], 1) * (1.0 if k != "integrate" else 1.0 / max(1, irate))) if frame_kernels else None
    out = {
        "metric": "frames/sec, KinectFusion computation (preprocess+track+integrate+raycast), %s, queued frames, "
                  "living-room stream %%dx%%d -c 1 @%%d^3" % ("PAGEABLE HOST INPUT (--host-input)" if args.host_input else "device-resident input") % (W, H, vres),
        "value": head["value"], "unit": "frames/s", "n_gpus": S.world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "timed_s": head["timed_s"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": head["data"],
        "config": {"workload": head["workload"],
                   "frames_per_gpu": args.steps * head["spg"], "streams_per_gpu": head["spg"], "icp_iterations_per_frame": head["icp_iterations_per_frame"],
                   "icp_solve_fallbacks": head["icp_solve_fallbacks"]},
        # the same queued loop with every frame handed over as pageable host memory - what Kfusion::computeFrame(const ushort*) means
        # (kernels.h:158-162); PCIe inside the loop, so by the bench contract it is reported beside `value`, never as it
        "value_host_input": head["value_host_input"], "fps_queued_host_input": head["value_host_input"],
        "fps_queued_host_input_error": head["fps_host_input_error"],
        "fps_total": head["fps_total"], "fps_total_error": head["fps_total_error"],
        "fps_host_buffers": head["fps_host_buffers"],
        "ate_m": head["ate_m"],
        "phase_us": head["phase_us"],
        "roofline": roof.get(dominant),
        "roofline_other": {k: v for k, v in roof.items() if k != dominant},
        "cpu_baseline": head["cpu_baseline"],
        "placement": placement,
        "configs": {k: compact(v) for k, v in extra.items()},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
