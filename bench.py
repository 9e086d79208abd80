#!/usr/bin/env python3
"""Throughput bench of the KinectFusion hot path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One *step* is one frame through preprocessing -> tracking (multi-level ICP) -> integration -> raycasting —
the reference's `computation` column (kfusion/src/benchmark.cpp:164), the figure its README quotes — on the
configuration BASELINE.json's metric is quoted on: 640x480 depth, `-c 1 -v 256 -y 10,5,4` and the reference
Makefile's trajectory-2 flags `-s 4.8 -p 0.34,0.5,0.24 -k 481.2,480,320,240 -r 2`.  Depth frames are a seeded
synthetic stream (no dataset offline) already resident in HBM when the timed region starts.  With N GPUs every
rank runs its own independent stream (frames shard by stream; no data-path collective) and RCCL is used only
for the barrier and the max-over-ranks time.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries `roofline` (dominant kernel,
algorithmic bytes / HIP-event time measured in this run), `cpu_baseline` (the OpenMP oracle timed on the host
cores on a bounded prefix of the same stream), `fps_total` (same loop plus the three renders and their
read-back) and the ATE of the estimated trajectory against the generator's ground truth.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

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


def icp_launches(pyramid, head=4):
    """launches of a frame's ICP loop: per level the first `head` iterations one each, the rest (if >= 2) share one
    (kf_pipeline.hip tail_length); the level-0 tail also closes the loop, otherwise k_icp_final does"""
    n, closed = 0, False
    for level, its in enumerate(pyramid):
        tail = its - head if its - head >= 2 else 0
        n += its - tail + (1 if tail else 0)
        closed = closed or (tail > 0 and level == 0)
    return n + (0 if closed else 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-total", action="store_true", help="skip the secondary loop with renders")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="HIP-event pairs around icp/integrate/raycast on every n-th timed frame (0 = none; 1 = every "
                         "frame, which costs ~23 us per frame in queue drains)")
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="independent RGB-D streams processed concurrently on each GPU (throughput mode; the headline is 1)")
    return ap.parse_args()


def run_gpu(kfusion, torch, stream, d_depth, cfg, device, n_warm, n_steps, with_renders, dist, extra_streams=0, prof_every=8):
    """Warm up on frames [0, n_warm), time frames [n_warm, n_warm + n_steps).  Returns (seconds, kf).
    extra_streams > 0 runs that many more independent contexts (same input stream) interleaved with the first one."""
    W, H, vres, pyramid, irate, zrate = cfg
    k = list(stream.k)
    kf = kfusion.Kfusion((W, H), (vres,) * 3, (4.8,) * 3, stream.init_pos, pyramid, device=device)
    others = [kfusion.Kfusion((W, H), (vres,) * 3, (4.8,) * 3, stream.init_pos, pyramid, device=device) for _ in range(extra_streams)]
    frame_bytes = W * H * 2
    base = d_depth.data_ptr()
    renders = [np.zeros((H, W, 4), np.uint8) for _ in range(3)] if with_renders else None
    import ctypes as C

    def step(f):
        kf.compute_frame(base + f * frame_bytes, (W, H), k, irate, TRACKING_RATE, ICP_THRESHOLD, MU, f)
        for o in others:
            o.compute_frame(base + f * frame_bytes, (W, H), k, irate, TRACKING_RATE, ICP_THRESHOLD, MU, f)
        if with_renders:   # benchmark.cpp:153-156
            L, h = kf.L, kf.h
            u8 = C.POINTER(C.c_uint8)
            L.kfh_render_depth(h, renders[0].ctypes.data_as(u8), W, H)
            L.kfh_render_track(h, renders[1].ctypes.data_as(u8), W, H)
            L.kfh_render_volume(h, renders[2].ctypes.data_as(u8), W, H, f, zrate, (C.c_float * 4)(*k), 0.75 * MU)

    for f in range(n_warm):
        step(f)
    kf.synchronise()
    for o in others:
        o.synchronise()
    if not with_renders and prof_every > 0:
        kf.set_profiling(["icp", "integrate", "raycast"], every=prof_every)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for f in range(n_warm, n_warm + n_steps):
        step(f)
    kf.synchronise()
    for o in others:
        o.synchronise()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for o in others:
        o.close()
    if dist is not None:
        dist.barrier()
    return elapsed, kf


def cpu_baseline(stream, cfg, budget_s, max_frames):
    """OpenMP oracle on the host cores over a bounded prefix of the same stream (computation column only)."""
    from oracle import orc
    orc.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # cgroup CPU quota of the box (a one-GPU share is 16 CPUs)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per))))
    except Exception:
        pass
    os.environ["OMP_NUM_THREADS"] = str(cores)
    W, H, vres, pyramid, irate, _ = cfg
    k = list(stream.k)
    o = orc.OracleKfusion((W, H), (vres,) * 3, (4.8,) * 3, stream.init_pos, pyramid, omp=True)
    n, spent = 0, 0.0
    while n < max_frames and (spent < budget_s or n < 8):
        t = time.perf_counter()
        o.preprocessing(stream.depth_mm[n]); o.tracking(k, ICP_THRESHOLD, TRACKING_RATE, n)
        o.integration(k, irate, MU, n); o.raycasting(k, MU, n)
        spent += time.perf_counter() - t
        n += 1
    # V of SURVEY §8(d) (distinct voxels a raycast samples) — counted on two more frames, outside the timing
    o.set_counting(True)
    vs = []
    for f in range(n, min(n + 2, max_frames)):
        o.preprocessing(stream.depth_mm[f]); o.tracking(k, ICP_THRESHOLD, TRACKING_RATE, f)
        o.integration(k, irate, MU, f); o.raycasting(k, MU, f)
        vs.append(o.counters()["raycast_distinct_voxels"])
    o.close()
    return {"value": n / spent, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "first %d frames of the same stream, OpenMP oracle (liborc_omp.so), %d threads, %.1f s" % (n, cores, spent)}, \
        (float(np.mean(vs)) if vs else None)


def main():
    args = parse()
    import torch
    from slambench_amd import checkpos, kfusion, multi, synth

    world, rank, local_rank = multi.env_rank()
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal on a one-GPU box: KFH_BENCH_BACKEND=gloo KFH_BENCH_ONE_DEVICE=1 runs the ranks side by side on cuda:0
        # with the collective on the host (RCCL refuses two ranks on one device); the driver's runs use neither
        backend = os.environ.get("KFH_BENCH_BACKEND", "nccl")
        if os.environ.get("KFH_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    coll_device = "cuda" if (dist is None or os.environ.get("KFH_BENCH_BACKEND", "nccl") == "nccl") else "cpu"

    cfg = CONFIGS[args.config]
    if os.environ.get("KFH_DEBUG_PYRAMID"):    # experiments only (cost of the queued-but-empty ICP launches)
        cfg = cfg[:3] + (tuple(int(v) for v in os.environ["KFH_DEBUG_PYRAMID"].split(",")),) + cfg[4:]
    W, H, vres, pyramid, irate, zrate = cfg
    n_frames = args.warmup + args.steps
    # one independent RGB-D stream per GPU: 4 distinct trajectories, reused x2 on 8 GPUs (BASELINE configs[4])
    stream = synth.cached_stream(n_frames, W, H, seed=1234, variant=multi.stream_variant(rank))
    d_depth = torch.from_numpy(stream.depth_mm.view(np.int16)).cuda()

    spg = max(1, args.streams_per_gpu)
    elapsed, kf = run_gpu(kfusion, torch, stream, d_depth, cfg, local_rank, args.warmup, args.steps, False, dist, spg - 1,
                          args.profile_every)
    prof = kf.read_profile()
    updated = kf.integrate_updated()
    icp_iters = kf.icp_iterations()
    icp_by_level = kf.icp_level_iterations()
    if kf.barrier_timeouts():
        raise RuntimeError("k_icp_tail: %d workgroups timed out at the grid barrier - the run is invalid" % kf.barrier_timeouts())
    fallbacks = kf.solve_fallbacks()
    poses, tracked, integrated = kf.frame_log(n_frames)
    kf.close()

    # ATE exactly as checkPos.py scores a log: row n holds the position before frame n (benchmark.cpp:127-131)
    pos_after = poses[:, :3, 3] - stream.init_pos.astype(np.float32)
    rows = np.vstack([np.zeros((1, 3), np.float32), pos_after[:-1]])
    errs = checkpos.ate([tuple(r) for r in rows], [tuple(g) for g in synth.ground_truth_rows(stream)])

    # whole-job aggregate: frames summed over ranks / max time over ranks (one small all-reduce on RCCL)
    job = multi.aggregate(multi.RankStats(args.steps * spg, elapsed, float(np.sum(errs)), float(np.max(errs)), int((tracked == 0).sum())),
                          dist, device=coll_device)
    t_max, total_frames = job.seconds, job.frames
    ate = {"mean": float(np.mean(errs)), "max": job.ate_max, "frames_rank0": len(errs), "untracked_all_ranks": job.untracked}

    fps_total = None
    if not args.no_total:
        e2, kf2 = run_gpu(kfusion, torch, stream, d_depth, cfg, local_rank, args.warmup, args.steps, True, dist)
        kf2.close()
        fps_total = multi.aggregate(multi.RankStats(args.steps, e2), dist, device=coll_device).fps

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    cpu, v_distinct = (None, None)
    if not args.no_cpu_baseline and world == 1:
        cpu, v_distinct = cpu_baseline(stream, cfg, args.cpu_seconds, n_frames)

    # HBM bytes per launch from the PMC passes of profiles/collect_traffic.sh (rocprofv3 cannot wrap this process from
    # inside); null when no measurement of this configuration is committed
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        for kname, v in tj.items():
            for short in ("integrate", "raycast"):
                if ("k_" + short) in kname:
                    traffic[short] = v["hbm_bytes_per_launch"]
        # the ICP loop of a frame = all k_icp_iter* / k_icp_tail / k_icp_final launches (+ k_control when it is a launch of
        # its own); frames of the run = launches of k_raycast (one per frame)
        frames_t = sum(v["launches"] for kname, v in tj.items() if "k_raycast" in kname)
        loop_t = sum(v.get("hbm_bytes_total", 0) for kname, v in tj.items() if "k_icp" in kname or "k_control" in kname)
        if frames_t and loop_t:
            traffic["icp"] = round(loop_t / frames_t)
    P = W * H
    roof = {}
    for name in ("integrate", "raycast"):
        ms, n = prof[name]
        if not n:
            continue
        avg_s = ms / n * 1e-3
        if name == "integrate":
            alg = 8.0 * (updated / max(1, int(integrated.sum()))) + 4.0 * P      # 8*U + 4*P  
        else:
            alg = 24.0 * P + 4.0 * (v_distinct if v_distinct else 0.0)            # 24*P + 4*V
        gbs = alg / avg_s / 1e9
        roof[name] = {"kernel": "k_" + name, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic.get(name), "avg_us": round(avg_s * 1e6, 2), "launches": n, "sampled_every_n_frames": args.profile_every,
                      "algorithmic_bytes_per_launch": round(alg)}
    # The ICP loop of a frame (k_icp_iter* launches, k_icp_tail for the late iterations of a level, k_icp_final): 48 bytes per associated pixel (two float4 input
    # maps, a 32-byte gather from the reference maps) for every EXECUTED iteration, plus - on the queued-frame path - the
    # next frame's bilateral filter (2P in, 8P out) and pyramid (4P in, 36 bytes per pixel of every level out) that ride
    # on it.  One "launch" of this entry is one frame's loop.
    ms, n = prof["icp"]
    if n:
        levels = len(pyramid)
        px = [(W >> l) * (H >> l) for l in range(levels)]
        alg = 48.0 * sum(icp_by_level[l] * px[l] for l in range(levels)) / max(1, n_frames)
        alg += 10.0 * P + 4.0 * P + 36.0 * sum(px)
        avg_s = ms / n * 1e-3
        gbs = alg / avg_s / 1e9
        roof["icp"] = {"kernel": "k_icp_iter* + k_icp_tail (one frame's ICP loop: %d queued iterations in %d launches)" % (sum(pyramid), icp_launches(pyramid)), "bound": "hbm",
                       "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                       "traffic": traffic.get("icp"), "avg_us": round(avg_s * 1e6, 2), "launches": n,
                       "sampled_every_n_frames": args.profile_every, "algorithmic_bytes_per_launch": round(alg),
                       "executed_iterations_per_frame_by_level": [round(v / max(1, n_frames), 2) for v in icp_by_level]}
    dominant = max(roof, key=lambda k: prof[k][0] / max(prof[k][1], 1) * (1.0 if k != "integrate" else 1.0 / max(1, irate))) if roof else None
    if spg > 1:
        args.steps_label = "%d frames x %d streams" % (args.steps, spg)
    out = {
        "metric": "frames/sec, KinectFusion computation (preprocess+track+integrate+raycast), living-room-like stream 640x480 -c 1 @256^3",
        "value": round(total_frames / t_max, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t_max / (args.steps * spg) * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%d depth, -c 1 -v %d -y %s -r %d -z %d -s 4.8 -p 0.34,0.5,0.24 -k 481.2,480,320,240 -m 0.1, one stream per GPU"
                               % (args.config, W, H, vres, ",".join(map(str, pyramid)), irate, zrate),
                   "frames_per_gpu": args.steps * spg, "streams_per_gpu": spg, "icp_iterations_per_frame": round(icp_iters / n_frames, 2), "icp_solve_fallbacks": fallbacks},
        "fps_total": None if fps_total is None else round(fps_total, 2),
        "ate_m": ate,
        "phase_us": {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in prof.items() if v[1]},
        "roofline": roof.get(dominant),
        "roofline_other": {k: v for k, v in roof.items() if k != dominant},
        "cpu_baseline": cpu,
    }
    if args.config != "cfg2":
        out["metric"] = out["metric"].replace("640x480 -c 1 @256^3", "%dx%d -c 1 @%d^3" % (W, H, vres))
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
