#!/usr/bin/env python3
"""bench.py -- throughput of the Tracker hot path on MI355X (particle-frames/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3]

One "step" = one frame update (evolve -> project -> search tile -> SSD -> spline -> weights ->
resample -> moments, track/tracker.py:331-357) for ALL tracked points of this GPU, on synthetic
frames that are already resident in HBM.  Metric (BASELINE.json): particle-frames/s
= points x particles x steps / wall, summed over GPUs (weak scaling: every GPU tracks its own
shard of points, no data-path collective; one RCCL gather of the posterior moments at the end).

Prints ONE JSON line on rank 0 (see the keys in DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3    # vector FP32 peak (= FP32-input MFMA rate on gfx950)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--points", type=int, default=None, help="override points per GPU")
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--burn-in", type=int, default=6,
                    help="untimed frames after initialisation, before the warm-up: the particle cloud starts "
                         "from its wide prior and reaches the tracking regime after a few updates")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo gathers host copies and "
                         "exists to exercise the multi-rank path where ranks share one GPU)")
    ap.add_argument("--max-search-dim", type=int, default=320, help="search-tile workspace side (pixels)")
    ap.add_argument("--frames-per-call", type=int, default=0,
                    help="frame updates per library call: 0 = all the timed steps in one glh_track call (the frame loop "
                         "of tracker.py:326-357 enqueued at once), 1 = one glh_step call per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    ap.add_argument("--cpu-workers", type=int, default=min(16, os.cpu_count() or 1),
                    help="processes of the parallel CPU baseline (the reference's parallel=True); 1 disables it")
    ap.add_argument("--seed", type=int, default=1234)
    return ap.parse_args()


class DevArray:
    """Zero-copy view of a library-owned device buffer for torch (RCCL gather)."""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


KERNEL_OF_STAGE = {"point_step": "k_point_step", "evolve_project": "k_evolve_project", "resample": "k_resample",
                   "weights": "k_weights", "ssd": "k_ssd", "tileprep": "k_tileprep", "spline_fit": "k_spline_fit"}


def pmc_traffic(wl, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    written by tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command),
    or None when no counters were collected for this workload shape."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    key = f"{wl.name}:{wl.P}x{wl.N}"
    entry = table.get(key, {}).get(kernel)
    return None if entry is None else entry.get("hbm_bytes_per_launch")


def algorithmic_bytes_per_step(P, N, O, tile, boxes, status):
    """P*(96 N) state + per observer (Ws*Hs + 20*tw*th + 96) per point."""
    tw, th = tile
    total = 96.0 * N * P
    for o in range(O):
        ok = status[o] == 0
        ws = (boxes[o, :, 2] - boxes[o, :, 0])[ok].astype(np.float64)
        hs = (boxes[o, :, 3] - boxes[o, :, 1])[ok].astype(np.float64)
        total += float((ws * hs).sum()) + ok.sum() * (20.0 * tw * th + 96.0)
    return total


def ssd_flops_per_step(O, tile, boxes, status):
    """3*tw*th*Wo*Ho FP32 per point-frame-observer."""
    tw, th = tile
    total = 0.0
    for o in range(O):
        ok = status[o] == 0
        wo = (boxes[o, :, 2] - boxes[o, :, 0] - tw + 1)[ok].astype(np.float64)
        ho = (boxes[o, :, 3] - boxes[o, :, 1] - th + 1)[ok].astype(np.float64)
        total += 3.0 * tw * th * float((wo * ho).sum())
    return total


def cpu_baseline(wl, frames, steps, target_seconds):
    """The oracle (CPU port of the reference path) on a bounded sample of the same workload."""
    from oracle import motion as omotion
    from oracle import tracker as otracker

    observers = [otracker.Observer(frames[o], np.tile(wl.cams[o], (len(frames[o]), 1)), wl.sigmas[o])
                 for o in range(wl.O)]
    nfr = 1 + steps
    matching = np.tile(np.arange(nfr)[:, None], (1, wl.O))
    taus = np.ones(nfr - 1)

    def model(p):
        q = wl.params[p]
        return omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13],
                                       axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=wl.N)

    np.random.seed(7)
    t0 = time.perf_counter()
    otracker.track([model(0)], observers, matching, taus, tile_size=wl.tile)
    per_point = time.perf_counter() - t0
    n_pts = int(max(1, min(wl.P - 1, round(target_seconds / max(per_point, 1e-3)))))
    first = 1 if wl.P > 1 else 0  # (a one-point workload, C1, times its only point again)
    t0 = time.perf_counter()
    otracker.track([model(first + p) for p in range(n_pts)], observers, matching, taus, tile_size=wl.tile)
    dt = time.perf_counter() - t0
    cpu_baseline.per_point_seconds = dt / n_pts
    return {
        "value": n_pts * wl.N * steps / dt,
        "unit": "particle-frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n_pts} of {wl.P} points x {wl.N} particles x {steps} steps of the same workload, "
                  f"oracle/ (NumPy {np.__version__} + SciPy + C SSD), {dt:.1f} s on 1 of {os.cpu_count()} host cores "
                  f"({cpu_model()})",
    }


def _cpu_worker(job):
    """One host process of the parallel CPU baseline (spawned: it never touches the GPU).  Rebuilds its
    inputs from the workload recipe, then tracks its block of points with the oracle."""
    name, n_points, n_particles, n_frames, lo, hi = job
    from glimpse_amd import workloads
    from oracle import motion as omotion
    from oracle import tracker as otracker

    wl = workloads.Workload(name, n_frames=n_frames, n_points=n_points, n_particles=n_particles, shard=0, seed=0)
    frames = [[wl.frame(o, t) for t in range(n_frames)] for o in range(wl.O)]
    observers = [otracker.Observer(frames[o], np.tile(wl.cams[o], (n_frames, 1)), wl.sigmas[o]) for o in range(wl.O)]
    matching = np.tile(np.arange(n_frames)[:, None], (1, wl.O))
    models = []
    for p in range(lo, hi):
        q = wl.params[p]
        models.append(omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10],
                                              axyz=q[10:13], axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=wl.N))
    np.random.seed(100 + lo)
    t0 = time.perf_counter()
    otracker.track(models, observers, matching, np.ones(n_frames - 1), tile_size=wl.tile)
    return time.perf_counter() - t0


def cpu_baseline_parallel(wl, steps, per_point_seconds, target_seconds, workers):
    """The reference's `parallel=True` (one process per block of tracks, tracker.py:381-387, helpers.py:2008-2017)
    restated with the oracle: `workers` spawned processes, each tracking its own block of points."""
    import multiprocessing as mp

    per_worker = int(max(1, round(target_seconds / max(per_point_seconds, 1e-3))))
    per_worker = min(per_worker, max(1, wl.P // workers))
    jobs = [(wl.name, wl.P, wl.N, 1 + steps, w * per_worker, (w + 1) * per_worker) for w in range(workers)]
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        times = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    busy = max(times)  # the tracking itself; `wall` also holds interpreter start-up and frame rendering
    n_pts = workers * per_worker
    return {
        "value": n_pts * wl.N * steps / busy,
        "unit": "particle-frames/s",
        "cores": workers,
        "kind": "port",
        "sample": f"{workers} processes x {per_worker} points x {wl.N} particles x {steps} steps (oracle/, one block of "
                  f"points per process like the reference's parallel=True); slowest process {busy:.1f} s, {wall:.1f} s "
                  f"with start-up, of {os.cpu_count()} host cores",
    }


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = int(os.environ.get("GLH_BENCH_DEVICE", local_rank))  # test hook: several ranks on one GPU
    use_nccl = args.dist_backend == "nccl"
    dist = None
    if world > 1 or os.environ.get("GLH_BENCH_FORCE_DIST") == "1":  # (test hook: the collective path with one rank)
        import torch
        import torch.distributed as dist

        if use_nccl:
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    from glimpse_amd import _lib, workloads

    K, W, B = args.steps, args.warmup, max(0, args.burn_in)
    T = 1 + B + W + K
    wl = workloads.Workload(args.workload, n_frames=T, n_points=args.points, n_particles=args.particles, shard=rank,
                            seed=0)
    frames = [wl.frames(o) for o in range(wl.O)]
    ctx = _lib.Context(wl.P, wl.N, wl.O, device_id=device, max_tile=max(wl.tile), max_search_dim=args.max_search_dim,
                       max_frames=T)
    workloads.setup_context(ctx, wl, frames)
    # one seed for the whole job: the device RNG is keyed on the GLOBAL point index, so the
    # sharded run draws what a single-GPU run of all points would draw
    ctx.set_point_offset(rank * wl.P)
    seed = args.seed
    images = lambda i: [i] * wl.O  # noqa: E731

    # frame 0: initialise particles + templates (tracker.py:327-342), untimed
    ctx.set_frame(0)
    ctx.init_particles(seed=seed)
    for o in range(wl.O):
        ctx.init_templates(o, 0)
    ctx.record_moments(0)
    F = K if args.frames_per_call <= 0 else min(args.frames_per_call, K)

    def run(first, count):
        """`count` consecutive frame updates from frame `first`: glh_track calls of up to F frames each (the frame
        loop of tracker.py:326-357, one kernel launch per frame), or glh_step per frame when F == 1."""
        i = first
        while i < first + count:
            n = min(F, first + count - i)
            if n == 1:
                ctx.step(i, 1.0, images(i), seed=seed)
            else:
                ctx.track(list(range(i, i + n)), [1.0] * n, [images(j) for j in range(i, i + n)], seed=seed)
            i += n

    run(1, B)  # burn-in, untimed
    run(1 + B, W)  # warm-up, untimed
    ctx.sync()

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch

            dist.barrier()
            if use_nccl:
                torch.cuda.synchronize()

    gather_list = None
    mom = None

    def gather_moments():
        """The one collective of a sequence: every rank's posterior moments [T][P][12] to rank 0."""
        import torch

        nonlocal gather_list, mom
        if use_nccl:
            if mom is None:  # zero-copy view of the library-owned history buffer
                ptr, nbytes = ctx.moments_device()
                mom = torch.as_tensor(DevArray(ptr, (T, wl.P, 12)), device=f"cuda:{device}")
            send = mom
        else:
            send = torch.from_numpy(ctx.get_moments(0, T))
        if rank == 0 and gather_list is None:
            gather_list = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list, dst=0)

    if dist is not None:
        gather_moments()  # warm the communicator up outside the timed region

    # HIP events around every kernel launch on the context's stream, over the timed region itself
    ctx.profile_enable(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    run(1 + B + W, K)
    if dist is not None:
        ctx.sync()
        gather_moments()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], device=f"cuda:{device}" if use_nccl else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms = ctx.profile_get()
    ctx.profile_enable(False)

    # health of the run: every point must still be tracked by every observer
    status = ctx.observer_status()
    pt_status = ctx.point_status()
    frac_ok = float((status == 0).mean())
    n_err = int((pt_status != 0).sum())
    # algorithmic bytes / SSD flops of one step, from the search boxes of the last timed step
    boxes = ctx.search_boxes()
    abytes = algorithmic_bytes_per_step(wl.P, wl.N, wl.O, wl.tile, boxes, status)
    flops = ssd_flops_per_step(wl.O, wl.tile, boxes, status)

    gathered_ok = None
    if dist is not None and rank == 0:
        # every shard's moments arrived and are finite for the timed frames
        import torch

        allm = torch.stack([g.cpu() for g in gather_list])  # (world, T, P, 12)
        gathered_ok = bool(torch.isfinite(allm[:, 1 + B + W:1 + B + W + K]).all())
    if rank == 0:
        value = world * wl.P * wl.N * K / elapsed
        out = {
            "metric": "particle-frames/s",
            "value": value,
            "unit": "particle-frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": dict(wl.describe(), rng="device Philox4x32-10", parallelism=f"points sharded x{world}",
                           frames_per_s=K / elapsed, burn_in_steps=B, frames_per_call=F),
            "health": {"observer_ok_fraction": frac_ok, "points_with_error_bits": n_err,
                       "gathered_moments_finite": gathered_ok},
        }
        tot = sum(ms for ms, _ in stage_ms.values())
        dom = max(stage_ms, key=lambda k: stage_ms[k][0])
        dom_ms, dom_n = stage_ms[dom]
        per_launch_ms = dom_ms / max(dom_n, 1)
        launches_per_step = dom_n / K
        # one launch of the dominant kernel processes P*N particle-frames (SURVEY 8(d) per-unit bytes)
        ach = abytes / launches_per_step / (per_launch_ms * 1e-3) / 1e9
        roof = {"kernel": KERNEL_OF_STAGE.get(dom, dom), "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(wl, KERNEL_OF_STAGE.get(dom, dom)),
                "avg_launch_ms": per_launch_ms, "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": abytes / launches_per_step,
                "algorithmic_bytes_per_particle_frame": abytes / (wl.P * wl.N),
                "ssd_fp32_tflops": flops / (tot / K * 1e-3) / 1e12,
                # the measured device-copy ceiling of this GPU beside the 8 TB/s spec
                "measured_copy_GBps": ctx.copy_bandwidth(1 << 30, 10)}
        roof["frac_of_measured_copy"] = ach / roof["measured_copy_GBps"]
        out["roofline"] = roof
        out["stage_ms_per_step"] = {k: ms / K for k, (ms, _) in stage_ms.items() if ms > 0}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, frames, min(K, 4), args.cpu_seconds)
            if args.cpu_workers > 1:
                out["cpu_baseline_parallel"] = cpu_baseline_parallel(
                    wl, min(K, 4), cpu_baseline.per_point_seconds, args.cpu_seconds, args.cpu_workers)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
