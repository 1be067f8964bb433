#!/usr/bin/env python3
"""bench.py -- throughput of the Tracker hot path on MI355X (particle-frames/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3]

One "step" = one frame update (evolve -> project -> search tile -> SSD -> spline -> weights ->
resample -> moments, track/tracker.py:331-357) for ALL tracked points of this GPU, on synthetic
frames that are already resident in HBM.  Metric (BASELINE.json): particle-frames/s
= points x particles x steps / wall, summed over GPUs.

Default workload = BASELINE config 3 as it is worded: 4096 points x 5000 particles tracked through a
100-frame sequence FROM THE PRIOR: frame 0 initialises particles and templates, every later frame is timed (the
first updates work on the wide prior cloud and run longer than the steady state, which is reported beside the
headline as `steady_ms_per_frame`).  The timed region is always that whole sequence: without --steps it is 99 steps
of one frame update; with `--steps K` it is K steps of F = round(99 / K) consecutive frame updates each (the driver's
`--steps 20`: 20 steps x 5 updates over a 101-frame sequence) -- a step is one pass of the hot path over one batch of
input, here F frames for all points.  The W warm-up steps run the first W steps of the same sequence, untimed; the
state is then re-initialised (untimed) and the K timed steps start from the prior.  `--burn-in B` (> 0) instead times
K single-frame steps after B untimed updates (the steady-state window used for A/B runs).  The timed region is repeated
`--repeats` R = 5 times inside the invocation (re-initialised, untimed, in between): `value`, `ms_per_step` and the
roofline figures are those of the MEDIAN repetition, `spread` holds the minimum / maximum / every repetition.

N > 1: one process per GPU, each tracking its own block of points (weak scaling; `--split strong` divides the
workload's points instead), no data-path collective, ONE RCCL gather of the posterior moments at the end of
the timed region (glh_gather_moments inside the library; no torch anywhere).  `--gpus N` without a launcher
starts the N ranks itself (children are started before anything touches a GPU); under torchrun
(RANK / WORLD_SIZE set) every process is one rank.

The default run (C3, one GPU) adds short SECONDARY legs after the headline, each over its whole sequence from the prior
on the frames already in memory: exact arithmetic at C3 (the arithmetic the oracle tests pin bit for bit), one GPU's
shard of C4 (1 250 x 10 000), C5 (2 048 x 5 000, two observers + DEM term), C2, and C3 on RGB frames -- `secondary` in
the JSON line.

Prints ONE JSON line on rank 0 (keys: DESIGN.md "Measurement") and exits non-zero when the run is unhealthy.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6290.0  # float4 copy kernel, same guide ("6.29 TB/s measured")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: one per frame update of the whole sequence).  With --burn-in 0 the "
                         "timed region is ALWAYS the configuration's whole sequence from the prior: K steps of F "
                         "consecutive frame updates each, F = round((frames - 1) / K) (--steps 20 on the 100-frame C3: 20 "
                         "steps x 5 updates)")
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="frame updates per timed step (0 = automatic, see --steps; 1 with --burn-in > 0)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--points", type=int, default=None, help="override points per GPU")
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--burn-in", type=int, default=0,
                    help="untimed frame updates between initialisation and the timed steps (0 = time the sequence "
                         "from the prior, as BASELINE words it; 6 = steady-state only)")
    ap.add_argument("--split", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU tracks the configuration's per-GPU share; strong: the configuration's "
                         "points are divided over the GPUs (C4: 10 000 points / N)")
    ap.add_argument("--transport", default=None, choices=["rccl", "host"],
                    help="collective transport for N > 1 (default: RCCL when every rank can make the communicator, "
                         "else host copies through the rendezvous directory -- reported in the JSON line)")
    ap.add_argument("--max-search-dim", type=int, default=None,
                    help="search-tile workspace side (pixels); default 320, 255 for --bits 16 (what the fused step takes)")
    ap.add_argument("--frames-per-call", type=int, default=0,
                    help="frame updates per library call: 0 = all the timed steps in one glh_track call (the frame loop "
                         "of tracker.py:326-357 enqueued at once), 1 = one glh_step call per frame")
    ap.add_argument("--streams", type=int, default=0, choices=(0, 1, 2, 3, 4),
                    help="streams of glh_track's frame loop (glh_set_track_streams): 0 = the library's choice (two for "
                         "batches of more points than the chip has compute units), 1 = one launch per frame, 2 = two "
                         "half-batches on two streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the glimpse_amd.Tracker.track() leg (api_ms_per_step)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="processes of the parallel CPU baseline (the reference's parallel=True = os.cpu_count(), "
                         "helpers.py:2008-2011); 0 = every core this process may use, 1 disables it")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--motion", default="cartesian",
                    choices=["cartesian", "cylindrical", "tangent_cartesian", "tangent_cylindrical"],
                    help="motion model of every tracked point (motion.py:92-522); the configurations of BASELINE.json "
                         "use CartesianMotion, real glacier runs TangentCartesianMotion")
    ap.add_argument("--dem", default="constant", choices=["constant", "gridded"],
                    help="surfaces of the motion models: the configuration's constants (BASELINE) or a gridded DEM + DEM "
                         "uncertainty (glimpse.Raster, what real runs bring): the same ground, sampled bilinearly per particle")
    ap.add_argument("--math", default="fast", choices=["fast", "exact"],
                    help="arithmetic of the device-RNG run: fast (GLH_MATH_FAST: FMA / reciprocal forms, what "
                         "Tracker.track(rng='philox') uses) or exact (NumPy rounding, what the host-RNG parity mode uses)")
    ap.add_argument("--bits", type=int, default=8, choices=[8, 16, 32],
                    help="frame samples: uint8 (BASELINE's configurations) or uint16 (the same scene on a 16-bit sensor)")
    ap.add_argument("--channels", type=int, default=1, choices=[1, 3],
                    help="frame channels: 1 (gray, BASELINE's configurations) or 3 (RGB uint8: what time-lapse JPEGs decode to)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (the K steps) is run this many times inside the one invocation, the state "
                         "re-initialised (untimed) in between; `value` / `ms_per_step` are the MEDIAN repetition's, "
                         "`spread` lists min / max / n (boxes of the pool differ by 3-5 %%, one 44-ms sample says little)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs of the default run (exact arithmetic at C3, the C4 shard, C5, C2)")
    ap.add_argument("--dump-moments", default=None, help="rank 0 saves the (gathered) posterior history (T, P, 12) here (.npy)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: `--gpus N` without torchrun
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args, argv=None, poll=0.05):
    """Start N fresh rank processes (this parent never touches a GPU), relay rank 0's JSON line, fail if any
    child fails.  A rank that dies takes the job down at once: the parent marks the rendezvous directory as aborted
    (ranks waiting in FileStore.get raise instead of polling until their timeout), stops the others and returns 1."""
    import tempfile
    import threading

    from glimpse_amd import sharding

    n = args.gpus
    port = _free_port()
    argv = sys.argv[1:] if argv is None else list(argv)
    procs = []
    out0 = tempfile.TemporaryFile()
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if rank == 0 else subprocess.DEVNULL))
    store_path = sharding.FileStore.default_path(f"{port}_{os.getpid()}")  # (Group.from_env: port + launcher's pid)
    codes = [None] * n
    failed = False
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if not failed and any(c not in (None, 0) for c in codes):
            failed = True
            os.makedirs(store_path, exist_ok=True)
            sharding.FileStore(store_path, -1, n).abort(f"ranks failed: {[(r, c) for r, c in enumerate(codes) if c]}")
            # the survivors see the abort mark at their next rendezvous; ranks stuck elsewhere are stopped after a grace
            # period (our own children, by pid)
            killer = threading.Timer(float(os.environ.get("GLH_LAUNCH_GRACE", "10")),
                                     lambda: [p.kill() for p in procs if p.poll() is None])
            killer.daemon = True
            killer.start()
        time.sleep(poll)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        import shutil

        shutil.rmtree(store_path, ignore_errors=True)
        print(f"bench.py: ranks failed: {bad}", file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------
# host-side helpers
# ------------------------------------------------------------------------------------------------
KERNEL_OF_STAGE = {"point_step": "k_point_step", "evolve_project": "k_evolve_project", "resample": "k_resample",
                   "weights": "k_weights", "ssd": "k_ssd", "tileprep": "k_tileprep", "spline_fit": "k_spline_fit"}


_T0 = time.perf_counter()


def _mark(label):
    """Section times of the run on stderr (the driver keeps stdout for the JSON line)."""
    print(f"[bench.py {time.perf_counter() - _T0:7.1f} s] {label}", file=sys.stderr, flush=True)


def usable_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota, whichever is smaller."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


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


def pmc_traffic_per_frame(wl, kernel):
    """HBM bytes of one frame update of all of `wl`'s points from the committed PMC passes: the per-launch figure scaled
    by the points a launch of the profiled run held (two streams: half of them), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(f"{wl.name}:{wl.P}x{wl.N}", {}).get(kernel)
    except (OSError, ValueError):
        return None
    if entry is None or "hbm_bytes_per_launch" not in entry:
        return None
    return entry["hbm_bytes_per_launch"] * wl.P / float(entry.get("points_per_launch", wl.P))


def pmc_traffic_source(wl, kernel):
    """Where `roofline.traffic` comes from: NOT this run -- the PMC passes are separate rocprofv3 runs of the same
    command on the builder's GPU box, committed under profiles/."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    entry = table.get(f"{wl.name}:{wl.P}x{wl.N}", {}).get(kernel)
    if entry is None:
        return None
    return ("committed profile, not measured in this run: " + entry.get("source", "profiles/pmc_traffic.json")
            + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the builder's gpurun box"
            + (", " + table["_meta"]["collected"] if "_meta" in table and "collected" in table["_meta"] else "") + ")")


SQ_COUNTER_FILES = ("r05_C3_sq_counters.json", "r04_C3_sq_counters.json", "r03_C3_sq_counters.json")  # (the newest committed pass)


def sq_counters(wl):
    """(VALU wave-instructions per 64 particle-frames of the fused kernel, file) from the committed SQ counter pass of
    this workload shape (C3 only: the pass is expensive), or (None, None)."""
    if (wl.name, wl.P, wl.N, wl.channels, wl.bits) != ("C3", 4096, 5000, 1, 8):
        return None, None
    for name in SQ_COUNTER_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return float(json.load(f)["derived"]["valu_wave_instructions_per_64_particle_frames"]), name
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def algorithmic_bytes_per_step(P, N, O, tile, boxes, status, channels=1):
    """P*(96 N) state + per observer (Ws*Hs*s_img + 20*tw*th + 96) per point (s_img = bytes per pixel)."""
    tw, th = tile
    total = 96.0 * N * P
    for o in range(O):
        ok = status[o] == 0
        ws = (boxes[o, :, 2] - boxes[o, :, 0])[ok].astype(np.float64)
        hs = (boxes[o, :, 3] - boxes[o, :, 1])[ok].astype(np.float64)
        total += channels * float((ws * hs).sum()) + ok.sum() * (20.0 * tw * th + 96.0)  # (channels x bytes per sample)
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


_SHARED = {}  # name -> ndarray over anonymous shared memory, allocated before the helpers fork
_WLS = {}  # name -> Workload (with its texture) the forked helpers render from: inherited, never pickled


def _shared_array(name, shape, dtype=np.uint8):
    import mmap

    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = mmap.mmap(-1, max(n, 1))  # MAP_SHARED | MAP_ANONYMOUS: forked helpers write, this process reads
    _SHARED[name] = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    return _SHARED[name]


def _ground_job(job):
    from glimpse_amd import synth

    if job[0] == "scene":
        return _WLS[job[1]].scene
    _, cam, r0, r1 = job
    return synth.ground_rows(cam, r0, r1)


def _render_job(job):
    kind, name, wl, o, t = job
    if kind == "rgb":  # the RGB frame of an already rendered gray one
        from glimpse_amd import synth

        _SHARED[name][t] = synth.gray_to_rgb(_SHARED[wl][t])
    else:
        _SHARED[name][t] = _WLS[wl].frame(o, t)
    return None


def _may_fork():
    """Forked helpers are only safe while this process has not initialised the GPU runtime; a profiler's preloaded
    library has done that before main() (rocprofv3 --pmc): render serially there."""
    from glimpse_amd import _lib as _l

    profiled = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    return not profiled and _l._lib is None


def render_all(wls, workers, rgb_of=None):
    """Frames of several workloads at once: {name: [O] arrays (T, H, W[, 3]) uint8}.  `wls` is {name: Workload};
    `rgb_of` = (name, source) adds the RGB frames of the one-observer gray workload `source` under `name` (the channel
    remap of synth.gray_to_rgb on its rendered frames -- what Workload.channels = 3 renders).

    The ground map of every distinct camera is computed once, in row bands, by forked helpers; a second set of
    helpers (forked after the maps exist, so they inherit them) renders the frames into shared memory.  All of this
    runs before the process touches the GPU."""
    from glimpse_amd import synth

    cache = os.environ.get("GLH_FRAME_CACHE")  # (A/B tooling: repeated runs of one workload on one box)

    def cache_key(wl, channels):
        return os.path.join(cache, f"{wl.name}_{wl.T}_{wl.imgsz[0]}x{wl.imgsz[1]}_{wl.O}_{channels}_{wl.bits}")

    out, todo = {}, {}
    for name, wl in wls.items():
        key = cache_key(wl, wl.channels) if cache else None
        if key and all(os.path.exists(f"{key}_{o}.npy") for o in range(wl.O)):
            out[name] = [np.load(f"{key}_{o}.npy", mmap_mode="r") for o in range(wl.O)]
        else:
            todo[name] = wl
    want_rgb = rgb_of is not None and rgb_of[1] in wls
    if want_rgb and cache and os.path.exists(cache_key(wls[rgb_of[1]], 3) + "_0.npy"):
        out[rgb_of[0]] = [np.load(cache_key(wls[rgb_of[1]], 3) + "_0.npy", mmap_mode="r")]
        want_rgb = False
    n_jobs = sum(wl.O * wl.T for wl in todo.values())
    fork = workers > 1 and n_jobs > 4 and _may_fork()
    if fork:
        import multiprocessing as mp

        cams = {}
        for wl in todo.values():
            for cam in wl.cams:
                if cam.tobytes() not in synth._GROUND_MAPS:
                    cams[cam.tobytes()] = cam
        # the textures (seconds each) first, then the ground maps in bands of rows
        _WLS.update(todo)
        first = [(name, ("scene", name)) for name, wl in todo.items() if wl._scene is None]
        for key, cam in cams.items():
            ny = int(cam[7])
            first += [(key, ("rows", cam, r, min(ny, r + 64))) for r in range(0, ny, 64)]
        if first:
            with mp.get_context("fork").Pool(min(workers, len(first))) as pool:
                parts = pool.map(_ground_job, [j for _, j in first], chunksize=1)
            for (key, job), part in zip(first, parts):
                if job[0] == "scene":
                    todo[key]._scene = part
            for key in cams:
                synth._GROUND_MAPS[key] = np.concatenate([p for (k, j), p in zip(first, parts)
                                                          if k == key and j[0] != "scene"])
    for wl in todo.values():
        wl.scene  # (the texture, before the helpers fork)
    _WLS.update(todo)
    jobs = []
    for name, wl in todo.items():
        shape = (wl.T, wl.imgsz[1], wl.imgsz[0]) + ((3,) if wl.channels == 3 else ())
        dtype = np.uint16 if wl.bits == 16 else np.uint8
        out[name] = []
        for o in range(wl.O):
            out[name].append(_shared_array(f"{name}/{o}", shape, dtype) if fork else np.empty(shape, dtype))
            _SHARED[f"{name}/{o}"] = out[name][o]
            jobs += [("frame", f"{name}/{o}", name, o, t) for t in range(wl.T)]
    rgb_jobs = []
    if want_rgb:
        src = wls[rgb_of[1]]
        assert src.O == 1 and src.channels == 1
        shape = (src.T, src.imgsz[1], src.imgsz[0], 3)
        out[rgb_of[0]] = [_shared_array(f"{rgb_of[0]}/0", shape) if fork else np.empty(shape, np.uint8)]
        _SHARED[f"{rgb_of[0]}/0"] = out[rgb_of[0]][0]
        if f"{rgb_of[1]}/0" not in _SHARED:
            _SHARED[f"{rgb_of[1]}/0"] = out[rgb_of[1]][0]
        rgb_jobs = [("rgb", f"{rgb_of[0]}/0", f"{rgb_of[1]}/0", 0, t) for t in range(src.T)]
    if fork:
        with mp.get_context("fork").Pool(min(workers, max(1, len(jobs) + len(rgb_jobs)))) as pool:
            pool.map(_render_job, jobs, chunksize=1)
            pool.map(_render_job, rgb_jobs, chunksize=1)
    else:
        for j in jobs + rgb_jobs:
            _render_job(j)
    if cache:
        os.makedirs(cache, exist_ok=True)
        for name, wl in todo.items():
            for o in range(wl.O):
                np.save(f"{cache_key(wl, wl.channels)}_{o}.npy", out[name][o])
        if want_rgb:
            np.save(cache_key(wls[rgb_of[1]], 3) + "_0.npy", out[rgb_of[0]][0])
    _SHARED.clear()
    _WLS.clear()
    return out


def render_frames(wl, workers):
    """All frames of one workload, [O] arrays (T, H, W[, 3]) uint8."""
    return render_all({"w": wl}, workers)["w"]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _oracle_models(wl, lo, hi):
    from oracle import motion as omotion

    out = []
    for p in range(lo, hi):
        q = wl.params[p]
        out.append(omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13],
                                           axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=wl.N))
    return out


def _oracle_track(wl, frames, lo, hi, n_frames, seed):
    """The oracle (CPU restatement of the reference path) on points [lo, hi) over frames 0 .. n_frames-1 from the
    prior: the same frame window as the GPU's timed region.  Returns seconds."""
    from oracle import tracker as otracker

    observers = [otracker.Observer(list(frames[o][:n_frames]), np.tile(wl.cams[o], (n_frames, 1)), wl.sigmas[o])
                 for o in range(wl.O)]
    matching = np.tile(np.arange(n_frames)[:, None], (1, wl.O))
    np.random.seed(seed)
    t0 = time.perf_counter()
    otracker.track(_oracle_models(wl, lo, hi), observers, matching, np.ones(n_frames - 1), tile_size=wl.tile)
    return time.perf_counter() - t0


def cpu_baseline(wl, frames, n_frames, target_seconds):
    """One host core, a bounded sample of the same workload: whole tracks (all `n_frames` frames from the prior)
    of as many points as fit in the time budget."""
    steps = n_frames - 1
    per_point = _oracle_track(wl, frames, 0, 1, n_frames, 7)
    n_pts = int(max(1, min(wl.P - 1, round(target_seconds / max(per_point, 1e-3)))))
    first = 1 if wl.P > 1 else 0  # (a one-point workload, C1, times its only point again)
    dt = _oracle_track(wl, frames, first, first + n_pts, n_frames, 8)
    cpu_baseline.per_point_seconds = dt / n_pts
    return {
        "value": n_pts * wl.N * steps / dt,
        "unit": "particle-frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n_pts} of {wl.P} points x {wl.N} particles x {steps} steps from the prior (the GPU's frame window), "
                  f"oracle/ (NumPy {np.__version__} + SciPy + C SSD), {dt:.1f} s on 1 of {os.cpu_count()} host cores "
                  f"({cpu_model()})",
    }


def _cpu_worker(job):
    """One host process of the parallel CPU baseline (spawned: it never touches the GPU)."""
    name, n_points, n_particles, n_total_frames, n_frames, lo, hi, frame_files = job
    from glimpse_amd import workloads

    wl = workloads.Workload(name, n_frames=n_total_frames, n_points=n_points, n_particles=n_particles, shard=0, seed=0)
    frames = [np.load(f, mmap_mode="r") for f in frame_files]
    return _oracle_track(wl, frames, lo, hi, n_frames, 100 + lo)


def cpu_baseline_parallel(wl, frames, n_frames, per_point_seconds, target_seconds, workers):
    """The reference's `parallel=True` (one process per block of tracks, tracker.py:381-387, helpers.py:2008-2017)
    restated with the oracle: `workers` spawned processes, each tracking its own block of points through the whole
    frame window; the rendered frames reach them as memory-mapped files."""
    import multiprocessing as mp
    import shutil
    import tempfile

    steps = n_frames - 1
    per_worker = int(max(1, round(target_seconds / max(per_point_seconds, 1e-3))))
    per_worker = min(per_worker, max(1, wl.P // workers))
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="glh_bench_frames_", dir=base)
    try:
        files = []
        for o in range(wl.O):
            files.append(os.path.join(tmp, f"frames_{o}.npy"))
            np.save(files[-1], np.ascontiguousarray(frames[o][:n_frames]))
        jobs = [(wl.name, wl.P, wl.N, wl.T, n_frames, w * per_worker, (w + 1) * per_worker, files) for w in range(workers)]
        ctx = mp.get_context("spawn")
        t0 = time.perf_counter()
        with ctx.Pool(workers) as pool:
            times = pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - t0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    busy = max(times)  # the tracking itself; `wall` also holds interpreter start-up
    n_pts = workers * per_worker
    return {
        "value": n_pts * wl.N * steps / busy,
        "unit": "particle-frames/s",
        "cores": workers,
        "kind": "port",
        "sample": f"{workers} processes x {per_worker} points x {wl.N} particles x {steps} steps from the prior (oracle/, one "
                  f"block of points per process like the reference's parallel=True); slowest process {busy:.1f} s, "
                  f"{wall:.1f} s with start-up; os.cpu_count() = {os.cpu_count()}, usable = {usable_cores()}",
    }


def _api_objects(wl, frames, n_frames):
    """Observers and motion models of `wl` as the drop-in API takes them."""
    import datetime

    import glimpse_amd as g

    t_start = datetime.datetime(2020, 1, 1)
    unit = datetime.timedelta(days=1)
    observers = []
    for o in range(wl.O):
        v = wl.cams[o]
        images = []
        for t in range(n_frames):
            cam = g.Camera(imgsz=v[6:8], f=v[8:10], c=v[10:12], k=v[12:18], p=v[18:20], xyz=v[0:3], viewdir=v[3:6])
            images.append(g.Image(cam=cam, datetime=t_start + t * unit, array=np.asarray(frames[o][t])))
        observers.append(g.Observer(images, sigma=wl.sigmas[o]))
    models = []
    for p in range(wl.P):
        q = wl.params[p]
        models.append(g.CartesianMotion(xy=q[0:2], time_unit=unit, dem=q[16], dem_sigma=q[17], n=wl.N, xy_sigma=q[2:4],
                                        vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13], axyz_sigma=q[13:16]))
    return observers, models


def api_leg(wl, frames, n_frames, seed, device, max_search_dim):
    """The same sequence through the drop-in Python API, glimpse_amd.Tracker.track(rng="philox"): wall time of the
    whole call (frame upload, the frame loop, per-frame status reads, result download) per frame update."""
    import glimpse_amd as g

    t00 = time.perf_counter()
    observers, models = _api_objects(wl, frames, n_frames)
    tracker = g.Tracker(observers, device=device)  # (workspaces sized from the prior: the API's default)
    t_setup = time.perf_counter() - t00
    t0 = time.perf_counter()
    tracks = tracker.track(models, tile_size=wl.tile, rng="philox", seed=seed)
    wall_cold = time.perf_counter() - t0  # creates the context (device allocations) and uploads every frame
    wall = None
    for _ in range(2):  # context and frames resident: the frame loop + results (the faster of two calls)
        t0 = time.perf_counter()
        tracks = tracker.track(models, tile_size=wl.tile, rng="philox", seed=seed)
        dt = time.perf_counter() - t0
        wall = dt if wall is None else min(wall, dt)
    ok = sum(e is None for e in tracks.errors)
    finite = bool(np.isfinite(tracks.means[:, -1]).all())
    out = {"api_ms_per_step": 1e3 * wall / (n_frames - 1), "api_track_seconds": wall,
           "api_first_call_seconds": wall_cold, "api_object_setup_seconds": t_setup,
           "api_tracks_ok": ok, "api_last_means_finite": finite}
    # Tracker.track(parallel=2): two persistent worker processes (here: sharing this one GPU), the frames shared with
    # them once through shared memory, the history collected by one RCCL exchange or -- two ranks on one device cannot
    # make a communicator -- through host memory.  What a call costs beyond the workers' own tracking.
    try:
        t0 = time.perf_counter()
        par = tracker.track(models, tile_size=wl.tile, rng="philox", seed=seed, parallel=2)
        cold = time.perf_counter() - t0  # starts the workers, shares the frames, makes their contexts
        warm, info = None, None
        for _ in range(2):
            t0 = time.perf_counter()
            par = tracker.track(models, tile_size=wl.tile, rng="philox", seed=seed, parallel=2)
            dt = time.perf_counter() - t0
            if warm is None or dt < warm:
                warm, info = dt, par.parallel_info
        out["api_parallel_2"] = {
            "workers": 2, "devices": min(2, max(1, _device_count())), "transport": par.transport,
            "first_call_seconds": cold, "call_seconds": warm, "ms_per_step": 1e3 * warm / (n_frames - 1),
            "worker_track_seconds": info["worker_track_seconds"], "worker_seconds": info["worker_seconds"],
            "parent_seconds": info["parent_seconds"],
            "overhead_seconds_per_call": warm - max(info["worker_track_seconds"]),
            "single_process_call_seconds": wall, "contexts_made_in_warm_call": int(sum(info["contexts_made"])),
            "shared_frame_bytes": info["shared_frame_bytes"],
            "same_as_single_process": bool(np.array_equal(par.means, tracks.means, equal_nan=True)
                                           and np.array_equal(par.sigmas, tracks.sigmas, equal_nan=True)),
            "note": "overhead = the call's wall time beyond the slower worker's own Tracker.track(): parameter tables out, "
                    "errors / warnings and the history back; on ONE GPU the two workers share the device, so the call "
                    "cannot be faster than the single-process one"}
    except Exception as e:  # noqa: BLE001
        out["api_parallel_2"] = {"error": repr(e)}
    tracker.close()
    return out


def from_files_leg(wl, frames, n_frames, seed, device, fmt, compute_only_seconds):
    """C3 as a run FROM IMAGE FILES (image.py:137-214): the frames are written to a directory as `fmt`
    ("jpeg": quality 95, what a time-lapse camera leaves; "tiff": uncompressed) -- untimed --, then
    Tracker.track(rng="philox") runs from glimpse_amd.Image(path=...) objects that hold no pixels: a pool of threads
    of decoder PROCESSES (glimpse_amd/ingest.py: Pillow, one per usable core but one) fills a shared-memory ring that is
    page-locked for the device, and the frame loop runs on the frames already resident while the later ones are decoded
    and copied (glh_observer_upload_frame_pinned on the copy stream).  Timed: the whole call on a Tracker whose context
    exists (a first call made it) and whose frames were forgotten (`forget_frames`), the median of three.  Checked: the
    tracks equal, bit for bit, those of the same pixels handed over as arrays."""
    import datetime
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    from PIL import Image as PILImage

    import glimpse_amd as g

    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="glh_bench_files_", dir=base)
    ext, kw = (".jpg", dict(format="JPEG", quality=95)) if fmt == "jpeg" else (".tif", dict(format="TIFF"))
    t_start, unit = datetime.datetime(2020, 1, 1), datetime.timedelta(days=1)
    try:
        paths = [[os.path.join(tmp, f"o{o}_{t:04d}{ext}") for t in range(n_frames)] for o in range(wl.O)]

        def write(job):
            o, t = job
            PILImage.fromarray(np.asarray(frames[o][t])).save(paths[o][t], **kw)

        with ThreadPoolExecutor(max_workers=usable_cores()) as pool:
            list(pool.map(write, [(o, t) for o in range(wl.O) for t in range(n_frames)]))
        file_bytes = sum(os.path.getsize(p) for po in paths for p in po)

        def observers(from_arrays):
            out = []
            for o in range(wl.O):
                v = wl.cams[o]
                images = []
                for t in range(n_frames):
                    cam = g.Camera(imgsz=v[6:8], f=v[8:10], c=v[10:12], k=v[12:18], p=v[18:20], xyz=v[0:3], viewdir=v[3:6])
                    if from_arrays:
                        with PILImage.open(paths[o][t]) as im:
                            images.append(g.Image(cam=cam, datetime=t_start + t * unit, array=np.asarray(im)))
                    else:
                        images.append(g.Image(path=paths[o][t], cam=cam, datetime=t_start + t * unit))
                out.append(g.Observer(images, sigma=wl.sigmas[o], cache=False))  # (a read leaves nothing on the image)
            return out

        _, models = _api_objects(wl, [f[:2] for f in frames], 2)
        tracker = g.Tracker(observers(False), device=device)
        t0 = time.perf_counter()
        tracks = tracker.track(models, tile_size=wl.tile, rng="philox", seed=seed)
        cold = time.perf_counter() - t0  # makes the context as well
        runs = []
        for _ in range(3):
            tracker.forget_frames()
            t0 = time.perf_counter()
            again = tracker.track(models, tile_size=wl.tile, rng="philox", seed=seed)
            runs.append((time.perf_counter() - t0, dict(tracker._feed_stats)))
        same_again = bool(np.array_equal(again.means, tracks.means, equal_nan=True))
        tracker.close()
        runs.sort(key=lambda r: r[0])
        wall, st = runs[1]
        # the same pixels as in-memory arrays (decoded here, untimed): the tracks must be the same numbers
        ref_tracker = g.Tracker(observers(True), device=device)
        ref = ref_tracker.track(models, tile_size=wl.tile, rng="philox", seed=seed)
        ref_tracker.close()
        same = bool(np.array_equal(ref.means, tracks.means, equal_nan=True) and np.array_equal(ref.sigmas, tracks.sigmas, equal_nan=True))
        return {
            "workload": wl.describe()["workload"], "format": fmt, "files": st["files"], "file_MB": file_bytes / 1e6,
            "decoded_MB": st["bytes"] / 1e6, "call_seconds": wall, "call_seconds_min_max": [runs[0][0], runs[-1][0]],
            "first_call_seconds": cold, "frames_per_s": n_frames / wall, "ms_per_frame": 1e3 * wall / (n_frames - 1),
            "decode_threads": st["threads"], "decode_processes": st.get("processes", 0),
            "pinned_ring": st.get("pinned_ring", False),
            "decode_ms_per_frame_per_core": 1e3 * st["decode_seconds"] / max(1, st["files"]),
            "decode_bound_seconds": st["decode_seconds"] / max(1, st["threads"] + st.get("processes", 0)),
            "upload_staging_seconds": st["upload_seconds"],
            "upload_staging_GBps": st["bytes"] / max(st["upload_seconds"], 1e-9) / 1e9,
            "frame_loop_waited_for_decoders_seconds": st["wait_seconds"], "glh_track_calls": st.get("track_calls"),
            "compute_only_call_seconds": compute_only_seconds,
            "gpu_idle_share": max(0.0, 1.0 - compute_only_seconds / wall),
            "same_tracks_as_arrays": same and same_again,
            "note": "gpu_idle_share = 1 - (the same call on frames already resident) / (this call): the share of the call the "
                    "device pipeline was not the limiter; decode_bound_seconds = the decoders' summed time / threads, what "
                    "the call cannot go below while decoding limits it"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _device_count():
    from glimpse_amd import _lib

    return _lib.device_count()


def apply_motion(ctx, wl, motion, dem="constant"):
    """The workload's points under another motion model than CartesianMotion (motion.py:207-522), and / or over gridded
    surfaces: a DEM raster of the scene's ground (z = 0 with centimetre relief, 2 m cells) and a DEM-uncertainty raster."""
    from glimpse_amd import Raster, _lib, workloads

    if motion == "cartesian" and dem == "constant":
        return
    full = np.zeros((wl.P, _lib.MOTION_FULL_LEN))
    full[:, :_lib.MOTION_LEN] = wl.params
    full[:, 18] = _lib.MOTION_KINDS[motion]
    full[:, 19] = 0.05  # slope_sigma (tangent models)
    if dem == "gridded":
        lo, hi = wl.xy.min(axis=0) - 200.0, wl.xy.max(axis=0) + 200.0
        nx, ny = (np.ceil((hi - lo) / 2.0)).astype(int)
        gx, gy = np.meshgrid(np.linspace(0, 6.0, nx), np.linspace(0, 4.0, ny))
        ctx.set_raster(0, Raster(0.02 * np.sin(gx) * np.cos(gy), x=(lo[0], hi[0]), y=(hi[1], lo[1])))
        ctx.set_raster(1, Raster(np.full((ny, nx), max(float(wl.params[0, 17]), 0.05)), x=(lo[0], hi[0]), y=(hi[1], lo[1])))
        full[:, 16:18] = 0.0
        full[:, 20:22] = 1.0
    if "cylindrical" in motion:  # (speed, direction, dz/dt) and their sigmas
        full[:, 4:7] = (workloads.VELOCITY[0], 0.0, 0.0)
        full[:, 7:10] = (workloads.SIGMA, 0.5, 0.0)
        full[:, 13:16] = (workloads.SIGMA / 4, 0.1, 0.0)
    ctx.set_motion(full)


# ------------------------------------------------------------------------------------------------
# secondary legs of the default run: the other configurations and the parity-grade arithmetic, short
# ------------------------------------------------------------------------------------------------
def measure_sequence(ctx, wl, n_frames, seed, math, warm=3, repeats=3):
    """The whole sequence of `wl` from the prior (frame 0 initialises, frames 1 .. n_frames-1 are timed) on a context that
    already holds its frames: one glh_track call, HIP events around every launch.  Returns the figures of a secondary
    leg: ms per frame update (wall and kernel), roofline fraction by SURVEY 8(d)'s algorithmic bytes, PMC traffic ratio
    from the committed counters."""
    images = lambda i: [i] * wl.O  # noqa: E731

    def initialise():
        ctx.set_frame(0)
        ctx.init_particles(seed=seed)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)

    def run(first, count):
        if count <= 0:
            return
        fr = list(range(first, first + count))
        ctx.track(fr, [1.0] * count, [images(j) for j in fr], seed=seed)

    ctx.set_math(math)
    initialise()
    run(1, min(warm, n_frames - 1))
    ctx.sync()

    def one_pass(profiled):
        initialise()
        ctx.sync()
        ctx.profile_enable(profiled)
        ctx.profile_reset()
        t0 = time.perf_counter()
        run(1, n_frames - 1)
        ctx.sync()
        return time.perf_counter() - t0

    # (like the headline: `repeats` passes without event timers, the median reported; one more with HIP events around
    # every launch for the launch durations and the GPU span)
    walls = sorted(one_pass(False) for _ in range(max(1, repeats)))
    wall = walls[(len(walls) - 1) // 2]
    profiled_wall = one_pass(True)
    stage_ms = ctx.profile_get()
    dom = max(stage_ms, key=lambda k: stage_ms[k][0])
    launch_ms = ctx.profile_launches(dom)
    span_ms = ctx.profile_span(dom)
    streams = ctx.last_track_streams()
    ctx.profile_enable(False)
    status = ctx.observer_status()
    boxes = ctx.search_boxes()
    abytes = algorithmic_bytes_per_step(wl.P, wl.N, wl.O, wl.tile, boxes, status, wl.channels * wl.bits // 8)
    dom_ms, dom_n = stage_ms[dom]
    per_launch = dom_ms / max(dom_n, 1)
    gpu_per_frame = span_ms / (n_frames - 1) if streams > 1 else per_launch  # (two streams: the launches overlap)
    kern = KERNEL_OF_STAGE.get(dom, dom)
    traffic = pmc_traffic_per_frame(wl, kern) if math == "fast" and wl.channels == 1 and wl.bits == 8 else None
    moments = ctx.get_moments(0, n_frames)
    leg = {
        "workload": wl.describe()["workload"], "math": math, "frames": n_frames, "kernel": kern,
        "variant": list(ctx.last_variant()),
        "ms_per_frame": 1e3 * wall / (n_frames - 1),
        "ms_per_frame_min_max": [round(1e3 * walls[0] / (n_frames - 1), 5), round(1e3 * walls[-1] / (n_frames - 1), 5)],
        "repeats": len(walls), "profiled_pass_ms_per_frame": 1e3 * profiled_wall / (n_frames - 1),
        "kernel_ms_per_launch": per_launch, "track_streams": streams, "gpu_ms_per_frame": gpu_per_frame,
        "value": wl.P * wl.N * (n_frames - 1) / wall,
        "frames_per_s": (n_frames - 1) / wall,
        "roofline_frac": abytes / (gpu_per_frame * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": abytes,
        "traffic_ratio": None if traffic is None else traffic / abytes,
        "observer_ok_fraction": float((status == 0).mean()),
        "points_with_error_bits": int((ctx.point_status() != 0).sum()),
        "final_means_finite": bool(np.isfinite(moments[n_frames - 1]).all()),
    }
    if len(launch_ms) == dom_n and dom_n == n_frames - 1:
        tail = launch_ms[-min(20, dom_n):]
        leg["steady_ms_per_frame"] = float(tail.mean())
        leg["first_steps_ms"] = [round(float(v), 4) for v in launch_ms[:6]]
    return leg


def secondary_legs(args, device, T, rendered, seed):
    """What the headline does not show, each over its whole sequence from the prior on the frames already rendered:
    exact arithmetic at C3 (the arithmetic the oracle tests pin bit for bit), one GPU's shard of C4, C5 (two observers
    + DEM term, all 2048 points on one GPU), C2, C3 on RGB frames (what time-lapse JPEGs decode to: 766 key bins
    instead of 256) and C3 on uint16 frames (the keys ranked in LDS instead of counted in 256 bins)."""
    from glimpse_amd import _lib, workloads

    legs = {}
    plan = [("C3_exact", "C3", None, "exact", T, "C3", 1, 8, "cartesian"),
            # the headline's configuration with ONE launch per frame (glh_set_track_streams(1)): the per-launch roofline
            # of rounds 1-3, where a launch has the chip to itself
            ("C3_one_stream", "C3", None, "fast", T, "C3", 1, 8, "cartesian"),
            ("C4_shard", "C4", None, "fast", T, "C3", 1, 8, "cartesian"),
            # BASELINE config 4 WHOLE on this one GPU (10 000 points x 10 000 particles, 2 x 4.8 GB of state): the N = 1
            # anchor of the strong-scaling curve; north_star asks for 50 frames/s of this on eight GPUs
            ("C4_full_1gpu", "C4", workloads.CONFIGS["C4"]["points"], "fast", T, "C3", 1, 8, "cartesian"),
            ("C5", "C5", workloads.CONFIGS["C5"]["points"], "fast", T, "C5", 1, 8, "cartesian"),
            # what each of C5's four GPUs runs: 512 of the 2 048 points
            ("C5_shard", "C5", None, "fast", T, "C5", 1, 8, "cartesian"),
            ("C2", "C2", None, "fast", min(T, workloads.CONFIGS["C2"]["frames"]), "C3", 1, 8, "cartesian"),
            # TangentCartesianMotion, the model of real glacier runs (motion.py:339-430): the general instantiation
            ("C3_tangent", "C3", None, "fast", T, "C3", 1, 8, "tangent_cartesian"),
            # ... over a gridded DEM + DEM uncertainty (glimpse.Raster, 2 m cells): what real runs bring
            ("C3_tangent_dem", "C3", None, "fast", T, "C3", 1, 8, "tangent_cartesian+dem"),
            ("C3_rgb", "C3", None, "fast", T, "C3_rgb", 3, 8, "cartesian"),
            ("C3_u16", "C3", None, "fast", T, "C3_u16", 1, 16, "cartesian"),
            # float32 frames (an orthophoto / reflectance observer): the C3 frames scaled to [0, 1]
            ("C3_f32", "C3", None, "fast", T, "C3", 1, 32, "cartesian")]
    for key, name, points, math, n_frames, frames_of, channels, bits, motion in plan:
        frames = rendered.get(frames_of)
        if frames is None:
            continue
        try:
            wl = workloads.Workload(name, n_frames=T, n_points=points, shard=0, seed=0)
            wl.channels, wl.bits = channels, bits
            if bits == 32:
                frames = [[np.asarray(f, dtype=np.float32) * np.float32(1.0 / 255.0) for f in fo] for fo in frames]
            # (uint16 frames take the fused step while a tile's pixel count fits a 16-bit key: workspaces up to 255 px)
            dim = min(args.max_search_dim, 255) if bits >= 16 else args.max_search_dim  # (uint16 / float: ranked keys)
            with _lib.Context(wl.P, wl.N, wl.O, device_id=device, max_tile=max(wl.tile), max_search_dim=dim,
                              max_frames=T) as ctx:
                workloads.setup_context(ctx, wl, frames)
                motion, _, dem = motion.partition("+")
                apply_motion(ctx, wl, motion, "gridded" if dem else "constant")
                ctx.set_track_streams(1 if key == "C3_one_stream" else args.streams)
                legs[key] = measure_sequence(ctx, wl, n_frames, seed, math, repeats=min(3, max(1, args.repeats)))
                if motion != "cartesian":
                    legs[key]["motion"] = motion
                if dem:
                    legs[key]["dem"] = "gridded"
                if motion != "cartesian" or dem:  # (the committed PMC passes are of the CartesianMotion runs over constants)
                    legs[key]["traffic_ratio"] = None
        except Exception as e:  # noqa: BLE001
            legs[key] = {"error": repr(e)}
    return legs


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def worker(args):
    from glimpse_amd import _lib, sharding, workloads

    if args.max_search_dim is None:
        args.max_search_dim = 255 if args.bits >= 16 else 320

    group = sharding.Group.from_env()
    rank, world = group.rank, group.world
    if "WORLD_SIZE" in os.environ and args.gpus != world and args.gpus != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cfg = workloads.CONFIGS[args.workload]
    W, B = max(0, args.warmup), max(0, args.burn_in)
    if args.steps is None:
        K, F = cfg["frames"] - 1 - B, 1
    else:
        K = args.steps
        F = args.frames_per_step if args.frames_per_step > 0 else \
            (max(1, round((cfg["frames"] - 1) / K)) if B == 0 else 1)
    if K < 1:
        raise SystemExit("need at least one timed step")
    T = 1 + B + K * F  # frames of the sequence: frame 0 initialises, then burn-in, then K steps of F updates
    W = min(W, (T - 1) // F)
    strong = args.split == "strong"
    if strong:
        # the configuration's points (or --points of them) divided over the ranks: every rank builds the same
        # point set and keeps its contiguous block
        total = args.points if args.points is not None else cfg["points"]
        lo, hi = sharding.shard_range(total, world, rank)
        wl = workloads.Workload(args.workload, n_frames=T, n_points=total, n_particles=args.particles, shard=0, seed=0)
        wl = wl.slice(lo, hi)
        point_offset = lo
        sizes = sharding.shard_sizes(total, world)
    else:
        wl = workloads.Workload(args.workload, n_frames=T, n_points=args.points, n_particles=args.particles, shard=rank,
                                seed=0)
        point_offset = rank * wl.P
        sizes = [wl.P] * world

    wl.channels, wl.bits = args.channels, (8 if args.bits == 32 else args.bits)  # (float32: the 8-bit scene, scaled below)
    # frames: rendered once per job (rank 0), shared with the other ranks as memory-mapped files -- by forked helpers,
    # hence BEFORE anything loads the HIP library (the device count below does)
    cores = usable_cores()
    secondary = (world == 1 and not args.no_secondary and args.workload == "C3" and args.points is None
                 and args.particles is None and args.motion == "cartesian" and args.dem == "constant" and B == 0 and args.channels == 1
                 and args.bits == 8)
    rendered = {}
    if world == 1:
        if secondary:
            # every secondary leg's frames with the headline's, by one set of helpers: C4 and C2 see the C3 frames, the
            # RGB leg their channel remap; C5 has its own scene (it covers the oblique camera's footprint); the 16-bit
            # leg sees the C3 scene on a 16-bit sensor
            c5 = workloads.Workload("C5", n_frames=T, n_points=workloads.CONFIGS["C5"]["points"], shard=0, seed=0)
            u16 = workloads.Workload("C3", n_frames=T, shard=0, seed=0)
            u16.bits = 16
            rendered = render_all({"C3": wl, "C5": c5, "C3_u16": u16}, cores, rgb_of=("C3_rgb", "C3"))
            frames = rendered["C3"]
        else:
            frames = render_frames(wl, cores)
    else:
        if rank == 0:
            frames = render_frames(wl, cores)
            for o in range(wl.O):
                group.store.put_array(f"frames_{o}", frames[o])
        else:
            frames = [group.store.get_array(f"frames_{o}", mmap=True) for o in range(wl.O)]

    if args.bits == 32:  # float32 frames (an orthophoto / reflectance observer): the scene's gray levels scaled to [0, 1]
        frames = [[np.asarray(f, dtype=np.float32) * np.float32(1.0 / 255.0) for f in fo] for fo in frames]
        wl.bits = 32
    # one GPU per rank (LOCAL_RANK); GLH_BENCH_DEVICE is a test hook (several ranks on one GPU)
    device = int(os.environ.get("GLH_BENCH_DEVICE", group.local_rank % max(1, _lib.device_count())))
    _mark("frames rendered")
    ctx = _lib.Context(wl.P, wl.N, wl.O, device_id=device, max_tile=max(wl.tile), max_search_dim=args.max_search_dim,
                       max_frames=T)
    workloads.setup_context(ctx, wl, frames)
    _mark("context ready, frames uploaded")
    # one seed for the whole job: the device RNG is keyed on the GLOBAL point index, so the
    # sharded run draws what a single-GPU run of all points would draw
    apply_motion(ctx, wl, args.motion, args.dem)
    ctx.set_point_offset(point_offset)
    ctx.set_math(args.math)
    ctx.set_track_streams(args.streams)
    transport = group.attach(ctx, args.transport)
    seed = args.seed
    images = lambda i: [i] * wl.O  # noqa: E731

    def initialise():
        """frame 0: particles + templates (tracker.py:327-342), untimed"""
        ctx.set_frame(0)
        ctx.init_particles(seed=seed)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)

    C = K * F if args.frames_per_call <= 0 else min(args.frames_per_call, K * F)

    def run(first, count):
        """`count` consecutive frame updates from frame `first`: glh_track calls of up to C frames each (the frame
        loop of tracker.py:326-357, one kernel launch per frame), or glh_step per frame when C == 1."""
        i = first
        while i < first + count:
            n = min(C, first + count - i)
            if n == 1:
                ctx.step(i, 1.0, images(i), seed=seed)
            else:
                ctx.track(list(range(i, i + n)), [1.0] * n, [images(j) for j in range(i, i + n)], seed=seed)
            i += n

    def gather():
        # (RCCL: the exchange only; the root copies the gathered history to the host after the timed region, like the
        # single-rank run reads its own history after it)
        return group.gather_moments(ctx, 0, T, sizes, download=False)

    initialise()
    run(1, W * F)  # warm-up: the first W steps of the same sequence, untimed
    if world > 1:
        gather()  # warm the communicator up outside the timed region
    ctx.sync()

    # The timed region -- the K steps -- R times (round 5): every repetition starts from the prior again (untimed
    # re-initialisation + burn-in) and is bracketed by barriers; `value` is the MEDIAN repetition's, `spread` lists all.
    # These repetitions carry NO event timers: a HIP event pair around every launch costs small batches 13 % of a
    # frame (C2: 0.052 ms per frame with them, 0.045 without -- profiles/ab_r05/r5j01_graph_bigframes.txt).  One MORE
    # pass of the same region is then run with HIP events around every kernel launch (on the streams the launches are
    # enqueued on): the roofline's launch durations and GPU span come from that pass, its wall time is reported beside.
    R = max(1, args.repeats)

    def timed_pass(profiled):
        initialise()  # back to the prior (untimed)
        run(1, B)  # burn-in, untimed
        ctx.sync()
        ctx.profile_enable(profiled)
        ctx.profile_reset()
        group.barrier()
        t0 = time.perf_counter()
        run(1 + B, K * F)
        got = None
        if world > 1:
            ctx.sync()
            got = gather()
        group.barrier()
        return group.max(time.perf_counter() - t0), got

    all_elapsed = []
    for _ in range(R):
        all_elapsed.append(timed_pass(False)[0])
    elapsed = sorted(all_elapsed)[(R - 1) // 2]  # the median repetition (the lower middle one of an even count)
    profiled_elapsed, gathered = timed_pass(True)  # (the gathered history: checked against the state this last pass left)
    stage_ms = ctx.profile_get()
    dom = max(stage_ms, key=lambda k: stage_ms[k][0])
    launch_ms = ctx.profile_launches(dom)
    span_ms = ctx.profile_span(dom)
    streams = ctx.last_track_streams() if C > 1 else 1
    ctx.profile_enable(False)
    _mark("headline timed")

    # health of the run: every point must still be tracked by every observer
    status = ctx.observer_status()
    pt_status = ctx.point_status()
    frac_ok = float((status == 0).mean())
    n_err = int((pt_status != 0).sum())
    # algorithmic bytes / SSD flops of one step, from the search boxes of the last timed step (the smallest tiles
    # of the sequence: the tile term, ~4 % of the bytes, is if anything understated for the first frames)
    boxes = ctx.search_boxes()
    abytes = algorithmic_bytes_per_step(wl.P, wl.N, wl.O, wl.tile, boxes, status, wl.channels * wl.bits // 8)
    flops = ssd_flops_per_step(wl.O, wl.tile, boxes, status)
    moments_local = ctx.get_moments(0, T)

    gathered_ok = None
    if world > 1 and rank == 0:
        if gathered is None:
            gathered = ctx.gathered()
        allm, allst = gathered
        gathered_ok = bool(allm.shape == (T, sum(sizes), 12) and np.isfinite(allm[1 + B:]).all()
                           and np.array_equal(allm[:, :wl.P], moments_local))
        n_err = int((allst != 0).sum())
    rc = 0
    if rank == 0 and args.dump_moments:
        np.save(args.dump_moments, gathered[0] if world > 1 else moments_local)
    if rank == 0:
        total_points = sum(sizes)
        value = total_points * wl.N * K * F / elapsed
        out = {
            "metric": "particle-frames/s",
            "value": value,
            "unit": "particle-frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "ms_per_frame": 1e3 * elapsed / (K * F),
            "spread": {"n": R, "statistic": "median of n repetitions of the timed region in this invocation",
                       "ms_per_step_min": 1e3 * min(all_elapsed) / K, "ms_per_step_max": 1e3 * max(all_elapsed) / K,
                       "value_min": total_points * wl.N * K * F / max(all_elapsed),
                       "value_max": total_points * wl.N * K * F / min(all_elapsed),
                       "ms_per_step_all": [round(1e3 * e / K, 5) for e in all_elapsed],
                       "profiled_pass_ms_per_step": 1e3 * profiled_elapsed / K,
                       "note": "the n repetitions run without event timers; one more pass of the same region with HIP "
                               "events around every launch gives the roofline's launch durations and GPU span"},
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": dict(wl.describe(args.motion), dem=args.dem, rng="device Philox4x32-7",
                           math=args.math, untimed_launches=W * F + B, untimed_kernel_launches=(W * F + B) * streams,
                           parallelism=f"points sharded x{world}",
                           total_points=total_points, frames_per_s=K * F / elapsed, burn_in_steps=B, frames_per_call=C,
                           frame_updates_per_step=F, track_streams=streams,
                           step=f"{F} consecutive frame update(s) of all {total_points} points",
                           timed_from="the prior: frame 0 initialises, all the later frames are timed "
                                      f"({K} steps x {F} updates)" if B == 0 else f"after {B} untimed updates"),
            "rccl_ranks": world if transport == "rccl" else 0,
            "collective": {"none": "none (1 rank)", "rccl": "RCCL ncclSend/ncclRecv gather inside libglimpse_hip.so",
                           "host": "host copies through the rendezvous directory (RCCL unavailable: "
                                   + getattr(group, "why_host", "") + ")"}[transport],
            "health": {"observer_ok_fraction": frac_ok, "points_with_error_bits": n_err,
                       "gathered_moments_finite": gathered_ok,
                       "final_means_finite": bool(np.isfinite(moments_local[T - 1]).all())},
        }
        tot = sum(ms for ms, _ in stage_ms.values())
        dom_ms, dom_n = stage_ms[dom]
        per_launch_ms = dom_ms / max(dom_n, 1)
        launches_per_step = dom_n / K
        launches_per_frame = dom_n / (K * F)
        # one launch of the dominant kernel processes P*N particle-frames (SURVEY 8(d) per-unit bytes); `abytes` are
        # the algorithmic bytes of ONE frame update
        kern = KERNEL_OF_STAGE.get(dom, dom)
        # `achieved`: algorithmic bytes of the dominant kernel's launches / the GPU time they take.  One launch per frame:
        # bytes of a launch / its mean duration.  glh_track on two streams (the two halves of the points: while one
        # half's launch drains, the other's fills the idle compute units): two launches are resident at a time and a
        # launch's own duration spans its neighbour's work, so the time is the GPU span of the timed launches (start
        # of the first to end of the last, HIP events on both streams) per frame; what ONE such launch achieves on its
        # share of the chip is under `per_launch`.
        # HBM bytes of a frame from the committed PMC passes: per launch x the launches of a frame (like `achieved`, which
        # is the bytes of a frame over the time of a frame)
        traffic = pmc_traffic_per_frame(wl, kern) if wl.channels == 1 and wl.bits == 8 else None
        span_per_frame_ms = span_ms / (K * F)
        per_launch_ach = abytes / launches_per_frame / (per_launch_ms * 1e-3) / 1e9
        ach = abytes / (span_per_frame_ms * 1e-3) / 1e9 if streams > 1 else per_launch_ach
        roof = {"kernel": kern, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": per_launch_ms, "launches_per_step": launches_per_step,
                "concurrent_launches": streams, "gpu_span_ms_per_frame": span_per_frame_ms,
                "per_launch": {"achieved": per_launch_ach, "frac": per_launch_ach / HBM_PEAK_GBS,
                               "points": wl.P / launches_per_frame},
                "algorithmic_bytes_per_launch": abytes / launches_per_frame,
                "algorithmic_bytes_per_particle_frame": abytes / (wl.P * wl.N),
                "ssd_fp32_tflops": flops / (tot / (K * F) * 1e-3) / 1e12,
                # what a float4 copy kernel reaches on this chip (MI355X_MICROARCH.md), and this GPU's own
                # hipMemcpyDtoD rate ("a measured device-copy ceiling")
                "achievable_GBps": HBM_ACHIEVABLE_GBS, "frac_of_achievable": ach / HBM_ACHIEVABLE_GBS,
                "measured_copy_GBps": ctx.copy_bandwidth(1 << 30, 10)}
        if streams > 1:
            roof["note"] = (f"{streams} launches of k_point_step are resident at a time (glh_track runs the halves of the "
                            "points on two HIP streams): the sum of the launch durations of a step exceeds the step's wall "
                            "time by design; `achieved` = algorithmic bytes of a frame / `gpu_span_ms_per_frame` (first "
                            "launch start to last launch end over the timed region, HIP events on both streams, per "
                            "frame); `avg_launch_ms` is one launch of half the points and is what rocprofv3's kernel trace "
                            "shows; `per_launch` is that launch's own rate; secondary.C3_one_stream is the same "
                            "configuration with one launch per frame")
        if roof["traffic"] is not None:
            roof["traffic_per"] = f"frame update ({launches_per_frame:g} launch(es) of {wl.P / launches_per_frame:g} points)"
            roof["traffic_over_algorithmic"] = traffic / abytes
            roof["traffic_source"] = pmc_traffic_source(wl, kern)
        out["roofline"] = roof
        # The other ceiling, for the record: the step is bound by VALU issue, not by memory.  Wave-level
        # VALU instructions per launch from the committed SQ counters of this workload (SQ_INSTS_VALU, rocprofv3 --pmc)
        # against what the chip can issue: 256 CUs x 4 SIMDs, one 64-lane float64 / 3-operand instruction per 4 cycles.
        sq, sq_file = sq_counters(wl)
        if sq is not None:
            insts = sq * wl.P * wl.N / 64.0
            peak_rate = 256 * 4 * 2.4e9 / 4.0
            roof["valu_issue"] = {"wave_instructions_per_frame": insts, "per_64_particle_frames": sq,
                                  "frac_of_issue_peak": insts / (peak_rate * span_per_frame_ms * 1e-3),
                                  "model": "1 wave-instruction / 4 cycles / SIMD at 2.4 GHz, 1024 SIMDs",
                                  "source": "profiles/" + sq_file,
                                  "source_run": "committed profile, not measured in this run: a separate rocprofv3 "
                                                "--pmc SQ_INSTS_VALU pass of this command on the builder's gpurun box"}
        out["stage_ms_per_step"] = {k: ms / K for k, (ms, _) in stage_ms.items() if ms > 0}
        if len(launch_ms) == dom_n and launches_per_frame == 1:
            tail = launch_ms[-min(20, K * F):]
            out["steady_ms_per_frame"] = float(tail.mean())
            out["steady_ms_per_step"] = float(tail.mean()) * F
            out["steady_value"] = wl.P * wl.N / (float(tail.mean()) * 1e-3) * world
            out["steady_roofline_frac"] = abytes / (float(tail.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["first_steps_ms"] = [round(float(v), 4) for v in launch_ms[:8]]
        elif len(launch_ms) == dom_n and streams == 2 and launches_per_frame == 2:
            # (the launches of a frame overlap: a frame's own time is not separable; the durations of its two
            # half-launches are listed as they are)
            out["first_steps_launch_ms"] = [round(float(v), 4) for v in launch_ms[:16]]
            out["steady_launch_ms"] = float(launch_ms[-min(40, dom_n):].mean())
    ctx_closed = False
    if rank == 0 and world == 1:
        # the side legs must not cost the headline: a failure is reported in the line, the run then exits non-zero
        if secondary:
            ctx.close()
            ctx_closed = True
            out["secondary"] = secondary_legs(args, device, T, rendered, seed)
            _mark("secondary legs")
        if not args.no_api:
            if not ctx_closed:
                ctx.close()
            ctx_closed = True
            try:
                out.update(api_leg(wl, frames, T, seed, device, args.max_search_dim))
                _mark("API leg")
            except Exception as e:  # noqa: BLE001
                out["api_error"] = repr(e)
            if secondary and "api_track_seconds" in out:
                for fmt in ("jpeg", "tiff"):  # (the run from files: decode pool -> pinned ring -> HBM while tracking)
                    try:
                        out["secondary"][f"C3_from_files_{fmt}"] = from_files_leg(wl, frames, T, seed, device, fmt,
                                                                                  out["api_track_seconds"])
                    except Exception as e:  # noqa: BLE001
                        out["secondary"][f"C3_from_files_{fmt}"] = {"error": repr(e)}
                _mark("run from files")
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(wl, frames, T, args.cpu_seconds)
                _mark("CPU baseline, one core")
                workers = args.cpu_workers if args.cpu_workers > 0 else usable_cores()
                if workers > 1 and wl.P >= 2 * workers:
                    out["cpu_baseline_parallel"] = cpu_baseline_parallel(
                        wl, frames, T, cpu_baseline.per_point_seconds, args.cpu_seconds, workers)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline_error"] = repr(e)
    if rank == 0:
        h = out["health"]
        # The legs that lean on the operating system (worker and decoder processes, shared memory, image files in a
        # temporary directory) report an exception under their own key and in `health.notes` -- a box without room in
        # /dev/shm is not an unhealthy tracker --; a WRONG result of theirs is as fatal as any other.
        host_legs = {k: v for k, v in out.get("secondary", {}).items() if k.startswith("C3_from_files")}
        host_legs["api_parallel_2"] = out.get("api_parallel_2", {})
        notes = [f"{k}: {v['error']}" for k, v in host_legs.items() if "error" in v]
        if notes:
            h["notes"] = notes
        wrong = any(v.get("same_tracks_as_arrays") is False or v.get("same_as_single_process") is False
                    for v in host_legs.values())
        if h["points_with_error_bits"] or h["observer_ok_fraction"] < 0.99 or not h["final_means_finite"] \
                or h["gathered_moments_finite"] is False or out.get("api_last_means_finite") is False \
                or "api_error" in out or "cpu_baseline_error" in out or wrong \
                or any(("error" in leg and k not in host_legs) or leg.get("points_with_error_bits")
                       or not leg.get("final_means_finite", True) or leg.get("observer_ok_fraction", 1.0) < 0.99
                       for k, leg in out.get("secondary", {}).items()):
            rc = 3
            out["health"]["verdict"] = "UNHEALTHY"
        _mark("done")
        print(json.dumps(out), flush=True)
    group.close()
    if not ctx_closed:
        ctx.close()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch(args)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
