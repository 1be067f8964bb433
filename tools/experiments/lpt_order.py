"""Experiment (round 5): does the ORDER of the points in a launch matter?  Workgroups are dispatched in block order; a launch
ends with its slowest point, and a point's workgroup lifetime is persistent from frame to frame (lifetime_persistence.py).
Longest-processing-time-first: the slow points at the head of the launch, the short ones fill its tail.

    python tools/experiments/lpt_order.py WORKLOAD POINTS FRAMES

1. lifetimes from phase stamps (one stream, frames 10..), 2. the same workload with its points permuted -- as they are,
slowest first (dealt alternately to the two halves glh_track runs on its two streams), fastest first -- timed over the whole
sequence, five interleaved repetitions each.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glimpse_amd import _lib, workloads  # noqa: E402


def main():
    name, P, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import bench  # noqa: E402

    wl = workloads.Workload(name, n_frames=T, n_points=P)
    frames = bench.render_frames(wl, bench.usable_cores())

    def context(w):
        ctx = _lib.Context(w.P, w.N, w.O, device_id=0, max_tile=max(w.tile), max_search_dim=320, max_frames=T)
        workloads.setup_context(ctx, w, frames)
        ctx.set_math("fast")
        return ctx

    def start(ctx, w):
        ctx.set_frame(0)
        ctx.init_particles(seed=1)
        for o in range(w.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        ctx.sync()

    ctx = context(wl)
    ctx.set_track_streams(1)
    start(ctx, wl)
    ctx.phase_stamps()  # arm
    life = []
    for i in range(1, min(T, 40)):
        ctx.step(i, 1.0, [i] * wl.O, seed=1)
        ctx.sync()
        st = ctx.phase_stamps().astype(np.int64)
        life.append(st[:, 9] - st[:, 0])
    ctx.close()
    life = np.array(life[9:], dtype=float).mean(axis=0)
    print(f"{name} {P}: lifetime median {np.median(life):.0f} max {life.max():.0f} min {life.min():.0f}", flush=True)
    desc = np.argsort(-life)
    half = (P + 1) // 2
    dealt = np.concatenate([desc[0::2], desc[1::2]])  # both halves slowest first
    assert len(dealt) == P and len(desc[0::2]) == half
    ids = np.arange(P)
    orders = {"as they are": ids, "slowest first": dealt, "fastest first": dealt[::-1].copy(),
              "random": np.random.default_rng(0).permutation(P),
              # the caller's order dealt alternately to the two halves (what an interleaved split of the streams would run,
              # but with each half contiguous in memory)
              "evens, odds": np.concatenate([ids[0::2], ids[1::2]]),
              # blocks of 64 neighbours dealt alternately: the streams balanced, neighbours still dispatched together
              "blocks of 64": np.concatenate([ids.reshape(-1, 64)[0::2].ravel(), ids.reshape(-1, 64)[1::2].ravel()])
              if P % 128 == 0 else ids,
              "blocks of 256": np.concatenate([ids.reshape(-1, 256)[0::2].ravel(), ids.reshape(-1, 256)[1::2].ravel()])
              if P % 512 == 0 else ids,
              # sorted but NOT dealt: the slow half on one stream, the fast half on the other
              "slow half, fast half": desc}
    if os.environ.get("LPT_ONLY"):
        orders = {k: v for k, v in orders.items() if k in os.environ["LPT_ONLY"].split(";")}
    # each half alone on one stream: are the halves of the caller's order equally expensive?
    for label, (lo, hi) in (("first half alone", (0, half)), ("second half alone", (half, P))):
        w = wl.slice(lo, hi)
        c = context(w)
        c.set_track_streams(1)
        ts = []
        for rep in range(3):
            start(c, w)
            fr = list(range(1, T))
            t0 = time.perf_counter()
            c.track(fr, [1.0] * len(fr), [[j] * w.O for j in fr], seed=1)
            c.sync()
            ts.append(1e3 * (time.perf_counter() - t0) / (T - 1))
        c.close()
        print(f"  {label:22s} {min(ts[1:]):.4f} ms/frame (one stream, {hi - lo} points)", flush=True)
    # ONE context alive at a time: every context brings its own two streams, and the streams of a process share a few
    # hardware queues -- with four contexts alive the last one's two streams shared a queue (first version of this script:
    # "random" 17 .. 65 % slower, an artefact)
    times = {k: [] for k in orders}
    for rnd in range(2):
        for label, order in orders.items():
            w = wl.slice(0, P)
            w.xy, w.params = wl.xy[order], wl.params[order]
            c = context(w)
            for rep in range(4):
                start(c, w)
                fr = list(range(1, T))
                t0 = time.perf_counter()
                c.track(fr, [1.0] * len(fr), [[j] * w.O for j in fr], seed=1)
                c.sync()
                if rep:
                    times[label].append(1e3 * (time.perf_counter() - t0) / (T - 1))
            c.close()
    for label in orders:
        t = sorted(times[label])
        print(f"  {label:22s} median {t[len(t) // 2]:.4f} ms/frame  min {t[0]:.4f} max {t[-1]:.4f}", flush=True)


if __name__ == "__main__":
    main()
