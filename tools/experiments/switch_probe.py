"""Experiment (round 5): wall time per frame of glh_track over a whole sequence from the prior with an environment switch
of the library off and on, same process, interleaved (the library reads its switches at every glh_track call):

    python tools/experiments/switch_probe.py WORKLOAD POINTS FRAMES VAR=VALUE [VAR=VALUE ...]

e.g.  switch_probe.py C2 256 50 GLH_TRACK_GRAPH=1        (the frame loop as one hipGraph launch)
      switch_probe.py C3 4096 30 GLH_PT_BIG_FRAMES=8     (1 024-thread workgroups for the wide first frames)
Prints the median of five interleaved repetitions per variant and the largest difference of the posterior history from
the plain run's.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glimpse_amd import _lib, workloads  # noqa: E402

name, P, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
switches = [a.split("=", 1) for a in sys.argv[4:]]
import bench  # noqa: E402  (its forked renderers: before anything touches the GPU; GLH_FRAME_CACHE keeps the frames)

wl = workloads.Workload(name, n_frames=T, n_points=P)
frames = bench.render_frames(wl, bench.usable_cores())

ctx = _lib.Context(wl.P, wl.N, wl.O, device_id=0, max_tile=max(wl.tile), max_search_dim=320, max_frames=T)
workloads.setup_context(ctx, wl, frames)
ctx.set_math("fast")


def once():
    # The sequence runs TWICE back to back and the second run is the one timed: a run that starts after the device has been
    # idle for some tens of milliseconds (the host comparing two histories, say) is 4-5 % slower than one that starts on
    # the heels of another -- the first version of this script timed single runs and charged that to whichever variant
    # came first in the loop (profiles/ab_r05/r5j52_switch_probe_bias.txt).
    for timed in (False, True):
        ctx.set_frame(0)
        ctx.init_particles(seed=1)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        ctx.sync()
        fr = list(range(1, T))
        t0 = time.perf_counter()
        ctx.track(fr, [1.0] * len(fr), [[j] * wl.O for j in fr], seed=1)
        ctx.sync()
        dt = time.perf_counter() - t0
    return 1e3 * dt / (T - 1), ctx.get_moments(0, T)


variants = [("plain", None)] + [(f"{k}={v}", (k, v)) for k, v in switches]
times = {v[0]: [] for v in variants}
ref, diff = None, {}
once()  # warm-up
for rep in range(5):
    for label, sw in variants:
        if sw:
            os.environ[sw[0]] = sw[1]
        ms, mom = once()
        if sw:
            del os.environ[sw[0]]
        times[label].append(ms)
        if label == "plain":
            ref = mom
        else:
            diff[label] = float(np.nanmax(np.abs(mom - ref) / (np.abs(ref) + 1e-9)))
for label, _ in variants:
    t = sorted(times[label])
    print(f"{name} {P}x{wl.N} {T} frames  {label:28s} median {t[len(t) // 2]:.4f} ms/frame  min {t[0]:.4f} max {t[-1]:.4f}"
          + (f"  max rel diff of the history {diff[label]:.2e}" if label in diff else ""), flush=True)
print("status bits:", int((ctx.point_status() != 0).sum()), "variant", list(ctx.last_variant()), flush=True)
ctx.close()
