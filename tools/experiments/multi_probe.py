"""Several frames per launch (glh_set_frames_per_launch): wall time of the whole sequence and bit-identity with the
frame-by-frame launches.   python tools/multi_probe.py [C3] [points] [particles] [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimpse_amd import _lib, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
P = int(sys.argv[2]) if len(sys.argv) > 2 else None
N = int(sys.argv[3]) if len(sys.argv) > 3 else None
T = int(sys.argv[4]) if len(sys.argv) > 4 else 100
wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N)
frames = [wl.frames(o) for o in range(wl.O)]
ref = None
with _lib.Context(wl.P, wl.N, wl.O, max_frames=T, max_search_dim=320) as ctx:
    workloads.setup_context(ctx, wl, frames)
    ctx.set_math("fast")
    for fpl in (1, 1, 5, 25, 0, 0, 1):
        ctx.set_frames_per_launch(fpl)
        ctx.set_frame(0)
        ctx.init_particles(seed=3)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        ctx.sync()
        t0 = time.perf_counter()
        fr = list(range(1, T))
        ctx.track(fr, [1.0] * len(fr), [[i] * wl.O for i in fr], seed=3)
        ctx.sync()
        dt = time.perf_counter() - t0
        mom, part, w = ctx.get_moments(0, T), ctx.get_particles(), ctx.get_weights()
        st = ctx.observer_status_frames(1, T - 1)
        same = "(reference)"
        if ref is None:
            ref = (mom, part, w, st)
        else:
            same = f"moments {np.array_equal(mom, ref[0], equal_nan=True)} particles {np.array_equal(part, ref[1])} " \
                   f"weights {np.array_equal(w, ref[2])} status {np.array_equal(st, ref[3])}"
        print(f"frames per launch {fpl:3d}: {1e3 * dt / (T - 1):.4f} ms/frame  variant {list(ctx.last_variant())}  {same}", flush=True)
