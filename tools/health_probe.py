"""Diagnostic: per-point status bits / observer statuses of a workload under the fused and staged paths."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimpse_amd import _lib, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
P = int(sys.argv[2]) if len(sys.argv) > 2 else None
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
wl = workloads.Workload(name, n_frames=T, n_points=P)
frames = [wl.frames(o) for o in range(wl.O)]
for mode in (1, 0):
    with _lib.Context(wl.P, wl.N, wl.O, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_fused(mode)
        ctx.set_frame(0)
        ctx.init_particles(seed=1234)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        print("mode", mode, "after templates: status bits", np.unique(ctx.point_status(), return_counts=True))
        for i in range(1, T):
            ctx.step(i, 1.0, [i] * wl.O, seed=1234)
            st, ob, ef = ctx.point_status(), ctx.observer_status(), ctx.point_error_frame()
            print(" frame", i, "pt bits", dict(zip(*np.unique(st, return_counts=True))), "obs", [dict(zip(*np.unique(ob[o], return_counts=True))) for o in range(wl.O)])
        bx = ctx.search_boxes()
        bad = np.nonzero(st)[0][:5]
        print(" example bad points", bad, "err frames", ef[bad], "boxes", bx[:, bad].tolist())
