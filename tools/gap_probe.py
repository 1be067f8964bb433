"""Diagnostic: wall time per frame of one glh_track call with the per-launch profiling events on and off."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from glimpse_amd import _lib, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = workloads.Workload(name)
T = workloads.CONFIGS[name]["frames"]
import bench
frames = bench.render_frames(wl, 8)
with _lib.Context(wl.P, wl.N, wl.O, max_frames=T) as ctx:
    workloads.setup_context(ctx, wl, frames)
    ctx.set_math("fast")
    for prof in (False, True, False, True):
        ctx.profile_enable(prof)
        ctx.set_frame(0); ctx.init_particles(seed=3)
        for o in range(wl.O): ctx.init_templates(o, 0)
        ctx.record_moments(0)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.track(list(range(1, T)), [1.0] * (T - 1), [[i] * wl.O for i in range(1, T)], seed=3)
        ctx.sync()
        dt = time.perf_counter() - t0
        print(name, "profiling", prof, "ms/frame", 1e3 * dt / (T - 1))
