import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from glimpse_amd import _lib as lib, workloads
T = 2
for PTS in (64, 1024, 4096):
    wl = workloads.Workload("C3", n_frames=T, n_points=PTS, n_particles=5000)
    frames = [wl.frames(0)]
    with lib.Context(wl.P, wl.N, 1, max_search_dim=160, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_frame(0); ctx.init_particles(seed=3)
        ctx.profile_enable(True)
        for rep in range(3):
            ctx.profile_reset()
            ctx.init_templates(0, 0)
            ctx.sync()
            print(PTS, rep, {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_get().items() if v[1]}, flush=True)
