import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from glimpse_amd import _lib as lib, workloads
T = 12
PTS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = workloads.Workload("C3", n_frames=T, n_points=PTS, n_particles=5000)
frames = [wl.frames(0)]
with lib.Context(wl.P, wl.N, 1, max_search_dim=160, max_frames=T) as ctx:
    workloads.setup_context(ctx, wl, frames)
    ctx.set_math("fast"); ctx.set_fused(0)
    ctx.set_frame(0); ctx.init_particles(seed=3); ctx.init_templates(0, 0); ctx.record_moments(0)
    ctx.profile_enable(True)
    for i in range(1, T):
        if i == T - 3:
            ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
        ctx.step(i, 1.0, [i], seed=3)
    ctx.sync()
    print("u8 staged ms/frame", 1e3 * (time.perf_counter() - t0) / 3, {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_get().items() if v[1]})
