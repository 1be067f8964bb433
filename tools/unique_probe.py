import sys, numpy as np
sys.path.insert(0,'/root/repo')
from glimpse_amd import _lib, workloads
wl = workloads.Workload("C3", n_frames=26, n_points=256, n_particles=5000)
frames=[wl.frames(0)]
with _lib.Context(wl.P, wl.N, 1, max_frames=26) as ctx:
    workloads.setup_context(ctx, wl, frames)
    ctx.set_frame(0); ctx.init_particles(seed=1); ctx.init_templates(0,0); ctx.record_moments(0)
    for i in range(1,26):
        ctx.step(i,1.0,[i],seed=1)
        if i in (1,2,3,5,8,12,25):
            p=ctx.get_particles()
            u=[len(np.unique(p[k,:,0])) for k in range(0,wl.P,16)]
            print(i, "unique fraction", np.mean(u)/wl.N)
