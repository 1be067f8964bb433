import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from glimpse_amd import _lib as lib, workloads
T = int(sys.argv[4]) if len(sys.argv) > 4 else 12
PTS = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
wl = workloads.Workload("C3", n_frames=T, n_points=PTS, n_particles=5000)
DT = np.float64 if len(sys.argv) > 1 and sys.argv[1] == "64" else np.float32
frames = [[f.astype(DT) for f in wl.frames(0)]]
with lib.Context(wl.P, wl.N, 1, max_search_dim=160, max_frames=T) as ctx:
    ctx.observer_init(0, T, wl.imgsz[0], wl.imgsz[1], 1, wl.sigmas[0])
    ctx.observer_set_depth(0, DT)
    ctx.observer_set_cameras(0, np.tile(wl.cams[0], (T, 1)))
    for t in range(T):
        ctx.observer_upload_frame(0, t, frames[0][t])
    ctx.begin_sequence(wl.P, wl.N, wl.tile)
    ctx.set_motion_cartesian(wl.params)
    ctx.set_math("fast")
    ctx.set_fused(int(sys.argv[3]) if len(sys.argv) > 3 else 1)  # (1: the fused kernel, 0: the staged kernels)
    ctx.set_frame(0); ctx.init_particles(seed=3); ctx.init_templates(0, 0); ctx.record_moments(0)
    ctx.profile_enable(True)
    for i in range(1, T):
        if i == T - 3:
            ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
        ctx.step(i, 1.0, [i], seed=3)
    ctx.sync()
    print("ms/frame (last 3)", 1e3 * (time.perf_counter() - t0) / 3, {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_get().items() if v[1]})
    box = ctx.search_boxes()[0]
    print("tile median", np.median(box[:, 2] - box[:, 0]), "ok", (ctx.observer_status() == 0).mean())
