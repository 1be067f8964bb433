import itertools, sys, numpy as np
sys.path.insert(0, "/root/repo")
from glimpse_amd import _lib as lib, workloads
T = 4
bad = 0
for N, tile, bits, math in itertools.product((64, 511, 513, 5120, 5121, 10240), (5, 15, 47), (8, 16), ("exact", "fast")):
    wl = workloads.Workload("C2", n_frames=T, n_points=5, n_particles=N, imgsz=(640, 640))
    wl.tile = (tile, tile)
    wl.bits = bits
    frames = [wl.frames(0)]
    res = []
    for mode in (1, 0):
        with lib.Context(wl.P, wl.N, 1, max_tile=48, max_search_dim=200, max_frames=T) as ctx:
            workloads.setup_context(ctx, wl, frames)
            ctx.set_math(math); ctx.set_fused(mode); ctx.set_debug(2)
            ctx.set_frame(0); ctx.init_particles(seed=2); ctx.init_templates(0, 0); ctx.record_moments(0)
            idx = []
            for i in range(1, T):
                ctx.step(i, 1.0, [i], seed=2)
                idx.append(ctx.resample_indices())
            stages = {k for k, v in ctx.profile_get().items() if v[1] > 0}
            res.append((ctx.get_particles(), ctx.get_weights(), ctx.get_moments(0, T), np.stack(idx), "point_step" in stages,
                        ctx.observer_status().copy(), ctx.point_status().copy()))
    same = all(np.array_equal(res[0][k], res[1][k]) for k in (0, 1, 3, 5, 6)) and np.allclose(res[0][2], res[1][2], rtol=1e-12, atol=1e-13, equal_nan=True)
    ok = (res[0][5] == 0).all()
    if not same:
        bad += 1
    print(N, tile, bits, math, "fused" if res[0][4] else "STAGED-ONLY", "same" if same else "DIFFERENT", "ok" if ok else f"status {res[0][5].ravel()}", flush=True)
print("bad", bad)
