import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glimpse_amd import _lib as lib
from oracle import motion as omotion
from oracle import tracker as otracker
g = dict(np.load("tests/golden/g8_c2mini.npz"))
frames, cams = g["obs0_frames"], g["obs0_cams"]
T = len(frames); N = int(g["n_particles"][0]); tile = tuple(int(v) for v in g["tile_size"])
rng = np.random.default_rng(5); P = 256
params = np.tile(g["params"][0], (P, 1)); params[:, 0:2] = rng.uniform(-6.0, 6.0, (P, 2))
def dev_run(seed, math, host=None):
    with lib.Context(P, N, 1, max_tile=31, max_search_dim=128, max_frames=T) as ctx:
        ctx.observer_init(0, T, frames.shape[2], frames.shape[1], 1, float(g["sigmas"][0]))
        ctx.observer_set_cameras(0, cams)
        for i, f in enumerate(frames): ctx.observer_upload_frame(0, i, f)
        ctx.begin_sequence(P, N, tile); ctx.set_motion_cartesian(params); ctx.set_math(math)
        ctx.set_frame(0)
        if host is None:
            ctx.init_particles(seed=seed)
        else:
            ctx.init_particles(normals=host.standard_normal((P, N, 6)))
        ctx.init_templates(0, 0); ctx.record_moments(0)
        for i in range(1, T):
            if host is None: ctx.step(i, 1.0, [i], seed=seed)
            else: ctx.step(i, 1.0, [i], normals=host.standard_normal((P, N, 3)), u=host.random(P))
        return ctx.get_moments(0, T)[-1]
observers = [otracker.Observer(list(frames), cams, float(g["sigmas"][0]))]
def oracle_run(seed):
    np.random.seed(seed)
    models = [omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13], axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=N) for q in params]
    res = otracker.track(models, observers, np.arange(T)[:, None], np.ones(T - 1), tile_size=tile)
    return np.concatenate((res["means"][:, -1], res["sigmas"][:, -1]), axis=1)
a, b = oracle_run(11), oracle_run(12)
runs = {"dev fast s1": dev_run(1, "fast"), "dev fast s2": dev_run(2, "fast"), "dev exact s1": dev_run(1, "exact"),
        "dev host-normals": dev_run(0, "exact", np.random.default_rng(77)), "oracle b": b}
for name, r in runs.items():
    for k in (0, 3):
        d = r[:, k] - a[:, k]
        print(name, "comp", k, "mean %.4g rms %.4g mad %.4g p95 %.4g max %.4g" % (d.mean(), np.sqrt((d**2).mean()), np.median(np.abs(d)), np.percentile(np.abs(d), 95), np.abs(d).max()), "sigma med %.4g" % np.median(r[:, 6 + k]))
