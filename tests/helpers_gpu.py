"""Drive the C-ABI context through a golden end-to-end fixture (host-fed RNG = parity mode)."""
import numpy as np

from glimpse_amd import _lib
from tests.helpers_golden import draws_from


def context_for(g, max_search_dim=128, debug=True):
    O = int(g["n_obs"])
    P = len(g["n_particles"])
    N = int(g["n_particles"][0])
    assert (g["n_particles"] == N).all()
    T = len(g["matching"])
    tile = tuple(int(v) for v in g["tile_size"])
    ctx = _lib.Context(P, N, O, max_tile=max(31, max(tile)), max_search_dim=max_search_dim, max_frames=T)
    for o in range(O):
        frames = g[f"obs{o}_frames"]
        ch = 1 if frames.ndim == 3 else frames.shape[3]
        ctx.observer_init(o, len(frames), frames.shape[2], frames.shape[1], ch, float(g["sigmas"][o]))
        ctx.observer_set_cameras(o, g[f"obs{o}_cams"])
        for i, f in enumerate(frames):
            ctx.observer_upload_frame(o, i, f)
    ctx.begin_sequence(P, N, tile)
    ctx.set_motion_cartesian(g["params"])
    if debug:
        ctx.set_debug(True)
    return ctx


def batched_draws(g):
    """Per-step host-fed normals [P][N][3] / u [P] from the reference's recorded stream."""
    draws = draws_from(g)
    P = len(draws)
    N = int(g["n_particles"][0])
    nsteps = max(len(d["evolve"]) for d in draws)
    init = np.stack([d["init"] for d in draws])
    ev = np.zeros((nsteps, P, N, 3))
    us = np.zeros((nsteps, P))
    for p, d in enumerate(draws):
        for s in range(len(d["evolve"])):
            ev[s, p] = d["evolve"][s]
            us[s, p] = d["u"][s]
    return init, ev, us


def run_free(g, ctx, collect=True):
    """Reference frame loop (tracker.py:326-357) for all points at once.  Returns per-step records."""
    matching = g["matching"]
    taus = np.diff(g["datetimes_days"])
    T, O = matching.shape
    template_indices = (matching >= 0).argmax(axis=0)
    observed = (matching >= 0).any(axis=1)
    first = int(np.argmax(observed))
    last = T - 1 - int(np.argmax(observed[::-1]))
    init, ev, us = batched_draws(g)
    records = []
    step = 0
    for i in range(first, last + 1):
        ctx.set_frame(i)
        if i == first:
            ctx.init_particles(normals=init)
        else:
            ctx.evolve(taus[i - 1], normals=ev[step])
        rec = {"i": i}
        if collect:
            rec["evolved"] = ctx.get_particles()
        for o in np.nonzero(template_indices == i)[0]:
            ctx.init_templates(int(o), int(matching[i][o]))
        if i > first:
            ctx.update_weights(matching[i])
            if collect:
                rec["weights"] = ctx.get_weights()
                rec["obs_status"] = ctx.observer_status()
                rec["dbg"] = [[ctx.likelihood_debug(o, p) for p in range(ctx.P)] for o in range(O)]
            ctx.resample(u=us[step])
            if collect:
                rec["idx"] = ctx.resample_indices()
                rec["particles"] = ctx.get_particles()
                rec["out_weights"] = ctx.get_weights()
            step += 1
        ctx.record_moments(i)
        records.append(rec)
    moments = ctx.get_moments(0, T)  # [T][P][12]
    return records, moments
