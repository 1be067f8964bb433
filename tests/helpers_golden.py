"""Shared helpers: rebuild oracle inputs from the g8_* end-to-end fixtures."""
import numpy as np

from oracle import motion as omotion
from oracle import tracker as otracker


def observers_from(g):
    obs = []
    for o in range(int(g["n_obs"])):
        obs.append(otracker.Observer(list(g[f"obs{o}_frames"]), g[f"obs{o}_cams"], float(g["sigmas"][o])))
    return obs


def models_from(g):
    models = []
    for p, n in zip(g["params"], g["n_particles"]):
        models.append(
            omotion.CartesianMotion(
                xy=p[0:2], xy_sigma=p[2:4], vxyz=p[4:7], vxyz_sigma=p[7:10], axyz=p[10:13],
                axyz_sigma=p[13:16], dem=p[16], dem_sigma=p[17], n=int(n),
            )
        )
    return models


def taus_from(g):
    return np.diff(g["datetimes_days"])


def draws_from(g):
    """Split the recorded legacy-RNG stream (reference call order) into per-track draws.

    Per track: randn(n,2), randn(n), randn(n,3), then per step randn(n,3) + random().
    A track that errors out before/at template initialisation consumed only its init draws.
    """
    randn = [g[f"randn{i}"] for i in range(int(g["n_randn"]))]
    rand = list(g["random"])
    starts = list(g["track_starts"]) + [int(g["n_steps"])]
    draws, ri, ui = [], 0, 0
    for t in range(len(g["n_particles"])):
        init = np.column_stack((randn[ri], randn[ri + 1], randn[ri + 2]))
        ri += 3
        nsteps = starts[t + 1] - starts[t] if t + 1 < len(starts) else 0
        ev, us = [], []
        for _ in range(nsteps):
            ev.append(randn[ri]); ri += 1
            us.append(rand[ui]); ui += 1
        draws.append({"init": init, "evolve": ev, "u": us})
    assert ri == len(randn) and ui == len(rand), (ri, len(randn), ui, len(rand))
    return draws
