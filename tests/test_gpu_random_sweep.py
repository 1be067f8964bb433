"""Seeded random configurations through the fused frame step (glh_step, C ABI) against the oracle's whole-track
restatement (oracle/tracker.py: tracker.py:305-374) on the same host-fed draws: cameras with random subsets of the
distortion terms (k1..k6, p1, p2, principal point offset, earth-curvature correction), oblique stations, odd / even /
non-square templates, particle counts that are not multiples of the wave or block size, gray and RGB frames,
fractional and negative time steps, a DEM likelihood term, one or two observers with missing images."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-7  # (BASELINE.json north_star asks for 1e-5 relative; the float32 SSD surface sets the floor)


def _random_case(seed):
    from glimpse_amd import synth

    rng = np.random.default_rng(1000 + seed)
    imgsz = (int(rng.integers(300, 520)), int(rng.integers(300, 520)))
    n_obs = 1 if seed % 3 else 2
    T = 4
    k = np.zeros(6)
    nk = int(rng.integers(0, 4))
    k[:nk] = rng.uniform(-0.05, 0.05, nk) * np.array([1.0, 0.3, 0.05])[:nk]
    if seed % 5 == 0:
        nden = int(rng.integers(1, 4))
        k[3:3 + nden] = rng.uniform(-0.01, 0.01, nden)
    p = rng.uniform(-0.002, 0.002, 2) if seed % 4 == 1 else np.zeros(2)
    c = rng.uniform(-6, 6, 2) if seed % 2 else np.zeros(2)
    correction = bool(seed % 7 == 3)
    cam0 = synth.pack_camera(imgsz=imgsz, f=(float(rng.uniform(700, 1100)), float(rng.uniform(700, 1100))), c=c, k=k,
                             p=p, xyz=(float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), 100.0),
                             viewdir=(float(rng.uniform(-20, 20)), -90.0 + float(rng.uniform(0, 4)), 0.0),
                             correction=correction)
    cams = [cam0]
    if n_obs == 2:
        cams.append(synth.pack_camera(imgsz=imgsz, f=900.0, k=(0.02, 0, 0), xyz=(12.0, -9.0, 95.0),
                                      viewdir=(-53.13, -82.0, 0.0)))
    channels = 3 if seed % 4 == 2 else 1
    velocity = (float(rng.uniform(0.05, 0.25)), float(rng.uniform(-0.1, 0.1)))
    scene = synth.default_scene(cams[-1], seed=seed, velocity=velocity, n_frames=T, margin=40.0)
    # with a negative first time step the scene is rendered at the times the filter will visit
    taus = np.array([1.0, float(rng.choice([1.0, 0.5, 1.5])), 1.0])
    times = np.concatenate(([0.0], np.cumsum(taus)))
    if seed % 6 == 4:
        taus, times = -taus, -times
    frames = [[scene.render(cam, float(t), channels=channels) for t in times] for cam in cams]
    tile = (int(rng.integers(9, 34)), int(rng.integers(9, 34)))
    N = int(rng.choice([37, 64, 100, 513, 777, 1500, 2049, 3000]))
    P = 3
    # points every camera sees with room for the template and the search tile
    xy = []
    margin = 0.5 * max(tile) + 45.0
    tries = 0
    while len(xy) < P:
        tries += 1
        assert tries < 10000
        cand = np.array([rng.uniform(-14, 14), rng.uniform(-14, 14)])
        ok = True
        for cam in cams:
            uv = synth.project(cam, np.array([[cand[0], cand[1], 0.0]]))[0]
            ok &= bool(margin < uv[0] < cam[6] - margin and margin < uv[1] < cam[7] - margin)
        if ok:
            xy.append(cand)
    dem_sigma = 0.4 if seed % 3 == 2 else 0.0
    params = np.zeros((P, 18))
    params[:, 0:2] = xy
    params[:, 2:4] = 0.15
    params[:, 4:7] = (velocity[0] * np.sign(taus[0]), velocity[1] * np.sign(taus[0]), 0.0)
    params[:, 7:10] = (0.1, 0.1, 0.03 if dem_sigma else 0.0)
    params[:, 13:16] = (0.04, 0.04, 0.01 if dem_sigma else 0.0)
    params[:, 17] = dem_sigma
    matching = np.tile(np.arange(T)[:, None], (1, n_obs))
    if n_obs == 2 and seed % 2 == 0:
        matching[2, 1] = -1  # the second station has no image for frame 2
    sigmas = [0.3, 0.45][:n_obs]
    return dict(imgsz=imgsz, cams=cams, frames=frames, channels=channels, tile=tile, N=N, P=P, params=params,
                matching=matching, taus=taus, sigmas=sigmas, T=T)


@pytest.mark.parametrize("math", ["exact", "fast"])
@pytest.mark.parametrize("seed", range(24))
def test_fused_step_matches_the_oracle_on_random_configurations(seed, math):
    """exact: the reference's rounding; fast: the arithmetic of device-RNG runs (GLH_MATH_FAST) on the same host-fed
    draws -- the general fast instantiation (flags == 3) -- meets the oracle just the same: indices bit for bit."""
    from glimpse_amd import _lib
    from oracle import motion as omotion
    from oracle import tracker as otracker

    cs = _random_case(seed)
    P, N, T, O = cs["P"], cs["N"], cs["T"], len(cs["cams"])
    rng = np.random.default_rng(seed)
    init = rng.standard_normal((P, N, 6))
    ev = rng.standard_normal((T - 1, P, N, 3))
    us = rng.random((T - 1, P))
    with _lib.Context(P, N, O, max_tile=max(31, max(cs["tile"])), max_search_dim=160, max_frames=T) as ctx:
        for o in range(O):
            ctx.observer_init(o, T, cs["imgsz"][0], cs["imgsz"][1], cs["channels"], cs["sigmas"][o])
            ctx.observer_set_cameras(o, np.tile(cs["cams"][o], (T, 1)))
            for t in range(T):
                ctx.observer_upload_frame(o, t, cs["frames"][o][t])
        ctx.begin_sequence(P, N, cs["tile"])
        ctx.set_motion_cartesian(cs["params"])
        ctx.set_math(math)
        ctx.set_frame(0)
        ctx.init_particles(normals=init)
        for o in range(O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        ctx.set_debug(2)  # resample indices of the fused kernel
        idx = []
        for i in range(1, T):
            ctx.step(i, cs["taus"][i - 1], cs["matching"][i], normals=ev[i - 1], u=us[i - 1])
            idx.append(ctx.resample_indices())
            assert ctx.last_variant()[3] == (3 if math == "fast" else 0)
        got = ctx.get_moments(0, T)
        status = ctx.point_status()
        obs_status = ctx.observer_status()
    assert (status == 0).all(), status
    assert (obs_status == _lib.OBS_OK).all() or (cs["matching"][T - 1] < 0).any()
    observers = [otracker.Observer(cs["frames"][o], np.tile(cs["cams"][o], (T, 1)), cs["sigmas"][o]) for o in range(O)]
    n_bad = 0
    for p in range(P):
        q = cs["params"][p]
        model = omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13],
                                        axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=N)
        draws = {"init": init[p], "evolve": [ev[s, p] for s in range(T - 1)], "u": [us[s, p] for s in range(T - 1)]}
        trace = []
        ref = otracker.track_one(model, observers, cs["matching"], cs["taus"], tile_size=cs["tile"], draws=draws,
                                 trace=trace)
        np.testing.assert_allclose(got[:, p, 0:6], ref["means"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(got[:, p, 6:12], ref["sigmas"], rtol=RTOL, atol=1e-8)
        steps = [tr for tr in trace if "idx" in tr]
        assert len(steps) == T - 1
        for s, tr in enumerate(steps):
            n_bad += int((idx[s][p] != tr["idx"]).sum())
    assert n_bad == 0, f"{n_bad} resample indices differ from the oracle"
