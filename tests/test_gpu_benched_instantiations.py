"""The kernel instantiations bench.py times meet the oracle DIRECTLY, at their own particle counts: host-fed draws
(exact arithmetic: the reference's rounding), BASELINE's frames (2048^2, k1-k3), 31x31 templates,

    C3:  N =  5 000  -> k_point_step<512, 10, 4, 1, ...>
    C4:  N = 10 000  -> k_point_step<1024, 10, 4, 1, ...>
    C5:  N =  5 000, two observers, DEM term -> k_point_step<512, 0, 4, 2, ...>

resample indices bit for bit, posteriors to 1e-7 (a few points are enough: the oracle is a NumPy loop)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T = 3


@pytest.mark.parametrize("name,N", [("C3", 5000), ("C3", 10000), ("C5", 5000)])
def test_benched_instantiation_matches_the_oracle(name, N):
    from glimpse_amd import _lib, workloads
    from oracle import motion as omotion
    from oracle import tracker as otracker

    P = 2
    wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N)
    assert wl.imgsz == (2048, 2048) and wl.tile == (31, 31)
    frames = [wl.frames(o) for o in range(wl.O)]
    rng = np.random.default_rng(N + wl.O)
    init = rng.standard_normal((P, N, 6))
    ev = rng.standard_normal((T - 1, P, N, 3))
    us = rng.random((T - 1, P))
    with _lib.Context(P, N, wl.O, max_tile=31, max_search_dim=200, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_frame(0)
        ctx.init_particles(normals=init)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        ctx.set_debug(2)  # keeps the resample indices; the step stays on the fused kernel
        idx = []
        for i in range(1, T):
            ctx.step(i, 1.0, [i] * wl.O, normals=ev[i - 1], u=us[i - 1])
            idx.append(ctx.resample_indices())
        got = ctx.get_moments(0, T)
        assert (ctx.point_status() == 0).all() and (ctx.observer_status() == _lib.OBS_OK).all()
        stages = {k: v for k, v in ctx.profile_get().items() if v[1] > 0}
    assert "point_step" in stages and "resample" not in stages  # the fused kernel took the steps
    observers = [otracker.Observer(frames[o], np.tile(wl.cams[o], (T, 1)), wl.sigmas[o]) for o in range(wl.O)]
    matching = np.tile(np.arange(T)[:, None], (1, wl.O))
    n_bad = 0
    for p in range(P):
        q = wl.params[p]
        model = omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13],
                                        axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=N)
        draws = {"init": init[p], "evolve": [ev[s, p] for s in range(T - 1)], "u": [us[s, p] for s in range(T - 1)]}
        trace = []
        ref = otracker.track_one(model, observers, matching, np.ones(T - 1), tile_size=wl.tile, draws=draws, trace=trace)
        np.testing.assert_allclose(got[:, p, 0:6], ref["means"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(got[:, p, 6:12], ref["sigmas"], rtol=1e-7, atol=1e-8)
        steps = [tr for tr in trace if "idx" in tr]
        assert len(steps) == T - 1
        for s, tr in enumerate(steps):
            n_bad += int((idx[s][p] != tr["idx"]).sum())
    assert n_bad == 0, f"{n_bad} resample indices differ from the oracle"


@pytest.mark.parametrize("name,P,N,variant", [("C3", 6, 5000, (512, 10, 1)), ("C4", 3, 10000, (1024, 10, 1)),
                                              ("C5", 4, 5000, (512, 10, 2)), ("C2", 8, 2000, (512, 4, 1))])
def test_which_instantiation_takes_which_step(name, P, N, variant):
    """bench.py's configurations in fast arithmetic run on the COMMON instantiation (flags == 5: fast | contract) from their second
    frame on -- the one the fast parity tests (fused == staged bit for bit, fast vs exact to rounding:
    test_gpu_fast_math.py, test_gpu_fullsize.py) therefore exercise with the same Philox / compact-state runs; the
    first frame (expanded input), host-fed draws and a frame without an image go to the general instantiation, exact
    arithmetic to the exact one."""
    from glimpse_amd import _lib, workloads

    T = 5
    wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
    frames = [wl.frames(o) for o in range(wl.O)]
    rng = np.random.default_rng(3)
    with _lib.Context(P, N, wl.O, max_tile=31, max_search_dim=200, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        assert ctx.last_variant() == (0, 0, 0, 0)
        for math, flags_later in (("fast", 5), ("exact", 0)):
            ctx.set_math(math)
            ctx.set_frame(0)
            ctx.init_particles(seed=5)
            for o in range(wl.O):
                ctx.init_templates(o, 0)
            ctx.step(1, 1.0, [1] * wl.O, seed=5)
            first = ctx.last_variant()
            ctx.step(2, 1.0, [2] * wl.O, seed=5)
            later = ctx.last_variant()
            assert first[:3] == variant and later[:3] == variant
            assert later[3] == flags_later
            assert first[3] == (3 if math == "fast" else 0)  # expanded input: general instantiation in fast arithmetic
            if math == "fast":
                ctx.step(3, 1.0, [3] * wl.O, normals=rng.standard_normal((P, N, 3)), u=rng.random(P))
                assert ctx.last_variant()[3] == 3  # host-fed draws
                ctx.step(4, 1.0, [4] + [-1] * (wl.O - 1), seed=5)
                assert ctx.last_variant()[3] == (3 if wl.O > 1 else 5)  # an observer without an image
        assert (ctx.point_status() == 0).all()
