"""What bench.py TIMES, pinned to the oracle DIRECTLY -- no chain through the exact arithmetic or the staged kernels.

The headline runs GLH_MATH_FAST on the common instantiation (k_point_step<..., fast, contract>, flags == 5) with the
device Philox streams.  Here the numbers of those very streams are read back through the C ABI (glh_debug_draws: the
initialisation normals, every frame's evolve normals and resampling offset, keyed exactly as the kernels key them) and
handed to the oracle's whole-track restatement (oracle/tracker.py: tracker.py:305-374) as its `draws` -- the role
np.random plays in the reference.  Then, for BASELINE's shapes (2048^2 frames, k1-k3, 31x31 templates; N = 5 000,
N = 10 000, two observers + DEM term):

* the resample indices of the device run equal the oracle's at EVERY step, bit for bit;
* the posterior means / sigmas agree to 1e-7 relative (north_star: 1e-5);
* the same over a LONG sequence -- 100 frames from the wide prior into the steady state, 99 resampling steps; one
  flipped index would propagate to every later frame -- in both arithmetics (exact with host-fed draws, fast with the
  device streams);
* at FULL size (C3: 4 096 x 5 000 x 100 frames) fast and exact arithmetic on one Philox stream: the posterior history
  to 1e-9 and the number of differing record indices of the final state, reported and bounded.

The SSD boundary.  cv2.matchTemplate is absent from the reference tree and from this image (parity
unpinned there); OpenCV's formula does not fix the precision of the accumulation.  The oracle restates it twice
(oracle/ssd.c): with a float64 accumulator (what the golden fixtures' stand-in does; the default of every other test)
and with the kernels' float32-along-a-row accumulation.  The two surfaces agree to a few float32 ulps -- and that alone
moves one resampling index every ~10-100 steps of a 5 000-particle filter (measured with the oracle against itself),
after which the two runs are different random realisations.  Index-for-index equality over 100 frames is therefore
asserted against the second restatement, which the kernels' surface equals bit for bit (tests/test_gpu_parity.py:
test_stage_ssd_matches_oracle): what these tests pin is everything around the SSD -- draws, evolve, projection, search
box, tile preparation, spline fit and sampling, weights, resampling, moments -- in the arithmetic bench.py times."""
import functools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 20261004
RTOL = 1e-7


@functools.lru_cache(maxsize=2)
def _workload(name, P, N, T, imgsz=None):
    from glimpse_amd import workloads

    wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=imgsz)
    frames = [np.stack(wl.frames(o)) for o in range(wl.O)]
    return wl, frames


def _oracle_tracks(wl, frames, T, init, ev, us):
    from oracle import motion as omotion
    from oracle import tracker as otracker

    # cv2.matchTemplate restated with the kernels' accumulation (float32 along a template row, float64 across rows:
    # bit for bit the kernels' surface, tests/test_gpu_parity.py): OpenCV does not specify the accumulation, and one
    # unit in the last place of a float32 surface value is enough to move an index once in ~10-100 steps
    observers = [otracker.Observer(list(frames[o][:T]), np.tile(wl.cams[o], (T, 1)), wl.sigmas[o], ssd="row_f32")
                 for o in range(wl.O)]
    matching = np.tile(np.arange(T)[:, None], (1, wl.O))
    means, sigmas, idx = [], [], []
    for p in range(wl.P):
        q = wl.params[p]
        model = omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13],
                                        axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=wl.N)
        draws = {"init": init[p], "evolve": [ev[s, p] for s in range(T - 1)], "u": [us[s, p] for s in range(T - 1)]}
        trace = []
        ref = otracker.track_one(model, observers, matching, np.ones(T - 1), tile_size=wl.tile, draws=draws, trace=trace)
        steps = [tr["idx"] for tr in trace if "idx" in tr]
        assert len(steps) == T - 1
        means.append(ref["means"])
        sigmas.append(ref["sigmas"])
        idx.append(np.stack(steps))
    # (T, P, 6), (T, P, 6), (T - 1, P, N)
    return np.stack(means, axis=1), np.stack(sigmas, axis=1), np.stack(idx, axis=1)


def _device_run(wl, frames, T, math, rng, host=None, max_search_dim=200):
    """The frame loop through the C ABI, one glh_step per frame (fused kernel, resample indices kept).
    rng == "philox": device streams, returned as the draws the oracle needs; rng == "host": `host` = (init, ev, us)."""
    from glimpse_amd import _lib, workloads

    with _lib.Context(wl.P, wl.N, wl.O, max_tile=31, max_search_dim=max_search_dim, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_math(math)
        ctx.set_debug(2)  # keeps the resample indices; the step stays on the fused kernel
        ctx.set_frame(0)
        if rng == "philox":
            ctx.init_particles(seed=SEED)
            init = ctx.debug_draws("init", SEED)
            ev = np.stack([ctx.debug_draws("evolve", SEED, step=i) for i in range(1, T)])
            us = np.stack([ctx.debug_draws("u", SEED, step=i) for i in range(1, T)])
        else:
            init, ev, us = host
            ctx.init_particles(normals=init)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        idx, variants = [], []
        for i in range(1, T):
            if rng == "philox":
                ctx.step(i, 1.0, [i] * wl.O, seed=SEED)
            else:
                ctx.step(i, 1.0, [i] * wl.O, normals=ev[i - 1], u=us[i - 1])
            idx.append(ctx.resample_indices())
            variants.append(ctx.last_variant())
        moments = ctx.get_moments(0, T)
        assert (ctx.point_status() == 0).all()
        assert (ctx.observer_status_frames(1, T - 1) == _lib.OBS_OK).all()
        stages = {k: v for k, v in ctx.profile_get().items() if v[1] > 0}
    assert "point_step" in stages and "resample" not in stages  # the fused kernel took every step
    return dict(moments=moments, idx=np.stack(idx), variants=variants, draws=(init, ev, us))


def _compare(dev, ref, T):
    means, sigmas, idx = ref
    bad_steps = [(s, int((dev["idx"][s] != idx[s]).sum())) for s in range(T - 1) if (dev["idx"][s] != idx[s]).any()]
    assert not bad_steps, f"resample indices differ from the oracle at (step, count): {bad_steps[:10]}"
    np.testing.assert_allclose(dev["moments"][..., 0:6], means, rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(dev["moments"][..., 6:12], sigmas, rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("name,P,N,variant", [("C3", 3, 5000, (512, 10, 1)), ("C4", 2, 10000, (1024, 10, 1)),
                                              ("C5", 3, 5000, (512, 10, 2)), ("C2", 4, 2000, (512, 4, 1))])
def test_benched_arithmetic_and_streams_match_the_oracle(name, P, N, variant):
    """fast arithmetic + device Philox on BASELINE's frames: from the second update on this is the common
    instantiation (flags == 5), the kernel of bench.py's headline, the C4 shard, C5 and C2."""
    T = 7
    wl, frames = _workload(name, P, N, T)
    assert wl.imgsz == (2048, 2048)
    dev = _device_run(wl, frames, T, "fast", "philox")
    assert dev["variants"][0] == variant + (3,)  # the first update reads the expanded prior: general fast code
    assert all(v == variant + (5,) for v in dev["variants"][1:]), dev["variants"]
    init, ev, us = dev["draws"]
    assert np.abs(init).max() < 6.7 and abs(init.std() - 1) < 0.02 and ((0 <= us) & (us < 1)).all()
    _compare(dev, _oracle_tracks(wl, frames, T, init, ev, us), T)


@pytest.mark.parametrize("name,P,N", [("C3", 2, 5000), ("C5", 2, 5000)])
@pytest.mark.parametrize("mode", ["exact-host", "fast-philox"])
def test_long_sequence_matches_the_oracle(name, P, N, mode):
    """100 frames (C3) / 60 frames (C5, two observers) from the wide prior: search tiles shrink from ~100 px to ~40 px
    (HBM-workspace tiles, then LDS tiles; coefficient form, then per-cell form of the sampling).  Indices equal at
    every step in the exact arithmetic on host-fed draws (the reference's rounding) AND in the benched fast arithmetic
    on the device streams."""
    T = 100 if name == "C3" else 60
    wl, frames = _workload(name, P, N, T)
    if mode == "exact-host":
        rng = np.random.default_rng(99)
        host = (rng.standard_normal((P, N, 6)), rng.standard_normal((T - 1, P, N, 3)), rng.random((T - 1, P)))
        dev = _device_run(wl, frames, T, "exact", "host", host=host)
        assert all(v[3] == 0 for v in dev["variants"])
    else:
        dev = _device_run(wl, frames, T, "fast", "philox")
        assert all(v[3] == 5 for v in dev["variants"][1:])
    init, ev, us = dev["draws"]
    _compare(dev, _oracle_tracks(wl, frames, T, init, ev, us), T)
    # the filter has converged on the scene's motion
    assert abs(np.median(dev["moments"][-1, :, 3]) - 0.15) < 0.03


def test_full_size_c3_fast_against_exact_on_one_stream():
    """BASELINE C3 whole -- 4 096 points x 5 000 particles x 100 frames from the prior, the bench's own sequence -- in
    both arithmetics on the same Philox stream: 2e9 resample decisions.  The posterior history agrees to 1e-9 wherever
    no decision differed; points whose final record indices differ (a cumulative weight within an ulp of a systematic
    position) are counted, reported and bounded."""
    from glimpse_amd import _lib, workloads

    T = 100
    wl = workloads.Workload("C3", n_frames=T)
    assert (wl.P, wl.N) == (4096, 5000)
    small, frames = _workload("C3", 2, 5000, T)  # (the frames of a configuration do not depend on its points)
    assert np.array_equal(small.cams[0], wl.cams[0])
    out = {}
    with _lib.Context(wl.P, wl.N, 1, max_tile=31, max_search_dim=320, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        for math in ("fast", "exact"):
            ctx.set_math(math)
            ctx.set_debug(2)
            ctx.set_frame(0)
            ctx.init_particles(seed=SEED)
            ctx.init_templates(0, 0)
            ctx.record_moments(0)
            fr = list(range(1, T))
            ctx.track(fr, [1.0] * len(fr), [[i] for i in fr], seed=SEED)
            assert (ctx.point_status() == 0).all()
            assert (ctx.observer_status_frames(1, T - 1) == _lib.OBS_OK).all()
            out[math] = dict(moments=ctx.get_moments(0, T), idx=ctx.resample_indices())
    same = (out["fast"]["idx"] == out["exact"]["idx"]).all(axis=1) & \
        np.isclose(out["fast"]["moments"], out["exact"]["moments"], rtol=1e-9, atol=1e-10).all(axis=(0, 2))
    n_diff = int((~same).sum())
    print(f"C3 full size, fast vs exact on one Philox stream: {n_diff} of {wl.P} points carry a differing decision "
          f"after {T - 1} steps ({(T - 1) * wl.P * wl.N:.2e} resample decisions)")
    np.testing.assert_allclose(out["fast"]["moments"][:, same], out["exact"]["moments"][:, same], rtol=1e-9, atol=1e-10)
    np.testing.assert_array_equal(out["fast"]["idx"][same], out["exact"]["idx"][same])
    # expected ~1e-10 flips per decision (an ulp-level tie between a cumulative weight and a position): a handful of
    # points at most over 2e9 decisions
    assert n_diff <= 4, n_diff
    # a point that did flip differs by ONE particle of 5 000: its posterior stays within the 1e-5 bar of north_star
    np.testing.assert_allclose(out["fast"]["moments"][..., 0:6], out["exact"]["moments"][..., 0:6], rtol=1e-5, atol=1e-6)
