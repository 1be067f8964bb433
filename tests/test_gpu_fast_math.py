"""GLH_MATH_FAST (fused multiply-adds, Newton reciprocals, table exp, unnormalised systematic resampling): the
arithmetic of device-RNG runs.  It has to (a) leave the fused and the staged kernels bit-identical to each other,
like the exact arithmetic does, and (b) stay within rounding distance of the exact arithmetic on the same Philox
stream -- the exact arithmetic being what the oracle / reference goldens pin in the host-RNG tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 11


def _run(name, P, N, T, math, fused, imgsz=(640, 640)):
    from glimpse_amd import _lib as lib
    from glimpse_amd import workloads

    wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=imgsz)
    frames = [wl.frames(o) for o in range(wl.O)]
    with lib.Context(P, N, wl.O, max_tile=31, max_search_dim=200, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_math(math)
        ctx.set_fused(fused)
        ctx.set_debug(2)
        ctx.set_frame(0)
        ctx.init_particles(seed=SEED)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        idx = []
        for i in range(1, T):
            ctx.step(i, 1.0, [i] * wl.O, seed=SEED)
            idx.append(ctx.resample_indices())
        assert (ctx.point_status() == 0).all() and (ctx.observer_status() == lib.OBS_OK).all()
        return dict(moments=ctx.get_moments(0, T), particles=ctx.get_particles(), weights=ctx.get_weights(),
                    idx=np.stack(idx))


@pytest.mark.parametrize("name,P,N", [("C2", 12, 2000), ("C3", 6, 5000), ("C5", 6, 3000), ("C4", 3, 10000),
                                      # particle counts that leave partial segments / odd segment lengths / no whole
                                      # 16-byte words: the guarded forms of the vectorised loops
                                      ("C2", 5, 777), ("C3", 4, 2049), ("C3", 3, 4999), ("C5", 3, 1501), ("C3", 2, 6007)])
def test_fast_fused_equals_fast_staged_bit_for_bit(name, P, N):
    T = 4
    fused = _run(name, P, N, T, "fast", 1)
    staged = _run(name, P, N, T, "fast", 0)
    np.testing.assert_array_equal(fused["idx"], staged["idx"])
    np.testing.assert_array_equal(fused["particles"], staged["particles"])
    np.testing.assert_array_equal(fused["weights"], staged["weights"])
    np.testing.assert_allclose(fused["moments"], staged["moments"], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("name,P,N", [("C2", 12, 2000), ("C3", 6, 5000), ("C5", 6, 3000)])
def test_fast_arithmetic_tracks_the_exact_arithmetic(name, P, N):
    """Same Philox stream, both arithmetics: the resample indices agree (an index flips only when a cumulative weight
    sits within ~1e-15 of a systematic position) and the posteriors agree to rounding."""
    T = 4
    fast = _run(name, P, N, T, "fast", 1)
    exact = _run(name, P, N, T, "exact", 1)
    flips = (fast["idx"] != exact["idx"]).mean()
    assert flips < 1e-6, flips
    np.testing.assert_allclose(fast["moments"], exact["moments"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(fast["particles"], exact["particles"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(fast["weights"], exact["weights"], rtol=1e-9, atol=1e-290)


@pytest.mark.parametrize("name,P,N", [("C3", 6, 5000), ("C5", 4, 5000)])
def test_fast_fused_equals_fast_staged_over_a_long_sequence(name, P, N):
    """40 frames from the wide prior into the steady state (search tiles shrink from ~100 px to ~40 px: the coefficient
    form, then the per-cell form of the sampling; the common instantiation from the second frame on): the fused and
    the staged kernels stay bit-identical all the way -- one flipped resample index would show in every later frame."""
    T = 40
    fused = _run(name, P, N, T, "fast", 1, imgsz=(1024, 1024))
    staged = _run(name, P, N, T, "fast", 0, imgsz=(1024, 1024))
    np.testing.assert_array_equal(fused["idx"], staged["idx"])
    np.testing.assert_array_equal(fused["particles"], staged["particles"])
    np.testing.assert_array_equal(fused["weights"], staged["weights"])
    np.testing.assert_allclose(fused["moments"], staged["moments"], rtol=1e-11, atol=1e-12)
    # the filter has converged on the scene's motion
    assert abs(np.median(fused["moments"][-1, :, 3]) - 0.15) < 0.02
