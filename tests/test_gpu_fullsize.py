"""BASELINE.json's configurations at FULL size (C3: 4 096 points x 5 000 particles; one GPU's shard of C4: 1 250 points
x 10 000 particles; 2048^2 frames, 31x31 templates) through properties that do not need the oracle to run at that size:

* the staged kernels, run on a 48-point slice of the same points with the same global RNG keys, reproduce the fused
  kernel's rows of the big run bit for bit (particles, weights, resample indices) -- the slice is small enough for
  the oracle-pinned staged path, the big run is what the bench times;
* systematic resampling returns every point's sources in nondecreasing order, each source index is in range, and the
  run-length compact state expands to exactly particles[idx] / weights[idx] of those indices;
* two half-size shards reproduce the unsharded posterior history (checksum of checksums);
* every point is tracked by the observer, nothing is flagged, and the filter follows the synthetic motion.

All of BASELINE.json's GPU configurations are covered: C2 (256 x 2 000, 15x15), C3 (4 096 x 5 000), one shard of C4
(1 250 x 10 000) and one shard of C5 (512 x 5 000, two observers, DEM term), in the arithmetic the bench times."""
import functools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T = 4
SEED = 77


def _run(lib, wl, frames, p0, p1, mode, keep_idx=False, math="fast"):
    from glimpse_amd import workloads

    sub = wl.slice(p0, p1)
    with lib.Context(sub.P, wl.N, wl.O, max_tile=31, max_search_dim=160, max_frames=T) as ctx:
        workloads.setup_context(ctx, sub, frames)
        ctx.set_point_offset(p0)
        ctx.set_fused(mode)
        ctx.set_math(math)
        if keep_idx:
            ctx.set_debug(2)
        ctx.set_frame(0)
        ctx.init_particles(seed=SEED)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        idx = None
        for i in range(1, T):
            ctx.step(i, 1.0, [i] * wl.O, seed=SEED)
        if keep_idx:
            idx = ctx.resample_indices()
        out = dict(moments=ctx.get_moments(0, T), particles=ctx.get_particles(), weights=ctx.get_weights(),
                   status=ctx.point_status(), obs=ctx.observer_status(), idx=idx)
    return out


@pytest.mark.parametrize("name,shape,math", [("C3", (4096, 5000), "fast"), ("C3", (4096, 5000), "exact"),
                                             ("C4", (1250, 10000), "fast"), ("C2", (256, 2000), "fast"),
                                             ("C5", (512, 5000), "fast"), ("C5", (512, 5000), "exact")])
def test_full_size_properties(name, shape, math):
    """C3 whole, C2 whole; C4 as one of its 8 shards (1 250 points x 10 000 particles: the 1 024-thread variant); C5 as
    one of its 4 shards (two observers, DEM likelihood term: the two-observer instantiation).  `fast` is the arithmetic
    bench.py times, `exact` the one the oracle pins in the host-RNG tests."""
    from glimpse_amd import _lib as lib
    from glimpse_amd import workloads

    wl = workloads.Workload(name, n_frames=T)
    assert (wl.P, wl.N) == shape and wl.imgsz == (2048, 2048)
    assert wl.tile == ((15, 15) if name == "C2" else (31, 31)) and wl.O == (2 if name == "C5" else 1)
    frames = [wl.frames(o) for o in range(wl.O)]
    run = functools.partial(_run, math=math)
    big = run(lib, wl, frames, 0, wl.P, 1, keep_idx=True)
    assert (big["status"] == 0).all() and (big["obs"] == lib.OBS_OK).all()
    assert np.isfinite(big["moments"]).all()
    # resample indices of the last step: sorted, in range
    idx = big["idx"]
    assert idx.min() >= 0 and idx.max() < wl.N
    assert (np.diff(idx, axis=1) >= 0).all()
    # the filter follows the synthetic motion (0.15 units/frame along x)
    vx = big["moments"][-1, :, 3]
    assert abs(np.median(vx) - 0.15) < 0.03
    # a slice of the points on the staged kernels, same global RNG keys: bit for bit the fused rows
    p0, p1 = wl.P // 4, wl.P // 4 + 48
    small = run(lib, wl, frames, p0, p1, 0, keep_idx=True)
    np.testing.assert_array_equal(small["idx"], idx[p0:p1])
    np.testing.assert_array_equal(small["particles"], big["particles"][p0:p1])
    np.testing.assert_array_equal(small["weights"], big["weights"][p0:p1])
    np.testing.assert_allclose(small["moments"], big["moments"][:, p0:p1], rtol=1e-11, atol=1e-12)
    # the compact state expanded == gather by the indices: copies of a source are identical records
    same = idx[:, 1:] == idx[:, :-1]
    assert (big["particles"][:, 1:][same] == big["particles"][:, :-1][same]).all()
    assert (big["weights"][:, 1:][same] == big["weights"][:, :-1][same]).all()
    # two shards == the unsharded run
    half = wl.P // 2
    lo = run(lib, wl, frames, 0, half, 1)
    hi = run(lib, wl, frames, half, wl.P, 1)
    np.testing.assert_array_equal(np.concatenate((lo["moments"], hi["moments"]), axis=1), big["moments"])
    np.testing.assert_array_equal(np.concatenate((lo["particles"], hi["particles"])), big["particles"])


def test_full_c4_on_one_gpu():
    """BASELINE config 4 WHOLE on one GPU -- 10 000 points x 10 000 particles (2 x 4.8 GB of state, 10^8 particles per
    frame): the N = 1 anchor of the strong-scaling curve, and every size limit at the shape north_star names (points per
    context, size_t indexing of the state, the history, the record-index tables).  Properties only, and host downloads
    kept small: the resample indices (0.4 GB), the posterior history, a few points' particles."""
    from glimpse_amd import _lib as lib
    from glimpse_amd import workloads

    T3 = 3
    wl = workloads.Workload("C4", n_frames=T3, n_points=10000)
    assert (wl.P, wl.N) == (10000, 10000) and wl.tile == (31, 31)
    frames = [wl.frames(0)]
    picks = (0, 1, 4999, 5000, 9998, 9999)

    def run(p0, p1, mode, want_idx=False, points=()):
        sub = wl.slice(p0, p1)
        with lib.Context(sub.P, wl.N, 1, max_tile=31, max_search_dim=160, max_frames=T3) as ctx:
            workloads.setup_context(ctx, sub, frames)
            ctx.set_point_offset(p0)
            ctx.set_fused(mode)
            ctx.set_math("fast")
            if want_idx:
                ctx.set_debug(2)
            ctx.set_frame(0)
            ctx.init_particles(seed=SEED)
            ctx.init_templates(0, 0)
            ctx.record_moments(0)
            if mode == 1 and not want_idx:
                ctx.track([1, 2], [1.0, 1.0], [[1], [2]], seed=SEED)  # (the frame loop in one call: two streams)
            else:
                for i in range(1, T3):
                    ctx.step(i, 1.0, [i], seed=SEED)
            return dict(moments=ctx.get_moments(0, T3), status=ctx.point_status(), obs=ctx.observer_status(),
                        idx=ctx.resample_indices() if want_idx else None,
                        state={p: ctx.get_point_state(p - p0) for p in points})

    big = run(0, wl.P, 1, want_idx=True, points=picks + tuple(range(2500, 2548)))
    assert (big["status"] == 0).all() and (big["obs"] == lib.OBS_OK).all()
    assert np.isfinite(big["moments"]).all()
    idx = big["idx"]
    assert idx.shape == (10000, 10000) and idx.min() >= 0 and idx.max() < wl.N
    assert (np.diff(idx, axis=1) >= 0).all()
    assert abs(np.median(big["moments"][-1, :, 3]) - 0.15) < 0.03
    # 48 points on the staged kernels, same global RNG keys: bit for bit the rows of the big run
    p0, p1 = 2500, 2548
    small = run(p0, p1, 0, want_idx=True, points=tuple(range(p0, p1)))
    np.testing.assert_array_equal(small["idx"], idx[p0:p1])
    for p in range(p0, p1):
        np.testing.assert_array_equal(small["state"][p][0], big["state"][p][0])
        np.testing.assert_array_equal(small["state"][p][1], big["state"][p][1])
    np.testing.assert_allclose(small["moments"], big["moments"][:, p0:p1], rtol=1e-11, atol=1e-12)
    del idx, small
    # two shards of 5 000 points, each through glh_track (two streams each): the unsharded history and the picked states
    lo = run(0, 5000, 1, points=picks[:3])
    hi = run(5000, 10000, 1, points=picks[3:])
    np.testing.assert_array_equal(np.concatenate((lo["moments"], hi["moments"]), axis=1), big["moments"])
    for p in picks:
        part = lo if p < 5000 else hi
        np.testing.assert_array_equal(part["state"][p][0], big["state"][p][0])
        np.testing.assert_array_equal(part["state"][p][1], big["state"][p][1])
