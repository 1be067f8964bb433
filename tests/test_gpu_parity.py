"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden vectors.

Tolerances (north_star): bit-exact for integer work (boxes, histograms, resample indices given
the same weights); float64 stages to ~1e-12; likelihoods / weights / posteriors within 1e-5
relative (the SSD runs in float32 exactly where the reference casts to float32).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import resample as oresample  # noqa: E402
from oracle import spline as ospline  # noqa: E402
from oracle import ssd as ossd  # noqa: E402
from oracle import tiles as otiles  # noqa: E402

RTOL = 1e-5  # north_star tolerance for likelihoods and posteriors


@pytest.fixture(scope="module")
def lib():
    from glimpse_amd import _lib

    assert _lib.device_count() >= 1
    return _lib


def test_stage_project_matches_reference(lib, golden):
    g = golden("g1_projection.npz")
    for cam, xyz, uv in zip(g["cams"], g["xyz"], g["uv"]):
        got = lib.stage_project(cam, xyz)
        assert np.array_equal(np.isnan(got), np.isnan(uv))
        ok = ~np.isnan(uv[:, 0])
        # float64 projection: differences only from sin/cos and dot-product association
        np.testing.assert_allclose(got[ok], uv[ok], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("name,key", [("gray", "gray"), ("rgb", "rgb"), ("coarse", "coarse")])
def test_stage_tiles_match_reference(lib, golden, name, key):
    g = golden("g2_tiles.npz")
    frames = g[key]
    for b in range(3):
        tbox, sbox = g[f"{name}_{b}_tbox"], g[f"{name}_{b}_sbox"]
        tile, (hv, hq) = lib.stage_template(frames[0], tbox)
        np.testing.assert_array_equal(hq, g[f"{name}_{b}_hist_q"])  # integer counts / size
        np.testing.assert_allclose(hv, g[f"{name}_{b}_hist_v"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(tile, g[f"{name}_{b}_tile"], rtol=1e-12, atol=1e-13)
        # search tile with the reference's histogram: LUT + integer median are exact
        hist = (g[f"{name}_{b}_hist_v"], g[f"{name}_{b}_hist_q"])
        search = lib.stage_search_tile(frames[1], sbox, hist)
        np.testing.assert_array_equal(search, g[f"{name}_{b}_search"].astype(np.float32))


def test_stage_ssd_matches_oracle(lib):
    rng = np.random.default_rng(7)
    for (hs, ws, th, tw) in [(19, 18, 15, 15), (40, 52, 15, 15), (64, 64, 31, 31), (35, 90, 31, 31),
                             (120, 131, 31, 31), (30, 33, 9, 11), (200, 36, 31, 31)]:
        s = rng.standard_normal((hs, ws)).astype(np.float32)
        t = rng.standard_normal((th, tw)).astype(np.float32)
        want = ossd.match_template_sqdiff(s, t)
        want *= 1 / (np.int64(tw) * np.int64(th))  # tracker.py:614
        got = lib.stage_ssd(s, t)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)
        # ... and BIT FOR BIT the restatement with the kernels' own accumulation (float32 fused multiply-adds along a
        # template row, float64 across rows: oracle/ssd.c oracle_ssd_f32_rows) -- OpenCV leaves the precision of the
        # accumulation open; the long-sequence index comparisons (test_gpu_pinned.py) are made against this one
        rows = ossd.match_template_sqdiff(s, t, accumulate="row_f32")
        rows *= 1 / (np.int64(tw) * np.int64(th))
        np.testing.assert_array_equal(got, rows)
        assert np.abs(rows / want - 1).max() < 5e-7


def test_stage_sample_matches_reference(lib, golden):
    g = golden("g4_spline.npz")
    for i in range(8):
        sse, box, uv, val = g[f"s{i}_sse"], g[f"s{i}_box"], g[f"s{i}_uv"], g[f"s{i}_val"]
        got, outside = lib.stage_sample(sse, box, uv)
        assert not outside.any()
        np.testing.assert_allclose(got, val, rtol=0, atol=5e-13)
    _, outside = lib.stage_sample(sse, box, uv + 1000.0)
    assert outside.all()


def test_stage_resample_bit_exact(lib, golden):
    g = golden("g5_resample.npz")
    for i, n in enumerate([1, 7, 100, 129, 1000, 2000, 5000, 10000]):
        w = g[f"r{i}_weights"]
        idx = lib.stage_resample(w, float(g[f"r{i}_u"]))
        np.testing.assert_array_equal(idx, g[f"r{i}_idx"])
    rng = np.random.default_rng(11)
    for n in [64, 255, 256, 257, 3000, 8192, 8193, 12000]:
        for trial in range(3):
            w = np.exp(-rng.random(n) * rng.choice([1.0, 20.0, 200.0])) + 1e-300
            u = rng.random()
            np.testing.assert_array_equal(lib.stage_resample(w, u), oresample.systematic(w, u))


@pytest.mark.parametrize("name", ["g8_c1.npz", "g8_c2mini.npz", "g8_c5mini.npz"])
def test_end_to_end_free_running(lib, golden, name):
    """Whole sequences through the C ABI with the reference's own random draws."""
    from tests.helpers_gpu import context_for, run_free

    g = golden(name)
    ctx = context_for(g)
    records, moments = run_free(g, ctx)
    P, N = ctx.P, ctx.N
    errors = g["errors"].astype(bool)
    starts = list(g["track_starts"]) + [int(g["n_steps"])]
    tw, th = ctx.tile
    # templates
    for p in range(P):
        for o in range(int(g["n_obs"])):
            if f"t{p}_o{o}_box" not in g:
                continue
            t = ctx.get_template(o, p)
            np.testing.assert_array_equal(t["box"], g[f"t{p}_o{o}_box"])
            np.testing.assert_allclose(t["duv"], g[f"t{p}_o{o}_duv"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(t["tile"], g[f"t{p}_o{o}_tile"], rtol=1e-11, atol=1e-12)
            np.testing.assert_array_equal(t["histogram"][1], g[f"t{p}_o{o}_hist_q"])
            np.testing.assert_allclose(t["histogram"][0], g[f"t{p}_o{o}_hist_v"], rtol=1e-11, atol=1e-12)
    # per-step intermediates
    step_recs = [r for r in records if "weights" in r]
    n_idx = n_bad = 0
    for p in range(P):
        if errors[p]:
            continue
        for k, rec in enumerate(step_recs):
            s = starts[p] + k
            np.testing.assert_allclose(rec["evolved"][p], g[f"s{s}_evolved"], rtol=RTOL, atol=1e-9)
            for o in range(int(g["n_obs"])):
                if f"s{s}_o{o}_sse" not in g:
                    assert rec["obs_status"][o, p] != lib.OBS_OK
                    continue
                assert rec["obs_status"][o, p] == lib.OBS_OK
                d = rec["dbg"][o][p]
                np.testing.assert_allclose(d["uv"], g[f"s{s}_o{o}_uv"], rtol=0, atol=1e-7)
                sb = g[f"s{s}_o{o}_sse_box"]
                ho, wo = g[f"s{s}_o{o}_sse"].shape
                assert d["sse"].shape == (ho, wo)
                np.testing.assert_allclose(d["search"], g[f"s{s}_o{o}_search_f32"], rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(d["sse"], g[f"s{s}_o{o}_sse"], rtol=RTOL, atol=1e-7)
                assert abs((d["box"][0] + tw / 2 - 0.5) - (sb[0] - g[f"t{p}_o{o}_duv"][0])) < 1e-6
            np.testing.assert_allclose(rec["weights"][p], g[f"s{s}_weights"], rtol=RTOL, atol=1e-290)
            # the resample step itself is bit-exact given its inputs ...
            want_idx = oresample.systematic(rec["weights"][p], float(g["random"][s]))
            np.testing.assert_array_equal(rec["idx"][p], want_idx)
            # ... and matches the reference's indices on these sequences
            n_idx += N
            n_bad += int((rec["idx"][p] != g[f"s{s}_idx"]).sum())
    assert n_bad == 0, f"{n_bad} of {n_idx} resample indices differ from the reference"
    # posterior means / sigmas (Tracks.means, Tracks.sigmas)
    means = np.transpose(moments[:, :, 0:6], (1, 0, 2))
    sigmas = np.transpose(moments[:, :, 6:12], (1, 0, 2))
    ok = ~errors
    np.testing.assert_allclose(means[ok], g["means"][ok], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(sigmas[ok], g["out_sigmas"][ok], rtol=RTOL, atol=1e-8)
    # failed tracks: flagged at the frame where the reference raised
    st = ctx.point_status()
    ef = ctx.point_error_frame()
    for p in range(P):
        if errors[p]:
            assert st[p] & lib.PT_TEMPLATE_OOB
            assert ef[p] == 0
        else:
            assert st[p] == 0 and ef[p] == lib.NO_ERROR_FRAME
    ctx.close()


@pytest.mark.parametrize("n", [7, 100, 129, 1000, 5000, 10000])
def test_residual_resampling_follows_the_reference_literally(lib, golden, n):
    """GLH_RESAMPLE_RESIDUAL (tracker.py:188-203 as written: repetition counts subtracted from the normalised
    weights, np.searchsorted's stateful bisection over a cumulative sum that is not monotone) against the oracle
    (which is pinned to the reference's own outputs, tests/test_oracle_golden.py) on several points at once:
    indices and the number of uniforms consumed, bit for bit."""
    rng = np.random.default_rng(n)
    P = 5
    particles = rng.standard_normal((P, n, 6))
    ll = rng.random((P, n)) * rng.choice([1, 5, 40], (P, n))
    weights = np.exp(-ll) + 1e-300
    weights[1] = 1.0  # uniform weights: every repetition count is 1 or 0 at the rounding boundary
    if n >= 100:
        weights[2, : n // 2] = 1e-300  # half of the particles at the floor
    u = rng.random((P, n))
    with lib.Context(P, n, 1, max_frames=2) as ctx:
        ctx.begin_sequence(P, n, (15, 15))
        ctx.set_debug(2)
        ctx.set_particles(particles)
        ctx.set_weights(weights)
        ctx.set_frame(0)
        ctx.resample(u=u, method="residual")
        idx = ctx.resample_indices()
        draws = ctx.residual_draws()
        out_p, out_w = ctx.get_particles(), ctx.get_weights()
    for p in range(P):
        want = oresample.residual(weights[p], u[p])
        np.testing.assert_array_equal(idx[p], want)
        wn = weights[p] / weights[p].sum()
        assert draws[p] == n - (n * wn).astype(int).sum()
        np.testing.assert_array_equal(out_p[p], particles[p][want])
        np.testing.assert_array_equal(out_w[p], weights[p][want])


def test_library_reports_errors(lib):
    with pytest.raises(lib.GlhError):
        lib.Context(0, 10)
    ctx = lib.Context(2, 64, 1, max_search_dim=64, max_frames=4)
    with pytest.raises(lib.GlhError):
        ctx.begin_sequence(3, 64, (15, 15))  # more points than capacity
    with pytest.raises(lib.GlhError):
        ctx.begin_sequence(2, 64, (15, 33))  # tile larger than max_tile
    ctx.close()


def test_highpass_window_sizes_match_reference(lib, golden):
    """Tracker(highpass={"size": ...}) (tracker.py:59, :530): median windows (3, 3), (7, 7), (3, 5) [rows, columns] and
    the int form, template tiles and search tiles of gray and RGB frames against the reference's extract_tile."""
    g = golden("g17_highpass.npz")
    frames = golden("g2_tiles.npz")
    tbox, sbox = g["tbox"], g["sbox"]
    for k, size in enumerate(g["sizes"]):
        size = tuple(int(v) for v in size)
        for name in ("gray", "rgb"):
            f = frames[name]
            tile, (hv, hq) = lib.stage_template(f[0], tbox, highpass=size)
            np.testing.assert_array_equal(hq, g[f"{name}_{k}_hist_q"])
            np.testing.assert_allclose(hv, g[f"{name}_{k}_hist_v"], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(tile, g[f"{name}_{k}_tile"], rtol=1e-12, atol=1e-13)
            hist = (g[f"{name}_{k}_hist_v"], g[f"{name}_{k}_hist_q"])
            search = lib.stage_search_tile(f[1], sbox, hist, highpass=size)
            np.testing.assert_array_equal(search, g[f"{name}_{k}_search"].astype(np.float32))
    with pytest.raises(lib.GlhError):
        lib.stage_template(frames["gray"][0], tbox, highpass=(4, 4))
    with pytest.raises(lib.GlhError):
        lib.stage_template(frames["gray"][0], tbox, highpass=(9, 9))


def test_highpass_boundary_modes_match_reference(lib, golden):
    """Tracker(highpass={"size": ..., "mode": ...}) (tracker.py:530 hands the dictionary to scipy.ndimage.median_filter):
    'nearest', 'mirror' and 'wrap' with the 5 x 5 default and a (3, 7) window, template tiles and search tiles of gray and
    RGB frames against the reference's extract_tile (g26)."""
    g = golden("g26_highpass_modes.npz")
    frames = golden("g2_tiles.npz")
    tbox, sbox = g["tbox"], g["sbox"]
    for k, (size, mode) in enumerate(zip(g["sizes"], g["modes"])):
        size, mode = tuple(int(v) for v in size), str(mode)
        for name in ("gray", "rgb"):
            f = frames[name]
            tile, (hv, hq) = lib.stage_template(f[0], tbox, highpass=size, mode=mode)
            np.testing.assert_array_equal(hq, g[f"{name}_{k}_hist_q"])
            np.testing.assert_allclose(hv, g[f"{name}_{k}_hist_v"], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(tile, g[f"{name}_{k}_tile"], rtol=1e-12, atol=1e-13)
            hist = (g[f"{name}_{k}_hist_v"], g[f"{name}_{k}_hist_q"])
            search = lib.stage_search_tile(f[1], sbox, hist, highpass=size, mode=mode)
            np.testing.assert_array_equal(search, g[f"{name}_{k}_search"].astype(np.float32))


def test_depth_limits_are_refused_up_front(lib):
    """glh_observer_set_depth refuses the combinations its tile kernels cannot serve (their LDS requests grow with the
    context's limits) instead of failing every launch later: 16-bit frames beyond a 1117-pixel search workspace, float64
    frames with templates beyond 73 pixels."""
    with lib.Context(2, 64, 1, max_tile=31, max_search_dim=1500, max_frames=2) as ctx:
        ctx.observer_init(0, 2, 64, 64, 1, 0.3)
        with pytest.raises(lib.GlhError, match="16-bit frames"):
            ctx.observer_set_depth(0, np.uint16)
        ctx.observer_set_depth(0, np.float64)  # (a 31-pixel template is fine)
    with lib.Context(2, 64, 1, max_tile=90, max_search_dim=300, max_frames=2) as ctx:
        ctx.observer_init(0, 2, 64, 64, 1, 0.3)
        with pytest.raises(lib.GlhError, match="float frames"):
            ctx.observer_set_depth(0, np.float64)
        ctx.observer_set_depth(0, np.uint16)


def test_stage_sample_orders_match_reference(lib, golden):
    """Observer.sample_tile for every order of RectBivariateSpline (1 .. 5, mixed; g23): the general kernels (banded
    fit of bandwidth k + evaluation with FITPACK's knots) against the reference's values."""
    g = golden("g23_orders.npz")
    for c in range(int(g["n_cases"])):
        kx, ky = (int(v) for v in g[f"c{c}_k"])
        got, outside = lib.stage_sample(g[f"c{c}_sse"], g[f"c{c}_box"], g[f"c{c}_uv"], orders=(kx, ky))
        assert not outside.any()
        np.testing.assert_allclose(got, g[f"c{c}_val"], rtol=0, atol=5e-12)
    with pytest.raises(lib.GlhError):
        lib.stage_sample(g["c0_sse"], g["c0_box"], g["c0_uv"], orders=(6, 3))
    with pytest.raises(lib.GlhError):
        lib.stage_sample(np.zeros((2, 9), np.float32), g["c0_box"], g["c0_uv"], orders=(2, 2))  # fewer than kx + 1 rows
