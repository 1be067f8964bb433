"""Pin the oracle (CPU restatement) against outputs of the reference itself.

The golden files were written by tools/make_golden.py, which imports
/root/reference under stubs.  These tests never touch the reference.
"""
import warnings

import numpy as np
import pytest

from oracle import camera, motion, resample, spline, ssd, tiles, tracker
from tests.helpers_golden import draws_from, models_from, observers_from, taus_from


def test_projection_matches_reference(golden):
    g = golden("g1_projection.npz")
    for cam, xyz, uv, R in zip(g["cams"], g["xyz"], g["uv"], g["R"]):
        assert np.array_equal(camera.rotation_matrix(cam[3:6]), R)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = camera.xyz_to_uv(cam, xyz)
        assert np.array_equal(np.isnan(got), np.isnan(uv))
        assert np.isnan(uv[250:]).all() and not np.isnan(uv[:250]).any()
        np.testing.assert_array_equal(got[:250], uv[:250])


def test_projection_reference_doctests():
    # camera.py:615-620: default camera projects (0, 10, 0) onto the image centre
    cam = camera.make_camera(imgsz=10, f=10)
    np.testing.assert_array_equal(camera.xyz_to_uv(cam, np.array([(0.0, 10.0, 0.0)])), [[5.0, 5.0]])
    # camera.py:683-694: behind the camera -> NaN; (1000, 10, 0) -> (1005, 5)
    xyz = np.array([(1000.0, 10, 0), (0, 10, 0), (0, 0, 0), (0, -10, 0)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        uv = camera.xyz_to_uv(cam, xyz)
    np.testing.assert_array_equal(uv[:2], [[1005.0, 5.0], [5.0, 5.0]])
    assert np.isnan(uv[2:]).all()
    # camera.py:712-715 inframe doctest
    cam = camera.make_camera(imgsz=(10, 12), f=10)
    uv = np.array([(-1, 1), (0, 0), (9, 11), (10, 15)], dtype=float)
    assert list(camera.inframe(cam, uv)) == [False, True, True, False]


def test_helpers_reference_doctests():
    # helpers.py:451-456, :482-487
    v, q = tiles.compute_cdf(np.array([3, 2, 1, 2]))
    assert list(v) == [1, 2, 3] and list(q) == [0.25, 0.75, 1.0]
    a = np.array([3, 2, 1, 2])
    b = np.array([4, 2, 1, 2, 4, 2, 1, 2])
    assert list(tiles.match_cdf(a, tiles.compute_cdf(b))) == [4.0, 2.0, 1.0, 2.0]
    # helpers.py:335-342
    x = tiles.normalize(np.array([0, 1, 2, 3]))
    assert x.mean() == 0.0 and x.std() == 1.0
    # helpers.py:827-829
    pts = np.array([(0, 0), (1, 1), (2, 2), (3, 3)])
    assert list(spline.in_box(pts, [1, 1, 2.5, 2.5])) == [False, True, True, False]


@pytest.mark.parametrize("name,frames_key,channels", [("gray", "gray", 1), ("rgb", "rgb", 3), ("coarse", "coarse", 1)])
def test_tiles_match_reference(golden, name, frames_key, channels):
    g = golden("g2_tiles.npz")
    frames = g[frames_key]
    for b in range(3):
        tbox, sbox = g[f"{name}_{b}_tbox"], g[f"{name}_{b}_sbox"]
        tile, hist = tiles.extract_tile(frames[0], tbox, return_histogram=True)
        np.testing.assert_array_equal(tile, g[f"{name}_{b}_tile"])
        np.testing.assert_array_equal(hist[0], g[f"{name}_{b}_hist_v"])
        np.testing.assert_array_equal(hist[1], g[f"{name}_{b}_hist_q"])
        search = tiles.extract_tile(frames[1], sbox, histogram=hist)
        np.testing.assert_array_equal(search, g[f"{name}_{b}_search"])
        # integer-key / LUT formulation implemented by the HIP kernels: bit-exact too
        key = tiles.gray_key(tiles.read_box(frames[0], tbox))
        t2, h2, _, _ = tiles.template_from_key(key, channels)
        np.testing.assert_array_equal(t2, g[f"{name}_{b}_tile"])
        np.testing.assert_array_equal(h2[0], g[f"{name}_{b}_hist_v"])
        np.testing.assert_array_equal(h2[1], g[f"{name}_{b}_hist_q"])
        skey = tiles.gray_key(tiles.read_box(frames[1], sbox))
        s2 = tiles.search_from_key(skey, channels, hist)
        np.testing.assert_array_equal(s2, g[f"{name}_{b}_search"])


def test_spline_matches_reference(golden):
    g = golden("g4_spline.npz")
    for i in range(8):
        sse, box, uv, val = g[f"s{i}_sse"], g[f"s{i}_box"], g[f"s{i}_uv"], g[f"s{i}_val"]
        got = spline.sample_tile(uv, sse, box)
        np.testing.assert_array_equal(got, val)
        # closed-form not-a-knot restatement (what the kernels do)
        cu, cv = spline.cell_centres(box, sse.shape)
        coef = spline.fit_notaknot(sse.astype(float))
        got2 = spline.eval_notaknot(coef, cv[0], cu[0], uv[:, 1], uv[:, 0])
        np.testing.assert_allclose(got2, val, rtol=0, atol=2e-13)
    with pytest.raises(ValueError):
        spline.sample_tile(uv + 100.0, sse, box)


def test_resample_matches_reference(golden):
    g = golden("g5_resample.npz")
    for i, n in enumerate([1, 7, 100, 129, 1000, 2000, 5000, 10000]):
        w = g[f"r{i}_weights"]
        assert len(w) == n
        idx = resample.systematic(w, float(g[f"r{i}_u"]))
        np.testing.assert_array_equal(idx, g[f"r{i}_idx"])
        assert resample.numpy_pairwise_sum(w) == w.sum()
        if n <= 1000:
            p = g[f"r{i}_particles"]
            np.testing.assert_array_equal(p[idx], g[f"r{i}_out_particles"])
            np.testing.assert_array_equal(w[idx], g[f"r{i}_out_weights"])
            mean = resample.particle_mean(p[idx], w[idx])
            np.testing.assert_array_equal(mean, g[f"r{i}_mean"])
            np.testing.assert_array_equal(resample.particle_sigma(p[idx], w[idx], mean), g[f"r{i}_sigma"])
            if n > 1:
                np.testing.assert_array_equal(resample.particle_covariance(p[idx], w[idx]), g[f"r{i}_cov"])
            sidx = resample.stratified(w, g[f"r{i}_strat_u"])
            np.testing.assert_array_equal(sidx, g[f"r{i}_strat_idx"])
            if n > 1:
                np.random.seed(int(g[f"r{i}_resid_seed"]))
                ridx = resample.residual(w, np.random.random)
                np.testing.assert_array_equal(p[ridx], g[f"r{i}_resid_out_particles"])


def test_motion_matches_reference(golden):
    g = golden("g7_motion.npz")
    for i in range(2):
        p = g[f"m{i}_params"]
        model = motion.CartesianMotion(
            xy=p[0:2], xy_sigma=p[2:4], vxyz=p[4:7], vxyz_sigma=p[7:10], axyz=p[10:13],
            axyz_sigma=p[13:16], dem=p[16], dem_sigma=p[17], n=len(g[f"m{i}_p0"]))
        p0, _ = model.initialize_particles(g[f"m{i}_init_normals"])
        np.testing.assert_array_equal(p0, g[f"m{i}_p0"])
        p1 = p0.copy()
        model.evolve_particles(p1, g[f"m{i}_taus"][0], g[f"m{i}_evolve_normals"][0])
        np.testing.assert_array_equal(p1, g[f"m{i}_p1"])
        p2 = p1.copy()
        model.evolve_particles(p2, g[f"m{i}_taus"][1], g[f"m{i}_evolve_normals"][1])
        np.testing.assert_array_equal(p2, g[f"m{i}_p2"])
        np.testing.assert_array_equal(model.compute_log_likelihoods(p2), g[f"m{i}_ll"])


@pytest.mark.parametrize("name", ["g8_c1.npz", "g8_c2mini.npz", "g8_c5mini.npz"])
def test_end_to_end_matches_reference(golden, name):
    """Replaying the recorded RNG draws, the oracle reproduces the reference bit for bit."""
    g = golden(name)
    observers = observers_from(g)
    models = models_from(g)
    draws = draws_from(g)
    traces = [[] for _ in models]
    res = tracker.track(
        models, observers, g["matching"], taus_from(g), tile_size=tuple(g["tile_size"]),
        draws=draws, return_particles=True,
    )
    np.testing.assert_array_equal(res["means"], g["means"])
    np.testing.assert_array_equal(res["sigmas"], g["out_sigmas"])
    assert [e is not None for e in res["errors"]] == [bool(e) for e in g["errors"]]
    for t, r in enumerate(res["results"]):
        if r["particles"] is not None and not g["errors"][t]:
            np.testing.assert_array_equal(r["particles"], g["out_particles"][t])
            np.testing.assert_array_equal(r["weights"], g["out_weights"][t])


@pytest.mark.parametrize("name", ["g8_c1.npz", "g8_c5mini.npz"])
def test_end_to_end_stage_traces(golden, name):
    """Per-step intermediates (uv, search tile, SSE, sampled ll, weights, indices)."""
    g = golden(name)
    observers = observers_from(g)
    models = models_from(g)
    draws = draws_from(g)
    starts = list(g["track_starts"]) + [int(g["n_steps"])]
    for t, model in enumerate(models):
        trace = []
        tracker.track_one(model, observers, g["matching"], taus_from(g), tile_size=tuple(g["tile_size"]),
                          draws=draws[t], trace=trace)
        steps = [tr for tr in trace if "weights" in tr]
        assert len(steps) == starts[t + 1] - starts[t]
        for k, tr in enumerate(steps):
            s = starts[t] + k
            np.testing.assert_array_equal(tr["evolved"], g[f"s{s}_evolved"])
            np.testing.assert_array_equal(tr["weights"], g[f"s{s}_weights"])
            np.testing.assert_array_equal(tr["idx"], g[f"s{s}_idx"])
            for o, ot in enumerate(tr["obs"]):
                if f"s{s}_o{o}_sse" not in g:
                    continue
                np.testing.assert_array_equal(ot["uv"], g[f"s{s}_o{o}_uv"])
                np.testing.assert_array_equal(ot["search_tile"].astype(np.float32), g[f"s{s}_o{o}_search_f32"])
                np.testing.assert_array_equal(ot["sse"], g[f"s{s}_o{o}_sse"])
                np.testing.assert_array_equal(ot["sse_box"], g[f"s{s}_o{o}_sse_box"])
                np.testing.assert_array_equal(ot["sampled"], g[f"s{s}_o{o}_sampled"])


def test_ssd_c_matches_numpy():
    rng = np.random.default_rng(3)
    for (hs, ws, th, tw) in [(20, 23, 15, 15), (40, 37, 31, 31), (16, 16, 15, 15), (50, 64, 5, 7)]:
        s = rng.standard_normal((hs, ws)).astype(np.float32)
        t = rng.standard_normal((th, tw)).astype(np.float32)
        np.testing.assert_array_equal(ssd.match_template_sqdiff(s, t), ssd.match_template_sqdiff_numpy(s, t))


@pytest.mark.parametrize("name,kw", [("g9_cov.npz", dict(return_covariances=True)),
                                     ("g9_stratified.npz", dict(resample_method="stratified")),
                                     ("g9_choice.npz", dict(resample_method="choice")),
                                     ("g22_residual.npz", dict(resample_method="residual"))])
def test_api_variants_match_reference(golden, name, kw):
    """Covariance output and the stratified / choice / residual resampling methods, end to end with the oracle
    consuming the legacy global stream from the reference's seed (residual: a weight-dependent number of uniforms per
    step, tracker.py:199-201)."""

    g = golden(name)
    np.random.seed(int(g["seed"]))
    res = tracker.track(models_from(g), observers_from(g), g["matching"], taus_from(g),
                         tile_size=tuple(int(v) for v in g["tile_size"]), **kw)
    np.testing.assert_allclose(res["means"], g["means"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(res["sigmas"], g["out_sigmas"], rtol=1e-9, atol=1e-14)


def _g11_models(g, module, day):
    """The three g11 motion models built from `module` (oracle.motion or glimpse_amd)."""
    def kw(name):
        return {k[len(name) + 4:]: g[k] for k in g if k.startswith(name + "_kw_")}

    c, t, y = kw("cyl"), kw("tcart"), kw("tcyl")
    if module is motion:
        return {
            "cyl": motion.CylindricalMotion(xy=c["xy"], dem=c["dem"], dem_sigma=c["dem_sigma"], n=int(c["n"]),
                                            xy_sigma=c["xy_sigma"], vxyz=c["vrthz"], vxyz_sigma=c["vrthz_sigma"],
                                            axyz=c["arthz"], axyz_sigma=c["arthz_sigma"]),
            "tcart": motion.TangentCartesianMotion(xy=t["xy"], dem=t["dem"], dem_sigma=t["dem_sigma"], n=int(t["n"]),
                                                   xy_sigma=t["xy_sigma"], vxy=t["vxy"], vxy_sigma=t["vxy_sigma"],
                                                   axy=t["axy"], axy_sigma=t["axy_sigma"], slope_sigma=t["slope_sigma"]),
            "tcyl": motion.TangentCylindricalMotion(xy=y["xy"], dem=y["dem"], dem_sigma=y["dem_sigma"], n=int(y["n"]),
                                                    xy_sigma=y["xy_sigma"], vxy=y["vrth"], vxy_sigma=y["vrth_sigma"],
                                                    axy=y["arth"], axy_sigma=y["arth_sigma"],
                                                    slope_sigma=y["slope_sigma"]),
        }
    raise ValueError(module)


def test_other_motion_models_match_reference(golden):
    """Cylindrical / TangentCartesian / TangentCylindrical (motion.py:207-522): init + two evolves with the
    reference's recorded draws."""
    g = golden("g11_motion.npz")
    models = _g11_models(g, motion, 1.0)
    for name, model in models.items():
        draws = [g[f"{name}_draw{i}"] for i in range(int(g[f"{name}_n_draws"]))]
        n = model.n
        if name == "cyl":
            init = np.column_stack((draws[0], draws[1], draws[2]))
            ev = [draws[3], draws[4]]
        else:
            init = np.column_stack((draws[0], draws[1], draws[2], np.zeros(n)))
            ev = [np.column_stack((draws[3], draws[4])), np.column_stack((draws[5], draws[6]))]
        p0, _ = model.initialize_particles(init)
        np.testing.assert_allclose(p0, g[f"{name}_p0"], rtol=1e-14, atol=1e-15)
        p1 = p0.copy()
        model.evolve_particles(p1, 1.5, ev[0])
        np.testing.assert_allclose(p1, g[f"{name}_p1"], rtol=1e-13, atol=1e-14)
        p2 = p1.copy()
        model.evolve_particles(p2, -0.75, ev[1])
        np.testing.assert_allclose(p2, g[f"{name}_p2"], rtol=1e-13, atol=1e-14)
        ll = model.compute_log_likelihoods(p2)
        assert (ll is not None) == bool(g[f"{name}_has_ll"])
        if ll is not None:
            np.testing.assert_allclose(ll, g[f"{name}_ll"], rtol=1e-13)


def test_raster_sampling_matches_reference(golden):
    """Raster.sample at points, orders 0 and 1, on grids with decreasing y / decreasing x (raster.py:913-1027)."""
    from oracle import raster as oraster

    g = golden("g12_raster.npz")
    for i in range(3):
        r = oraster.Raster(g[f"r{i}_z"], x=g[f"r{i}_xlim"], y=g[f"r{i}_ylim"])
        np.testing.assert_array_equal(r._centres(0), g[f"r{i}_x"])
        np.testing.assert_array_equal(r._centres(1), g[f"r{i}_y"])
        np.testing.assert_allclose(r.sample(g[f"r{i}_xy"]), g[f"r{i}_linear"], rtol=1e-13, atol=1e-14)
        np.testing.assert_array_equal(r.sample(g[f"r{i}_xy"], order=0), g[f"r{i}_nearest"])
        with pytest.raises(ValueError):
            r.sample(g[f"r{i}_mixed_xy"])


def _e2e_rasters(g, mod):
    dem = mod.Raster(g["dem"], x=g["xlim"], y=g["ylim"])
    dem_sigma = mod.Raster(g["dem_sigma"], x=g["xlim"], y=g["ylim"])
    viewshed = mod.Raster(g["viewshed"], x=g["xlim"], y=g["ylim"])
    return dem, dem_sigma, viewshed


def test_gridded_surfaces_end_to_end_match_reference(golden):
    """Tracks on a gridded dem / dem_sigma, a tangent model on a gridded dem, a viewshed, and surfaces
    that do not cover a track (ValueError captured, NaN rows), with the oracle on the reference's seeds."""
    from oracle import raster as oraster

    g = golden("g12_raster_e2e.npz")
    dem, dem_sigma, viewshed = _e2e_rasters(g, oraster)
    T = len(g["frames"])
    observers = [tracker.Observer(list(g["frames"]), np.tile(g["cam"], (T, 1)), 0.3)]
    matching, taus = np.arange(T)[:, None], np.ones(T - 1)
    cart = dict(n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.02), axyz=(0, 0, 0),
                axyz_sigma=(0.05, 0.05, 0.01))
    cases = {
        "cart": (1301, [motion.CartesianMotion(xy=xy, dem=dem, dem_sigma=dem_sigma, **cart)
                        for xy in [(0.5, -0.5), (-2.0, 1.5), (7.5, 0.0)]], {}),
        "tcart": (1302, [motion.TangentCartesianMotion(xy=xy, dem=dem, dem_sigma=0.2, n=150, xy_sigma=(0.2, 0.2),
                                                       vxy=(0.15, 0.0), vxy_sigma=(0.2, 0.2), axy=(0, 0),
                                                       axy_sigma=(0.05, 0.05), slope_sigma=0.1)
                         for xy in [(-1.0, 1.0), (1.5, 0.5)]], {}),
        "view": (1303, [motion.CartesianMotion(xy=xy, dem=0.0, dem_sigma=0.3, **cart) for xy in [(0.5, -0.5), (3.5, 1.0)]],
                 dict(viewshed=viewshed)),
    }
    for name, (seed, models, kw) in cases.items():
        np.random.seed(seed)
        res = tracker.track(models, observers, matching, taus, tile_size=(15, 15), **kw)
        errors = g[f"{name}_errors"].astype(bool)
        assert [e is not None for e in res["errors"]] == list(errors)
        ok = ~errors
        np.testing.assert_allclose(res["means"][ok], g[f"{name}_means"][ok], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(res["sigmas"][ok], g[f"{name}_sigmas"][ok], rtol=1e-9, atol=1e-14)
        assert np.isnan(res["means"][errors]).all()


def test_motion_models_with_their_own_rasters_match_reference(golden):
    """Five tracks whose motion models carry different dem / dem_sigma rasters (motion.py:136-141)."""
    from oracle import raster as oraster

    g = golden("g22_rasters.npz")
    T = len(g["frames"])
    R = lambda key: oraster.Raster(g[key], x=g["xlim"], y=g["ylim"])  # noqa: E731
    dem, sig = {"a": R("dem_a"), "b": R("dem_b")}, {"a": R("sigma_a"), "b": R("sigma_b")}
    cart = dict(n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.02), axyz=(0, 0, 0),
                axyz_sigma=(0.05, 0.05, 0.01))
    models = [motion.CartesianMotion(xy=xy, dem=0.1 if k == "s" else dem[k], dem_sigma=0.25 if k == "s" else sig[k], **cart)
              for xy, k in zip(g["xy"], g["kinds"])]
    observers = [tracker.Observer(list(g["frames"]), np.tile(g["cam"], (T, 1)), 0.3)]
    np.random.seed(int(g["seed"]))
    res = tracker.track(models, observers, np.arange(T)[:, None], np.ones(T - 1), tile_size=(15, 15))
    assert all(e is None for e in res["errors"])
    np.testing.assert_allclose(res["means"], g["means"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(res["sigmas"], g["sigmas"], rtol=1e-9, atol=1e-14)


def test_orthophoto_observer_matches_reference(golden):
    """Raster images as observer images: Grid.xyz_to_uv (raster.py:423-445) and a two-track run."""
    g = golden("g13_ortho.npz")
    T = len(g["frames"])
    vec = camera.grid_vector((192, 192), g["xlim"], g["ylim"])
    np.testing.assert_allclose(camera.xyz_to_uv(vec, g["xyz"]), g["uv"], rtol=1e-15, atol=1e-13)
    odd = camera.grid_vector((5, 7), g["odd_xlim"], g["odd_ylim"])
    np.testing.assert_allclose(camera.xyz_to_uv(odd, g["xyz"] + [95, 10, 0]), g["odd_uv"], rtol=1e-15, atol=1e-13)
    observers = [tracker.Observer(list(g["frames"]), np.tile(vec, (T, 1)), 0.3)]
    models = [motion.CartesianMotion(xy=xy, dem=0.0, dem_sigma=0.0, n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0),
                                     vxyz_sigma=(0.2, 0.2, 0), axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0))
              for xy in [(0.5, -0.5), (-2.0, 1.5)]]
    np.random.seed(1314)
    res = tracker.track(models, observers, np.arange(T)[:, None], np.ones(T - 1), tile_size=(15, 15))
    np.testing.assert_allclose(res["means"], g["means"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(res["sigmas"], g["sigmas"], rtol=1e-9, atol=1e-14)


def test_inverse_projection_matches_reference(golden):
    """Camera.uv_to_xyz (camera.py:630-663) incl. the k1 closed form and the Oulu undistortion."""
    g = golden("g14_unproject.npz")
    for vec, uv, depth, xd, xa in zip(g["cams"], g["uv"], g["depth"], g["xyz_directions"], g["xyz_absolute"]):
        np.testing.assert_allclose(camera.uv_to_xyz(vec, uv), xd, rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(camera.uv_to_xyz(vec, uv, directions=False, depth=depth), xa, rtol=1e-13,
                                   atol=1e-10)


def test_oracle_highpass_window_sizes(golden):
    """oracle.tiles.extract_tile(highpass_size=...) against the reference's Tracker(highpass={"size": ...}) tiles."""
    from oracle import tiles as otiles

    g = golden("g17_highpass.npz")
    frames = golden("g2_tiles.npz")
    for k, size in enumerate(g["sizes"]):
        size = tuple(int(v) for v in size)
        for name in ("gray", "rgb"):
            f = frames[name]
            tile, hist = otiles.extract_tile(f[0], g["tbox"], return_histogram=True, highpass_size=size)
            np.testing.assert_allclose(tile, g[f"{name}_{k}_tile"], rtol=1e-13, atol=1e-14)
            search = otiles.extract_tile(f[1], g["sbox"], histogram=hist, highpass_size=size)
            np.testing.assert_allclose(search, g[f"{name}_{k}_search"], rtol=1e-13, atol=1e-14)


def test_oracle_highpass_boundary_modes(golden):
    """oracle.tiles.extract_tile(highpass_mode=...) against the reference's Tracker(highpass={"size": .., "mode": ..})."""
    from oracle import tiles as otiles

    g = golden("g26_highpass_modes.npz")
    frames = golden("g2_tiles.npz")
    for k, (size, mode) in enumerate(zip(g["sizes"], g["modes"])):
        size, mode = tuple(int(v) for v in size), str(mode)
        for name in ("gray", "rgb"):
            f = frames[name]
            tile, hist = otiles.extract_tile(f[0], g["tbox"], return_histogram=True, highpass_size=size, highpass_mode=mode)
            np.testing.assert_allclose(tile, g[f"{name}_{k}_tile"], rtol=1e-13, atol=1e-14)
            search = otiles.extract_tile(f[1], g["sbox"], histogram=hist, highpass_size=size, highpass_mode=mode)
            np.testing.assert_allclose(search, g[f"{name}_{k}_search"], rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("name,channels", [("gray", 1), ("rgb", 3)])
def test_oracle_on_uint16_frames(golden, name, channels):
    """uint16 frames (tracker.py:494-534 works on any dtype): the oracle's whole-track loop on np.random against the
    reference run with the same seed (g18; the scene is regenerated from its recipe and checked by checksum)."""
    from glimpse_amd import synth
    from oracle import motion as omotion
    from oracle import tracker as otracker

    g = golden("g18_uint16.npz")
    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    scene = synth.default_scene(cam, seed=12, velocity=(0.15, 0.0), n_frames=5)
    frames = [scene.render(cam, float(t), channels=channels, bits=16) for t in range(5)]
    assert sum(int(f.astype(np.int64).sum()) for f in frames) == int(g[f"{name}_checksum"])
    observers = [otracker.Observer(frames, np.tile(cam, (5, 1)), 0.3)]
    models = [omotion.CartesianMotion(xy=xy, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0),
                                      axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0), dem=0.0, dem_sigma=0.0, n=200)
              for xy in g[f"{name}_xy"]]
    np.random.seed(41)
    res = otracker.track(models, observers, np.arange(5)[:, None], np.ones(4), tile_size=(15, 15))
    np.testing.assert_allclose(res["means"], g[f"{name}_means"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(res["sigmas"], g[f"{name}_sigmas"], rtol=1e-9, atol=1e-10)


def float64_scene():
    """The float64 scene of g20 (tools/make_golden.py: scene64)."""
    from glimpse_amd import synth

    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    scene = synth.default_scene(cam, seed=12, velocity=(0.15, 0.0), n_frames=5)
    frames = [np.power(scene.render(cam, float(t), channels=1, bits=16).astype(np.float64) / 65535.0, 0.8) * 3.5 - 1.25
              for t in range(5)]
    return cam, frames


def float_scenes():
    """The float scenes of g24 (tools/make_golden.py: float_scenes)."""
    cam, frames64 = float64_scene()

    def rgb(f):
        return np.stack([f, 0.9 * f + 0.05, np.roll(f, 1, axis=1) * 1.1 - 0.1], axis=2)

    return cam, {"f32": [f.astype(np.float32) for f in frames64], "f32rgb": [rgb(f).astype(np.float32) for f in frames64],
                 "f64rgb": [rgb(f) for f in frames64]}


@pytest.mark.parametrize("tag", ["f32", "f32rgb", "f64rgb"])
def test_oracle_on_float32_and_multichannel_float_frames(golden, tag):
    """float32 frames and three-channel float frames (tracker.py:494-534 works on any dtype; the normalisation then runs
    in the frame's dtype): the oracle's tiles for explicit boxes and its whole tracks against the reference (g24, under this
    container's NumPy)."""
    from oracle import motion as omotion
    from oracle import tracker as otracker

    g = golden("g24_float_frames.npz")
    cam, scenes = float_scenes()
    frames = scenes[tag]
    assert np.float64(sum(float(np.asarray(f, dtype=np.float64).sum()) for f in frames)) == g[f"{tag}_checksum"]
    tile, hist = tiles.extract_tile(frames[0], g["tbox"], return_histogram=True)
    assert tile.dtype == g[f"{tag}_tile"].dtype and hist[0].dtype == g[f"{tag}_hist_v"].dtype
    np.testing.assert_array_equal(tile, g[f"{tag}_tile"])
    np.testing.assert_array_equal(hist[0], g[f"{tag}_hist_v"])
    np.testing.assert_array_equal(hist[1], g[f"{tag}_hist_q"])
    np.testing.assert_array_equal(tiles.extract_tile(frames[1], g["sbox"], histogram=hist), g[f"{tag}_search"])
    T = len(frames)
    observers = [otracker.Observer(frames, np.tile(cam, (T, 1)), 0.3)]
    models = [omotion.CartesianMotion(xy=xy, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                      axyz_sigma=(0.05, 0.05, 0.0), dem=0.0, dem_sigma=0.0, n=200) for xy in g["xy"]]
    np.random.seed(4300 + len(tag))
    res = otracker.track(models, observers, np.arange(T)[:, None], np.ones(T - 1), tile_size=(15, 15))
    np.testing.assert_allclose(res["means"], g[f"{tag}_means"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(res["sigmas"], g[f"{tag}_sigmas"], rtol=1e-9, atol=1e-10)


def test_oracle_on_float64_frames(golden):
    """float64 frames (tracker.py:494-534 works on any dtype): the oracle's tiles for explicit boxes and its
    whole-track loop on np.random against the reference run with the same seed (g20)."""
    from oracle import motion as omotion
    from oracle import tiles as otiles
    from oracle import tracker as otracker

    g = golden("g20_float64.npz")
    cam, frames = float64_scene()
    assert abs(sum(float(f.sum()) for f in frames) - float(g["checksum"])) < 1e-6 * abs(float(g["checksum"]))
    tile, hist = otiles.extract_tile(frames[0], g["tbox"], return_histogram=True)
    np.testing.assert_allclose(tile, g["tile"], rtol=1e-13, atol=1e-14)
    np.testing.assert_array_equal(hist[1], g["hist_q"])
    np.testing.assert_allclose(hist[0], g["hist_v"], rtol=1e-13, atol=1e-14)
    search = otiles.extract_tile(frames[1], g["sbox"], histogram=hist)
    np.testing.assert_allclose(search, g["search"], rtol=1e-13, atol=1e-14)
    observers = [otracker.Observer(frames, np.tile(cam, (5, 1)), 0.3)]
    models = [omotion.CartesianMotion(xy=xy, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0),
                                      axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0), dem=0.0, dem_sigma=0.0, n=200)
              for xy in g["xy"]]
    np.random.seed(43)
    res = otracker.track(models, observers, np.arange(5)[:, None], np.ones(4), tile_size=(15, 15))
    np.testing.assert_allclose(res["means"], g["means"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(res["sigmas"], g["sigmas"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("tag", ["wide", "tight"])
def test_oracle_with_bilinear_interpolation(golden, tag):
    """Tracker(interpolation={"kx": 1, "ky": 1}) (tracker.py:60, :585-590, :623): the oracle's whole-track loop against
    the reference run with the same seed (g21; `tight`: clouds narrower than a pixel, 2 x 2 surfaces)."""
    from oracle import motion as omotion
    from oracle import tracker as otracker

    g = golden("g21_bilinear.npz")
    scene = golden("g15_ragged.npz")
    observers = [otracker.Observer(list(scene["frames"]), np.tile(scene["cam"], (6, 1)), 0.3, interp=(1, 1))]
    wide = tag == "wide"
    models = [omotion.CartesianMotion(xy=xy, xy_sigma=(0.2, 0.2) if wide else (0.004, 0.004), vxyz=(0.15, 0, 0),
                                      vxyz_sigma=(0.2, 0.2, 0.0) if wide else (0.002, 0.002, 0.0), axyz=(0, 0, 0),
                                      axyz_sigma=(0.05, 0.05, 0.0) if wide else (0.0005, 0.0005, 0.0), dem=0.0,
                                      dem_sigma=0.0, n=200) for xy in g["xy"]]
    np.random.seed(47)
    res = otracker.track(models, observers, np.arange(6)[:, None], np.ones(5), tile_size=(15, 15))
    np.testing.assert_allclose(res["means"], g[f"{tag}_means"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(res["sigmas"], g[f"{tag}_sigmas"], rtol=1e-9, atol=1e-10)


def test_spline_orders_closed_form_matches_reference(golden):
    """Every order RectBivariateSpline takes (1 .. 5, mixed), g23: the SciPy call the oracle makes AND the closed form the
    general kernels implement (oracle/spline.py: fit_general / eval_general -- FITPACK's knots, banded solves, fpbspl)
    against Observer.sample_tile of the reference, down to the least surface size (kx + 1) x (ky + 1)."""
    from oracle import spline as ospline

    g = golden("g23_orders.npz")
    for c in range(int(g["n_cases"])):
        kx, ky = (int(v) for v in g[f"c{c}_k"])
        sse, box, uv, val = g[f"c{c}_sse"].astype(float), g[f"c{c}_box"], g[f"c{c}_uv"], g[f"c{c}_val"]
        np.testing.assert_allclose(ospline.sample_tile(uv, sse, box, kx=kx, ky=ky), val, rtol=0, atol=1e-13)
        cu, cv = ospline.cell_centres(box, sse.shape)
        coef = ospline.fit_general(sse, kx, ky)
        got = ospline.eval_general(coef, kx, ky, cv[0], cu[0], uv[:, 1], uv[:, 0])
        np.testing.assert_allclose(got, val, rtol=0, atol=5e-12)
    z = np.random.default_rng(0).random((9, 11))
    np.testing.assert_array_equal(ospline.fit_general(z, 3, 3), ospline.fit_notaknot(z))


@pytest.mark.parametrize("orders", [(2, 2), (5, 5), (3, 1), (4, 2)])
@pytest.mark.parametrize("tag", ["wide", "tight"])
def test_oracle_with_other_interpolation_orders(golden, orders, tag):
    """Tracker(interpolation={"kx": .., "ky": ..}) for orders other than (3, 3) and (1, 1): the oracle's whole-track loop
    against the reference run with the same seed (g23)."""
    from oracle import motion as omotion
    from oracle import tracker as otracker

    g = golden("g23_orders.npz")
    scene = golden("g15_ragged.npz")
    kx, ky = orders
    observers = [otracker.Observer(list(scene["frames"]), np.tile(scene["cam"], (6, 1)), 0.3, interp=(kx, ky))]
    wide = tag == "wide"
    models = [omotion.CartesianMotion(xy=xy, xy_sigma=(0.2, 0.2) if wide else (0.004, 0.004), vxyz=(0.15, 0, 0),
                                      vxyz_sigma=(0.2, 0.2, 0.0) if wide else (0.002, 0.002, 0.0), axyz=(0, 0, 0),
                                      axyz_sigma=(0.05, 0.05, 0.0) if wide else (0.0005, 0.0005, 0.0), dem=0.0,
                                      dem_sigma=0.0, n=200) for xy in g["xy"]]
    np.random.seed(4700 + 10 * kx + ky)
    res = otracker.track(models, observers, np.arange(6)[:, None], np.ones(5), tile_size=(15, 15))
    np.testing.assert_allclose(res["means"], g[f"k{kx}{ky}_{tag}_means"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(res["sigmas"], g[f"k{kx}{ky}_{tag}_sigmas"], rtol=1e-9, atol=1e-10)
