"""The drop-in Python API (Tracker / Observer / Camera) on the GPU against reference goldens."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import glimpse_amd  # noqa: E402
from tests.helpers_api import DAY, T0, camera_from, models_from, observers_from  # noqa: E402

RTOL = 1e-5


@pytest.mark.parametrize("name", ["g8_c1.npz", "g8_c2mini.npz", "g8_c5mini.npz"])
def test_tracker_track_reproduces_reference(golden, name):
    """np.random.seed(s); Tracker.track(...) == the reference run with the same seed."""
    g = golden(name)
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128)
    models = models_from(g)
    np.random.seed(int(g["seed"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=tuple(int(v) for v in g["tile_size"]), return_particles=True)
    errors = g["errors"].astype(bool)
    assert [e is not None for e in tracks.errors] == list(errors)
    for p in np.nonzero(errors)[0]:
        assert isinstance(tracks.errors[p], IndexError)
        assert np.isnan(tracks.means[p]).all()
    ok = ~errors
    got_images = np.array([[-1 if v is None else int(v) for v in row] for row in tracks.images])
    np.testing.assert_array_equal(got_images, g["matching"])
    np.testing.assert_allclose(tracks.means[ok], g["means"][ok], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas[ok], g["out_sigmas"][ok], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.particles[ok], g["out_particles"][ok], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.weights[ok], g["out_weights"][ok], rtol=RTOL, atol=1e-290)
    assert tracks.means.shape == (len(models), len(g["matching"]), 6)
    assert tracks.params["tile_size"] == tuple(int(v) for v in g["tile_size"])
    assert list(tracks.success) == list(ok)


def test_single_track_error_is_raised(golden):
    g = golden("g8_c2mini.npz")
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128)
    bad = models_from(g)[3]  # starts outside the image
    with pytest.raises(IndexError):
        tracker.track([bad])


def test_public_step_methods(golden):
    """initialize_template / compute_observer_log_likelihoods / update_weights / resample_particles."""
    g = golden("g8_c1.npz")
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128)
    model = models_from(g)[0]
    # state before the first update: reference's evolved particles of step 0
    tracker.particles = g["out_particles"][0, 0].copy()
    tracker.initialize_weights()
    np.testing.assert_allclose(tracker.particle_mean, g["means"][0, 0], rtol=1e-12)
    np.testing.assert_allclose(tracker.compute_particle_sigma(), g["out_sigmas"][0, 0], rtol=1e-10)
    tracker.initialize_template(obs=0, img=0, tile_size=(15, 15))
    np.testing.assert_array_equal(tracker.templates[0]["box"], g["t0_o0_box"])
    np.testing.assert_allclose(tracker.templates[0]["tile"], g["t0_o0_tile"], rtol=1e-11, atol=1e-12)
    tracker.particles = g["s0_evolved"].copy()
    ll = tracker.compute_observer_log_likelihoods(0, 1)
    np.testing.assert_allclose(ll, g["s0_o0_ll"], rtol=RTOL)
    assert tracker.compute_observer_log_likelihoods(0, None) is None
    tracker.update_weights([1], motion_model=model)
    np.testing.assert_allclose(tracker.weights, g["s0_weights"], rtol=RTOL)
    np.random.seed(0)
    u = np.random.random()
    np.random.seed(0)
    tracker.resample_particles()
    from oracle import resample as oresample

    idx = oresample.systematic(g["s0_weights"], u)
    np.testing.assert_allclose(tracker.particles, g["s0_evolved"][idx], rtol=0, atol=0)
    with pytest.raises(IndexError):
        tracker.particles = g["s0_evolved"] + np.array([500.0, 0, 0, 0, 0, 0])
        tracker.initialize_template(obs=0, img=0, tile_size=(15, 15))


def test_observer_mask_and_late_start(golden):
    """observer_mask restricts a track to one observer; its window shrinks accordingly."""
    g = golden("g8_c5mini.npz")
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128)
    models = models_from(g)
    mask = np.array([[True, True], [False, True]])
    np.random.seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), observer_mask=mask)
    # track 1 only sees observer 1, whose first image is at datetime index 1
    assert np.isnan(tracks.means[1, 0]).all() and not np.isnan(tracks.means[1, 1:]).any()
    assert not np.isnan(tracks.means[0]).any()
    # the oracle with the same draws (track-major legacy stream) agrees
    from oracle import tracker as otracker
    from tests.helpers_golden import models_from as omodels
    from tests.helpers_golden import observers_from as oobs

    np.random.seed(3)
    res = otracker.track(omodels(g), oobs(g), g["matching"], np.diff(g["datetimes_days"]), tile_size=(15, 15),
                         observer_mask=mask)
    np.testing.assert_allclose(tracks.means, res["means"], rtol=RTOL, atol=1e-8, equal_nan=True)
    np.testing.assert_allclose(tracks.sigmas, res["sigmas"], rtol=RTOL, atol=1e-8, equal_nan=True)


def test_philox_tracking_recovers_velocity(golden):
    """Device RNG: physically sane result (the scene moves at 0.15 units/day along x)."""
    g = golden("g8_c2mini.npz")
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128)
    models = models_from(g)[:3]
    for m in models:
        m.n = 2000
    tracks = tracker.track(models, tile_size=(15, 15), rng="philox", seed=5)
    v = tracks.vxyz[:, -1]
    assert np.all(np.abs(v[:, 0] - 0.15) < 0.05), v
    assert np.all(np.abs(v[:, 1]) < 0.05), v


def test_philox_tracking_returns_covariances_in_one_run(golden):
    """return_covariances=True on the device RNG (tracker.py:307-308, :352): the frames still go out as one glh_track run
    (glh_track_covariances), the covariances are read from the compact state -- and equal np.cov of the particles the
    same seed returns with return_particles=True (the frame-by-frame path)."""
    g = golden("g8_c2mini.npz")
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128)
    models = models_from(g)[:3]
    for m in models:
        m.n = 1500
    cov = tracker.track(models, tile_size=(15, 15), rng="philox", seed=5, return_covariances=True)
    full = tracker.track(models, tile_size=(15, 15), rng="philox", seed=5, return_particles=True)
    np.testing.assert_array_equal(cov.means, full.means)
    assert cov.sigmas is None and cov.covariances.shape == full.means.shape + (6,)
    for p in range(len(models)):
        for t in range(full.means.shape[1]):
            want = np.cov(full.particles[p, t].T, aweights=full.weights[p, t], ddof=0)
            np.testing.assert_allclose(cov.covariances[p, t], want, rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("method", ["stratified", "residual", "choice"])
def test_philox_tracking_with_the_other_resampling_methods(golden, method):
    """Every resampling method of tracker.py:151-223 drives a whole run on the device RNG (residual resampling on
    the np.random stream: test_tracker_api_variants_reproduce_reference)."""
    g = golden("g8_c2mini.npz")
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128, resample_method=method)
    models = models_from(g)[:3]
    for m in models:
        m.n = 1500
    tracks = tracker.track(models, tile_size=(15, 15), rng="philox", seed=9)
    assert all(e is None for e in tracks.errors)
    v = tracks.vxyz[:, -1]
    assert np.all(np.abs(v[:, 0] - 0.15) < 0.06), v
    assert np.all(np.abs(v[:, 1]) < 0.06), v


@pytest.mark.parametrize("name,tracker_kw,track_kw", [
    ("g9_cov.npz", {}, dict(return_covariances=True)),
    ("g9_stratified.npz", dict(resample_method="stratified"), {}),
    ("g9_choice.npz", dict(resample_method="choice"), {}),
    # residual resampling on the legacy stream (tracker.py:188-203): every step of every track draws a weight-dependent
    # number of uniforms, so the tracks run one after another through the per-track loop, like the reference's
    ("g22_residual.npz", dict(resample_method="residual"), {}),
])
def test_tracker_api_variants_reproduce_reference(golden, name, tracker_kw, track_kw):
    """return_covariances (tracker.py:307-308, :352) and resample_method='stratified' / 'choice' / 'residual'
    (tracker.py:178-209) against reference runs with the same seed."""
    g = golden(name)
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128, **tracker_kw)
    np.random.seed(int(g["seed"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models_from(g), tile_size=tuple(int(v) for v in g["tile_size"]),
                               return_particles=True, **track_kw)
    assert all(e is None for e in tracks.errors)
    np.testing.assert_allclose(tracks.means, g["means"], rtol=RTOL, atol=1e-8)
    if track_kw.get("return_covariances"):
        assert tracks.sigmas is None and tracks.covariances.shape == g["out_sigmas"].shape
        np.testing.assert_allclose(tracks.covariances, g["out_sigmas"], rtol=RTOL, atol=1e-9)
        # the device reduction itself, against NumPy on the device's own particles
        want = np.cov(tracks.particles[0, -1].T, aweights=tracks.weights[0, -1], ddof=0)
        np.testing.assert_allclose(tracks.covariances[0, -1], want, rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(tracks.xyz_sigma, np.sqrt(g["out_sigmas"][:, :, (0, 1, 2), (0, 1, 2)]), rtol=RTOL)
    else:
        np.testing.assert_allclose(tracks.sigmas, g["out_sigmas"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.particles, g["out_particles"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.weights, g["out_weights"], rtol=RTOL, atol=1e-290)


def test_single_track_covariance_and_resample_methods(golden):
    """Tracker.particle_covariance and resample_particles(method) on explicit populations (g5 fixtures)."""
    g5 = golden("g5_resample.npz")
    g = golden("g8_c1.npz")
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128)
    for i in (2, 3, 4):  # n = 100, 129, 1000
        particles, weights = g5[f"r{i}_particles"], g5[f"r{i}_weights"]
        # covariance of the reference's systematic resample output
        tracker.particles = g5[f"r{i}_out_particles"].copy()
        tracker.weights = g5[f"r{i}_out_weights"].copy()
        np.testing.assert_allclose(tracker.particle_covariance, g5[f"r{i}_cov"], rtol=1e-9, atol=1e-18)
        # stratified: same uniforms as the reference -> same indices
        tracker.particles, tracker.weights = particles.copy(), weights.copy()
        np.random.seed(1000 + i)
        tracker.resample_particles(method="stratified")
        idx = g5[f"r{i}_strat_idx"]
        np.testing.assert_array_equal(tracker.particles, particles[idx])
        np.testing.assert_array_equal(tracker.weights, weights[idx])
        # choice: against NumPy's own RandomState.choice from the same state
        tracker.particles, tracker.weights = particles.copy(), weights.copy()
        np.random.seed(2000 + i)
        tracker.resample_particles(method="choice")
        np.random.seed(2000 + i)
        p = weights / weights.sum()
        want = np.random.choice(np.arange(len(p)), size=(len(p),), replace=True, p=p)
        np.testing.assert_array_equal(tracker.particles, particles[want])
        # residual: as written in the reference (tracker.py:188-203), same np.random stream -> same particles,
        # and the stream is left where the reference leaves it (n - sum(repetitions) uniforms consumed)
        tracker.particles, tracker.weights = particles.copy(), weights.copy()
        np.random.seed(int(g5[f"r{i}_resid_seed"]))
        tracker.resample_particles(method="residual")
        np.testing.assert_array_equal(tracker.particles, g5[f"r{i}_resid_out_particles"])
        after = np.random.random()
        np.random.seed(int(g5[f"r{i}_resid_seed"]))
        from oracle import resample as oresample
        oresample.residual(weights, np.random.random)
        assert after == np.random.random()


def test_other_motion_models_on_the_device(golden):
    """Cylindrical / TangentCartesian / TangentCylindrical motion (motion.py:207-522) on the device:
    initialise + evolve with the reference's recorded draws (C ABI), then whole tracks through
    Tracker.track against reference runs with the same seed."""
    import datetime

    from glimpse_amd import _lib
    from tests.test_host_logic import _api_models

    g = golden("g11_motion.npz")
    for name, model in _api_models(g).items():
        draws = [g[f"{name}_draw{i}"] for i in range(int(g[f"{name}_n_draws"]))]
        n = model.n
        if name == "cyl":
            init = np.column_stack((draws[0], draws[1], draws[2]))
            ev = [draws[3], draws[4]]
        else:
            init = np.column_stack((draws[0], draws[1], draws[2], np.zeros(n)))
            ev = [np.column_stack((draws[3], draws[4])), np.column_stack((draws[5], draws[6]))]
        with _lib.Context(1, n, 1, max_search_dim=64, max_frames=2) as ctx:
            ctx.observer_init(0, 1, 64, 64, 1, 0.3)
            ctx.begin_sequence(1, n, (15, 15))
            ctx.set_motion(model.params_full()[None])
            ctx.init_particles(normals=init[None])
            np.testing.assert_allclose(ctx.get_particles()[0], g[f"{name}_p0"], rtol=1e-13, atol=1e-14)
            ctx.evolve(1.5, normals=ev[0][None])
            np.testing.assert_allclose(ctx.get_particles()[0], g[f"{name}_p1"], rtol=1e-12, atol=1e-13)
            ctx.evolve(-0.75, normals=ev[1][None])
            np.testing.assert_allclose(ctx.get_particles()[0], g[f"{name}_p2"], rtol=1e-12, atol=1e-13)
    e = golden("g11_motion_e2e.npz")
    t0, day = datetime.datetime(2020, 1, 1), datetime.timedelta(days=1)
    from tests.helpers_api import camera_from

    images = [glimpse_amd.Image("synthetic", cam=camera_from(e["cam"]), datetime=t0 + i * day, array=e["frames"][i])
              for i in range(len(e["frames"]))]
    models = {
        "cyl": glimpse_amd.CylindricalMotion(xy=(0.5, -0.5), time_unit=day, dem=0.0, dem_sigma=0.3, n=150,
                                             xy_sigma=(0.2, 0.2), vrthz=(0.15, 0.0, 0.0), vrthz_sigma=(0.1, 0.5, 0.02),
                                             arthz=(0, 0, 0), arthz_sigma=(0.03, 0.2, 0.01)),
        "tcart": glimpse_amd.TangentCartesianMotion(xy=(-1.0, 1.0), time_unit=day, dem=0.0, dem_sigma=0.2, n=150,
                                                    xy_sigma=(0.2, 0.2), vxy=(0.15, 0.0), vxy_sigma=(0.2, 0.2),
                                                    axy=(0, 0), axy_sigma=(0.05, 0.05), slope_sigma=0.1),
        "tcyl": glimpse_amd.TangentCylindricalMotion(xy=(1.5, 0.5), time_unit=day, dem=0.0, dem_sigma=0.2, n=150,
                                                     xy_sigma=(0.2, 0.2), vrth=(0.15, 0.0), vrth_sigma=(0.1, 0.5),
                                                     arth=(0, 0), arth_sigma=(0.03, 0.2), slope_sigma=0.1),
    }
    for name, model in models.items():
        tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
        np.random.seed(910)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track([model], tile_size=(15, 15), return_particles=True)
        np.testing.assert_allclose(tracks.means, e[f"{name}_means"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(tracks.sigmas, e[f"{name}_sigmas"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(tracks.particles, e[f"{name}_particles"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(tracks.weights, e[f"{name}_weights"], rtol=RTOL, atol=1e-290)


def test_raster_sample_on_the_device(golden):
    """glimpse_amd.Raster.sample (glh_stage_raster_sample) against the reference: bilinear and nearest, grids
    with decreasing y / decreasing x, half-cell border extrapolation, bounds errors (raster.py:913-1027)."""
    g = golden("g12_raster.npz")
    for i in range(3):
        r = glimpse_amd.Raster(g[f"r{i}_z"], x=g[f"r{i}_xlim"], y=g[f"r{i}_ylim"])
        np.testing.assert_array_equal(r.x, g[f"r{i}_x"])
        np.testing.assert_array_equal(r.y, g[f"r{i}_y"])
        np.testing.assert_allclose(r.d, g[f"r{i}_d"], rtol=1e-15)
        np.testing.assert_allclose(r.sample(g[f"r{i}_xy"]), g[f"r{i}_linear"], rtol=1e-12, atol=1e-13)
        np.testing.assert_array_equal(r.sample(g[f"r{i}_xy"], order=0), g[f"r{i}_nearest"])
        np.testing.assert_array_equal(r.inbounds_xy(g[f"r{i}_mixed_xy"]), g[f"r{i}_mixed_in"])
        with pytest.raises(ValueError):
            r.sample(g[f"r{i}_mixed_xy"])
        filled = r.sample(g[f"r{i}_mixed_xy"], bounds_error=False, fill_value=-7.0)
        assert (filled[~g[f"r{i}_mixed_in"]] == -7.0).all() and (filled[g[f"r{i}_mixed_in"]] != -7.0).all()


def test_raster_coordinates_must_be_a_uniform_grid():
    """The kernels find a sample's cell from the cell size (glh_math.h: raster_interval), which is scipy's find_indices only
    for cell centres of a uniform grid -- what every glimpse Raster has.  The C ABI refuses anything else, loudly."""
    from glimpse_amd import _lib

    class Warped(glimpse_amd.Raster):
        def device_args(self):
            z, nx, ny, gx, gy, sx, sy, x0, x1, y0, y1 = super().device_args()
            gx = gx.copy()
            gx[3:] += 0.3 * abs(float(self.d[0]))  # (ascending still, but not uniform)
            return z, nx, ny, gx, gy, sx, sy, x0, x1, y0, y1

    z = np.arange(12.0 * 9).reshape(9, 12)
    ok = glimpse_amd.Raster(z, x=(0.0, 24.0), y=(18.0, 0.0))
    assert np.isfinite(ok.sample([[5.0, 5.0]])).all()
    with pytest.raises(_lib.GlhError, match="uniform grid"):
        Warped(z, x=(0.0, 24.0), y=(18.0, 0.0)).sample([[5.0, 5.0]])
    with _lib.Context(2, 64, 1, max_frames=2) as ctx:
        ctx.set_raster(_lib.RASTER_DEM, ok)
        with pytest.raises(_lib.GlhError, match="uniform grid"):
            ctx.set_raster(_lib.RASTER_DEM_SIGMA, Warped(z, x=(0.0, 24.0), y=(18.0, 0.0)))


def test_gridded_surfaces_end_to_end(golden):
    """Tracker.track on a gridded dem / dem_sigma (CartesianMotion), a tangent model on a gridded dem, and a
    viewshed (tracker.py:114-117), incl. tracks the surfaces do not cover (ValueError captured, NaN rows),
    against reference runs with the same seeds."""
    import datetime

    from tests.helpers_api import camera_from

    g = golden("g12_raster_e2e.npz")
    t0, day = datetime.datetime(2020, 1, 1), datetime.timedelta(days=1)
    images = [glimpse_amd.Image("synthetic", cam=camera_from(g["cam"]), datetime=t0 + i * day, array=g["frames"][i])
              for i in range(len(g["frames"]))]
    dem = glimpse_amd.Raster(g["dem"], x=g["xlim"], y=g["ylim"])
    dem_sigma = glimpse_amd.Raster(g["dem_sigma"], x=g["xlim"], y=g["ylim"])
    viewshed = glimpse_amd.Raster(g["viewshed"], x=g["xlim"], y=g["ylim"])
    cart = dict(time_unit=day, n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.02),
                axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.01))
    cases = {
        "cart": (1301, [glimpse_amd.CartesianMotion(xy=xy, dem=dem, dem_sigma=dem_sigma, **cart)
                        for xy in [(0.5, -0.5), (-2.0, 1.5), (7.5, 0.0)]], {}),
        "tcart": (1302, [glimpse_amd.TangentCartesianMotion(xy=xy, time_unit=day, dem=dem, dem_sigma=0.2, n=150,
                                                            xy_sigma=(0.2, 0.2), vxy=(0.15, 0.0), vxy_sigma=(0.2, 0.2),
                                                            axy=(0, 0), axy_sigma=(0.05, 0.05), slope_sigma=0.1)
                         for xy in [(-1.0, 1.0), (1.5, 0.5)]], {}),
        "view": (1303, [glimpse_amd.CartesianMotion(xy=xy, dem=0.0, dem_sigma=0.3, **cart)
                        for xy in [(0.5, -0.5), (3.5, 1.0)]], dict(viewshed=viewshed)),
    }
    for name, (seed, models, kw) in cases.items():
        tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128, **kw)
        np.random.seed(seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
        errors = g[f"{name}_errors"].astype(bool)
        assert [e is not None for e in tracks.errors] == list(errors), name
        for p in np.nonzero(errors)[0]:
            assert type(tracks.errors[p]).__name__ == str(g[f"{name}_error_types"][p])
            assert np.isnan(tracks.means[p]).all()
        ok = ~errors
        np.testing.assert_allclose(tracks.means[ok], g[f"{name}_means"][ok], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(tracks.sigmas[ok], g[f"{name}_sigmas"][ok], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(tracks.particles[ok], g[f"{name}_particles"][ok], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(tracks.weights[ok], g[f"{name}_weights"][ok], rtol=RTOL, atol=1e-290)
        # the same tracks on two worker processes (device RNG: the parallel run draws what the single-process run draws);
        # the surfaces' arrays reach the workers through shared memory (tests/test_parallel_pool.py)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            one = tracker.track(models, tile_size=(15, 15), rng="philox", seed=4)
            two = tracker.track(models, tile_size=(15, 15), rng="philox", seed=4, parallel=2)
        np.testing.assert_array_equal(two.means, one.means)
        np.testing.assert_array_equal(two.sigmas, one.sigmas)
        assert [type(e) for e in two.errors] == [type(e) for e in one.errors]
        assert isinstance(dem.array, np.ndarray) and isinstance(viewshed.array, np.ndarray)
        tracker.close()


def test_motion_models_with_their_own_rasters(golden):
    """Every motion model may carry its OWN dem / dem_sigma rasters (motion.py:136-141).  A device context holds one
    raster per surface, so consecutive models that share theirs form a batch (constant surfaces mix freely) and the
    batches run in track order -- np.random is consumed like the reference consumes it: same seed, same tracks."""
    import datetime

    from glimpse_amd.tracker import _batches
    from tests.helpers_api import camera_from

    g = golden("g22_rasters.npz")
    t0, day = datetime.datetime(2020, 1, 1), datetime.timedelta(days=1)
    images = [glimpse_amd.Image("synthetic", cam=camera_from(g["cam"]), datetime=t0 + i * day, array=g["frames"][i])
              for i in range(len(g["frames"]))]
    R = lambda key: glimpse_amd.Raster(g[key], x=g["xlim"], y=g["ylim"])  # noqa: E731
    dem, sig = {"a": R("dem_a"), "b": R("dem_b")}, {"a": R("sigma_a"), "b": R("sigma_b")}
    cart = dict(time_unit=day, n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.02),
                axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.01))
    models = [glimpse_amd.CartesianMotion(xy=xy, dem=0.1 if k == "s" else dem[k], dem_sigma=0.25 if k == "s" else sig[k],
                                          **cart) for xy, k in zip(g["xy"], g["kinds"])]
    assert list(g["kinds"]) == ["a", "a", "b", "s", "a"] and _batches(models) == [0, 2, 4]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
    np.random.seed(int(g["seed"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
    assert all(e is None for e in tracks.errors)
    np.testing.assert_allclose(tracks.means, g["means"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g["sigmas"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.particles, g["particles"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.weights, g["weights"], rtol=RTOL, atol=1e-290)


def test_orthophoto_observer_on_the_device(golden):
    """Observer([Raster, ...]) (observer.py:26): projection through Grid.xyz_to_uv (raster.py:423-445) on the
    device and a two-track run against the reference with the same seed."""
    import datetime

    g = golden("g13_ortho.npz")
    t0, day = datetime.datetime(2020, 1, 1), datetime.timedelta(days=1)
    rasters = [glimpse_amd.Raster(g["frames"][i], x=g["xlim"], y=g["ylim"], datetime=t0 + i * day)
               for i in range(len(g["frames"]))]
    np.testing.assert_allclose(rasters[0].xyz_to_uv(g["xyz"]), g["uv"], rtol=1e-15, atol=1e-13)
    odd = glimpse_amd.Raster(np.zeros((7, 5)), x=g["odd_xlim"], y=g["odd_ylim"])
    np.testing.assert_allclose(odd.xyz_to_uv(g["xyz"] + [95, 10, 0]), g["odd_uv"], rtol=1e-15, atol=1e-13)
    assert rasters[0].inbounds(np.array([[0, 0], [192, 192], [193, 5]])).tolist() == [True, True, False]
    models = [glimpse_amd.CartesianMotion(xy=xy, time_unit=day, dem=0.0, dem_sigma=0.0, n=150, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0)) for xy in [(0.5, -0.5), (-2.0, 1.5)]]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(rasters, sigma=0.3)], max_search_dim=128)
    np.random.seed(1314)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
    assert all(e is None for e in tracks.errors)
    np.testing.assert_allclose(tracks.means, g["means"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g["sigmas"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.particles, g["particles"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.weights, g["weights"], rtol=RTOL, atol=1e-290)


def _pixel_centres(imgsz):
    u, v = np.meshgrid(np.arange(imgsz[0]) + 0.5, np.arange(imgsz[1]) + 0.5)
    return np.column_stack((u.ravel(), v.ravel()))


def test_inverse_projection_on_the_device(golden):
    """Camera.uv_to_xyz (camera.py:630-663) on the device against the reference's outputs, and the reference's
    own round-trip tests (tests/test_camera.py:34-88) restated: project(unproject(pixel centres)) returns
    the pixel centres within 1e-14 (pinhole) / 1e-12 (k1..k6, p1, p2, and extreme k1 = +-2)."""
    g = golden("g14_unproject.npz")
    for vec, uv, depth, xd, xa in zip(g["cams"], g["uv"], g["depth"], g["xyz_directions"], g["xyz_absolute"]):
        cam = camera_from(vec)
        np.testing.assert_allclose(cam.uv_to_xyz(uv), xd, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(cam.uv_to_xyz(uv, directions=False, depth=depth), xa, rtol=1e-11, atol=1e-9)

        # xyz_to_uv(return_depth=True): unprojecting to depth d along the optical axis and projecting back
        # returns d (camera.py:1468-1469)
        uv2, d2 = cam.xyz_to_uv(xa, return_depth=True)
        np.testing.assert_array_equal(uv2, cam.xyz_to_uv(xa))
        if not vec[20]:  # (uv_to_xyz does not undo the earth-curvature correction, camera.py:1483-1497)
            np.testing.assert_allclose(d2, np.broadcast_to(depth, d2.shape), rtol=1e-12)
    # the doctest of camera.py:622-627, and a point behind the camera: NaN uv, negative depth
    cam = glimpse_amd.Camera(imgsz=10, f=10)
    uv, d = cam.xyz_to_uv(np.array([(0.0, 10.0, 0.0), (0.0, -4.0, 0.0)]), return_depth=True)
    np.testing.assert_array_equal(uv[0], [5.0, 5.0])
    assert np.isnan(uv[1]).all() and list(d) == [10.0, -4.0]

    def reprojection_errors(cam):
        uv = _pixel_centres(cam.imgsz.astype(int))
        return np.linalg.norm(cam.xyz_to_uv(cam.uv_to_xyz(uv), directions=True) - uv, axis=1)

    kw = dict(imgsz=(100, 100), f=(100, 100))
    assert reprojection_errors(glimpse_amd.Camera(**kw)).max() < 1e-14
    for extra in (dict(k=0.1), dict(k=-0.1), dict(k=[0.1] * 6), dict(p=[0.01] * 2), dict(k=[0.1] * 6, p=[0.01] * 2),
                  dict(k=2), dict(k=-2)):
        assert reprojection_errors(glimpse_amd.Camera(**kw, **extra)).max() < 1e-12, extra
    # default camera: the image centre looks along +y (camera.py:650-655 doctest)
    cam = glimpse_amd.Camera(imgsz=10, f=10)
    np.testing.assert_allclose(cam.uv_to_xyz(np.array([(5, 5)])), [[0, 1, 0]], atol=1e-15)
    np.testing.assert_allclose(cam.uv_to_xyz(np.array([(5, 5)]), depth=10), [[0, 10, 0]], atol=1e-14)


def test_tracking_from_image_files_equals_tracking_from_arrays(golden, tmp_path):
    """Frames that live in files (decoded by a thread pool, staged through pinned buffers, uploaded on a copy
    stream) give the run of the same frames held as arrays, bit for bit."""
    PIL = pytest.importorskip("PIL.Image")
    g = golden("g8_c5mini.npz")
    tile = tuple(int(v) for v in g["tile_size"])

    def run(observers):
        tracker = glimpse_amd.Tracker(observers, max_search_dim=128)
        np.random.seed(int(g["seed"]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return tracker.track(models_from(g), tile_size=tile)

    from_arrays = run(observers_from(g))
    observers = observers_from(g)
    for o, obs in enumerate(observers):
        for i, img in enumerate(obs.images):
            path = tmp_path / f"o{o}_{i:03d}.png"
            PIL.fromarray(img.array).save(path)
            img.path, img.array = str(path), None
        obs.cache = bool(o % 2)  # one observer keeps the decoded frames, the other does not
    from_files = run(observers)
    np.testing.assert_array_equal(from_files.means, from_arrays.means)
    np.testing.assert_array_equal(from_files.sigmas, from_arrays.sigmas)
    assert (observers[0].images[1].array is None) and (observers[1].images[1].array is not None)
    # Device RNG: the frame loop (glh_track) runs on the frames already resident while the later files are still being
    # decoded (tracker.py: _FrameFeed) -- the sequence goes to the device in several calls, the status words of every
    # frame are read afterwards; same tracks as from arrays, again after the frames were forgotten.
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = glimpse_amd.Tracker(observers_from(g), max_search_dim=128).track(models_from(g), tile_size=tile, rng="philox", seed=3)
        for obs in observers:
            obs.cache = False
            for img in obs.images:
                img.array = None
        tracker = glimpse_amd.Tracker(observers, max_search_dim=128)
        got = tracker.track(models_from(g), tile_size=tile, rng="philox", seed=3)
        stats = dict(tracker._feed_stats)
        assert stats["files"] == stats["frames"] == sum(len(obs.images) for obs in observers)
        assert stats["threads"] + stats["processes"] >= 1  # (a pool of decoder processes from eight files on)
        assert stats["bytes"] == sum(img.read(cache=False).nbytes for obs in observers for img in obs.images)
        np.testing.assert_array_equal(got.means, ref.means)
        np.testing.assert_array_equal(got.sigmas, ref.sigmas)
        assert [type(e) for e in got.errors] == [type(e) for e in ref.errors]
        resident = tracker.track(models_from(g), tile_size=tile, rng="philox", seed=3)  # nothing is read again
        assert tracker._feed_stats["frames"] == 0
        tracker.forget_frames()
        again = tracker.track(models_from(g), tile_size=tile, rng="philox", seed=3)
        assert tracker._feed_stats["files"] == stats["files"]
        for t in (resident, again):
            np.testing.assert_array_equal(t.means, ref.means)
        tracker.close()


def test_ragged_particle_counts_reproduce_reference(golden):
    """Motion models with different n in one Tracker.track call (tracker.py:305-314): runs of equal n are batched,
    the batches consume np.random in track order, so means, sigmas, the captured error of the track that starts
    outside the image, reduce_particles results and the position of the stream afterwards match the reference."""
    g = golden("g15_ragged.npz")
    cam = camera_from(g["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f) for i, f in enumerate(g["frames"])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
    models = [glimpse_amd.CartesianMotion(xy=xy, time_unit=DAY, dem=0.0, dem_sigma=0.0, n=int(n), xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy, n in zip(g["xy"], g["n_particles"])]
    np.random.seed(int(g["seed"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15),
                               reduce_particles=lambda p, w: (p.shape, float(np.nansum(w))))
    assert np.random.random() == float(g["random_after"])
    errors = g["errors"].astype(bool)
    assert [e is not None for e in tracks.errors] == list(errors)
    assert isinstance(tracks.errors[3], IndexError)
    np.testing.assert_allclose(tracks.means, g["means"], rtol=RTOL, atol=1e-8, equal_nan=True)
    np.testing.assert_allclose(tracks.sigmas, g["sigmas"], rtol=RTOL, atol=1e-8, equal_nan=True)
    assert [r[0][1] for r in tracks.reduced] == list(g["reduced_n"])
    np.testing.assert_allclose([r[1] for r in tracks.reduced], g["reduced_w"], rtol=RTOL)
    assert tracks.particles is None and tracks.params["motion_models"] is models


def test_parallel_workers_reproduce_the_single_process_run(golden):
    """Tracker.track(parallel=N) (tracker.py:236, :381-387; glimpse_amd/parallel.py): N PERSISTENT worker processes, each
    with its own context and its contiguous block of tracks, the frames shared with them once through shared memory.  The
    device RNG is keyed on the global track index, so the parallel run IS the single-process run, value for value;
    warnings, errors and the NaN rows of a failing track come back in order.  A second call finds the workers, their
    contexts and their uploaded frames in place.  (Several workers on ONE GPU cannot make an RCCL communicator: the
    history then comes through host memory, and the result says so.)"""
    import time

    g = golden("g8_c2mini.npz")
    tile = tuple(int(v) for v in g["tile_size"])
    models = models_from(g)  # the 4th starts outside the image
    tracker = glimpse_amd.Tracker(observers_from(g), max_search_dim=128)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        serial = tracker.track(models, tile_size=tile, rng="philox", seed=5)
        last_p, last_w = tracker.particles.copy(), tracker.weights.copy()
        for n in (2, 4):
            t0 = time.perf_counter()
            par = tracker.track(models, tile_size=tile, rng="philox", seed=5, parallel=n)
            first_call = time.perf_counter() - t0
            np.testing.assert_array_equal(par.means, serial.means)
            np.testing.assert_array_equal(par.sigmas, serial.sigmas)
            assert [type(e) for e in par.errors] == [type(e) for e in serial.errors]
            assert isinstance(par.errors[3], IndexError) and np.isnan(par.means[3]).all()
            assert par.params["parallel"] == n and par.means.shape == serial.means.shape
            np.testing.assert_array_equal(tracker.particles, last_p)
            np.testing.assert_array_equal(tracker.weights, last_w)
            info = par.parallel_info
            assert par.transport in ("rccl", "host") and info["workers"] == n and info["frames_shared_now"]
            if n > glimpse_amd._lib.device_count():  # (workers that share a GPU: host transport, said so, never tried)
                assert par.transport == "host"
            assert all(info["contexts_made"])
            pids = [p.pid for p in tracker._pool.procs]
            # again: the same processes, no context is made, no frame is shared or uploaded again
            t0 = time.perf_counter()
            again = tracker.track(models, tile_size=tile, rng="philox", seed=5, parallel=n)
            second_call = time.perf_counter() - t0
            np.testing.assert_array_equal(again.means, serial.means)
            np.testing.assert_array_equal(again.sigmas, serial.sigmas)
            assert [p.pid for p in tracker._pool.procs] == pids
            assert not any(again.parallel_info["contexts_made"]) and not again.parallel_info["frames_shared_now"]
            assert second_call < first_call, (first_call, second_call)
        # covariances and particles are per-worker host downloads, in track order
        cov = tracker.track(models[:3], tile_size=tile, rng="philox", seed=5, parallel=2, return_covariances=True,
                            return_particles=True)
        ref = tracker.track(models[:3], tile_size=tile, rng="philox", seed=5, return_covariances=True,
                            return_particles=True)
        np.testing.assert_array_equal(cov.means, ref.means)
        np.testing.assert_array_equal(cov.covariances, ref.covariances)
        np.testing.assert_array_equal(cov.particles, ref.particles)
        np.testing.assert_array_equal(cov.weights, ref.weights)
        assert cov.sigmas is None
        # host-fed draws: each worker consumes its own np.random stream -- a valid filter run, not the serial stream
        np.random.seed(3)
        par = tracker.track(models[:3], tile_size=tile, parallel=2)
        assert np.isfinite(par.means).all() and abs(np.median(par.means[:, -1, 3]) - 0.15) < 0.05
        par = tracker.track(models[:3], tile_size=tile, parallel=True)  # True = one worker per GPU (here: one: this process)
        assert np.isfinite(par.means).all()
    tracker.close()
    assert tracker._pool is None
    assert glimpse_amd.Tracker._parse_parallel(False, 10) == 0 and glimpse_amd.Tracker._parse_parallel(8, 3) == 3


def test_parallel_workers_with_blocks_that_are_not_one_batch(golden):
    """`track(parallel=2)` when a worker's tracks are NOT one device batch -- ragged particle counts, a user-defined motion
    model among them: the workers get the model objects themselves (not a parameter table) and hand their host results over;
    with the device RNG (keyed on the global track index) the ragged run equals the single-process run value for value,
    `reduce_particles` runs in the workers, errors come back in track order."""
    from tests import custom_motion as cm

    g = golden("g15_ragged.npz")
    cam = camera_from(g["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f) for i, f in enumerate(g["frames"])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
    models = [glimpse_amd.CartesianMotion(xy=xy, time_unit=DAY, dem=0.0, dem_sigma=0.0, n=int(n), xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy, n in zip(g["xy"], g["n_particles"])]
    assert len(set(int(n) for n in g["n_particles"])) > 1
    reduce = cm.shape_and_weight
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        serial = tracker.track(models, tile_size=(15, 15), rng="philox", seed=2, reduce_particles=reduce)
        par = tracker.track(models, tile_size=(15, 15), rng="philox", seed=2, reduce_particles=reduce, parallel=2)
        assert par.transport == "host" and par.parallel_info["workers"] == 2
        assert [type(e) for e in par.errors] == [type(e) for e in serial.errors]
        for a, b in zip(par.means, serial.means):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(par.sigmas, serial.sigmas):
            np.testing.assert_array_equal(a, b)
        assert par.reduced == serial.reduced
        # a user-defined model among the tracks: its methods run on the host of whichever worker has it
        mixed = [models[0], cm.SpeedPriorMotion(tuple(g["xy"][1]), DAY, n=150), models[2], models[4]]
        np.random.seed(4)
        out = tracker.track(mixed, tile_size=(15, 15), parallel=2)
        assert [e is None for e in out.errors] == [True] * 4
        assert all(np.isfinite(np.asarray(m)).all() for m in out.means) and len(out.means) == 4
    tracker.close()


def test_user_defined_motion_models_reproduce_reference(golden):
    """Motion models the library does not know (the duck type of motion.py:13-89) in one Tracker.track call with a
    built-in one: their own initialize / evolve / log-likelihood methods run on the host, everything between them on
    the device, np.random consumed in the reference's order -- the reference's tracks (g16: generated by running the
    reference Tracker on the same classes)."""
    from tests import custom_motion as cm

    g = golden("g16_custom_motion.npz")
    cam = camera_from(g["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f) for i, f in enumerate(g["frames"])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
    xy = g["xy"]
    models = [cm.DriftMotion(tuple(xy[0]), DAY, n=150), cm.SpeedPriorMotion(tuple(xy[1]), DAY, n=150),
              glimpse_amd.CartesianMotion(xy=tuple(xy[2]), time_unit=DAY, dem=0.0, dem_sigma=0.0, n=150,
                                          xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0),
                                          axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0)),
              cm.NoTermMotion(tuple(xy[3]), DAY, n=150)]
    np.random.seed(int(g["seed"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
    assert np.random.random() == float(g["random_after"])
    assert [e is not None for e in tracks.errors] == list(g["errors"].astype(bool))
    assert isinstance(tracks.errors[0], AttributeError)  # DriftMotion has no compute_log_likelihoods (tracker.py:143)
    np.testing.assert_allclose(tracks.means, g["means"], rtol=RTOL, atol=1e-8, equal_nan=True)
    np.testing.assert_allclose(tracks.sigmas, g["sigmas"], rtol=RTOL, atol=1e-8, equal_nan=True)
    np.testing.assert_allclose(tracks.particles[1][-1], g["last_particles_1"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.weights[1][-1], g["last_weights_1"], rtol=RTOL, atol=1e-290)
    with pytest.raises(TypeError):
        tracker.track([type("NotAModel", (), {"time_unit": DAY, "n": 10})()])


@pytest.mark.parametrize("size", [(3, 3), (7, 7)])
def test_tracking_with_another_highpass_window_reproduces_reference(golden, size):
    """Tracker(highpass={"size": size}) end to end against the reference run with the same np.random seed."""
    g = golden("g17_highpass.npz")
    scene = golden("g15_ragged.npz")
    cam = camera_from(scene["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f)
              for i, f in enumerate(scene["frames"])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], highpass={"size": size}, max_search_dim=128)
    models = [glimpse_amd.CartesianMotion(xy=tuple(xy), time_unit=DAY, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in g["e2e_xy"]]
    np.random.seed(31)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15))
    np.testing.assert_allclose(tracks.means, g[f"e2e_means_{size[0]}"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g[f"e2e_sigmas_{size[0]}"], rtol=RTOL, atol=1e-8)
    with pytest.raises(NotImplementedError):
        glimpse_amd.Tracker([glimpse_amd.Observer(images)], highpass={"size": (4, 4)})


@pytest.mark.parametrize("tag,highpass", [("nearest", {"size": (5, 5), "mode": "nearest"}),
                                          ("mirror", {"size": 3, "mode": "mirror"})])
def test_tracking_with_another_highpass_boundary_mode_reproduces_reference(golden, tag, highpass):
    """Tracker(highpass={"size": .., "mode": ..}) end to end against the reference run with the same np.random seed (g26);
    the fused kernel takes such runs on its general instantiations."""
    g = golden("g26_highpass_modes.npz")
    scene = golden("g15_ragged.npz")
    cam = camera_from(scene["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f)
              for i, f in enumerate(scene["frames"])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], highpass=highpass, max_search_dim=128)
    models = [glimpse_amd.CartesianMotion(xy=tuple(xy), time_unit=DAY, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in g["e2e_xy"]]
    np.random.seed(31)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15))
    np.testing.assert_allclose(tracks.means, g[f"e2e_means_{tag}"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g[f"e2e_sigmas_{tag}"], rtol=RTOL, atol=1e-8)
    with pytest.raises(NotImplementedError):
        glimpse_amd.Tracker([glimpse_amd.Observer(images)], highpass={"size": 5, "mode": "constant"})
    with pytest.raises(NotImplementedError):
        glimpse_amd.Tracker([glimpse_amd.Observer(images)], highpass={"size": 5, "origin": 1})


@pytest.mark.parametrize("name,channels", [("gray", 1), ("rgb", 3)])
def test_tracking_on_uint16_frames_reproduces_reference(golden, name, channels):
    """16-bit frames, gray and RGB (Tracker.extract_tile works on any dtype, tracker.py:494-534): whole tracks against
    the reference run with the same np.random seed, and the last track's template (tile, histogram, box)."""
    from glimpse_amd import synth

    g = golden("g18_uint16.npz")
    cam_vec = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    scene = synth.default_scene(cam_vec, seed=12, velocity=(0.15, 0.0), n_frames=5)
    frames = [scene.render(cam_vec, float(t), channels=channels, bits=16) for t in range(5)]
    assert frames[0].dtype == np.uint16 and sum(int(f.astype(np.int64).sum()) for f in frames) == int(g[f"{name}_checksum"])
    cam = camera_from(cam_vec)
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f) for i, f in enumerate(frames)]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
    models = [glimpse_amd.CartesianMotion(xy=tuple(xy), time_unit=DAY, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in g[f"{name}_xy"]]
    np.random.seed(41)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15))
    assert all(e is None for e in tracks.errors)
    np.testing.assert_allclose(tracks.means, g[f"{name}_means"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g[f"{name}_sigmas"], rtol=RTOL, atol=1e-8)
    tpl = tracker._ctx.get_template(0, len(models) - 1)
    np.testing.assert_array_equal(tpl["box"], g[f"{name}_tpl_box"])
    np.testing.assert_array_equal(tpl["histogram"][1], g[f"{name}_tpl_hist_q"])
    np.testing.assert_allclose(tpl["histogram"][0], g[f"{name}_tpl_hist_v"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(tpl["tile"], g[f"{name}_tpl_tile"], rtol=1e-12, atol=1e-13)
    # the device RNG and a 7 x 7 high-pass window run on the same (staged) kernels
    t2 = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128, highpass={"size": 7})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = t2.track(models, tile_size=(15, 15), rng="philox", seed=3)
    assert np.isfinite(tr.means).all() and np.all(np.abs(tr.means[:, -1, 3] - 0.15) < 0.06)
    # other sample types are refused, not cast
    bad = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f.astype(np.int32))
           for i, f in enumerate(frames)]
    with pytest.raises(NotImplementedError):
        glimpse_amd.Tracker([glimpse_amd.Observer(bad, sigma=0.3)]).track(models, tile_size=(15, 15))


@pytest.mark.parametrize("name", ["gray", "rgb"])
def test_tracker_extract_tile_matches_reference(golden, name):
    """Tracker.extract_tile (tracker.py:494-534) as the reference's own callers use it: return_histogram=True for a
    template (tracker.py:554-556) and histogram= for a search tile (tracker.py:608)."""
    g = golden("g2_tiles.npz")
    cam = camera_from(g["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f) for i, f in enumerate(g[name][:2])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images)])
    for b in range(3):
        tbox, sbox = g[f"{name}_{b}_tbox"], g[f"{name}_{b}_sbox"]
        tile, (hv, hq) = tracker.extract_tile(0, 0, tbox, return_histogram=True)
        assert tile.dtype == np.float64
        np.testing.assert_array_equal(hq, g[f"{name}_{b}_hist_q"])
        np.testing.assert_allclose(hv, g[f"{name}_{b}_hist_v"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(tile, g[f"{name}_{b}_tile"], rtol=1e-12, atol=1e-13)
        np.testing.assert_array_equal(tracker.extract_tile(0, 0, tbox), tile)
        search = tracker.extract_tile(0, 1, sbox, histogram=(hv, hq))
        assert search.dtype == np.float64
        np.testing.assert_allclose(search, g[f"{name}_{b}_search"], rtol=0, atol=2e-6)  # float32 search tile
    with pytest.raises(NotImplementedError):
        tracker.extract_tile(0, 1, sbox, histogram=(hv, hq), return_histogram=True)


def test_tracking_on_float64_frames_reproduces_reference(golden):
    """One-channel float64 frames (Tracker.extract_tile works on any dtype, tracker.py:494-534; the distinct values of
    a tile by counting instead of a key histogram): whole tracks against the reference run with the same np.random
    seed, and the last track's template (tile, histogram -- one value of the 225 repeats, box)."""
    from tests.test_oracle_golden import float64_scene

    g = golden("g20_float64.npz")
    cam_vec, frames = float64_scene()
    assert frames[0].dtype == np.float64
    cam = camera_from(cam_vec)
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f) for i, f in enumerate(frames)]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
    models = [glimpse_amd.CartesianMotion(xy=tuple(xy), time_unit=DAY, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in g["xy"]]
    np.random.seed(43)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15))
    assert all(e is None for e in tracks.errors)
    np.testing.assert_allclose(tracks.means, g["means"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g["sigmas"], rtol=RTOL, atol=1e-8)
    tpl = tracker._ctx.get_template(0, len(models) - 1)
    np.testing.assert_array_equal(tpl["box"], g["tpl_box"])
    assert len(tpl["histogram"][0]) == len(g["tpl_hist_v"]) < tpl["tile"].size  # (a repeated value is merged)
    np.testing.assert_array_equal(tpl["histogram"][1], g["tpl_hist_q"])
    np.testing.assert_allclose(tpl["histogram"][0], g["tpl_hist_v"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(tpl["tile"], g["tpl_tile"], rtol=1e-11, atol=1e-12)
    # the device RNG runs on the same (staged) kernels
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = tracker.track(models, tile_size=(15, 15), rng="philox", seed=3)
    assert np.isfinite(tr.means).all() and np.all(np.abs(tr.means[:, -1, 3] - 0.15) < 0.06)


@pytest.mark.parametrize("tag", ["f32", "f32rgb", "f64rgb"])
def test_tracking_float32_and_multichannel_float_frames_reproduces_reference(golden, tag):
    """float32 frames (one or three channels) and three-channel float64 frames (tracker.py:494-534 works on any dtype):
    the reference normalises a tile in the frame's own dtype -- a float32 mean, standard deviation and scaling, summed in
    NumPy's order (row by row over the strided view of a one-channel tile, flat over the channel mean of an RGB one) --
    and so does the device: the template of a float32 frame (tile and sorted distinct values) equals the reference's BIT
    FOR BIT, the tracks follow (g24, reference run under this container's NumPy, same np.random seed)."""
    from tests.test_oracle_golden import float_scenes

    g = golden("g24_float_frames.npz")
    cam_vec, scenes = float_scenes()
    frames = scenes[tag]
    cam = camera_from(cam_vec)
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f) for i, f in enumerate(frames)]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
    models = [glimpse_amd.CartesianMotion(xy=tuple(xy), time_unit=DAY, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in g["xy"]]
    np.random.seed(4300 + len(tag))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15))
    assert all(e is None for e in tracks.errors)
    tpl = tracker._ctx.get_template(0, len(models) - 1)
    np.testing.assert_array_equal(tpl["box"], g[f"{tag}_tpl_box"])
    np.testing.assert_array_equal(tpl["histogram"][1], g[f"{tag}_tpl_hist_q"])
    if tag.startswith("f32"):
        np.testing.assert_array_equal(tpl["histogram"][0], g[f"{tag}_tpl_hist_v"].astype(np.float64))
        np.testing.assert_array_equal(tpl["tile"], g[f"{tag}_tpl_tile"].astype(np.float64))
    else:
        np.testing.assert_allclose(tpl["histogram"][0], g[f"{tag}_tpl_hist_v"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(tpl["tile"], g[f"{tag}_tpl_tile"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(tracks.means, g[f"{tag}_means"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g[f"{tag}_sigmas"], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("tag", ["wide", "tight"])
def test_tracking_with_bilinear_interpolation_reproduces_reference(golden, tag):
    """Tracker(interpolation={"kx": 1, "ky": 1}) (tracker.py:60, :585-590, :623: a degree-1 RectBivariateSpline over
    the SSD surface, the search box widened to 2 cells only) against the reference run with the same np.random seed."""
    g = golden("g21_bilinear.npz")
    scene = golden("g15_ragged.npz")
    cam = camera_from(scene["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f)
              for i, f in enumerate(scene["frames"])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], interpolation={"kx": 1, "ky": 1},
                                  max_search_dim=128)
    wide = tag == "wide"
    models = [glimpse_amd.CartesianMotion(xy=tuple(xy), time_unit=DAY, dem=0.0, dem_sigma=0.0, n=200,
                                          xy_sigma=(0.2, 0.2) if wide else (0.004, 0.004), vxyz=(0.15, 0, 0),
                                          vxyz_sigma=(0.2, 0.2, 0.0) if wide else (0.002, 0.002, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0) if wide else (0.0005, 0.0005, 0.0))
              for xy in g["xy"]]
    np.random.seed(47)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15))
    assert all(e is None for e in tracks.errors)
    np.testing.assert_allclose(tracks.means, g[f"{tag}_means"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g[f"{tag}_sigmas"], rtol=RTOL, atol=1e-8)
    with pytest.raises(NotImplementedError):
        glimpse_amd.Tracker([glimpse_amd.Observer(images)], interpolation={"kx": 6, "ky": 1})
    with pytest.raises(NotImplementedError):
        glimpse_amd.Tracker([glimpse_amd.Observer(images)], interpolation={"kx": 3, "ky": 3, "s": 0.1})


@pytest.mark.parametrize("orders", [(2, 2), (5, 5), (3, 1), (4, 2)])
@pytest.mark.parametrize("tag", ["wide", "tight"])
def test_tracking_with_other_interpolation_orders_reproduces_reference(golden, orders, tag):
    """Tracker(interpolation={"kx": .., "ky": ..}) with any orders RectBivariateSpline takes (tracker.py:60, :585-590,
    :623; kx along the rows, ky along the columns -- the orders also set the least size of the surface) against reference
    runs with the same np.random seed (g23)."""
    g = golden("g23_orders.npz")
    scene = golden("g15_ragged.npz")
    kx, ky = orders
    cam = camera_from(scene["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f)
              for i, f in enumerate(scene["frames"])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], interpolation={"kx": kx, "ky": ky})
    wide = tag == "wide"
    models = [glimpse_amd.CartesianMotion(xy=tuple(xy), time_unit=DAY, dem=0.0, dem_sigma=0.0, n=200,
                                          xy_sigma=(0.2, 0.2) if wide else (0.004, 0.004), vxyz=(0.15, 0, 0),
                                          vxyz_sigma=(0.2, 0.2, 0.0) if wide else (0.002, 0.002, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0) if wide else (0.0005, 0.0005, 0.0))
              for xy in g["xy"]]
    np.random.seed(4700 + 10 * kx + ky)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15))
    assert all(e is None for e in tracks.errors)
    np.testing.assert_allclose(tracks.means, g[f"k{kx}{ky}_{tag}_means"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g[f"k{kx}{ky}_{tag}_sigmas"], rtol=RTOL, atol=1e-8)


def test_search_workspaces_size_themselves(golden, monkeypatch):
    """Tracker(max_search_dim=None), the default: the per-point search-tile workspaces are sized from the prior's
    projected spread, and a run whose search tiles outgrow them is repeated with larger ones (same draws) -- the tracks
    never depend on the guess.  Forced here by a guess that is far too small."""
    g = golden("g8_c2mini.npz")
    tile = tuple(int(v) for v in g["tile_size"])

    def run(**kw):
        tracker = glimpse_amd.Tracker(observers_from(g), **kw)
        np.random.seed(int(g["seed"]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track(models_from(g), tile_size=tile, return_particles=True)
        return tracker, tracks

    _, explicit = run(max_search_dim=128)
    tracker, auto = run()
    assert tracker._ctx_key[-1] % 16 == 0 and tile[0] + 16 <= tracker._ctx_key[-1] <= 160, tracker._ctx_key
    monkeypatch.setattr(glimpse_amd.Tracker, "_estimate_search_dim", lambda self, *a: max(tile) + 16)
    tracker, grown = run()
    assert tracker._ctx_key[-1] > max(tile) + 16  # the first attempt overflowed and was repeated
    for tracks in (auto, grown):
        assert [e is None for e in tracks.errors] == [e is None for e in explicit.errors]
        np.testing.assert_array_equal(tracks.means, explicit.means)
        np.testing.assert_array_equal(tracks.particles, explicit.particles)
        assert all(w is None or not any("max_search_dim" in str(x) for x in w) for w in tracks.warnings)
    np.testing.assert_allclose(auto.means[:3], g["means"][:3], rtol=RTOL, atol=1e-8)


def test_reference_base_motion_model(golden):
    """glimpse.Motion, the reference's minimal model (motion.py:13-89), used as it is: it has no device twin, so it runs
    like any user-defined model -- its own NumPy methods on the host, in the reference's draw order, templates /
    likelihoods / weights / resampling / moments on the device.  Same seed, same tracks (g25)."""
    g = golden("g25_base_motion.npz")
    scene = golden("g15_ragged.npz")
    cam = camera_from(scene["cam"])
    images = [glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + i * DAY, array=f)
              for i, f in enumerate(scene["frames"])]
    tracker = glimpse_amd.Tracker([glimpse_amd.Observer(images, sigma=0.3)], max_search_dim=128)
    models = [glimpse_amd.Motion(xy=tuple(xy), time_unit=DAY, n=300, vxyz_sigma=(0.3, 0.2, 0.0)) for xy in g["xy"]]
    np.random.seed(2501)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
    assert all(e is None for e in tracks.errors)
    np.testing.assert_allclose(tracks.means, g["means"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.sigmas, g["sigmas"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.particles, g["particles"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(tracks.weights, g["weights"], rtol=RTOL, atol=1e-290)


def test_sample_tile_on_a_grid_and_with_other_orders(golden):
    """Observer.sample_tile(grid=True) and its kx / ky arguments (observer.py:178-214 passes them to
    scipy.interpolate.RectBivariateSpline, which is what this test calls directly)."""
    import scipy.interpolate

    obs = observers_from(golden("g8_c1.npz"))[0]
    rng = np.random.default_rng(11)
    tile = rng.normal(size=(13, 17)).astype(np.float32)
    box = np.array([100.0, 40.0, 117.0, 53.0])
    cu, cv = np.arange(box[0] + 0.5, box[2]), np.arange(box[1] + 0.5, box[3])
    u, v = np.sort(rng.uniform(box[0], box[2], 9)), np.sort(rng.uniform(box[1], box[3], 7))
    for kw in ({}, {"kx": 1, "ky": 1}, {"kx": 2, "ky": 4}, {"kx": 5, "ky": 5}, {"s": 0}):
        f = scipy.interpolate.RectBivariateSpline(cv, cu, tile.astype(float), **kw)
        got = obs.sample_tile((u, v), tile, box, grid=True, **kw)
        assert got.shape == (7, 9)
        np.testing.assert_allclose(got, f(v, u, grid=True), rtol=0, atol=5e-12)
        pts = np.column_stack((rng.uniform(box[0], box[2], 25), rng.uniform(box[1], box[3], 25)))
        np.testing.assert_allclose(obs.sample_tile(pts, tile, box, **kw), f(pts[:, 1], pts[:, 0], grid=False), rtol=0,
                                   atol=5e-12)
    with pytest.raises(ValueError, match="outside box"):
        obs.sample_tile((u + 100.0, v), tile, box, grid=True)
    with pytest.raises(ValueError, match="sorted"):
        obs.sample_tile((u[::-1], v), tile, box, grid=True)
    with pytest.raises(NotImplementedError):
        obs.sample_tile((u, v), tile, box, grid=True, s=2.0)
