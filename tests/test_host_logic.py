"""Host-side logic of the drop-in API that needs no GPU: datetime matching, containers, guards."""
import datetime

import numpy as np
import pytest

import glimpse_amd
from tests.helpers_api import DAY, T0, models_from, observers_from


@pytest.mark.parametrize("name", ["g8_c1.npz", "g8_c2mini.npz", "g8_c5mini.npz"])
def test_match_datetimes_equals_reference(golden, name):
    g = golden(name)
    tracker = glimpse_amd.Tracker(observers_from(g))
    datetimes = tracker.datetimes
    assert [(d - T0).total_seconds() / 86400.0 for d in datetimes] == list(g["datetimes_days"])
    m = tracker.match_datetimes(datetimes, maxdt=datetime.timedelta(days=float(g["maxdt_days"])))
    got = np.array([[-1 if v is None else int(v) for v in row] for row in m])
    np.testing.assert_array_equal(got, g["matching"])


def test_parse_datetimes_rules(golden):
    g = golden("g8_c1.npz")
    tracker = glimpse_amd.Tracker(observers_from(g))
    dts = list(tracker.datetimes)
    with pytest.raises(ValueError, match="monotonic"):
        tracker.parse_datetimes([dts[0], dts[2], dts[1]])
    with pytest.warns(UserWarning, match="duplicate"):
        out = tracker.parse_datetimes([dts[0], dts[0], dts[1]])
    assert len(out) == 2
    with pytest.warns(UserWarning, match="not matching"):
        out = tracker.parse_datetimes([dts[0], dts[1], dts[1] + datetime.timedelta(hours=5)])
    assert len(out) == 2
    with pytest.raises(ValueError, match="Fewer than two"):
        with pytest.warns(UserWarning):
            tracker.parse_datetimes([dts[0], dts[0] + datetime.timedelta(hours=5)])
    # backward (monotone decreasing) sequences are accepted (tracker.py:446-450)
    assert list(tracker.parse_datetimes(dts[::-1])) == dts[::-1]


def test_observer_validation(golden):
    g = golden("g8_c1.npz")
    obs = observers_from(g)[0]
    with pytest.raises(ValueError, match="two or greater"):
        glimpse_amd.Observer(obs.images[:1])
    with pytest.raises(ValueError, match="increasing"):
        glimpse_amd.Observer(obs.images[::-1])
    assert obs.index(obs.images[2]) == 2
    assert obs.index(obs.datetimes[3]) == 3
    with pytest.raises(ValueError, match="out of range"):
        obs.index(obs.datetimes[3] + datetime.timedelta(hours=1))
    # tile_box == Grid.snap_box (raster.py:390-421): integer edges, IndexError outside the image
    np.testing.assert_array_equal(obs.tile_box((100.4, 50.6), (15, 15), 0), [93, 43, 108, 58])
    with pytest.raises(IndexError):
        obs.tile_box((3.0, 50.0), (15, 15), 0)
    np.testing.assert_array_equal(obs.extract_tile((10, 20, 14, 23), 1), g["obs0_frames"][1][20:23, 10:14])


def test_camera_constructor_contract():
    cam = glimpse_amd.Camera(imgsz=(800, 536), sensorsz=(23.6, 15.8), fmm=20)
    np.testing.assert_allclose(cam.f, 20 * np.array([800, 536]) / np.array([23.6, 15.8]))
    np.testing.assert_allclose(cam.fmm, [20, 20])
    with pytest.raises(ValueError):
        glimpse_amd.Camera(imgsz=(10, 10), fmm=20)
    with pytest.raises(ValueError):
        glimpse_amd.Camera(imgsz=(10, 10), f=5, fmm=20, sensorsz=(1, 1))
    with pytest.raises(ValueError):
        glimpse_amd.Camera(imgsz=(10.5, 10), f=5)
    with pytest.raises(ValueError):
        glimpse_amd.Camera(imgsz=10)
    cam = glimpse_amd.Camera(imgsz=10, f=10, correction=True)
    assert cam.correction == {"radius": 6.3781e6, "refraction": 0.13}
    v = cam.vector24
    assert v[20] == 1 and v[21] == 6.3781e6 and v[22] == 0.13
    # camera.py:712-715 doctest
    cam = glimpse_amd.Camera(imgsz=(10, 12), f=10)
    assert list(cam.inframe(np.array([(-1, 1), (0, 0), (9, 11), (10, 15)]))) == [False, True, True, False]
    # (the camera.py:651-656 uv_to_xyz doctest runs on the device: tests/test_gpu_api.py)


def test_tracker_guards(golden):
    g = golden("g8_c1.npz")
    observers = observers_from(g)
    with pytest.raises(ValueError):
        glimpse_amd.Tracker(observers, resample_method="multinomial")
    for bad in ({"size": (4, 4)}, {"size": (9, 9)}, {"size": (5, 5), "mode": "constant"}, {"size": (5, 5), "origin": 1},
                {"size": (3, 3, 3)}):
        with pytest.raises(NotImplementedError):
            glimpse_amd.Tracker(observers, highpass=bad)
    assert glimpse_amd.Tracker(observers, highpass={"size": 3})._highpass_size == (3, 3)
    assert glimpse_amd.Tracker(observers, highpass={"size": (3, 7)})._highpass_size == (3, 7)
    tracker = glimpse_amd.Tracker(observers)
    models = models_from(g)
    other = glimpse_amd.CartesianMotion(xy=(0, 0), time_unit=datetime.timedelta(hours=1), dem=0, dem_sigma=0)
    with pytest.raises(ValueError, match="equal time units"):
        tracker.track(models + [other])
    with pytest.raises(NotImplementedError):
        glimpse_amd.CartesianMotion(xy=(0, 0), time_unit=DAY, dem=0, dem_sigma=None)


def test_motion_host_methods_match_reference(golden):
    g = golden("g7_motion.npz")
    for i in range(2):
        p = g[f"m{i}_params"]
        n = len(g[f"m{i}_p0"])
        model = glimpse_amd.CartesianMotion(xy=p[0:2], time_unit=DAY, dem=p[16], dem_sigma=p[17], n=n,
                                            xy_sigma=p[2:4], vxyz=p[4:7], vxyz_sigma=p[7:10], axyz=p[10:13],
                                            axyz_sigma=p[13:16])
        np.testing.assert_array_equal(model.params(), p)
        np.random.seed(700 + i)
        p0 = model.initialize_particles()
        np.testing.assert_array_equal(p0, g[f"m{i}_p0"])
        model.evolve_particles(p0, datetime.timedelta(days=1.5))
        np.testing.assert_array_equal(p0, g[f"m{i}_p1"])
        model.evolve_particles(p0, datetime.timedelta(days=-0.75))
        np.testing.assert_array_equal(p0, g[f"m{i}_p2"])
        np.testing.assert_array_equal(model.compute_log_likelihoods(p0), g[f"m{i}_ll"])


def test_tracks_container():
    means = np.full((2, 3, 6), np.nan)
    means[0, 1:] = 1.0
    tr = glimpse_amd.Tracks(datetimes=[T0, T0 + DAY, T0 + 2 * DAY], time_unit=DAY, means=means,
                            sigmas=np.ones((2, 3, 6)), errors=[None, ValueError("x")])
    assert tr.xyz.shape == (2, 3, 3) and tr.vxyz_sigma.shape == (2, 3, 3)
    valid, first, last = tr.endpoints
    assert list(valid) == [True, False] and list(first) == [1] and list(last) == [2]
    assert list(tr.success) == [True, False]


def test_tracks_merge_average_reverse_match_reference(golden):
    """Tracks.from_multiple / average / reverse (tracks.py:131-213) against the reference's outputs."""
    import datetime

    from glimpse_amd import Tracks

    g = golden("g10_tracks.npz")
    t0, day = datetime.datetime(2020, 1, 1), datetime.timedelta(days=1)
    dts = [t0 + i * day for i in range(g["run0_means"].shape[1])]
    runs = [Tracks(datetimes=dts, time_unit=day, means=g[f"run{r}_means"].copy(), sigmas=g[f"run{r}_sigmas"].copy())
            for r in range(3)]
    for flag in (0, 1):
        merged = Tracks.from_multiple(runs, ignore_nan=bool(flag))
        np.testing.assert_allclose(merged.means, g[f"merged_means_{flag}"], rtol=1e-13, equal_nan=True)
        np.testing.assert_allclose(merged.sigmas, g[f"merged_sigmas_{flag}"], rtol=1e-13, equal_nan=True)
        m, s = merged.average(ignore_nan=bool(flag))
        np.testing.assert_allclose(m, g[f"merged_avg_means_{flag}"], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(s, g[f"merged_avg_sigmas_{flag}"], rtol=1e-12, equal_nan=True)
        m, s = runs[0].average(ignore_nan=bool(flag))
        np.testing.assert_allclose(m, g[f"run0_avg_means_{flag}"], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(s, g[f"run0_avg_sigmas_{flag}"], rtol=1e-12, equal_nan=True)
    runs[0].reverse()
    np.testing.assert_array_equal(runs[0].means, g["rev_means"])
    assert [(d - t0).days for d in runs[0].datetimes] == list(g["rev_days"])
    other = Tracks(datetimes=dts[1:], time_unit=day, means=g["run1_means"][:, 1:], sigmas=g["run1_sigmas"][:, 1:])
    with pytest.raises(ValueError):
        Tracks.from_multiple([runs[1], other])


def _api_models(g, names=("cyl", "tcart", "tcyl")):
    def kw(name):
        return {k[len(name) + 4:]: g[k] for k in g if k.startswith(name + "_kw_")}

    cls = {"cyl": glimpse_amd.CylindricalMotion, "tcart": glimpse_amd.TangentCartesianMotion,
           "tcyl": glimpse_amd.TangentCylindricalMotion}
    out = {}
    for name in names:
        k = kw(name)
        k["n"] = int(k["n"])
        for s in ("dem", "dem_sigma", "slope_sigma"):
            if s in k:
                k[s] = float(k[s])
        out[name] = cls[name](time_unit=DAY, **k)
    return out


def test_other_motion_models_host_methods_match_reference(golden):
    """CylindricalMotion / TangentCartesianMotion / TangentCylindricalMotion host methods consume the
    legacy stream like the reference (motion.py:207-522)."""
    g = golden("g11_motion.npz")
    for name, model in _api_models(g).items():
        np.random.seed(900)
        p0 = model.initialize_particles()
        np.testing.assert_allclose(p0, g[f"{name}_p0"], rtol=1e-14, atol=1e-15)
        p1 = p0.copy()
        model.evolve_particles(p1, dt=datetime.timedelta(days=1.5))
        np.testing.assert_allclose(p1, g[f"{name}_p1"], rtol=1e-13, atol=1e-14)
        p2 = p1.copy()
        model.evolve_particles(p2, dt=datetime.timedelta(days=-0.75))
        np.testing.assert_allclose(p2, g[f"{name}_p2"], rtol=1e-13, atol=1e-14)
        ll = model.compute_log_likelihoods(p2)
        assert (ll is not None) == bool(g[f"{name}_has_ll"])
        assert model.params_full().shape == (24,) and model.params_full()[18] == model.KIND


def test_image_reads_files_like_arrays(tmp_path):
    """Image.read (image.py:137-214) from 8-bit PNG / TIFF / JPEG files: same samples as the decoded file, crops
    equal slices of the whole image, and the cache semantics of the reference."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    cam = glimpse_amd.Camera(imgsz=(40, 30), f=100)
    gray = rng.integers(0, 256, (30, 40), dtype=np.uint8)
    rgb = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    for name, a in (("g.png", gray), ("c.png", rgb), ("g.tif", gray), ("c.tif", rgb)):
        path = tmp_path / name
        PIL.fromarray(a).save(path)
        img = glimpse_amd.Image(path, cam=cam, datetime=T0)
        np.testing.assert_array_equal(img.read(cache=False), a)
        assert img.array is None  # cache=False leaves nothing behind (image.py:211-213)
        np.testing.assert_array_equal(img.read(box=(3, 5, 20, 17)), a[5:17, 3:20])
        assert img.array is not None and img.array.shape == a.shape
    path = tmp_path / "c.jpg"
    PIL.fromarray(rgb).save(path, quality=95)
    img = glimpse_amd.Image(path, cam=cam, datetime=T0)
    with PIL.open(path) as im:
        want = np.asarray(im)
    np.testing.assert_array_equal(img.read(), want)
    # a palette image comes back as its RGB samples
    path = tmp_path / "p.png"
    PIL.fromarray(rgb).convert("P").save(path)
    assert glimpse_amd.Image(path, cam=cam, datetime=T0).read().shape == (30, 40, 3)
    # a camera of another size than the file: the read is resized (test_image_reads_at_the_camera_size)
    assert glimpse_amd.Image(tmp_path / "g.png", cam=glimpse_amd.Camera(imgsz=(20, 15), f=100), datetime=T0).read().shape[:2] == (15, 20)
    with pytest.raises(ValueError):
        glimpse_amd.Image(cam=cam, datetime=T0)
    # an in-memory image under a resized camera: the caller's pixels stay, the resampled copy is cached beside them
    cam2 = glimpse_amd.Camera(imgsz=(40, 30), f=100)
    mem = glimpse_amd.Image(cam=cam2, datetime=T0, array=gray.copy() if "gray" in dir() else np.arange(1200, dtype=np.uint8).reshape(30, 40))
    full = mem.read().copy()
    cam2.resize(0.5)
    half = mem.read()
    assert half.shape == (15, 20) and mem.array.shape == (30, 40)
    np.testing.assert_array_equal(mem.read(), half)
    cam2.resize(1)
    np.testing.assert_array_equal(mem.read(), full)


def test_nearest_in_sorted_equals_the_distance_matrix_argmin():
    """The searchsorted matcher picks what np.argmin over the reference's pairwise distance matrix picks
    (helpers.py:1831-1854, tracker.py:479-484), ties included."""
    from glimpse_amd.tracker import nearest_in_sorted

    rng = np.random.default_rng(0)
    for _ in range(200):
        times = np.unique(rng.integers(0, 50, int(rng.integers(1, 12)))) * 1000
        queries = rng.integers(-10, 60, int(rng.integers(1, 20))) * 500  # half-steps: exact ties
        idx, dist = nearest_in_sorted(times, queries)
        d = np.abs(queries[:, None] - times[None, :])
        np.testing.assert_array_equal(idx, d.argmin(axis=1))
        np.testing.assert_array_equal(dist, d.min(axis=1))


def _schedule_observer(secs):
    blank = np.zeros((8, 8), dtype=np.uint8)
    cam = glimpse_amd.Camera(imgsz=(8, 8), f=(10, 10))
    return glimpse_amd.Observer([glimpse_amd.Image("synthetic", cam=cam, datetime=T0 + datetime.timedelta(seconds=int(s)),
                                                   array=blank) for s in secs])


def test_observer_subset_split_and_shift_equal_reference(golden):
    """Observer.subset / split / shift_tile (observer.py:146-176, 455-493) against the reference's own results on an
    irregular schedule (tools/make_golden.py --g19)."""
    from glimpse_amd.timeutil import select_datetimes
    g = golden("g19_observer_helpers.npz")
    secs = g["secs"]
    obs = _schedule_observer(secs)
    when = lambda s: T0 + datetime.timedelta(seconds=int(s))  # noqa: E731
    position = {d: i for i, d in enumerate(obs.datetimes)}
    for k, (a, b, snap, maxdt) in enumerate(g["cases"]):
        kw = {}
        if a >= 0:
            kw["start"] = when(a)
        if b >= 0:
            kw["end"] = when(b)
        if snap:
            kw["snap"] = datetime.timedelta(seconds=int(snap))
            if maxdt >= 0:
                kw["maxdt"] = datetime.timedelta(seconds=int(maxdt))
        want = g[f"mask_{k}"]
        np.testing.assert_array_equal(select_datetimes(obs.datetimes, **kw), want, err_msg=f"case {k}: {kw}")
        if want.sum() >= 2:
            sub = obs.subset(**kw)
            assert [position[d] for d in sub.datetimes] == list(np.flatnonzero(want))
            assert sub.sigma == obs.sigma and sub.cache == obs.cache
        else:
            with pytest.raises(ValueError, match="two or greater"):
                obs.subset(**kw)
    with pytest.raises(ValueError, match="after end"):
        obs.subset(start=when(secs[5]), end=when(secs[2]))

    def spans(parts):
        return [[position[p.datetimes[0]], position[p.datetimes[-1]], len(p.images)] for p in parts]

    for n, overlap in [(3, 1), (4, 0), (5, 2)]:
        assert spans(obs.split(n, overlap=overlap)) == g[f"split_{n}_{overlap}"].tolist()
    assert spans(obs.split([when(s) for s in g["breaks_secs"]])) == g["split_breaks"].tolist()

    for i, duv in enumerate(g["duv"]):
        np.testing.assert_allclose(obs.shift_tile(g["tile"].copy(), duv), g[f"shift_{i}"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(obs.shift_tile(g["rgb"].copy(), duv), g[f"shift_rgb_{i}"], rtol=0, atol=1e-13)
    with pytest.raises(ValueError, match="0.5"):
        obs.shift_tile(g["tile"].copy(), (0.6, 0.0))


def test_observer_image_cache_helpers(tmp_path):
    """Observer.cache_images / clear_images (observer.py:256-274) on images read from files."""
    import PIL.Image
    frames = [np.full((6, 7), 10 * i, dtype=np.uint8) for i in range(3)]
    cam = glimpse_amd.Camera(imgsz=(7, 6), f=(10, 10))
    images = []
    for i, f in enumerate(frames):
        path = tmp_path / f"f{i}.png"
        PIL.Image.fromarray(f).save(path)
        images.append(glimpse_amd.Image(str(path), cam=cam, datetime=T0 + i * DAY))
    obs = glimpse_amd.Observer(images)
    assert all(img.array is None for img in images)
    obs.cache_images([0, 2])
    assert images[0].array is not None and images[1].array is None and images[2].array is not None
    obs.cache_images()
    np.testing.assert_array_equal(images[1].array, frames[1])
    obs.clear_images(slice(1, None))
    assert images[0].array is not None and images[1].array is None and images[2].array is None


def test_batches_of_motion_models():
    """Which consecutive motion models one device batch can hold (glimpse_amd.tracker._batches): equal particle counts, at
    most one gridded dem and one gridded dem_sigma (constant surfaces mix freely), user-defined models alone."""
    import datetime

    import glimpse_amd as g
    from glimpse_amd.tracker import _batches
    from tests.custom_motion import DriftMotion

    day = datetime.timedelta(days=1)
    A, B = g.Raster(np.zeros((4, 4)), x=(0, 4), y=(4, 0)), g.Raster(np.ones((4, 4)), x=(0, 4), y=(4, 0))

    def cart(dem=0.0, sigma=0.1, n=100):
        return g.CartesianMotion(xy=(1, 1), time_unit=day, dem=dem, dem_sigma=sigma, n=n)

    assert _batches([cart(), cart(), cart()]) == [0]
    assert _batches([cart(n=100), cart(n=200), cart(n=200)]) == [0, 1]
    assert _batches([cart(A), cart(), cart(A), cart(B), cart(B, A), cart(sigma=B)]) == [0, 3, 5]
    assert _batches([cart(), cart(A), cart(sigma=A), cart(A, A), cart(sigma=B)]) == [0, 4]
    custom = DriftMotion.__new__(DriftMotion)
    custom.n = 100
    assert _batches([cart(), custom, cart(), cart()]) == [0, 1, 2]


def test_camera_file_format_and_resizing(tmp_path):
    """The Camera conveniences either side of the path, pinned by the reference's own doctests (camera.py:399-589,
    :665-683): reset, to_array, to_dict, to_json / from_json, idealize, resize (aspect-ratio check), infront."""
    import json

    import glimpse_amd as g

    cam = g.Camera(imgsz=1, f=1)
    cam.f[0] += 1
    cam.reset()
    assert cam.f[0] == 1
    cam = g.Camera(xyz=(1, 2, 3), viewdir=(4, 5, 6), imgsz=(7, 8), f=(9, 10), c=(11, 12), k=(13, 14, 15, 16, 17, 18), p=(19, 20))
    np.testing.assert_array_equal(cam.to_array(), np.arange(1.0, 21.0))
    cam = g.Camera(imgsz=(8, 6), f=(7.9, 6.1))
    d = cam.to_dict()
    assert d["imgsz"] == [8, 6] and d["f"] == [7.9, 6.1] and d["correction"] is False and set(d) == set(g.Camera._FIELDS)
    assert cam.to_dict(("imgsz", "f")) == {"imgsz": [8, 6], "f": [7.9, 6.1]}
    assert json.loads(cam.to_json())["k"] == [0.0] * 6
    path = tmp_path / "cam.json"
    full = g.Camera(imgsz=(80, 60), f=(79.0, 61.0), c=(1.5, -2.0), k=(0.1, 0.02), p=(0.001, 0.002), xyz=(5, 6, 7),
                    viewdir=(10, -20, 3), correction=True)
    full.to_json(path, indent=2)
    back = g.Camera.from_json(path)
    np.testing.assert_array_equal(back.to_array(), full.to_array())
    assert back.correction == full.correction
    assert g.Camera.from_json(path, f=(100, 100)).f.tolist() == [100, 100]  # (keyword arguments override the file)
    (tmp_path / "sparse.json").write_text(json.dumps({"imgsz": [8, 6], "f": None, "fmm": [20, 20], "sensorsz": [16, 12]}))
    assert g.Camera.from_json(tmp_path / "sparse.json").f.tolist() == [10.0, 10.0]   # (null entries count as absent)
    cam = g.Camera(imgsz=1, f=1, c=(0.1, 0.2), k=(0.1, 0.2), p=(0.1, 0.2))
    cam.idealize()
    assert all(cam.c == 0) and all(cam.k == 0) and all(cam.p == 0)
    cam = g.Camera(imgsz=(10, 20), f=(1, 2), c=(0.1, 0.2))
    cam.resize(2)
    assert cam.imgsz.tolist() == [20, 40] and cam.f.tolist() == [2.0, 4.0] and cam.c.tolist() == [0.2, 0.4]
    cam.resize(1)
    assert cam.imgsz.tolist() == [10, 20] and cam.f.tolist() == [1.0, 2.0]
    with pytest.raises(ValueError, match="aspect ratio"):
        cam.resize((11, 20))
    cam.resize((11, 20), force=True)
    assert cam.imgsz.tolist() == [11, 20]
    cam = g.Camera(imgsz=(100, 50), f=(10, 10))
    cam.resize((33, 17))  # (no single factor gives 33 / 100 = 17 / 50 exactly, but round(0.335 * (100, 50)) does)
    assert cam.imgsz.tolist() == [33, 17]
    cam = g.Camera(imgsz=10, f=10)
    xyz = np.array([(1000, 10, 0), (0, 10, 0), (0, 0, 0), (0, -10, 0)], dtype=float)
    assert cam.infront(xyz).tolist() == [True, True, False, False]


def test_image_reads_at_the_camera_size(tmp_path):
    """Image.read (image.py:137-214): the image is resized as needed to the camera's image size (the reference's doctest:
    cam.resize(0.5) -> a.shape == (268, 400), back to 1 -> (536, 800)), a box is a slice of that array whether or not the
    read is cached, and a cached read of another size is not reused.  Nearest neighbour like GDAL's default RasterIO
    resampling (GDAL is absent here: the sampling rule is restated, not pinned)."""
    import datetime

    from PIL import Image as PILImage

    import glimpse_amd

    rng = np.random.default_rng(5)
    full = rng.integers(0, 256, size=(536, 800, 3), dtype=np.uint8)
    path = tmp_path / "frame.png"
    PILImage.fromarray(full).save(path)
    cam = glimpse_amd.Camera(imgsz=(800, 536), f=(1000, 1000))
    img = glimpse_amd.Image(str(path), cam=cam, datetime=datetime.datetime(2020, 1, 1))
    np.testing.assert_array_equal(img.read(), full)
    img.cam.resize(0.5)
    assert tuple(img.cam.imgsz) == (400, 268)
    half = img.read()
    assert half.shape == (268, 400, 3)
    np.testing.assert_array_equal(half, full[1::2, 1::2])  # floor((i + 0.5) * 2) = 2 i + 1
    box = (0, 5, 100, 94)
    np.testing.assert_array_equal(img.read(box), half[5:94, 0:100])
    img.array = None
    np.testing.assert_array_equal(img.read(box, cache=False), half[5:94, 0:100])
    assert img.array is None
    img.read()
    img.cam.resize(1)
    back = img.read()  # (the half-size array in the cache is not what a full-size read returns)
    np.testing.assert_array_equal(back, full)
    # a size that is not a divisor
    img.cam.resize(0.3)
    small = img.read()
    w, h = (int(v) for v in img.cam.imgsz)
    assert small.shape == (h, w, 3)
    rows = np.floor((np.arange(h) + 0.5) * 536 / h + 1e-10).astype(int)
    cols = np.floor((np.arange(w) + 0.5) * 800 / w + 1e-10).astype(int)
    np.testing.assert_array_equal(small, full[rows][:, cols])
