"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in this container.

Build-container only: imports /root/reference/src (glimpse 0.1.1) under the
stub modules of tools/refstubs.py.  The fixtures hold inputs and the
reference's outputs only (no reference source); the GPU box never sees the
reference.  Re-run with:  python tools/make_golden.py

Fixtures:
  g1_projection.npz   Camera.xyz_to_uv for 5 cameras x 256 points
  g2_tiles.npz        Tracker.extract_tile / initialize_template on gray / RGB tiles
  g4_spline.npz       Observer.sample_tile on random SSE surfaces
  g5_resample.npz     Tracker.resample_particles (all four methods) + moments
  g7_motion.npz       CartesianMotion init / evolve / log-likelihoods
  g8_c1.npz           end-to-end config 1 (1 pt x 100 particles, 5 frames 512^2, pinhole)
  g8_c2mini.npz       k1-k3 distortion, 3 pts x 200 particles, 6 frames 256^2
  g8_c5mini.npz       2 observers (nadir + oblique), dem_sigma > 0, 2 pts x 200 particles
Every g8 file stores frames, cameras, motion parameters, the recorded legacy
RNG draws (in the reference's order), per-step traces (uv, box, search tile,
sse, sampled ll, weights, searchsorted indices, particles) and Tracks.means /
.sigmas.
"""
import datetime
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refstubs  # noqa: E402

glimpse = refstubs.import_reference()
from glimpse_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ref_camera(vec):
    """24-vector -> reference Camera."""
    corr = False
    if vec[20]:
        corr = {"radius": vec[21], "refraction": vec[22]}
    return glimpse.Camera(
        imgsz=vec[6:8], f=vec[8:10], c=vec[10:12], k=vec[12:18], p=vec[18:20],
        xyz=vec[0:3], viewdir=vec[3:6], correction=corr,
    )


def ref_image(frame, vec, when):
    cam = ref_camera(vec)
    img = glimpse.Image("synthetic", cam=cam, datetime=when)
    img.array = frame
    return img


# ---------------------------------------------------------------- G1 projection
def g1_projection():
    rng = np.random.default_rng(101)
    cams = [
        synth.pack_camera(imgsz=(512, 384), f=(600, 610), xyz=(1, 2, 3), viewdir=(10, -5, 2)),
        synth.pack_camera(imgsz=(2048, 2048), f=1000, k=(0.05, -0.01, 0.002), xyz=(0, 0, 100),
                          viewdir=(0, -90, 0)),
        synth.pack_camera(imgsz=(800, 536), f=(700, 690), c=(3.5, -2.25),
                          k=(0.1, -0.05, 0.01, 0.02, -0.01, 0.003), p=(0.001, -0.002),
                          xyz=(-5, 4, 20), viewdir=(30, -40, 5)),
        synth.pack_camera(imgsz=(1000, 700), f=1200, k=(0.03, 0, 0), xyz=(40, -30, 90),
                          viewdir=(-53.13, -60.9, 0), correction=True),
        synth.pack_camera(imgsz=(4288, 2848), f=(3000, 3010), c=(10, -7), k=(-0.1, 0.02, 0),
                          p=(0.0005, 0.0003), xyz=(499000.5, 6781000.25, 450.0),
                          viewdir=(120, -12, 1.5), correction={"radius": 6.3e6, "refraction": 0.2}),
        # only denominator radial terms (k4..k6), exercises the `dr /= temp` branch alone
        synth.pack_camera(imgsz=(640, 480), f=500, k=(0, 0, 0, 0.05, 0, 0.001), xyz=(0, 0, 0),
                          viewdir=(0, 0, 0)),
    ]
    xyz_all, uv_all = [], []
    for vec in cams:
        cam = ref_camera(vec)
        R = cam.R
        # points in front of the camera around its optical axis, plus some behind
        d = np.column_stack((rng.uniform(-0.6, 0.6, 256), rng.uniform(-0.6, 0.6, 256), np.ones(256)))
        depth = rng.uniform(5, 500, 256)
        xyz = vec[0:3] + (d * depth[:, None]) @ R
        xyz[250:] = vec[0:3] - (d[250:] * depth[250:, None]) @ R  # behind -> NaN
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            uv = cam.xyz_to_uv(xyz.copy())
        xyz_all.append(xyz)
        uv_all.append(uv)
    # doctest known answers (camera.py:615-620, :683-694)
    np.savez_compressed(
        os.path.join(OUT, "g1_projection.npz"),
        cams=np.stack(cams), xyz=np.stack(xyz_all), uv=np.stack(uv_all),
        R=np.stack([ref_camera(v).R for v in cams]),
    )


# ---------------------------------------------------------------- G2 tiles
def g2_tiles():
    rng = np.random.default_rng(202)
    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0)
    frames, _ = synth.make_sequence(cam, 2, seed=5)
    rgb, _ = synth.make_sequence(cam, 2, seed=6, channels=3)
    obs_imgs = [ref_image(frames[i], cam, datetime.datetime(2020, 1, 1 + i)) for i in range(2)]
    obs_rgb = [ref_image(rgb[i], cam, datetime.datetime(2020, 1, 1 + i)) for i in range(2)]
    tracker = glimpse.Tracker([glimpse.Observer(obs_imgs), glimpse.Observer(obs_rgb)])
    out = {"gray": np.stack(frames), "rgb": np.stack(rgb), "cam": cam}
    # low-entropy frame (few distinct values, many ties) to stress the CDF / median paths
    coarse = (frames[0] // 32 * 32).astype(np.uint8)
    obs_coarse = [ref_image(coarse, cam, datetime.datetime(2020, 1, 1)),
                  ref_image((frames[1] // 32 * 32).astype(np.uint8), cam, datetime.datetime(2020, 1, 2))]
    tracker.observers.append(glimpse.Observer(obs_coarse))
    out["coarse"] = np.stack([coarse, (frames[1] // 32 * 32).astype(np.uint8)])
    boxes_t = [(100, 90, 115, 105), (20, 30, 51, 61), (200, 10, 215, 31)]
    boxes_s = [(90, 80, 130, 121), (5, 12, 80, 70), (180, 0, 256, 60)]
    for o, name in enumerate(["gray", "rgb", "coarse"]):
        for b, (bt, bs) in enumerate(zip(boxes_t, boxes_s)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                tile, hist = tracker.extract_tile(obs=o, img=0, box=np.array(bt), return_histogram=True)
                search = tracker.extract_tile(obs=o, img=1, box=np.array(bs), histogram=hist)
            out[f"{name}_{b}_tbox"] = np.array(bt)
            out[f"{name}_{b}_sbox"] = np.array(bs)
            out[f"{name}_{b}_tile"] = tile
            out[f"{name}_{b}_hist_v"] = hist[0]
            out[f"{name}_{b}_hist_q"] = hist[1]
            out[f"{name}_{b}_search"] = search
    np.savez_compressed(os.path.join(OUT, "g2_tiles.npz"), **out)


# ---------------------------------------------------------------- G4 spline
def g4_spline():
    rng = np.random.default_rng(404)
    cam = synth.nadir_camera((64, 64))
    imgs = [ref_image(np.zeros((64, 64), np.uint8), cam, datetime.datetime(2020, 1, 1 + i)) for i in range(2)]
    obs = glimpse.Observer(imgs)
    out = {}
    for i, (ho, wo) in enumerate([(4, 4), (4, 7), (5, 5), (6, 9), (8, 8), (9, 12), (23, 17), (64, 51)]):
        sse = rng.random((ho, wo)).astype(np.float32)
        l, t = rng.integers(0, 1000, 2)
        duv = rng.uniform(-0.5, 0.5, 2)
        box = np.array([l + 7.5 - 0.5, t + 7.5 - 0.5, l + 7.5 - 0.5 + wo, t + 7.5 - 0.5 + ho]) + np.tile(duv, 2)
        n = 300
        uv = np.column_stack((rng.uniform(box[0], box[2], n), rng.uniform(box[1], box[3], n)))
        # exact edges and corners too (clamped evaluation)
        uv[0] = box[0:2]
        uv[1] = box[2:4]
        uv[2] = (box[0], box[3])
        out[f"s{i}_sse"] = sse
        out[f"s{i}_box"] = box
        out[f"s{i}_uv"] = uv
        out[f"s{i}_val"] = obs.sample_tile(uv, tile=sse, box=box, grid=False, kx=3, ky=3)
    np.savez_compressed(os.path.join(OUT, "g4_spline.npz"), **out)


# ---------------------------------------------------------------- G5 resample / moments
def g5_resample():
    cam = synth.nadir_camera((64, 64))
    imgs = [ref_image(np.zeros((64, 64), np.uint8), cam, datetime.datetime(2020, 1, 1 + i)) for i in range(2)]
    tracker = glimpse.Tracker([glimpse.Observer(imgs)])
    rng = np.random.default_rng(505)
    out = {}
    real_searchsorted = np.searchsorted
    for i, n in enumerate([1, 7, 100, 129, 1000, 2000, 5000, 10000]):
        particles = rng.standard_normal((n, 6)) * [1, 1, 0.1, 0.2, 0.2, 0.01] + [5e5, 6.7e6, 400, 0.1, 0, 0]
        ll = rng.random(n) * rng.choice([1, 5, 40], n)
        weights = np.exp(-ll) + 1e-300
        big = n > 1000  # keep the fixture small: large cases store weights + indices only
        if not big:
            out[f"r{i}_particles"] = particles
        out[f"r{i}_weights"] = weights
        for method in ("systematic", "stratified", "residual"):
            tracker.particles = particles.copy()
            tracker.weights = weights.copy()
            np.random.seed(1000 + i)
            state = np.random.get_state()
            log = []

            def spy(a, v, *args, **kw):
                r = real_searchsorted(a, v, *args, **kw)
                log.append(np.array(r))
                return r

            np.searchsorted = spy
            try:
                tracker.resample_particles(method=method)
            finally:
                np.searchsorted = real_searchsorted
            np.random.set_state(state)
            if method == "systematic":
                out[f"r{i}_u"] = np.random.random()
                out[f"r{i}_idx"] = log[0]
                if not big:
                    out[f"r{i}_out_particles"] = tracker.particles.copy()
                    out[f"r{i}_out_weights"] = tracker.weights.copy()
                out[f"r{i}_mean"] = tracker.particle_mean
                out[f"r{i}_sigma"] = tracker.compute_particle_sigma()
                out[f"r{i}_cov"] = np.atleast_2d(tracker.particle_covariance) if n > 1 else np.zeros((6, 6))
            elif method == "stratified":
                if not big:
                    out[f"r{i}_strat_u"] = np.random.random(n)
                    out[f"r{i}_strat_idx"] = log[0]
            else:
                out[f"r{i}_resid_seed"] = 1000 + i
                if not big:
                    out[f"r{i}_resid_out_particles"] = tracker.particles.copy()
    np.savez_compressed(os.path.join(OUT, "g5_resample.npz"), **out)


# ---------------------------------------------------------------- G7 motion
def g7_motion():
    out = {}
    day = datetime.timedelta(days=1)
    configs = [
        dict(xy=(10.0, -4.0), dem=0.0, dem_sigma=0.0, n=300, xy_sigma=(0.2, 0.3), vxyz=(0.15, 0, 0),
             vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0)),
        dict(xy=(499000.5, 6781000.25), dem=450.0, dem_sigma=0.5, n=257, xy_sigma=(0.5, 0.5),
             vxyz=(1.0, -2.0, 0.1), vxyz_sigma=(0.3, 0.2, 0.05), axyz=(0.01, 0.02, 0.0),
             axyz_sigma=(0.05, 0.05, 0.01)),
    ]
    for i, kw in enumerate(configs):
        model = glimpse.CartesianMotion(time_unit=day, **kw)
        np.random.seed(700 + i)
        state = np.random.get_state()
        p0 = model.initialize_particles()
        p1 = p0.copy()
        model.evolve_particles(p1, dt=datetime.timedelta(days=1.5))
        p2 = p1.copy()
        model.evolve_particles(p2, dt=datetime.timedelta(days=-0.75))
        ll = model.compute_log_likelihoods(p2)
        np.random.set_state(state)
        n = kw["n"]
        out[f"m{i}_params"] = np.concatenate([kw["xy"], kw["xy_sigma"], kw["vxyz"], kw["vxyz_sigma"],
                                              kw["axyz"], kw["axyz_sigma"], [kw["dem"], kw["dem_sigma"]]])
        out[f"m{i}_init_normals"] = np.column_stack(
            (np.random.randn(n, 2), np.random.randn(n), np.random.randn(n, 3)))
        out[f"m{i}_evolve_normals"] = np.stack([np.random.randn(n, 3), np.random.randn(n, 3)])
        out[f"m{i}_taus"] = np.array([1.5, -0.75])
        out[f"m{i}_p0"], out[f"m{i}_p1"], out[f"m{i}_p2"], out[f"m{i}_ll"] = p0, p1, p2, ll
    np.savez_compressed(os.path.join(OUT, "g7_motion.npz"), **out)


# ---------------------------------------------------------------- G8 end to end
class Recorder:
    """Hooks the reference Tracker to log draws and per-step intermediates."""

    def __init__(self, tracker):
        self.tracker = tracker
        self.steps = []
        self.randn = []
        self.random = []

    def __enter__(self):
        t = self.tracker
        self._randn, self._random = np.random.randn, np.random.random
        self._searchsorted = np.searchsorted
        rec = self

        def randn(*shape):
            r = rec._randn(*shape)
            rec.randn.append(np.array(r))
            return r

        def random(*a):
            r = rec._random(*a)
            rec.random.append(np.array(r))
            return r

        def searchsorted(a, v, *args, **kw):
            r = rec._searchsorted(a, v, *args, **kw)
            if rec.steps and isinstance(rec.steps[-1], dict) and "weights" in rec.steps[-1] \
                    and "idx" not in rec.steps[-1] and np.ndim(v) == 1 and len(v) == len(a) \
                    and len(a) == len(t.particles):
                rec.steps[-1]["idx"] = np.array(r)
            return r

        np.random.randn, np.random.random, np.searchsorted = randn, random, searchsorted
        self._ll = t.compute_observer_log_likelihoods
        self._update = t.update_weights
        self._resample = t.resample_particles
        self._init_t = t.initialize_template
        self._sample = [o.sample_tile for o in t.observers]
        self._cv2 = sys.modules["cv2"].matchTemplate

        def update_weights(imgs, motion_model=None):
            rec.steps.append({"evolved": t.particles.copy(), "imgs": [(-1 if i is None else int(i)) for i in imgs],
                              "obs": {}})
            rec._update(imgs=imgs, motion_model=motion_model)
            rec.steps[-1]["weights"] = t.weights.copy()

        def compute_ll(obs, img):
            cur = {}
            rec.steps[-1]["obs"][int(obs)] = cur
            rec._cur = cur
            if img is not None:
                cur["uv"] = t.observers[obs].xyz_to_uv(t.particles[:, 0:3], img=img)
            r = rec._ll(obs, img)
            cur["ll"] = None if r is None else np.array(r)
            return r

        def matchTemplate(image, templ, method=0):
            r = rec._cv2(image, templ, method)
            rec._cur["search_f32"] = np.array(image)
            rec._cur["sse_raw"] = np.array(r)
            return r

        def make_sample(o, orig):
            def sample_tile(uv, tile, box, grid=False, **kw):
                r = orig(uv, tile=tile, box=box, grid=grid, **kw)
                rec._cur["sse"] = np.array(tile)
                rec._cur["sse_box"] = np.array(box)
                rec._cur["sampled"] = np.array(r)
                return r
            return sample_tile

        def resample_particles(method=None):
            rec._resample(method)
            rec.steps[-1]["particles"] = t.particles.copy()
            rec.steps[-1]["out_weights"] = t.weights.copy()

        def initialize_template(obs, img, tile_size):
            rec._init_t(obs=obs, img=img, tile_size=tile_size)
            rec.templates.setdefault(len(rec.track_starts) - 1, {})[int(obs)] = {
                k: np.array(v) if k != "histogram" else (np.array(v[0]), np.array(v[1]))
                for k, v in t.templates[obs].items()
            }

        self.templates = {}
        self.track_starts = []
        t.update_weights = update_weights
        t.compute_observer_log_likelihoods = compute_ll
        t.resample_particles = resample_particles
        t.initialize_template = initialize_template
        sys.modules["cv2"].matchTemplate = matchTemplate
        for o, orig in zip(t.observers, self._sample):
            o.sample_tile = make_sample(o, orig)
        self._init_w = t.initialize_weights

        def initialize_weights():
            rec.track_starts.append(len(rec.steps))
            rec._init_w()

        t.initialize_weights = initialize_weights
        return self

    def __exit__(self, *a):
        np.random.randn, np.random.random, np.searchsorted = self._randn, self._random, self._searchsorted
        sys.modules["cv2"].matchTemplate = self._cv2
        return False


def run_e2e(name, cams_per_obs, frames_per_obs, sigmas, models_kw, tile_size, seed, maxdt_days=0.0,
            obs_day_offsets=None, observer_mask=None, return_covariances=False, resample_method="systematic"):
    """Run reference Tracker.track and save everything needed to replay it."""
    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    observers = []
    nobs = len(frames_per_obs)
    for o in range(nobs):
        offs = obs_day_offsets[o] if obs_day_offsets is not None else np.arange(len(frames_per_obs[o]))
        imgs = [ref_image(frames_per_obs[o][i], cams_per_obs[o][i], t0 + float(offs[i]) * day)
                for i in range(len(frames_per_obs[o]))]
        observers.append(glimpse.Observer(imgs, sigma=sigmas[o]))
    tracker = glimpse.Tracker(observers, resample_method=resample_method)
    models = [glimpse.CartesianMotion(time_unit=day, **kw) for kw in models_kw]
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with Recorder(tracker) as rec:
            tracks = tracker.track(
                models, maxdt=datetime.timedelta(days=maxdt_days), tile_size=tile_size,
                return_particles=True, observer_mask=observer_mask,
                return_covariances=return_covariances,
            )
    out = {
        "seed": seed,
        "tile_size": np.array(tile_size),
        "n_obs": nobs,
        "sigmas": np.array(sigmas, dtype=float),
        "means": tracks.means,
        "out_sigmas": tracks.covariances if return_covariances else tracks.sigmas,
        "out_particles": tracks.particles,
        "out_weights": tracks.weights,
        "matching": np.array([[(-1 if v is None else int(v)) for v in row] for row in tracks.images]),
        "datetimes_days": np.array([(d - t0).total_seconds() / 86400.0 for d in tracks.datetimes]),
        "errors": np.array([0 if e is None else 1 for e in tracks.errors]),
        "params": np.stack([np.concatenate([kw["xy"], kw["xy_sigma"], kw["vxyz"], kw["vxyz_sigma"],
                                             kw["axyz"], kw["axyz_sigma"], [kw["dem"], kw["dem_sigma"]]])
                            for kw in models_kw]),
        "n_particles": np.array([kw["n"] for kw in models_kw]),
        "maxdt_days": maxdt_days,
    }
    if observer_mask is not None:
        out["observer_mask"] = np.asarray(observer_mask)
    for o in range(nobs):
        out[f"obs{o}_frames"] = np.stack(frames_per_obs[o])
        out[f"obs{o}_cams"] = np.stack(cams_per_obs[o])
        offs = obs_day_offsets[o] if obs_day_offsets is not None else np.arange(len(frames_per_obs[o]))
        out[f"obs{o}_days"] = np.asarray(offs, dtype=float)
    # RNG draws in call order
    out["n_randn"] = len(rec.randn)
    for i, r in enumerate(rec.randn):
        out[f"randn{i}"] = r
    if all(np.size(r) == 1 for r in rec.random):
        out["random"] = np.array([float(r) for r in rec.random])
    elif len({np.size(r) for r in rec.random}) == 1:  # stratified resampling draws random(n) per step
        out["random_n"] = np.stack([np.asarray(r, dtype=float) for r in rec.random])
    else:  # residual resampling draws n - sum(repetitions) uniforms per step
        out["random_counts"] = np.array([np.size(r) for r in rec.random])
        out["random_flat"] = np.concatenate([np.ravel(np.asarray(r, dtype=float)) for r in rec.random])
    out["resample_method"] = resample_method
    # per-step traces; steps are grouped per track in order
    out["track_starts"] = np.array(rec.track_starts)
    out["n_steps"] = len(rec.steps)
    for s, st in enumerate(rec.steps):
        out[f"s{s}_evolved"] = st["evolved"]
        out[f"s{s}_imgs"] = np.array(st["imgs"])
        out[f"s{s}_weights"] = st["weights"]
        if "idx" in st:
            out[f"s{s}_idx"] = st["idx"]
        out[f"s{s}_particles"] = st["particles"]
        out[f"s{s}_out_weights"] = st["out_weights"]
        for o, cur in st["obs"].items():
            for k in ("uv", "ll", "search_f32", "sse_raw", "sse", "sse_box", "sampled"):
                if k in cur and cur[k] is not None:
                    out[f"s{s}_o{o}_{k}"] = cur[k]
    for tr, d in rec.templates.items():
        for o, tpl in d.items():
            out[f"t{tr}_o{o}_box"] = tpl["box"]
            out[f"t{tr}_o{o}_duv"] = tpl["duv"]
            out[f"t{tr}_o{o}_tile"] = tpl["tile"]
            out[f"t{tr}_o{o}_hist_v"] = tpl["histogram"][0]
            out[f"t{tr}_o{o}_hist_q"] = tpl["histogram"][1]
            out[f"t{tr}_o{o}_img"] = tpl["img"]
    np.savez_compressed(os.path.join(OUT, name), **out)
    return tracks


def g8_c1():
    cam = synth.nadir_camera((512, 512), f=1000.0, height=100.0)
    frames, _ = synth.make_sequence(cam, 5, seed=11, velocity=(0.15, 0.0))
    kw = dict(xy=(0.0, 0.0), dem=0.0, dem_sigma=0.0, n=100, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0),
              vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0))
    tr = run_e2e("g8_c1.npz", [[cam] * 5], [frames], [0.3], [kw], (15, 15), seed=42)
    print("c1 vx:", tr.means[0, :, 3])


def g8_c2mini():
    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    frames, _ = synth.make_sequence(cam, 6, seed=12, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam, 3, border_px=70.0, seed=3)
    kws = [dict(xy=tuple(p), dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0),
                vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0)) for p in pts]
    # a 4th track that starts outside the image -> IndexError captured, NaN rows
    kws.append(dict(kws[0], xy=(100.0, 100.0)))
    tr = run_e2e("g8_c2mini.npz", [[cam] * 6], [frames], [0.3], kws, (15, 15), seed=43)
    print("c2mini vx:", tr.means[:, -1, 3], "errors:", [e is not None for e in tr.errors])


def g8_c5mini():
    cam0 = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    cam1 = synth.pack_camera(imgsz=(256, 256), f=1200.0, k=(0.03, 0, 0), xyz=(40, -30, 90),
                             viewdir=(-53.13, -60.9, 0))
    scene = synth.default_scene(cam1, seed=13, velocity=(0.15, 0.0), n_frames=8, margin=20.0)
    f0 = [scene.render(cam0, float(t)) for t in range(6)]
    # second observer starts one frame late and has RGB frames
    days1 = np.arange(1, 6)
    f1 = [scene.render(cam1, float(t), channels=3) for t in days1]
    kws = [dict(xy=xy, dem=0.0, dem_sigma=0.5, n=200, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0),
                vxyz_sigma=(0.2, 0.2, 0.05), axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.01))
           for xy in [(0.5, -0.3), (-2.0, 1.5)]]
    tr = run_e2e("g8_c5mini.npz", [[cam0] * 6, [cam1] * 5], [f0, f1], [0.3, 0.4], kws, (15, 15), seed=44,
                 obs_day_offsets=[np.arange(6), days1])
    print("c5mini v:", tr.means[:, -1, 3:6], "sz:", tr.sigmas[:, -1, 2])


def g9_variants():
    """Small end-to-end runs of the API variants that the g8 files do not cover: covariance output
    (return_covariances=True) and the stratified / choice resampling methods."""
    cam = synth.nadir_camera((192, 192), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    frames, _ = synth.make_sequence(cam, 4, seed=21, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam, 2, border_px=70.0, seed=4)
    kws = [dict(xy=tuple(p), dem=0.0, dem_sigma=0.3, n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0),
                vxyz_sigma=(0.2, 0.2, 0.02), axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.01)) for p in pts]
    args = ([[cam] * 4], [frames], [0.3], kws, (15, 15))
    tr = run_e2e("g9_cov.npz", *args, seed=51, return_covariances=True)
    print("g9_cov:", tr.covariances.shape)
    tr = run_e2e("g9_stratified.npz", *args, seed=52, resample_method="stratified")
    print("g9_stratified vx:", tr.means[:, -1, 3])
    tr = run_e2e("g9_choice.npz", *args, seed=53, resample_method="choice")
    print("g9_choice vx:", tr.means[:, -1, 3])


def g11_motion_models():
    """Cylindrical / TangentCartesian / TangentCylindrical motion (motion.py:207-522) with scalar
    surfaces: unit vectors (init, two evolves) with the draws recorded in call order, and one small
    end-to-end track per model."""
    out = {}
    day = datetime.timedelta(days=1)
    unit = [
        ("cyl", glimpse.CylindricalMotion, dict(xy=(10.0, -4.0), dem=3.0, dem_sigma=0.4, n=211, xy_sigma=(0.2, 0.3),
                                               vrthz=(0.5, 0.3, 0.02), vrthz_sigma=(0.1, 0.2, 0.01),
                                               arthz=(0.01, 0.02, 0.0), arthz_sigma=(0.05, 0.1, 0.01))),
        ("tcart", glimpse.TangentCartesianMotion, dict(xy=(499000.5, 6781000.25), dem=450.0, dem_sigma=0.5, n=190,
                                                      xy_sigma=(0.5, 0.5), vxy=(1.0, -2.0), vxy_sigma=(0.3, 0.2),
                                                      axy=(0.01, 0.02), axy_sigma=(0.05, 0.05), slope_sigma=0.1)),
        ("tcyl", glimpse.TangentCylindricalMotion, dict(xy=(-3.0, 8.0), dem=0.0, dem_sigma=0.2, n=130,
                                                       xy_sigma=(0.2, 0.2), vrth=(0.4, -1.0), vrth_sigma=(0.1, 0.3),
                                                       arth=(0.0, 0.01), arth_sigma=(0.05, 0.05), slope_sigma=0.05)),
    ]
    real_randn = np.random.randn
    for name, cls, kw in unit:
        model = cls(time_unit=day, **kw)
        log = []

        def spy(*shape):
            r = real_randn(*shape)
            log.append(np.array(r))
            return r

        np.random.seed(900)
        np.random.randn = spy
        try:
            p0 = model.initialize_particles()
            n_init = len(log)
            p1 = p0.copy()
            model.evolve_particles(p1, dt=datetime.timedelta(days=1.5))
            p2 = p1.copy()
            model.evolve_particles(p2, dt=datetime.timedelta(days=-0.75))
        finally:
            np.random.randn = real_randn
        ll = model.compute_log_likelihoods(p2)
        out[f"{name}_p0"], out[f"{name}_p1"], out[f"{name}_p2"] = p0, p1, p2
        out[f"{name}_has_ll"] = ll is not None
        if ll is not None:
            out[f"{name}_ll"] = ll
        out[f"{name}_n_init_draws"] = n_init
        for i, r in enumerate(log):
            out[f"{name}_draw{i}"] = r
        out[f"{name}_n_draws"] = len(log)
        for k, v in kw.items():
            out[f"{name}_kw_{k}"] = np.asarray(v, dtype=float)
    np.savez_compressed(os.path.join(OUT, "g11_motion.npz"), **out)
    # end to end
    cam = synth.nadir_camera((192, 192), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    frames, _ = synth.make_sequence(cam, 4, seed=21, velocity=(0.15, 0.0))
    t0 = datetime.datetime(2020, 1, 1)
    imgs = [ref_image(frames[i], cam, t0 + i * day) for i in range(4)]
    e2e = {"frames": np.stack(frames), "cam": cam}
    models = {
        "cyl": glimpse.CylindricalMotion(xy=(0.5, -0.5), time_unit=day, dem=0.0, dem_sigma=0.3, n=150,
                                         xy_sigma=(0.2, 0.2), vrthz=(0.15, 0.0, 0.0), vrthz_sigma=(0.1, 0.5, 0.02),
                                         arthz=(0, 0, 0), arthz_sigma=(0.03, 0.2, 0.01)),
        "tcart": glimpse.TangentCartesianMotion(xy=(-1.0, 1.0), time_unit=day, dem=0.0, dem_sigma=0.2, n=150,
                                                xy_sigma=(0.2, 0.2), vxy=(0.15, 0.0), vxy_sigma=(0.2, 0.2),
                                                axy=(0, 0), axy_sigma=(0.05, 0.05), slope_sigma=0.1),
        "tcyl": glimpse.TangentCylindricalMotion(xy=(1.5, 0.5), time_unit=day, dem=0.0, dem_sigma=0.2, n=150,
                                                 xy_sigma=(0.2, 0.2), vrth=(0.15, 0.0), vrth_sigma=(0.1, 0.5),
                                                 arth=(0, 0), arth_sigma=(0.03, 0.2), slope_sigma=0.1),
    }
    for name, model in models.items():
        tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)])
        np.random.seed(910)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track([model], tile_size=(15, 15), return_particles=True)
        e2e[f"{name}_means"], e2e[f"{name}_sigmas"] = tracks.means, tracks.sigmas
        e2e[f"{name}_particles"], e2e[f"{name}_weights"] = tracks.particles, tracks.weights
    np.savez_compressed(os.path.join(OUT, "g11_motion_e2e.npz"), **e2e)
    print("g11 e2e vx:", {k: e2e[f"{k}_means"][0, -1, 3] for k in models})


def _dem_field(nx, ny, xlim, ylim, seed):
    """A gentle sloping surface with bumps on the cell centres of a grid."""
    rng = np.random.default_rng(seed)
    r = glimpse.Raster(np.zeros((ny, nx)), x=xlim, y=ylim)
    X, Y = r.X, r.Y
    return 0.05 * X - 0.03 * Y + 0.2 * np.sin(0.7 * X) * np.cos(0.5 * Y) + 0.02 * rng.standard_normal((ny, nx))


def g12_rasters():
    """Gridded surfaces (raster.py:613-1027): Raster.sample unit vectors (orders 0 and 1, decreasing y,
    decreasing x, half-cell border extrapolation, out-of-bounds mask) and end-to-end tracks with a
    gridded dem / dem_sigma, a tangent model on a gridded dem, a viewshed and surfaces that do not
    cover every track."""
    rng = np.random.default_rng(1212)
    out = {}
    specs = [((25, 21), (-12.0, 12.0), (10.0, -10.0)),      # north-up: y decreases along rows
             ((8, 5), (100.0, 108.0), (50.0, 44.0)),
             ((6, 9), (3.0, -3.0), (0.0, 4.5))]             # x decreases along columns
    for i, ((nx, ny), xlim, ylim) in enumerate(specs):
        Z = _dem_field(nx, ny, xlim, ylim, 40 + i)
        r = glimpse.Raster(Z, x=xlim, y=ylim)
        lo, hi = r.min, r.max
        xy = lo + (hi - lo) * rng.random((300, 2))
        xy[:8] = [lo, hi, (lo[0], hi[1]), (hi[0], lo[1]), (lo + hi) / 2, r.xy[0] if hasattr(r, "xy") else (lo + hi) / 2,
                  (r.x[0], r.y[0]), (r.x[-1], r.y[-1])]
        xy[8:12] = [(r.x[1], r.y[2]), (r.x[2], 0.5 * (r.y[1] + r.y[2])), (0.5 * (r.x[0] + r.x[1]), r.y[1]),
                    (r.x[0], r.y[-1])]
        out[f"r{i}_z"], out[f"r{i}_xlim"], out[f"r{i}_ylim"] = Z, np.array(xlim), np.array(ylim)
        out[f"r{i}_x"], out[f"r{i}_y"], out[f"r{i}_d"] = r.x, r.y, r.d
        out[f"r{i}_xy"] = xy
        out[f"r{i}_linear"] = r.sample(xy)
        out[f"r{i}_nearest"] = r.sample(xy, order=0)
        far = np.vstack((xy[:20] + (hi - lo) * [1.2, 0], xy[20:40] - (hi - lo) * [0, 1.1], xy[40:60]))
        out[f"r{i}_mixed_xy"] = far
        out[f"r{i}_mixed_in"] = r.inbounds_xy(far)
    np.savez_compressed(os.path.join(OUT, "g12_raster.npz"), **out)

    # ---- end to end
    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    cam = synth.nadir_camera((192, 192), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    frames, _ = synth.make_sequence(cam, 4, seed=21, velocity=(0.15, 0.0))
    imgs = [ref_image(frames[i], cam, t0 + i * day) for i in range(4)]
    xlim, ylim = (-6.0, 6.0), (6.0, -6.0)
    Zd = 0.1 * _dem_field(24, 24, xlim, ylim, 7)
    Zs = 0.2 + 0.05 * np.abs(_dem_field(24, 24, xlim, ylim, 8))
    dem, dem_sigma = glimpse.Raster(Zd, x=xlim, y=ylim), glimpse.Raster(Zs, x=xlim, y=ylim)
    vis = np.ones((12, 12))
    vis[:, 9:] = 0  # the eastern quarter of the scene is not visible
    viewshed = glimpse.Raster(vis, x=xlim, y=ylim)
    e = {"frames": np.stack(frames), "cam": cam, "dem": Zd, "dem_sigma": Zs, "viewshed": vis,
         "xlim": np.array(xlim), "ylim": np.array(ylim)}

    def run(name, models, seed, **tracker_kw):
        tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)], **tracker_kw)
        np.random.seed(seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
        e[f"{name}_means"], e[f"{name}_sigmas"] = tracks.means, tracks.sigmas
        e[f"{name}_particles"], e[f"{name}_weights"] = tracks.particles, tracks.weights
        e[f"{name}_errors"] = np.array([0 if x is None else 1 for x in tracks.errors])
        e[f"{name}_error_types"] = np.array(["" if x is None else type(x).__name__ for x in tracks.errors])
        print(name, "errors", e[f"{name}_errors"], "vx", tracks.means[:, -1, 3])

    cart = dict(time_unit=day, n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.02),
                axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.01))
    # A: gridded dem + gridded dem_sigma; the third track starts outside the rasters (ValueError at init)
    run("cart", [glimpse.CartesianMotion(xy=xy, dem=dem, dem_sigma=dem_sigma, **cart)
                 for xy in [(0.5, -0.5), (-2.0, 1.5), (7.5, 0.0)]], 1301)
    # B: gridded dem, scalar dem_sigma, tangent model
    run("tcart", [glimpse.TangentCartesianMotion(xy=xy, time_unit=day, dem=dem, dem_sigma=0.2, n=150,
                                                 xy_sigma=(0.2, 0.2), vxy=(0.15, 0.0), vxy_sigma=(0.2, 0.2),
                                                 axy=(0, 0), axy_sigma=(0.05, 0.05), slope_sigma=0.1)
                  for xy in [(-1.0, 1.0), (1.5, 0.5)]], 1302)
    # C: viewshed: the second track sits in the non-visible strip (ValueError from test_particles)
    run("view", [glimpse.CartesianMotion(xy=xy, dem=0.0, dem_sigma=0.3, **cart) for xy in [(0.5, -0.5), (3.5, 1.0)]],
        1303, viewshed=viewshed)
    np.savez_compressed(os.path.join(OUT, "g12_raster_e2e.npz"), **e)


def g22_variants():
    """Two inputs the reference accepts that need their own batching on the device: (a) resample_method='residual'
    with the legacy np.random stream for several tracks (tracker.py:188-203: every step of every track draws
    n - sum(repetitions) uniforms, so the stream position of a track depends on the weights of the tracks before it);
    (b) motion models that each carry their OWN dem / dem_sigma rasters (motion.py:136-141)."""
    cam = synth.nadir_camera((192, 192), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    frames, _ = synth.make_sequence(cam, 5, seed=21, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam, 3, border_px=70.0, seed=4)
    kws = [dict(xy=tuple(p), dem=0.0, dem_sigma=0.3, n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0),
                vxyz_sigma=(0.2, 0.2, 0.02), axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.01)) for p in pts]
    tr = run_e2e("g22_residual.npz", [[cam] * 5], [frames], [0.3], kws, (15, 15), seed=2201, resample_method="residual")
    print("g22_residual vx:", tr.means[:, -1, 3])

    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    imgs = [ref_image(frames[i], cam, t0 + i * day) for i in range(5)]
    xlim, ylim = (-6.0, 6.0), (6.0, -6.0)
    Za, Zb = 0.1 * _dem_field(24, 24, xlim, ylim, 17), 0.3 + 0.1 * _dem_field(16, 20, xlim, ylim, 18)
    Sa, Sb = 0.2 + 0.05 * np.abs(_dem_field(24, 24, xlim, ylim, 19)), 0.4 + 0.02 * np.abs(_dem_field(12, 12, xlim, ylim, 20))
    R = lambda Z: glimpse.Raster(Z, x=xlim, y=ylim)  # noqa: E731
    cart = dict(time_unit=day, n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.02),
                axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.01))
    # five tracks: rasters A, A, B (both surfaces differ), scalar surfaces, A again
    spec = [((0.5, -0.5), "a"), ((-2.0, 1.5), "a"), ((1.0, 1.0), "b"), ((-1.0, -1.5), "s"), ((2.0, -1.0), "a")]
    dem = {"a": R(Za), "b": R(Zb)}
    sig = {"a": R(Sa), "b": R(Sb)}
    models = [glimpse.CartesianMotion(xy=xy, dem=0.1 if k == "s" else dem[k], dem_sigma=0.25 if k == "s" else sig[k], **cart)
              for xy, k in spec]
    tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)])
    np.random.seed(2202)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
    assert all(e is None for e in tracks.errors)
    e = {"frames": np.stack(frames), "cam": cam, "xlim": np.array(xlim), "ylim": np.array(ylim), "dem_a": Za, "dem_b": Zb,
         "sigma_a": Sa, "sigma_b": Sb, "xy": np.array([xy for xy, _ in spec]), "kinds": np.array([k for _, k in spec]),
         "seed": 2202, "means": tracks.means, "sigmas": tracks.sigmas, "particles": tracks.particles,
         "weights": tracks.weights}
    print("g22_rasters vx:", tracks.means[:, -1, 3])
    np.savez_compressed(os.path.join(OUT, "g22_rasters.npz"), **e)


def g13_ortho():
    """Raster images as Observer images (orthophoto tracking; observer.py:26, Grid.xyz_to_uv raster.py:423-445):
    projection unit vectors and a two-track run (one RGB orthophoto sequence, y decreasing along rows)."""
    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    cam = synth.nadir_camera((192, 192), f=1000.0, height=100.0)
    frames, _ = synth.make_sequence(cam, 4, seed=21, velocity=(0.15, 0.0), channels=1)
    xlim, ylim = (-9.6, 9.6), (9.6, -9.6)  # the nadir pinhole frame is an orthophoto at 10 px per unit
    rasters = [glimpse.Raster(frames[i], x=xlim, y=ylim, datetime=t0 + i * day) for i in range(4)]
    rng = np.random.default_rng(1313)
    xyz = np.column_stack((rng.uniform(-12, 12, 64), rng.uniform(-12, 12, 64), rng.uniform(-1, 1, 64)))
    out = {"frames": np.stack(frames), "xlim": np.array(xlim), "ylim": np.array(ylim), "xyz": xyz,
           "uv": rasters[0].xyz_to_uv(xyz)}
    odd = glimpse.Raster(np.zeros((7, 5)), x=(100.0, 90.0), y=(3.0, 17.0))  # x decreases along columns
    out["odd_xlim"], out["odd_ylim"], out["odd_uv"] = np.array((100.0, 90.0)), np.array((3.0, 17.0)), odd.xyz_to_uv(xyz + [95, 10, 0])
    models = [glimpse.CartesianMotion(xy=xy, time_unit=day, dem=0.0, dem_sigma=0.0, n=150, xy_sigma=(0.2, 0.2),
                                      vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0), axyz=(0, 0, 0),
                                      axyz_sigma=(0.05, 0.05, 0)) for xy in [(0.5, -0.5), (-2.0, 1.5)]]
    tracker = glimpse.Tracker([glimpse.Observer(rasters, sigma=0.3)])
    np.random.seed(1314)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
    out["means"], out["sigmas"] = tracks.means, tracks.sigmas
    out["particles"], out["weights"] = tracks.particles, tracks.weights
    print("g13 vx:", tracks.means[:, -1, 3], "errors", tracks.errors)
    np.savez_compressed(os.path.join(OUT, "g13_ortho.npz"), **out)


def g14_unproject():
    """Camera.uv_to_xyz (camera.py:630-663) for the g1 cameras (without their elevation correction:
    it is not inverted by uv_to_xyz) plus the distortion cases of tests/test_camera.py:42-88."""
    rng = np.random.default_rng(1414)
    g1 = np.load(os.path.join(OUT, "g1_projection.npz"))
    cams = [c.copy() for c in g1["cams"]]
    for k in ([0.1], [-0.1], [0.1] * 6, [0, 0, 0, 0, 0, 0], [0.1] * 6, [2.0], [-2.0]):
        cams.append(synth.pack_camera(imgsz=(100, 100), f=(100, 100), k=k + [0] * (6 - len(k))))
    cams[-4][18:20] = 0.01  # tangential only
    cams[-3][18:20] = 0.01  # all distortion
    out = {"cams": np.stack(cams)}
    uv_all, d_all, xyz_dir, xyz_abs = [], [], [], []
    for vec in cams:
        cam = ref_camera(vec)
        uv = rng.uniform(0, 1, (200, 2)) * vec[6:8]
        depth = rng.uniform(2, 300, 200)
        uv_all.append(uv)
        d_all.append(depth)
        xyz_dir.append(cam.uv_to_xyz(uv))
        xyz_abs.append(cam.uv_to_xyz(uv, directions=False, depth=depth))
    out["uv"], out["depth"] = np.stack(uv_all), np.stack(d_all)
    out["xyz_directions"], out["xyz_absolute"] = np.stack(xyz_dir), np.stack(xyz_abs)
    np.savez_compressed(os.path.join(OUT, "g14_unproject.npz"), **out)


def g10_tracks():
    """Tracks.reverse / from_multiple / average (tracks.py:131-213) on synthetic result arrays with
    missing rows, e.g. merging a forward and a backward run."""
    rng = np.random.default_rng(77)
    t0 = datetime.datetime(2020, 1, 1)
    day = datetime.timedelta(days=1)
    P, T = 5, 7
    dts = [t0 + i * day for i in range(T)]
    out = {}
    runs = []
    for r in range(3):
        means = rng.standard_normal((P, T, 6)) * [1, 1, 0.1, 0.2, 0.2, 0.01] + [10, 20, 5, 0.1, 0, 0]
        sigmas = rng.uniform(0.05, 0.5, (P, T, 6))
        if r == 0:
            means[1, :2] = np.nan; sigmas[1, :2] = np.nan
        if r == 1:
            means[1, 1:4] = np.nan; sigmas[1, 1:4] = np.nan
            means[3] = np.nan; sigmas[3] = np.nan
        if r == 2:
            means[3] = np.nan; sigmas[3] = np.nan
            means[4, -1] = np.nan; sigmas[4, -1] = np.nan
        if r != 2:
            pass
        out[f"run{r}_means"], out[f"run{r}_sigmas"] = means, sigmas
        runs.append(glimpse.Tracks(datetimes=dts, time_unit=day, means=means.copy(), sigmas=sigmas.copy()))
    # runs 0 and 1 have rows that are all-NaN in both for track 3? make track 3 NaN in run 0 too
    for flag in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            merged = glimpse.Tracks.from_multiple(runs, ignore_nan=flag)
            am, asd = merged.average(ignore_nan=flag)
            bm, bsd = runs[0].average(ignore_nan=flag)
        out[f"merged_means_{int(flag)}"], out[f"merged_sigmas_{int(flag)}"] = merged.means, merged.sigmas
        out[f"merged_avg_means_{int(flag)}"], out[f"merged_avg_sigmas_{int(flag)}"] = am, asd
        out[f"run0_avg_means_{int(flag)}"], out[f"run0_avg_sigmas_{int(flag)}"] = bm, bsd
    rev = glimpse.Tracks(datetimes=dts, time_unit=day, means=out["run0_means"].copy(), sigmas=out["run0_sigmas"].copy())
    rev.reverse()
    out["rev_means"] = rev.means
    out["rev_days"] = np.array([(d - t0).days for d in rev.datetimes])
    np.savez_compressed(os.path.join(OUT, "g10_tracks.npz"), **out)


def g15_ragged():
    """Motion models with different particle counts in one Tracker.track call (each track of the reference has its
    own n, tracker.py:305-314): the frames of g8_c2mini, five tracks with n = 150, 150, 90, 200, 200 (the fourth
    starts outside the image), np.random seeded once."""
    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    frames, _ = synth.make_sequence(cam, 6, seed=12, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam, 4, border_px=70.0, seed=3)
    ns = [150, 150, 90, 200, 200]
    xys = [tuple(pts[0]), tuple(pts[1]), tuple(pts[2]), (100.0, 100.0), tuple(pts[3])]
    kws = [dict(xy=xy, dem=0.0, dem_sigma=0.0, n=n, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0),
                axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0)) for xy, n in zip(xys, ns)]
    imgs = [ref_image(frames[i], cam, t0 + i * day) for i in range(6)]
    tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)])
    models = [glimpse.CartesianMotion(time_unit=day, **kw) for kw in kws]
    np.random.seed(77)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), reduce_particles=lambda p, w: (p.shape, float(np.nansum(w))))
    after = np.random.random()
    out = {
        "seed": 77, "n_particles": np.array(ns), "xy": np.array(xys), "means": tracks.means, "sigmas": tracks.sigmas,
        "errors": np.array([0 if e is None else 1 for e in tracks.errors]),
        "reduced_n": np.array([r[0][1] for r in tracks.reduced]),
        "reduced_w": np.array([r[1] for r in tracks.reduced]),
        "random_after": after, "frames": np.stack(frames), "cam": cam,
    }
    np.savez_compressed(os.path.join(OUT, "g15_ragged.npz"), **out)
    print("g15 vx:", tracks.means[:, -1, 3], "errors:", out["errors"])


def g16_custom_motion():
    """User-defined motion models (the duck type of motion.py:13-89) through the reference Tracker: the frames of
    g8_c2mini, four tracks in one call -- tests/custom_motion.py's DriftMotion (no likelihood method),
    SpeedPriorMotion (array), a built-in CartesianMotion in between, NoTermMotion (None) -- np.random seeded once."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import custom_motion as cm

    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    frames, _ = synth.make_sequence(cam, 6, seed=12, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam, 4, border_px=70.0, seed=3)
    imgs = [ref_image(frames[i], cam, t0 + i * day) for i in range(6)]
    tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)])
    cart = dict(xy=tuple(pts[2]), dem=0.0, dem_sigma=0.0, n=150, xy_sigma=(0.2, 0.2), vxyz=(0.15, 0, 0),
                vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0))
    models = [cm.DriftMotion(tuple(pts[0]), day, n=150), cm.SpeedPriorMotion(tuple(pts[1]), day, n=150),
              glimpse.CartesianMotion(time_unit=day, **cart), cm.NoTermMotion(tuple(pts[3]), day, n=150)]
    np.random.seed(123)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
    after = np.random.random()
    out = {"seed": 123, "xy": np.array([pts[0], pts[1], pts[2], pts[3]]), "means": tracks.means, "sigmas": tracks.sigmas,
           "errors": np.array([0 if e is None else 1 for e in tracks.errors]), "random_after": after,
           "frames": np.stack(frames), "cam": cam, "n": np.array([m.n for m in models]),
           "last_particles_1": tracks.particles[1][-1], "last_weights_1": tracks.weights[1][-1]}
    np.savez_compressed(os.path.join(OUT, "g16_custom_motion.npz"), **out)
    print("g16 vx:", tracks.means[:, -1, 3], "errors:", out["errors"])


def g17_highpass():
    """Tracker(highpass={"size": ...}) other than the default (5, 5) (tracker.py:59, :530): tiles of the reference's
    extract_tile for sizes (3, 3), (7, 7), (3, 5) [rows, columns] and 3 on gray and RGB frames, and whole tracks
    (the g8_c2mini scene) for (3, 3) and (7, 7)."""
    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0)
    frames, _ = synth.make_sequence(cam, 2, seed=5)
    rgb, _ = synth.make_sequence(cam, 2, seed=6, channels=3)
    t0 = datetime.datetime(2020, 1, 1)
    day = datetime.timedelta(days=1)
    # (the frames are those of g2_tiles.npz and g15_ragged.npz: not stored again)
    g2 = np.load(os.path.join(OUT, "g2_tiles.npz"))
    assert np.array_equal(g2["gray"], np.stack(frames)) and np.array_equal(g2["rgb"], np.stack(rgb))
    out = {}
    bt, bs = (20, 30, 51, 61), (5, 12, 80, 70)
    out["tbox"], out["sbox"] = np.array(bt), np.array(bs)
    sizes = [(3, 3), (7, 7), (3, 5), 3]
    out["sizes"] = np.array([(s, s) if np.isscalar(s) else s for s in sizes])
    for k, size in enumerate(sizes):
        tracker = glimpse.Tracker([glimpse.Observer([ref_image(frames[i], cam, t0 + i * day) for i in range(2)]),
                                   glimpse.Observer([ref_image(rgb[i], cam, t0 + i * day) for i in range(2)])],
                                  highpass={"size": size})
        for o, name in enumerate(["gray", "rgb"]):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                tile, hist = tracker.extract_tile(obs=o, img=0, box=np.array(bt), return_histogram=True)
                search = tracker.extract_tile(obs=o, img=1, box=np.array(bs), histogram=hist)
            out[f"{name}_{k}_tile"], out[f"{name}_{k}_hist_v"], out[f"{name}_{k}_hist_q"] = tile, hist[0], hist[1]
            out[f"{name}_{k}_search"] = search
    # whole tracks
    cam2 = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    seq, _ = synth.make_sequence(cam2, 6, seed=12, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam2, 3, border_px=70.0, seed=3)
    g15 = np.load(os.path.join(OUT, "g15_ragged.npz"))
    assert np.array_equal(g15["frames"], np.stack(seq)) and np.array_equal(g15["cam"], cam2)
    out["e2e_xy"] = pts
    for size in [(3, 3), (7, 7)]:
        imgs = [ref_image(seq[i], cam2, t0 + i * day) for i in range(6)]
        tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)], highpass={"size": size})
        models = [glimpse.CartesianMotion(xy=tuple(xy), time_unit=day, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in pts]
        np.random.seed(31)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track(models, tile_size=(15, 15))
        out[f"e2e_means_{size[0]}"], out[f"e2e_sigmas_{size[0]}"] = tracks.means, tracks.sigmas
        print("g17", size, "vx:", tracks.means[:, -1, 3])
    np.savez_compressed(os.path.join(OUT, "g17_highpass.npz"), **out)


def g26_highpass_modes():
    """Tracker(highpass={"size": ..., "mode": ...}) (tracker.py:59, :530: the dictionary goes to
    scipy.ndimage.median_filter): the boundary modes 'nearest', 'mirror', 'wrap' -- tiles of the reference's extract_tile
    on gray and RGB frames for the 5 x 5 default and a (3, 7) window, and whole tracks (the g15 scene) for 'nearest' with
    the default window and 'mirror' with (3, 3)."""
    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0)
    frames, _ = synth.make_sequence(cam, 2, seed=5)
    rgb, _ = synth.make_sequence(cam, 2, seed=6, channels=3)
    t0 = datetime.datetime(2020, 1, 1)
    day = datetime.timedelta(days=1)
    g2 = np.load(os.path.join(OUT, "g2_tiles.npz"))  # (the frames are those of g2_tiles.npz: not stored again)
    assert np.array_equal(g2["gray"], np.stack(frames)) and np.array_equal(g2["rgb"], np.stack(rgb))
    out = {}
    bt, bs = (20, 30, 51, 61), (5, 12, 80, 70)
    out["tbox"], out["sbox"] = np.array(bt), np.array(bs)
    cases = [((5, 5), "nearest"), ((5, 5), "mirror"), ((5, 5), "wrap"), ((3, 7), "nearest"), ((3, 7), "mirror"),
             ((3, 7), "wrap")]
    out["sizes"] = np.array([c[0] for c in cases])
    out["modes"] = np.array([c[1] for c in cases])
    for k, (size, mode) in enumerate(cases):
        tracker = glimpse.Tracker([glimpse.Observer([ref_image(frames[i], cam, t0 + i * day) for i in range(2)]),
                                   glimpse.Observer([ref_image(rgb[i], cam, t0 + i * day) for i in range(2)])],
                                  highpass={"size": size, "mode": mode})
        for o, name in enumerate(["gray", "rgb"]):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                tile, hist = tracker.extract_tile(obs=o, img=0, box=np.array(bt), return_histogram=True)
                search = tracker.extract_tile(obs=o, img=1, box=np.array(bs), histogram=hist)
            out[f"{name}_{k}_tile"], out[f"{name}_{k}_hist_v"], out[f"{name}_{k}_hist_q"] = tile, hist[0], hist[1]
            out[f"{name}_{k}_search"] = search
    cam2 = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    seq, _ = synth.make_sequence(cam2, 6, seed=12, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam2, 3, border_px=70.0, seed=3)
    g15 = np.load(os.path.join(OUT, "g15_ragged.npz"))
    assert np.array_equal(g15["frames"], np.stack(seq)) and np.array_equal(g15["cam"], cam2)
    out["e2e_xy"] = pts
    for tag, hp in (("nearest", {"size": (5, 5), "mode": "nearest"}), ("mirror", {"size": 3, "mode": "mirror"})):
        imgs = [ref_image(seq[i], cam2, t0 + i * day) for i in range(6)]
        tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)], highpass=hp)
        models = [glimpse.CartesianMotion(xy=tuple(xy), time_unit=day, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in pts]
        np.random.seed(31)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track(models, tile_size=(15, 15))
        out[f"e2e_means_{tag}"], out[f"e2e_sigmas_{tag}"] = tracks.means, tracks.sigmas
        print("g26", tag, "vx:", tracks.means[:, -1, 3])
    np.savez_compressed(os.path.join(OUT, "g26_highpass_modes.npz"), **out)


def scene16(channels):
    """The 16-bit scene of g18 (tests regenerate it from the same recipe and check `checksum`)."""
    cam = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    scene = synth.default_scene(cam, seed=12, velocity=(0.15, 0.0), n_frames=5)
    frames = [scene.render(cam, float(t), channels=channels, bits=16) for t in range(5)]
    return cam, frames


def g18_uint16():
    """uint16 frames (Tracker.extract_tile works on any dtype, tracker.py:494-534): whole tracks on a gray and an RGB
    16-bit scene, plus the last track's template (tile, histogram) as the reference holds it after the run."""
    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    out = {}
    for name, channels in (("gray", 1), ("rgb", 3)):
        cam, frames = scene16(channels)
        assert frames[0].dtype == np.uint16
        pts = synth.grid_points(cam, 3, border_px=70.0, seed=3)
        imgs = [ref_image(frames[i], cam, t0 + i * day) for i in range(len(frames))]
        tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)])
        models = [glimpse.CartesianMotion(xy=tuple(xy), time_unit=day, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in pts]
        np.random.seed(41)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track(models, tile_size=(15, 15))
        tpl = tracker.templates[0]
        out.update({f"{name}_means": tracks.means, f"{name}_sigmas": tracks.sigmas, f"{name}_xy": pts,
                    f"{name}_checksum": np.int64(sum(int(f.astype(np.int64).sum()) for f in frames)),
                    f"{name}_tpl_tile": tpl["tile"], f"{name}_tpl_hist_v": tpl["histogram"][0],
                    f"{name}_tpl_hist_q": tpl["histogram"][1], f"{name}_tpl_box": np.asarray(tpl["box"])})
        print("g18", name, "vx:", tracks.means[:, -1, 3], "distinct template values:", len(tpl["histogram"][0]))
    np.savez_compressed(os.path.join(OUT, "g18_uint16.npz"), **out)


def g19_observer_helpers():
    """Observer.subset / split (observer.py:455-493 over helpers.select_datetimes, helpers.py:1883-1951) on an
    irregular image schedule, and Observer.shift_tile (observer.py:146-176).  Times are stored as integer seconds
    from 2020-01-01."""
    t0 = datetime.datetime(2020, 1, 1)
    rng = np.random.default_rng(19)
    secs = np.cumsum(rng.integers(1, 5000, 40)).astype(np.int64)
    cam = synth.nadir_camera((32, 32), f=100.0, height=10.0)
    blank = np.zeros((32, 32), dtype=np.uint8)
    obs = glimpse.Observer([ref_image(blank, cam, t0 + datetime.timedelta(seconds=int(s))) for s in secs])
    out = {"secs": secs}
    when = lambda s: t0 + datetime.timedelta(seconds=int(s))  # noqa: E731
    index_of = {d: i for i, d in enumerate(obs.datetimes)}
    cases = []
    for k in range(24):
        a, b = sorted(rng.integers(secs[0] - 3000, secs[-1] + 3000, 2))
        snap = [None, 3600, 7200, 1800][k % 4]
        maxdt = [None, None, 600, 0][(k // 4) % 4] if snap else None
        kw = {}
        if k % 3 != 0:
            kw["start"] = when(a)
        if k % 5 != 0:
            kw["end"] = when(b)
        if snap:
            kw["snap"] = datetime.timedelta(seconds=snap)
            if maxdt is not None:
                kw["maxdt"] = datetime.timedelta(seconds=maxdt)
        try:
            sub = obs.subset(**kw)
            keep = np.zeros(len(secs), dtype=bool)
            keep[[index_of[d] for d in sub.datetimes]] = True
        except ValueError:  # fewer than two images left
            keep = glimpse.helpers.select_datetimes(obs.datetimes, **kw)
        cases.append([a if "start" in kw else -1, b if "end" in kw else -1, snap or 0, -1 if maxdt is None else maxdt])
        out[f"mask_{k}"] = keep
    out["cases"] = np.array(cases, dtype=np.int64)
    for n, overlap in [(3, 1), (4, 0), (5, 2)]:
        parts = obs.split(n, overlap=overlap)
        out[f"split_{n}_{overlap}"] = np.array([[index_of[p.datetimes[0]], index_of[p.datetimes[-1]], len(p.images)]
                                                for p in parts])
    brk = [when(secs[9] + 7), when(secs[25])]
    parts = obs.split(brk, overlap=1)
    out["split_breaks"] = np.array([[index_of[p.datetimes[0]], index_of[p.datetimes[-1]], len(p.images)] for p in parts])
    out["breaks_secs"] = np.array([secs[9] + 7, secs[25]])
    tile = rng.standard_normal((9, 11))
    rgb = rng.standard_normal((7, 8, 3))
    out["tile"], out["rgb"] = tile.copy(), rgb.copy()
    out["duv"] = np.array([[0.3, -0.2], [-0.5, 0.5], [0.0, 0.25]])
    for i, duv in enumerate(out["duv"]):
        out[f"shift_{i}"] = obs.shift_tile(tile.copy(), duv)
        out[f"shift_rgb_{i}"] = obs.shift_tile(rgb.copy(), duv)
    np.savez_compressed(os.path.join(OUT, "g19_observer_helpers.npz"), **out)
    print("g19", {k: int(v.sum()) for k, v in out.items() if k.startswith("mask_")})
    print("g19 splits", out["split_3_1"].tolist(), out["split_4_0"].tolist(), out["split_breaks"].tolist())


def scene64():
    """The float64 scene of g20: the 16-bit gray scene mapped to reflectance-like doubles (a gamma curve: almost every
    pixel of a tile is a distinct value -- with a few exact repeats from the 16-bit source, which np.unique merges)."""
    cam, frames16 = scene16(1)
    frames = [np.power(f.astype(np.float64) / 65535.0, 0.8) * 3.5 - 1.25 for f in frames16]
    return cam, frames


def g20_float64():
    """float64 frames (Tracker.extract_tile works on any dtype, tracker.py:494-534): whole tracks, the last track's
    template, and the reference's own tiles for explicit boxes (template + histogram, search tile)."""
    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    cam, frames = scene64()
    assert frames[0].dtype == np.float64 and frames[0].ndim == 2
    pts = synth.grid_points(cam, 3, border_px=70.0, seed=3)
    imgs = [ref_image(frames[i], cam, t0 + i * day) for i in range(len(frames))]
    tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)])
    models = [glimpse.CartesianMotion(xy=tuple(xy), time_unit=day, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                      vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                      axyz_sigma=(0.05, 0.05, 0.0)) for xy in pts]
    np.random.seed(43)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15))
    tpl = tracker.templates[0]
    out = {"means": tracks.means, "sigmas": tracks.sigmas, "xy": pts,
           "checksum": np.float64(sum(float(f.sum()) for f in frames)),
           "tpl_tile": tpl["tile"], "tpl_hist_v": tpl["histogram"][0], "tpl_hist_q": tpl["histogram"][1],
           "tpl_box": np.asarray(tpl["box"])}
    bt, bs = np.array((100, 90, 115, 105)), np.array((92, 80, 126, 117))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tile, hist = tracker.extract_tile(obs=0, img=0, box=bt, return_histogram=True)
        search = tracker.extract_tile(obs=0, img=1, box=bs, histogram=hist)
    out.update({"tbox": bt, "sbox": bs, "tile": tile, "hist_v": hist[0], "hist_q": hist[1], "search": search})
    print("g20 vx:", tracks.means[:, -1, 3], "distinct template values:", len(tpl["histogram"][0]), "of", tpl["tile"].size)
    np.savez_compressed(os.path.join(OUT, "g20_float64.npz"), **out)


def g21_bilinear():
    """Tracker(interpolation={"kx": 1, "ky": 1}) (tracker.py:60, :585-590, :623): the SSD surface sampled by a degree-1
    RectBivariateSpline, the search box widened to 2 cells only -- whole tracks on the g15 scene."""
    t0 = datetime.datetime(2020, 1, 1)
    day = datetime.timedelta(days=1)
    cam2 = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    seq, _ = synth.make_sequence(cam2, 6, seed=12, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam2, 3, border_px=70.0, seed=3)
    g15 = np.load(os.path.join(OUT, "g15_ragged.npz"))
    assert np.array_equal(g15["frames"], np.stack(seq)) and np.array_equal(g15["cam"], cam2)
    imgs = [ref_image(seq[i], cam2, t0 + i * day) for i in range(6)]
    tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)], interpolation={"kx": 1, "ky": 1})
    out = {"xy": pts}
    for tag, sig in (("wide", (0.2, 0.2)), ("tight", (0.004, 0.004))):
        # (tight: a cloud narrower than a pixel, where the least surface size -- 2 x 2 instead of 4 x 4 -- matters)
        models = [glimpse.CartesianMotion(xy=tuple(xy), time_unit=day, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=sig,
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0) if tag == "wide" else (0.002, 0.002, 0.0),
                                          axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0) if tag == "wide" else (0.0005, 0.0005, 0.0))
                  for xy in pts]
        np.random.seed(47)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track(models, tile_size=(15, 15))
        out[f"{tag}_means"], out[f"{tag}_sigmas"] = tracks.means, tracks.sigmas
        print("g21", tag, "vx:", tracks.means[:, -1, 3])
    np.savez_compressed(os.path.join(OUT, "g21_bilinear.npz"), **out)


def g23_orders():
    """Tracker(interpolation={"kx": ..., "ky": ...}) for every order RectBivariateSpline takes (1 .. 5, mixed): (a)
    Observer.sample_tile on random surfaces down to the least size (kx + 1 rows, ky + 1 columns) with points on the edges
    and corners; (b) whole tracks on the g15 scene, clouds wider and narrower than a pixel (the order also sets the least
    size of the surface: tracker.py:585-590)."""
    rng = np.random.default_rng(2323)
    cam = synth.nadir_camera((64, 64))
    imgs = [ref_image(np.zeros((64, 64), np.uint8), cam, datetime.datetime(2020, 1, 1 + i)) for i in range(2)]
    obs = glimpse.Observer(imgs)
    out = {}
    orders = [(2, 2), (4, 4), (5, 5), (1, 3), (3, 2), (5, 1), (2, 4), (3, 5), (1, 2)]
    out["orders"] = np.array(orders)
    case = 0
    for kx, ky in orders:
        for ho, wo in [(kx + 1, ky + 1), (kx + 2, ky + 3), (9, 8), (23, 17), (40, 33)]:
            sse = rng.random((ho, wo)).astype(np.float32)
            l, t = rng.integers(0, 1000, 2)
            duv = rng.uniform(-0.5, 0.5, 2)
            box = np.array([l + 7.5 - 0.5, t + 7.5 - 0.5, l + 7.5 - 0.5 + wo, t + 7.5 - 0.5 + ho]) + np.tile(duv, 2)
            n = 200
            uv = np.column_stack((rng.uniform(box[0], box[2], n), rng.uniform(box[1], box[3], n)))
            uv[0], uv[1], uv[2] = box[0:2], box[2:4], (box[0], box[3])
            # knots of the even orders sit between the cell centres: points exactly on centres and on midpoints
            uv[3] = (box[0] + 0.5 + wo // 2, box[1] + 0.5 + ho // 2)
            uv[4] = (box[0] + 1.0 + wo // 2 - 1, box[1] + 1.0 + ho // 2 - 1)
            out[f"c{case}_k"] = np.array([kx, ky])
            out[f"c{case}_sse"], out[f"c{case}_box"], out[f"c{case}_uv"] = sse, box, uv
            out[f"c{case}_val"] = obs.sample_tile(uv, tile=sse, box=box, grid=False, kx=kx, ky=ky)
            case += 1
    out["n_cases"] = case

    t0 = datetime.datetime(2020, 1, 1)
    day = datetime.timedelta(days=1)
    cam2 = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    seq, _ = synth.make_sequence(cam2, 6, seed=12, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam2, 3, border_px=70.0, seed=3)
    g15 = np.load(os.path.join(OUT, "g15_ragged.npz"))
    assert np.array_equal(g15["frames"], np.stack(seq)) and np.array_equal(g15["cam"], cam2)
    frames_imgs = [ref_image(seq[i], cam2, t0 + i * day) for i in range(6)]
    out["xy"] = pts
    e2e = [(2, 2), (5, 5), (3, 1), (4, 2)]
    out["e2e_orders"] = np.array(e2e)
    for kx, ky in e2e:
        tracker = glimpse.Tracker([glimpse.Observer(frames_imgs, sigma=0.3)], interpolation={"kx": kx, "ky": ky})
        for tag, sig in (("wide", (0.2, 0.2)), ("tight", (0.004, 0.004))):
            models = [glimpse.CartesianMotion(xy=tuple(xy), time_unit=day, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=sig,
                                              vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0) if tag == "wide" else (0.002, 0.002, 0.0),
                                              axyz=(0, 0, 0), axyz_sigma=(0.05, 0.05, 0.0) if tag == "wide" else (0.0005, 0.0005, 0.0))
                      for xy in pts]
            np.random.seed(4700 + 10 * kx + ky)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                tracks = tracker.track(models, tile_size=(15, 15))
            assert all(e is None for e in tracks.errors)
            out[f"k{kx}{ky}_{tag}_means"], out[f"k{kx}{ky}_{tag}_sigmas"] = tracks.means, tracks.sigmas
            print("g23", (kx, ky), tag, "vx:", tracks.means[:, -1, 3])
    np.savez_compressed(os.path.join(OUT, "g23_orders.npz"), **out)


def float_scenes():
    """The float scenes of g24 (tests regenerate them from the same recipe: glimpse_amd.synth only): float32 one channel,
    float32 three channels, float64 three channels -- the reflectance-like doubles of g20, narrowed / spread over
    channels that differ by a gain, an offset and a one-pixel shift (so that the channel mean is not any one channel)."""
    cam, frames64 = scene64()
    f32 = [f.astype(np.float32) for f in frames64]

    def rgb(f):
        return np.stack([f, 0.9 * f + 0.05, np.roll(f, 1, axis=1) * 1.1 - 0.1], axis=2)

    return cam, {"f32": f32, "f32rgb": [rgb(f).astype(np.float32) for f in frames64], "f64rgb": [rgb(f) for f in frames64]}


def g24_float_frames():
    """float32 frames and multi-channel float frames (Tracker.extract_tile works on any dtype, tracker.py:494-534; the
    arithmetic then runs in the frame's dtype under NumPy's rules -- these are outputs of the reference under the
    container's NumPy 2.2): whole tracks, the last track's template, tiles for explicit boxes."""
    day = datetime.timedelta(days=1)
    t0 = datetime.datetime(2020, 1, 1)
    cam, scenes = float_scenes()
    pts = synth.grid_points(cam, 3, border_px=70.0, seed=3)
    out = {"xy": pts, "numpy": np.array(np.__version__)}
    for tag, frames in scenes.items():
        imgs = [ref_image(frames[i], cam, t0 + i * day) for i in range(len(frames))]
        tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)])
        models = [glimpse.CartesianMotion(xy=tuple(xy), time_unit=day, dem=0.0, dem_sigma=0.0, n=200, xy_sigma=(0.2, 0.2),
                                          vxyz=(0.15, 0, 0), vxyz_sigma=(0.2, 0.2, 0.0), axyz=(0, 0, 0),
                                          axyz_sigma=(0.05, 0.05, 0.0)) for xy in pts]
        np.random.seed(4300 + len(tag))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks = tracker.track(models, tile_size=(15, 15))
        assert all(e is None for e in tracks.errors)
        tpl = tracker.templates[0]
        out.update({f"{tag}_means": tracks.means, f"{tag}_sigmas": tracks.sigmas,
                    f"{tag}_checksum": np.float64(sum(float(np.asarray(f, dtype=np.float64).sum()) for f in frames)),
                    f"{tag}_tpl_tile": tpl["tile"], f"{tag}_tpl_hist_v": tpl["histogram"][0],
                    f"{tag}_tpl_hist_q": tpl["histogram"][1], f"{tag}_tpl_box": np.asarray(tpl["box"])})
        bt, bs = np.array((100, 90, 115, 105)), np.array((92, 80, 126, 117))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tile, hist = tracker.extract_tile(obs=0, img=0, box=bt, return_histogram=True)
            search = tracker.extract_tile(obs=0, img=1, box=bs, histogram=hist)
        out.update({f"{tag}_tile": tile, f"{tag}_hist_v": hist[0], f"{tag}_hist_q": hist[1], f"{tag}_search": search})
        out["tbox"], out["sbox"] = bt, bs
        print("g24", tag, "vx:", tracks.means[:, -1, 3], "tile dtype", tile.dtype, "hist dtype", hist[0].dtype, "search dtype",
              search.dtype, "distinct", len(hist[0]), "of", tile.size)
    np.savez_compressed(os.path.join(OUT, "g24_float_frames.npz"), **out)


def g25_base_motion():
    """The reference's minimal motion model used as it is (motion.py:13-89: particles start AT xy, z = 0, velocities
    drawn around zero, no likelihood term) on the g15 scene -- the per-track loop with a model that has no device twin."""
    t0 = datetime.datetime(2020, 1, 1)
    day = datetime.timedelta(days=1)
    cam2 = synth.nadir_camera((256, 256), f=1000.0, height=100.0, k=(0.05, -0.01, 0.002))
    seq, _ = synth.make_sequence(cam2, 6, seed=12, velocity=(0.15, 0.0))
    pts = synth.grid_points(cam2, 3, border_px=70.0, seed=3)
    imgs = [ref_image(seq[i], cam2, t0 + i * day) for i in range(6)]
    tracker = glimpse.Tracker([glimpse.Observer(imgs, sigma=0.3)])
    from glimpse.track.motion import Motion as RefMotion  # (not re-exported at the package top level)
    models = [RefMotion(xy=tuple(xy), time_unit=day, n=300, vxyz_sigma=(0.3, 0.2, 0.0)) for xy in pts]
    np.random.seed(2501)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tracks = tracker.track(models, tile_size=(15, 15), return_particles=True)
    assert all(e is None for e in tracks.errors)
    print("g25 vx:", tracks.means[:, -1, 3])
    np.savez_compressed(os.path.join(OUT, "g25_base_motion.npz"), xy=pts, means=tracks.means, sigmas=tracks.sigmas,
                        particles=tracks.particles, weights=tracks.weights)


if __name__ == "__main__":
    if "--g25" in sys.argv:
        g25_base_motion()
        sys.exit(0)
    if "--g24" in sys.argv:
        g24_float_frames()
        sys.exit(0)
    if "--g23" in sys.argv:
        g23_orders()
        sys.exit(0)
    if "--g22" in sys.argv:
        g22_variants()
        sys.exit(0)
    if "--g21" in sys.argv:
        g21_bilinear()
        sys.exit(0)
    if "--g20" in sys.argv:
        g20_float64()
        sys.exit(0)
    if "--g19" in sys.argv:
        g19_observer_helpers()
        sys.exit(0)
    if "--g18" in sys.argv:
        g18_uint16()
        sys.exit(0)
    if "--g26" in sys.argv:
        g26_highpass_modes()
        sys.exit(0)
    if "--g17" in sys.argv:
        g17_highpass()
        sys.exit(0)
    if "--g16" in sys.argv:
        g16_custom_motion()
        sys.exit(0)
    if "--g15" in sys.argv:
        g15_ragged()
        sys.exit(0)
    if "--g14" in sys.argv:
        g14_unproject()
        sys.exit(0)
    if "--g13" in sys.argv:
        g13_ortho()
        sys.exit(0)
    if "--g12" in sys.argv:
        g12_rasters()
        sys.exit(0)
    if "--g11" in sys.argv:
        g11_motion_models()
        sys.exit(0)
    if "--g10" in sys.argv:
        g10_tracks()
        sys.exit(0)
    if "--g9" in sys.argv:
        g9_variants()
        sys.exit(0)
    g1_projection()
    g2_tiles()
    g4_spline()
    g5_resample()
    g7_motion()
    g8_c1()
    g8_c2mini()
    g8_c5mini()
    g9_variants()
    g10_tracks()
    g11_motion_models()
    g12_rasters()
    g13_ortho()
    g14_unproject()
    g15_ragged()
    g16_custom_motion()
    g17_highpass()
    g18_uint16()
    g19_observer_helpers()
    g20_float64()
    g21_bilinear()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
