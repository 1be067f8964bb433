"""Oracle: the per-track particle-filter loop (test infrastructure only).

Follows /root/reference/src/glimpse/track/tracker.py:
  * `Tracker.track` / `process`               tracker.py:225-417 (frame loop :326-357)
  * `update_weights`                           tracker.py:126-149
  * `compute_observer_log_likelihoods`         tracker.py:563-625
  * `initialize_template`                      tracker.py:536-561
  * `resample_particles`                       tracker.py:151-223
  * `test_particles` (NaN check)               tracker.py:118-119
Datetime matching (tracker.py:425-492) is host logic outside this module: the
oracle takes the matched image table `matching[i][o]` (image index or -1) and
the step lengths `taus[i]` = dts[i] / time_unit directly.

Observers are plain dicts: {"frames": [ndarray uint8 (H,W[,3])...],
"cams": ndarray (n_images, 24) (oracle.camera layout), "sigma": float}.
"""
import warnings

import numpy as np

from . import camera, resample, spline, ssd, tiles


class Observer(dict):
    def __init__(self, frames, cams, sigma=0.3, interp=(3, 3), ssd="f64"):
        """`interp` = (kx, ky) of Tracker(interpolation=...) (tracker.py:60, :585-590, :623); `ssd`: which
        restatement of cv2.matchTemplate's accumulation (oracle/ssd.py) -- "f64" or "row_f32"."""
        super().__init__(frames=list(frames), cams=np.asarray(cams, dtype=float), sigma=sigma, interp=tuple(interp),
                         ssd=ssd)


def observer_log_likelihoods(obs, img, template, particles, trace=None):
    """tracker.py:563-625.  Returns (n,) float64 or None (out-of-bounds search box)."""
    if img is None or img < 0:
        return None
    cam = obs["cams"][img]
    frame = obs["frames"][img]
    size = np.asarray(template["tile"].shape[0:2][::-1])
    uv = camera.xyz_to_uv(cam, particles[:, 0:3])
    halfsize = size * 0.5
    kx, ky = obs.get("interp", (3, 3))
    box = tiles.search_box(uv, size, kx=kx, ky=ky)
    if trace is not None:
        trace["uv"] = uv
        trace["box"] = box.ravel().copy()
    if not all(camera.inframe(cam, box)):
        warnings.warn("Particles too close to or beyond image bounds, skipping image")
        return None
    box = box.ravel()
    search_tile = tiles.extract_tile(frame, box, histogram=template["histogram"])
    sse = ssd.match_template_sqdiff(
        search_tile.astype(np.float32), template["tile"].astype(np.float32), accumulate=obs.get("ssd", "f64")
    )
    sse *= 1 / (size[0] * size[1])
    box_edge = halfsize - 0.5
    sse_box = box + np.concatenate((box_edge, -box_edge))
    sse_box += np.tile(template["duv"], 2)
    sampled = spline.sample_tile(uv, sse, sse_box, kx=kx, ky=ky)
    if trace is not None:
        trace["search_tile"] = search_tile
        trace["sse"] = sse
        trace["sse_box"] = sse_box
        trace["sampled"] = sampled
    return sampled * (1 / (2 * obs["sigma"] ** 2))


def update_weights(observers, imgs, templates, particles, weights, motion_model, trace=None):
    """tracker.py:126-149.  Returns the new weights (or the old ones unchanged)."""
    log_likelihoods = []
    for o, img in enumerate(imgs):
        tr = None
        if trace is not None:
            tr = {}
            trace.setdefault("obs", []).append(tr)
        log_likelihoods.append(
            observer_log_likelihoods(observers[o], img, templates[o], particles, tr)
            if img is not None and img >= 0
            else None
        )
    if motion_model is not None:
        log_likelihoods.append(motion_model.compute_log_likelihoods(particles))
    log_likelihoods = [x for x in log_likelihoods if x is not None]
    if log_likelihoods:
        likelihoods = np.exp(-sum(log_likelihoods))
        return likelihoods + 1e-300
    return weights


def _test_visible(viewshed, particles):
    """Tracker.test_particles, viewshed half (tracker.py:114-117): nearest-cell lookup."""
    if viewshed is not None:
        is_visible = viewshed.sample(particles[:, 0:2], order=0)
        if not all(is_visible):
            raise ValueError("Some particles are on non-visible viewshed cells")


def track_one(
    motion_model,
    observers,
    matching,
    taus,
    tile_size=(15, 15),
    observer_mask=None,
    return_covariances=False,
    return_particles=False,
    draws=None,
    trace=None,
    capture_errors=False,
    resample_method="systematic",
    viewshed=None,
):
    """One track: tracker.py:305-374 (`process`).

    `draws`: None -> draw from the global legacy np.random stream in the
    reference's order; a dict -> record the draws into it (keys 'init' (n,6),
    'evolve' list of (n,3), 'u' list of float) or replay them when the keys
    already exist.
    Raises like the reference for a single track (caller decides what to catch).
    """
    matching = np.asarray(matching)
    ntimes, nobs = matching.shape
    if observer_mask is None:
        observer_mask = np.ones(nobs, dtype=bool)
    observer_mask = np.asarray(observer_mask, dtype=bool)
    n = motion_model.n
    means = np.full((ntimes, 6), np.nan)
    sigmas = np.full((ntimes, 6, 6) if return_covariances else (ntimes, 6), np.nan)
    out_particles = np.full((ntimes, n, 6), np.nan) if return_particles else None
    out_weights = np.full((ntimes, n), np.nan) if return_particles else None
    replay = draws is not None and "init" in draws
    if draws is not None and not replay:
        draws["evolve"] = []
        draws["u"] = []
    step = 0
    template_indices = (matching >= 0).argmax(axis=0)
    observed = (matching[:, observer_mask] >= 0).any(axis=1)
    first = int(np.argmax(observed))
    last = len(observed) - 1 - int(np.argmax(observed[::-1]))
    templates = [None] * nobs
    particles = None
    weights = None
    error = None
    try:
        for i in range(first, last + 1):
            tr = None
            if trace is not None:
                tr = {"i": i}
                trace.append(tr)
            if i == first:
                particles, normals = motion_model.initialize_particles(
                    draws["init"] if replay else None
                )
                if draws is not None and not replay:
                    draws["init"] = normals
                if np.isnan(particles).any():
                    raise ValueError("Some particles have missing (NaN) values")
                _test_visible(viewshed, particles)
                weights = np.ones(n)
            else:
                normals = motion_model.evolve_particles(
                    particles, taus[i - 1], draws["evolve"][step] if replay else None
                )
                if draws is not None and not replay:
                    draws["evolve"].append(normals)
                if np.isnan(particles).any():
                    raise ValueError("Some particles have missing (NaN) values")
                _test_visible(viewshed, particles)
            if tr is not None:
                tr["evolved"] = particles.copy()
            at_template = observer_mask & (template_indices == i)
            for o in np.nonzero(at_template)[0]:
                img = matching[i][o]
                mean = resample.particle_mean(particles, weights)
                templates[o] = tiles.initialize_template(
                    observers[o]["frames"][img], observers[o]["cams"][img], mean, tile_size
                )
                if tr is not None:
                    tr.setdefault("templates", {})[int(o)] = templates[o]
            if i > first:
                imgs = [int(img) if m and img >= 0 else None for img, m in zip(matching[i], observer_mask)]
                weights = update_weights(observers, imgs, templates, particles, weights, motion_model, tr)
                if tr is not None:
                    tr["weights"] = weights.copy()
                if replay:
                    u = draws["u"][step]
                else:
                    # one uniform (systematic) or n of them (stratified; np.random.choice draws its n
                    # uniforms from the same global stream)
                    if resample_method == "systematic":
                        u = np.random.random()
                    elif resample_method == "residual":
                        u = np.random.random  # tracker.py:199-201 draws n - sum(repetitions) uniforms: known inside
                    else:
                        u = np.random.random(n)
                    if draws is not None:
                        draws["u"].append(u)
                idx = resample.METHODS[resample_method](weights, u)
                if tr is not None:
                    tr["idx"] = idx.copy()
                particles = particles[idx]
                weights = weights[idx]
                step += 1
            means[i] = resample.particle_mean(particles, weights)
            if return_covariances:
                sigmas[i] = resample.particle_covariance(particles, weights)
            else:
                sigmas[i] = resample.particle_sigma(particles, weights, means[i])
            if return_particles:
                out_particles[i] = particles
                out_weights[i] = weights
    except Exception as e:  # noqa: BLE001  (tracker.py:360-368)
        if not capture_errors:
            raise
        error = e
    return {
        "means": means,
        "sigmas": sigmas,
        "particles": out_particles,
        "weights": out_weights,
        "error": error,
    }


def track(motion_models, observers, matching, taus, **kwargs):
    """All tracks, one after another (tracker.py:381-387 with parallel=False).

    Errors are re-raised for a single track and captured (NaN rows) for >= 2
    tracks, like tracker.py:283, :360-368.
    """
    results = []
    errors = []
    draws_list = kwargs.pop("draws", None)
    masks = kwargs.pop("observer_mask", None)
    for p, model in enumerate(motion_models):
        kw = dict(kwargs)
        if draws_list is not None:
            kw["draws"] = draws_list[p]
        if masks is not None:
            kw["observer_mask"] = masks[p]
        kw["capture_errors"] = len(motion_models) >= 2
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            results.append(track_one(model, observers, matching, taus, **kw))
        errors.append(results[-1]["error"])
    return {
        "means": np.stack([r["means"] for r in results]),
        "sigmas": np.stack([r["sigmas"] for r in results]),
        "errors": errors,
        "results": results,
    }
