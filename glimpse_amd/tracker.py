"""`Tracker`: drop-in for `glimpse.Tracker` (/root/reference/src/glimpse/track/tracker.py).

Same constructor, `track()` signature, result container and error behaviour as the
reference, but the per-frame particle-filter step (tracker.py:326-357) runs on an MI355X
through libglimpse_hip.so for ALL tracks at once: the reference loops tracks outermost
(tracker.py:381-387), this loops frames outermost -- legal because tracks are independent.

There is no CPU fallback: without the HIP library every method that computes raises.

Random numbers.  The reference draws from the legacy global `np.random` stream, one whole
track after another.  `rng="numpy"` (default) reproduces exactly that
stream on the host and feeds it to the device, so `np.random.seed(s); tracker.track(...)`
gives the reference's particles, indices and posteriors.  `rng="philox"` draws on the device
(counter-based Philox4x32, fast arithmetic: `glh_set_math`) and is what large runs use.
"""
import datetime
import operator
import os
import warnings as _warnings

import numpy as np

from . import _lib
from .motion import (CartesianMotion, CylindricalMotion, ModelBlock, _RowView, TangentCartesianMotion, TangentCylindricalMotion,
                     params_table)
from .raster import Raster
from .timeutil import _US, _offsets_us, nearest_in_sorted  # noqa: F401  (re-exported)
from .tracks import Tracks

_ERRORS = (
    (_lib.PT_NAN, ValueError, "Some particles have missing (NaN) values"),
    (_lib.PT_NOT_VISIBLE, ValueError, "Some particles are on non-visible viewshed cells"),
    (_lib.PT_RASTER_OOB, ValueError, "Some of the sampling coordinates are out of bounds"),
    (_lib.PT_TEMPLATE_OOB, IndexError, "Box extends beyond grid bounds"),
    (_lib.PT_CONST_TILE, ValueError, "Template tile has zero variance"),
    (_lib.PT_SAMPLE_OUTSIDE, ValueError, "Some sampling points are outside box"),
    (_lib.PT_RESAMPLE_CLAMP, IndexError, "Resampling index out of bounds (weights do not sum to 1)"),
)
_OOB_WARNING = "Particles too close to or beyond image bounds, skipping image"
_MAX_HOST_DRAWS_BYTES = 4 << 30


_DEVICE_MODELS = (CartesianMotion, CylindricalMotion, TangentCartesianMotion, TangentCylindricalMotion)


def _on_device(model):
    """The four motion models of the reference are initialised and evolved on the device.  Anything else -- a
    subclass that overrides them, or any object with the interface of motion.py:13-89 -- is a user-defined model: its
    own initialize_particles / evolve_particles / compute_log_likelihoods run on the host, as the user wrote them."""
    return type(model) in _DEVICE_MODELS or isinstance(model, _RowView)


_GET_N = operator.attrgetter("n")
_GET_TIME_UNIT = operator.attrgetter("time_unit")


def _any_raster(models, attr):
    """Some model's `attr` is a Raster (the types of the attribute values, collected at C speed)."""
    return any(issubclass(t, Raster) for t in set(map(type, map(operator.attrgetter(attr), models))))


def _any_raster_safe(models, attr):
    try:
        return _any_raster(models, attr)
    except AttributeError:  # (some model without the attribute: the caller's own loop decides)
        return True


def _batches(models):
    """First index of every run of consecutive motion models that one device batch can hold: device models (`_on_device`)
    with the same particle count and at most ONE gridded dem and ONE gridded dem_sigma among them (constant surfaces are
    parameters of the point and mix freely); a user-defined model is a run of its own."""
    # the usual batch -- thousands of device models with one particle count and constant surfaces -- is recognised by
    # sets built at C speed; anything else takes the loop below
    if isinstance(models, ModelBlock):  # (made of one batch: motion.ModelBlock)
        return [0]
    models = list(models)
    if models and set(map(type, models)) <= set(_DEVICE_MODELS) and len(set(map(_GET_N, models))) == 1 \
            and not _any_raster(models, "dem") and not _any_raster(models, "dem_sigma"):
        return [0]
    starts, n, dem, sigma, device = [], None, None, None, False
    for i, m in enumerate(models):
        d = m.dem if isinstance(getattr(m, "dem", None), Raster) else None
        s = m.dem_sigma if isinstance(getattr(m, "dem_sigma", None), Raster) else None
        fits = (i > 0 and device and _on_device(m) and m.n == n
                and (d is None or dem is None or d is dem) and (s is None or sigma is None or s is sigma))
        if fits:
            dem, sigma = dem if d is None else d, sigma if s is None else s
        else:
            starts.append(i)
            n, dem, sigma, device = m.n, d, s, _on_device(m)
    return starts


def _vector24(img):
    """Camera vector of an Observer image: an Image's camera, or a Raster's own grid (observer.py:26)."""
    return img.cam.vector24 if hasattr(img, "cam") else img.vector24


def _usable_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota, whichever is smaller."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


class _FrameFeed:
    """The frames a run touches, on their way to HBM in the order the sequence needs them (image.py:137-214: the
    reference reads an image when a track first asks for it).  Images that still live in files are decoded WHILE the
    frame loop runs on the frames already resident: `need(i)` waits for the frames of time steps <= i only, hands them
    to the device (pinned staging + copy stream: glh_observer_upload_frame_async) together with every later frame that has
    been decoded meanwhile, and returns the last time step whose frames are all resident -- the frame loop runs up to
    there in one call.  Decoders: a pool of PROCESSES writing into a shared-memory ring (glimpse_amd/ingest.py) for
    sequences of at least `PROCESS_MIN` files, threads below that (Pillow releases the GIL only inside the decode proper:
    sixteen threads reached 4x one for JPEG, 1.2x for uncompressed TIFF).  In-memory images are all uploaded by the
    first call, as before."""

    PROCESS_MIN = 8

    def __init__(self, tracker, ctx, matching):
        self.tracker, self.ctx = tracker, ctx
        first_use = {}
        for i in range(len(matching)):
            for o, m in enumerate(matching[i]):
                if m is not None and (o, int(m)) not in tracker._uploaded:
                    first_use.setdefault((o, int(m)), i)
        self.jobs = sorted(first_use.items(), key=lambda kv: (kv[1], kv[0]))  # [((observer, image), first time step)]
        self.ntimes = len(matching)
        self.pos = 0  # jobs [0, pos) are on the device
        self.done = [False] * len(self.jobs)
        self.pool = None      # threads
        self.futures = None
        self.procs = None     # ingest.DecodePool
        self.submitted = 0
        self.decoding = 0     # images queued with the decoders, not yet back
        self.inflight = []    # (upload ticket, slot) of the copies enqueued from the page-locked ring, oldest first
        self.stats = tracker._feed_stats = dict(files=0, frames=len(self.jobs), decode_seconds=0.0, upload_seconds=0.0,
                                                bytes=0, threads=0, processes=0, waits=0, wait_seconds=0.0)
        on_disk = [k for k, (job, _) in enumerate(self.jobs) if self._image(job).__dict__.get("array") is None]
        self.on_disk = set(on_disk)
        self.stats["files"] = len(on_disk)
        cores = _usable_cores()
        if len(on_disk) >= self.PROCESS_MIN and cores >= 4 and not os.environ.get("GLH_DECODE_THREADS"):
            from . import ingest

            slot = max(nbytes for _, _, nbytes in tracker._frame_dtypes())
            want = max(2, min(cores - 1, len(on_disk)))
            pool = tracker._decoders
            if pool is None or not pool.alive() or pool.slot_bytes < slot or pool.n < want:
                if pool is not None:
                    pool.close()
                try:
                    pool = tracker._decoders = ingest.DecodePool(want, slot)
                except OSError:  # (no room for the ring in shared memory: the threads below)
                    pool = tracker._decoders = None
            else:
                pool.drain()
            if pool is not None:
                self.procs = pool
                self.stats["processes"] = pool.n
                self.stats["pinned_ring"] = bool(pool.pinned)
        if self.procs is None and len(on_disk) > 1:
            from concurrent.futures import ThreadPoolExecutor

            self.stats["threads"] = min(cores, len(on_disk))
            self.pool = ThreadPoolExecutor(max_workers=self.stats["threads"])
            self.futures = {k: self.pool.submit(self._pixels, self.jobs[k][0]) for k in on_disk}

    def _image(self, job):
        return self.tracker.observers[job[0]].images[job[1]]

    def _pixels(self, job):
        import time

        t0 = time.perf_counter()
        obs = self.tracker.observers[job[0]]
        a = np.ascontiguousarray(obs.images[job[1]].read(cache=obs.cache))  # (dtype checked by the upload)
        return a, time.perf_counter() - t0

    def _to_device(self, k, a, seconds):
        import time

        job = self.jobs[k][0]
        self.stats["decode_seconds"] += seconds
        t0 = time.perf_counter()
        self.ctx.observer_upload_frame_async(job[0], job[1], a)
        self.stats["upload_seconds"] += time.perf_counter() - t0
        self.stats["bytes"] += a.nbytes
        self.tracker._uploaded.add(job)
        self.done[k] = True

    def _wait(self, fn):
        import time

        t0 = time.perf_counter()
        out = fn()
        self.stats["waits"] += 1
        self.stats["wait_seconds"] += time.perf_counter() - t0
        return out

    def _reclaim(self, wait_one=False):
        """Slots whose host-to-device copy (from the page-locked ring) has finished go back to the decoders."""
        while self.inflight and self.ctx.upload_done(self.inflight[0][0], wait=wait_one):
            self.procs.release(self.inflight.pop(0)[1])
            wait_one = False

    def _pump_processes(self, until):
        """Keep the decoders fed; upload what they have finished; block until jobs [pos, until] are on the device."""
        import time

        pool = self.procs
        while True:
            self._reclaim()
            while self.submitted < len(self.jobs):  # (in the order of use; in-memory images go straight to the device)
                k = self.submitted
                if k not in self.on_disk:
                    a, dt = self._pixels(self.jobs[k][0])
                    self._to_device(k, a, dt)
                elif pool.submit(k, self._image(self.jobs[k][0])):
                    self.decoding += 1
                else:
                    break
                self.submitted += 1
            block = not all(self.done[self.pos:until + 1])  # (the decoders finish in any order)
            if self.decoding == 0:
                if block and self.inflight:  # (every slot is waiting for its copy: the oldest one first)
                    self._wait(lambda: self._reclaim(wait_one=True))
                    continue
                return
            got = self._wait(lambda: pool.result(True)) if block else pool.result(False)
            if got is None:
                return
            self.decoding -= 1
            k, view, slot, seconds = got
            job = self.jobs[k][0]
            if self.tracker.observers[job[0]].cache:  # (Observer.cache: the decoded pixels stay on the image, image.py:211-213)
                self._image(job).array = np.array(view)
            if pool.pinned:
                self.stats["decode_seconds"] += seconds
                t0 = time.perf_counter()
                self.inflight.append((self.ctx.observer_upload_frame_pinned(job[0], job[1], view), slot))
                self.stats["upload_seconds"] += time.perf_counter() - t0
                self.stats["bytes"] += view.nbytes
                self.tracker._uploaded.add(job)
                self.done[k] = True
            else:
                self._to_device(k, view, seconds)  # (copied into pinned staging before the call returns)
                pool.release(slot)
            del view

    def _upload(self, k, wait):
        if self.done[k]:
            return True
        if self.futures is not None and k in self.futures:
            fut = self.futures[k]
            if not wait and not fut.done():
                return False
            a, dt = fut.result() if fut.done() else self._wait(fut.result)
            del self.futures[k]  # (the pixels are not kept here)
        else:
            a, dt = self._pixels(self.jobs[k][0])
        self._to_device(k, a, dt)
        return True

    def need(self, i):
        """Every frame of time steps <= i is resident (or on its way, ahead of the next launch); returns the last time
        step for which that holds."""
        last = -1
        for k in range(self.pos, len(self.jobs)):
            if self.jobs[k][1] > i:
                break
            last = k
        if self.procs is not None:
            self._pump_processes(last)
        else:
            for k in range(self.pos, last + 1):
                self._upload(k, True)
            k = max(self.pos, last + 1)
            while k < len(self.jobs) and self._upload(k, self.futures is None):
                k += 1
        while self.pos < len(self.jobs) and self.done[self.pos]:
            self.pos += 1
        if self.pos >= len(self.jobs):
            self.close()
            return self.ntimes - 1
        return self.jobs[self.pos][1] - 1

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=False, cancel_futures=True)
            self.pool = None
        if self.procs is not None:
            while self.inflight:  # (the copies read the ring: its slots are free when they are over)
                try:
                    self._reclaim(wait_one=True)
                except Exception:  # noqa: BLE001  (the context is gone: so are its copies)
                    for _, slot in self.inflight:
                        self.procs.release(slot)
                    self.inflight = []
            if self.pos < len(self.jobs):
                self.procs.drain()  # (a run that ended early: nothing of it stays in the ring)
            self.procs = None


class Tracker:
    def __init__(self, observers, viewshed=None, resample_method="systematic", highpass={"size": (5, 5)},  # noqa: B006
                 interpolation={"kx": 3, "ky": 3}, device=0, max_search_dim=None):  # noqa: B006
        """tracker.py:52-70 (+ `device`: GPU ordinal; `max_search_dim`: side of the per-point search-tile workspaces in
        pixels -- None sizes them from the prior's projected spread and re-runs the sequence with larger ones if a
        search tile outgrows them, up to what the kernels take (2000 pixels; 1117 for 16-bit frames) and half of the
        device's free memory allows; beyond that, as with a fixed size, the observer is skipped on that frame with a
        warning.  The per-track step methods (`initialize_template`, `update_weights`, ...) use `max_search_dim or 320`
        and do not grow)."""
        self.observers = list(observers)
        if viewshed is not None and not isinstance(viewshed, Raster):
            raise TypeError("viewshed must be a glimpse_amd.Raster")
        if resample_method not in _lib.RESAMPLE:
            raise ValueError(f"resample_method {resample_method!r}: expected one of {sorted(_lib.RESAMPLE)}")
        # tracker.py:59, :530: the dict goes to scipy.ndimage.median_filter; `size` (an int or (rows, columns), odd, up
        # to 7) and `mode` ("reflect", "nearest", "mirror", "wrap" and scipy's grid-* aliases) are what the device
        # implements -- a footprint, an origin or the "constant" mode (a fill value in the matched tile's units) are not
        size = highpass.get("size", (5, 5))
        size = (int(size), int(size)) if np.isscalar(size) else tuple(int(v) for v in size)
        mode = highpass.get("mode", "reflect")
        if set(highpass) - {"size", "mode"} or len(size) != 2 or any(v < 1 or v > 7 or v % 2 == 0 for v in size) \
                or mode not in _lib.HIGHPASS_MODES:
            raise NotImplementedError("the high-pass filter is a median with odd size up to (7, 7) and mode 'reflect', "
                                      f"'nearest', 'mirror' or 'wrap': highpass={highpass!r}")
        self._highpass_size = size
        self._highpass_mode = mode
        orders = (int(interpolation.get("kx", 3)), int(interpolation.get("ky", 3)))
        if not all(1 <= k <= 5 for k in orders) or set(interpolation) - {"kx", "ky"}:
            # (tracker.py:60, :623: the dict goes to scipy RectBivariateSpline; its orders kx, ky are what the device
            # implements -- a smoothing factor `s` or a `bbox` would be another spline)
            raise NotImplementedError("sub-pixel interpolation: interpolating splines of orders kx, ky in 1 .. 5 "
                                      f"(the reference default is 3, 3), not {interpolation}")
        self._orders = orders
        self.viewshed = viewshed
        self.resample_method = resample_method
        self.highpass = highpass
        self.interpolation = interpolation
        self.device = device
        self.max_search_dim = max_search_dim
        self._last_state = None
        self._particles = None
        self._weights = None
        self.templates = None
        self._ctx = None
        self._ctx_key = None
        self._feed = None
        self._decoders = None  # ingest.DecodePool: processes that decode image files, kept between runs
        self._feed_stats = None  # decode / upload figures of the last run's frame feed (`_FrameFeed.stats`)
        self._uploaded = set()
        self._pool = None  # worker processes of track(parallel=N), kept between calls (glimpse_amd/parallel.py)

    # ---- host logic shared with the reference -------------------------------------------
    @property
    def datetimes(self):
        """tracker.py:84-87."""
        return np.unique(np.concatenate([obs.datetimes for obs in self.observers]))

    # `particles` / `weights`: the single-track state of the reference (tracker.py:61-62).  After a batch run they hold
    # the LAST track's final particles, like the reference leaves them -- fetched from the device when first read
    # (nobody reads them after a run of thousands of tracks; the download was 0.5 ms of every call).
    def _fetch_last_state(self):
        pending, self._last_state = self._last_state, None
        if pending is not None:
            ctx, point = pending
            if ctx is self._ctx and ctx.handle:
                self._particles, self._weights = ctx.get_point_state(point)

    @property
    def particles(self):
        self._fetch_last_state()
        return self._particles

    @particles.setter
    def particles(self, value):
        self._fetch_last_state()  # (the weights of a pending state are still wanted)
        self._particles = value

    @property
    def weights(self):
        self._fetch_last_state()
        return self._weights

    @weights.setter
    def weights(self, value):
        self._fetch_last_state()
        self._weights = value

    def reset(self):
        """tracker.py:419-423."""
        self._last_state = None
        self._particles = None
        self._weights = None
        self.templates = None

    def parse_datetimes(self, datetimes, maxdt=datetime.timedelta(0)):
        """Behaviour of tracker.py:425-464: the sequence must be monotone (either direction: backward tracking
        passes decreasing datetimes), repeated datetimes and datetimes further than |maxdt| from every Observer
        image are dropped with the reference's warnings, at least two must remain."""
        datetimes = np.asarray(datetimes)
        ref = datetimes[0] if len(datetimes) else None
        t = _offsets_us(datetimes, ref) if len(datetimes) else np.zeros(0, dtype=np.int64)
        steps = np.diff(t)
        if (steps < 0).any() and (steps > 0).any():
            raise ValueError("Datetimes must be monotonic")
        keep = np.ones(len(t), dtype=bool)
        keep[1:] = steps != 0
        if not keep.all():
            _warnings.warn("Dropping duplicate datetimes")
            datetimes, t = datetimes[keep], t[keep]
        pool = np.sort(_offsets_us(self.datetimes, ref)) if len(t) else t
        _, distance = nearest_in_sorted(pool, t)
        keep = distance <= abs(maxdt // _US)
        if not keep.all():
            _warnings.warn("Dropping datetimes not matching any Observers")
            datetimes = datetimes[keep]
        if len(datetimes) < 2:
            raise ValueError("Fewer than two valid datetimes")
        return datetimes

    def match_datetimes(self, datetimes, maxdt=datetime.timedelta(0)):
        """tracker.py:466-492: (len(datetimes), len(observers)) object array, the index of each Observer's image
        nearest to each datetime, None where it is further than |maxdt|."""
        datetimes = np.asarray(datetimes)
        matches = np.full((len(datetimes), len(self.observers)), None)
        if not len(datetimes):
            return matches
        ref = datetimes[0]
        t = _offsets_us(datetimes, ref)
        limit = abs(maxdt // _US)
        for o, observer in enumerate(self.observers):  # (Observer datetimes are strictly increasing)
            index, distance = nearest_in_sorted(_offsets_us(observer.datetimes, ref), t)
            column = index.astype(object)
            column[distance > limit] = None
            matches[:, o] = column
        return matches

    # ---- device context ---------------------------------------------------------------------
    def _context(self, n_points, n_particles, n_frames, tile_size, search_dim, keep_old=False):
        O = len(self.observers)
        key = (n_points, n_particles, n_frames, max(tile_size), search_dim)
        if self._ctx is not None and self._ctx_key == key:
            return self._ctx
        if self._ctx is not None and not keep_old:
            # (closed first: two contexts of a large batch need not fit side by side; the fields are cleared at once, so
            # that a failure below cannot leave a destroyed context behind for the next run of this shape)
            old, self._ctx, self._ctx_key = self._ctx, None, None
            old.close()
        ctx = _lib.Context(n_points, n_particles, O, device_id=self.device, max_tile=max(31, max(tile_size)),
                           max_search_dim=search_dim, max_frames=n_frames)
        try:
            for o, obs in enumerate(self.observers):
                first = obs.images[0].read(cache=obs.cache)  # (an observer that does not cache keeps no pixels)
                if first.dtype not in (np.uint8, np.uint16, np.float32, np.float64) or \
                        (first.ndim == 3 and first.shape[2] not in (1, 3)):
                    raise NotImplementedError("frames must be uint8, uint16, float32 or float64 with one or three channels "
                                              f"on the GPU path, not {first.dtype} {first.shape}")
                h, w = first.shape[:2]
                ch = 1 if first.ndim == 2 else first.shape[2]
                ctx.observer_init(o, len(obs.images), w, h, ch, obs.sigma)
                ctx.observer_set_depth(o, first.dtype)
                ctx.observer_set_cameras(o, np.stack([_vector24(img) for img in obs.images]))
            ctx.set_highpass(self._highpass_size, self._highpass_mode)
            ctx.set_interpolation(*self._orders)
        except Exception:
            ctx.close()
            raise
        if self._ctx is not None:  # (keep_old: the previous context served until its successor was ready)
            self._ctx.close()
        self._ctx, self._ctx_key = ctx, key
        self._uploaded = set()
        return ctx

    def _frame_dtypes(self):
        """dtype (and channel count) of every observer's frames, read from its first image once (a read of a file-backed,
        uncached image decodes the file)."""
        key = tuple(id(obs.images[0]) for obs in self.observers)
        if getattr(self, "_dtypes_key", None) != key:
            firsts = [np.asarray(obs.images[0].read(cache=obs.cache)) for obs in self.observers]
            self._dtypes = [(a.dtype, 1 if a.ndim == 2 else a.shape[2], a.nbytes) for a in firsts]
            self._dtypes_key = key
        return self._dtypes

    def _sixteen_bit(self):
        """Some observer's frames are uint16: no kernel takes those beyond 1117-pixel workspaces."""
        return any(dt == np.uint16 for dt, _, _ in self._frame_dtypes())

    def _ranked_keys(self):
        """Some observer's frames are uint16 or float: the fused step ranks a tile's pixels and takes those frames
        while the workspaces are at most 255 pixels (the count of a tile's pixels must fit a 16-bit key)."""
        return any(dt != np.uint8 for dt, _, _ in self._frame_dtypes())

    def _dim_limit(self, n_points):
        """The largest workspace side the automatic growth may ask for: what the kernels take (2000 pixels; 1117 for
        16-bit frames, glh_observer_set_depth) and what fits a memory budget -- search tile, keys and surface are
        about 14 dim^2 bytes per point and observer, float frames add the 16 dim^2 bytes of their normalisation workspace
        (allocated on first use, inside the run), uint16 frames a key histogram of 262 KB per point and channel whatever
        the side -- and half of the device's free memory may go to all of them."""
        limit = 1117 if self._sixteen_bit() else 2000
        try:
            free, _ = _lib.device_memory(self.device)
            P = max(1, n_points)
            per, fixed = 0.0, 0.0
            for dt, channels, _ in self._frame_dtypes():
                per += (14.0 + (16.0 if dt in (np.float32, np.float64) else 0.0)) * P
                if dt == np.uint16:
                    fixed += 4.0 * (65535 * channels + 1) * P
            budget = 0.5 * free - fixed
            limit = min(limit, int(np.sqrt(max(budget, 0.0) / max(per, 1.0))))
        except Exception:  # noqa: BLE001  (no device to ask: the kernels' limits alone)
            pass
        return limit

    def _fit_dim(self, need, limit):
        """Workspace side for a need of `need` pixels: a multiple of 16, except that 16-bit and float frames stay at 255
        while the need allows it (256 would send the whole run to the staged kernels, several times slower)."""
        dim = int(16 * np.ceil(need / 16))
        if need <= 255 < dim and self._ranked_keys():
            dim = 255
        return int(min(limit, dim))

    def _estimate_search_dim(self, motion_models, matching, taus, tile_size):
        """Side (pixels) of the search-tile workspaces a run is likely to need: the template plus five standard
        deviations on either side of the particle cloud at its widest -- the prior, widened by a few frames of free
        drift (the filter needs a few updates to pin the velocity down) -- projected through every observer's camera.
        A guess, not a bound: `track` re-runs with larger workspaces when a search tile outgrows it."""
        free = 4.0 * (float(np.max(np.abs(taus))) if len(taus) else 1.0)  # time units of unconstrained drift
        t = params_table(motion_models)
        v, vs, a, as_, kind = t[:, 4:7], t[:, 7:10], t[:, 10:13], t[:, 13:16], t[:, 18]
        polar = (kind == CylindricalMotion.KIND) | (kind == TangentCylindricalMotion.KIND)
        sv = np.where(polar, np.hypot(vs[:, 0], np.abs(v[:, 0]) * vs[:, 1]), np.maximum(vs[:, 0], vs[:, 1]))
        sa = np.where(polar, np.hypot(as_[:, 0], np.abs(a[:, 0]) * as_[:, 1]), np.maximum(as_[:, 0], as_[:, 1]))
        s_h = np.sqrt(np.max(t[:, 2:4], axis=1) ** 2 + (free * sv) ** 2 + (0.5 * free ** 2 * sa) ** 2)
        s_z = np.sqrt(t[:, 17] ** 2 + (free * (vs[:, 2] + t[:, 19] * sv)) ** 2 + (0.5 * free ** 2 * as_[:, 2]) ** 2)
        z0 = t[:, 16].copy()
        # rows over gridded surfaces (columns 20 / 21): the surface under the starting point, ONE sample call per Raster
        for col, attr in ((20, "dem"), (21, "dem_sigma")):
            by_raster = {}
            for p in np.nonzero(t[:, col])[0]:
                r = getattr(motion_models[int(p)], attr)
                by_raster.setdefault(id(r), (r, []))[1].append(int(p))
            for r, rows in by_raster.values():
                z = np.asarray(r.sample(t[rows, 0:2], bounds_error=False), dtype=float)
                z = np.where(np.isfinite(z), z, 0.0)
                if attr == "dem":
                    z0[rows] = z
                else:
                    s_z[rows] = np.hypot(s_z[rows], z)
        P = len(motion_models)
        base = np.column_stack((t[:, 0:2], z0))
        pts = np.concatenate((base, base + np.column_stack((s_h, np.zeros(P), np.zeros(P))),
                              base + np.column_stack((np.zeros(P), s_h, np.zeros(P))),
                              base + np.column_stack((np.zeros(P), np.zeros(P), s_z))))
        spread = 0.0
        for o, obs in enumerate(self.observers):
            imgs = [m for m in matching[:, o] if m is not None]
            if not imgs:
                continue
            cam = _vector24(obs.images[int(imgs[0])])
            uv = _lib.stage_project(cam, pts, device_id=self.device).reshape(4, P, 2)
            sigma_px = np.sqrt(((uv[1:] - uv[0]) ** 2).sum(axis=0))  # (P, 2)
            # (a point that starts outside the image has no template: its track fails whatever the workspaces)
            seen = (uv[0] >= 0).all(axis=1) & (uv[0] <= cam[6:8]).all(axis=1) & np.isfinite(sigma_px).all(axis=1)
            if seen.any():
                spread = max(spread, float(sigma_px[seen].max()))
        dim = max(tile_size) + 2 * 5.0 * spread + 8
        return self._fit_dim(max(max(tile_size) + 16, dim), self._dim_limit(len(motion_models)))

    def _frame_feed(self, ctx, matching):
        """Frames the run will touch -> HBM, once (`_FrameFeed`)."""
        previous = getattr(self, "_feed", None)
        if previous is not None:
            previous.close()  # (a run that ended in an exception: its decoders stop here)
        self._feed = _FrameFeed(self, ctx, matching)
        return self._feed

    def _upload_surfaces(self, ctx, motion_models):
        """One gridded dem, one dem_sigma (shared by every model of the batch that uses a raster: `_batches` splits the
        tracks accordingly) and the viewshed."""
        for which, attr in ((_lib.RASTER_DEM, "dem"), (_lib.RASTER_DEM_SIGMA, "dem_sigma")):
            if isinstance(motion_models, ModelBlock):
                ctx.set_raster(which, motion_models.raster(attr))
                continue
            rasters = {}
            if _any_raster_safe(motion_models, attr):  # (checked on the attribute types first: thousands of models)
                rasters = {id(getattr(m, attr)): getattr(m, attr) for m in motion_models
                           if isinstance(getattr(m, attr), Raster)}
            assert len(rasters) <= 1, "a batch shares one raster per surface"
            ctx.set_raster(which, next(iter(rasters.values())) if rasters else None)
        ctx.set_raster(_lib.RASTER_VIEWSHED, self.viewshed)

    # ---- the tracking loop (tracker.py:225-417) ------------------------------------------------
    def track(self, motion_models, datetimes=None, maxdt=datetime.timedelta(0), tile_size=(15, 15),
              observer_mask=None, return_covariances=False, return_particles=False, reduce_particles=None,
              parallel=False, rng="numpy", seed=0, point_offset=0, _catch_errors=None):
        if reduce_particles:
            return_particles = True
        params = dict(motion_models=motion_models, datetimes=datetimes, maxdt=maxdt, tile_size=tile_size,
                      observer_mask=observer_mask, return_covariances=return_covariances,
                      return_particles=return_particles, reduce_particles=reduce_particles, parallel=parallel)
        block = isinstance(motion_models, ModelBlock)  # (a worker's tracks as one parameter table: checked where it was made)
        time_unit = motion_models.time_unit if block else motion_models[0].time_unit
        if not block and len(set(map(_GET_TIME_UNIT, motion_models))) > 1:  # (equal timedeltas hash alike: one pass at C speed)
            raise ValueError("Motion models must have equal time units")
        if not block and not set(map(type, motion_models)) <= set(_DEVICE_MODELS):
            for model in motion_models:
                if not _on_device(model) and not (callable(getattr(model, "initialize_particles", None))
                                                  and callable(getattr(model, "evolve_particles", None))):
                    raise TypeError(f"{type(model).__name__} is not a motion model: it needs initialize_particles() "
                                    "and evolve_particles(particles, dt) (motion.py:13-89)")
        self.reset()
        ntracks = len(motion_models)
        raise_errors = ntracks < 2 if _catch_errors is None else not _catch_errors
        workers = self._parse_parallel(parallel, ntracks)
        if workers > 1:
            return self._track_parallel(workers, motion_models, params, datetimes=datetimes, maxdt=maxdt,
                                        tile_size=tile_size, observer_mask=observer_mask,
                                        return_covariances=return_covariances, return_particles=return_particles,
                                        reduce_particles=reduce_particles, rng=rng, seed=seed, point_offset=point_offset)
        n = motion_models[0].n
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        # tracker.py:199-201 draws n - sum(repetitions) uniforms: how far the legacy global stream advances at every
        # frame of a track depends on that track's weights, so the stream cannot be staged ahead for a batch of tracks
        # -- with np.random the tracks then run one after another, like the reference runs them (the device RNG has no
        # such coupling and keeps the batch)
        serial = self.resample_method == "residual" and rng == "numpy"
        if serial or len(_batches(motion_models)) > 1 or not _on_device(motion_models[0]):
            # Motion models that cannot share one batch -- different particle counts (each track of the reference has
            # its own n, tracker.py:305-314), their own dem / dem_sigma rasters (motion.py:136-141), user-defined
            # models: consecutive compatible models form one batch, the batches run in order -- so the legacy
            # np.random stream is consumed track after track like the reference -- and are merged.
            return self._track_runs(motion_models, params, datetimes=datetimes, maxdt=maxdt, tile_size=tile_size,
                                    observer_mask=observer_mask, return_covariances=return_covariances,
                                    return_particles=return_particles, reduce_particles=reduce_particles, rng=rng,
                                    seed=seed, point_offset=point_offset, serial=serial)
        if datetimes is None:
            datetimes = self.datetimes
        else:
            datetimes = self.parse_datetimes(datetimes=datetimes, maxdt=maxdt)
        nobs = len(self.observers)
        if observer_mask is None:
            observer_mask = np.ones((ntracks, nobs), dtype=bool)
        observer_mask = np.asarray(observer_mask, dtype=bool)
        matching = self.match_datetimes(datetimes=datetimes, maxdt=maxdt)
        has = np.not_equal(matching, None)
        template_indices = has.argmax(axis=0)
        ntimes = len(datetimes)
        dts = np.diff(datetimes)
        taus = np.array([dt.total_seconds() / time_unit.total_seconds() for dt in dts])
        # per-track [first, last] window (tracker.py:321-325)
        observed = (has[None, :, :] & observer_mask[:, None, :]).any(axis=2)  # (P, T)
        first = observed.argmax(axis=1)
        last = ntimes - 1 - observed[:, ::-1].argmax(axis=1)
        empty = ~observed.any(axis=1)
        first[empty], last[empty] = 0, -1

        # search-tile workspaces: the caller's size, or a guess from the prior that grows (and re-runs) on demand
        dim = self.max_search_dim
        if dim is None and self._ctx is not None and self._ctx_key[:4] == (ntracks, n, ntimes, max(tile_size)):
            dim = self._ctx_key[4]  # the workspaces that served the last run of this shape
        elif dim is None:
            dim = max(self._estimate_search_dim(motion_models, matching, taus, tile_size), max(31, max(tile_size)) + 16)
        ctx = self._context(ntracks, n, ntimes, tile_size, dim)
        outgrown = [False]  # a search tile did not fit the workspaces (this attempt)
        uniform = bool(observer_mask.all()) and bool((first == first[0]).all()) and bool((last == last[0]).all())

        warn_log = {}  # track -> its warnings (most tracks have none)
        images_of = lambda i: [m if m is not None else -1 for m in matching[i]]  # noqa: E731

        def set_active(mask):
            ctx.set_active(None if (uniform and mask.all()) else mask.astype(np.uint8))

        lo, hi = int(first[~empty].min()) if (~empty).any() else 0, int(last.max())
        method = self.resample_method
        systematic = method == "systematic"

        def run(draws):
            """The frame loop (tracker.py:326-357) for all tracks at once."""
            ctx.begin_sequence(ntracks, n, tile_size)
            feed = self._frame_feed(ctx, matching)  # (frames from files are decoded while the frames before them are tracked)
            self._upload_surfaces(ctx, motion_models)
            ctx.set_motion(params_table(motion_models))
            ctx.set_observer_mask(None if observer_mask.all() else observer_mask.astype(np.uint8))
            ctx.set_point_offset(point_offset)
            # device-RNG runs have no reference stream to be bit-exact with: fast arithmetic (GLH_MATH_FAST)
            ctx.set_math("fast" if draws is None else "exact")
            ctx.track_covariances(bool(return_covariances))  # (runs of frames in one call record them on the way)
            warn_log.clear()
            out_p = np.full((ntracks, ntimes, n, 6), np.nan) if return_particles else None
            out_w = np.full((ntracks, ntimes, n), np.nan) if return_particles else None
            def note_skips(running, status):
                for o in range(nobs):
                    for p in np.nonzero(running & (status[o] == _lib.OBS_OUT_OF_BOUNDS))[0]:
                        warn_log.setdefault(int(p), []).append(UserWarning(_OOB_WARNING))
                    for p in np.nonzero(running & (status[o] == _lib.OBS_TILE_TOO_LARGE))[0]:
                        outgrown[0] = True
                        warn_log.setdefault(int(p), []).append(RuntimeWarning(
                            f"search tile exceeds max_search_dim={dim}; observer {o} skipped"))

            def common(i):
                """Every track is running and no template starts at frame i: ONE fused launch does evolve +
                likelihood + resample + moments (glh_step)."""
                return (uniform and systematic and bool(((first < i) & (i <= last)).all())
                        and not (template_indices == i).any())

            i = lo
            deferred = []  # runs of common frames whose status words are read after the loop
            while i <= hi:
                through = feed.need(i)  # the frames of time steps <= through are resident
                ctx.set_frame(i)
                starting = (first == i) & ~empty
                running = (first < i) & (i <= last)
                window = starting | running
                if common(i):
                    set_active(window)
                    if draws is None and not return_particles:
                        # device RNG: the whole run of common frames in one call (glh_track: the launches are
                        # enqueued back to back, no host round trip per frame) -- as far as the frames are resident:
                        # a sequence read from files is tracked while its later frames are still being decoded.  Every
                        # frame keeps its own status words, so the per-frame warnings are read afterwards.
                        j = i
                        while j + 1 <= min(hi, through) and common(j + 1):
                            j += 1
                        ctx.track(list(range(i, j + 1)), taus[i - 1:j], [images_of(k) for k in range(i, j + 1)], seed=seed)
                        feed.stats["track_calls"] = feed.stats.get("track_calls", 0) + 1
                        deferred.append((i, j))
                        i = j + 1
                        continue
                    if draws is None:
                        ctx.step(i, taus[i - 1], images_of(i), seed=seed)
                    else:
                        ctx.step(i, taus[i - 1], images_of(i), normals=draws["evolve"][i], u=draws["u"][i])
                    note_skips(running, ctx.observer_status())
                else:
                    if starting.any():
                        set_active(starting)
                        if draws is None:
                            ctx.init_particles(seed=seed)
                        else:
                            ctx.init_particles(normals=draws["init"])
                    if running.any():
                        set_active(running)
                        if draws is None:
                            ctx.evolve(taus[i - 1], seed=seed, step=i)
                        else:
                            ctx.evolve(taus[i - 1], normals=draws["evolve"][i])
                    set_active(window)
                    for o in np.nonzero(template_indices == i)[0]:
                        if has[i, o]:
                            ctx.init_templates(int(o), int(matching[i][o]))
                    if running.any():
                        set_active(running)
                        ctx.update_weights(images_of(i))
                        note_skips(running, ctx.observer_status())
                        if draws is None:
                            ctx.resample(seed=seed, step=i, method=method)
                        else:
                            ctx.resample(u=draws["u"][i], method=method)
                    set_active(window)
                    ctx.record_moments(i)
                if return_covariances:
                    set_active(window)
                    ctx.record_covariances(i)
                if return_particles:
                    P_, W_ = ctx.get_particles(), ctx.get_weights()
                    out_p[window, i] = P_[window]
                    out_w[window, i] = W_[window]
                i += 1
            feed.close()
            for a, b in deferred:
                statuses = ctx.observer_status_frames(a, b - a + 1)
                # (one test for the whole run; the per-frame bookkeeping only where something was skipped)
                skipped = (statuses == _lib.OBS_OUT_OF_BOUNDS) | (statuses == _lib.OBS_TILE_TOO_LARGE)
                for k in np.nonzero(skipped.any(axis=(1, 2)))[0]:
                    note_skips(np.ones(ntracks, dtype=bool), statuses[k])  # (a common frame: every track is running)
            return out_p, out_w, ctx.point_status(), ctx.point_error_frame()

        state0 = np.random.get_state() if rng == "numpy" else None
        while True:
            outgrown[0] = False
            out_particles, out_weights, status, err_frame = self._attempt(run, rng, state0, ntracks, n, first, last,
                                                                          systematic, motion_models)
            if not outgrown[0] or self.max_search_dim is not None:
                break
            # an automatic workspace was too small for some search tile (that observer was skipped on that frame):
            # nothing of this attempt is kept -- same draws, larger workspaces, within what the kernels take and the
            # device's memory allows (beyond that the run stands as it is, with its 'observer skipped' warnings).  Only
            # the context of this shape remembers the size: a diverging track does not enlarge later, unrelated runs.
            grown = self._fit_dim(2 * dim, self._dim_limit(ntracks))
            if grown <= dim:
                break
            try:
                # (the context in use stays until the larger one exists: if that cannot be made, its results stand)
                bigger = self._context(ntracks, n, ntimes, tile_size, grown, keep_old=True)
            except (_lib.GlhError, MemoryError):
                break
            ctx, dim = bigger, grown

        means, sigmas = ctx.get_tracks(0, ntimes)  # (P, T, 6) each, laid out on the device
        covariances = None
        if return_covariances:  # tracker.py:307-308, :352: (P, T, 6, 6) instead of sigmas
            covariances = np.ascontiguousarray(np.transpose(ctx.get_covariances(0, ntimes), (1, 0, 2, 3)))
        errors = [None] * ntracks
        for p in np.nonzero(status)[0]:
            if status[p]:
                for bit, cls, msg in _ERRORS:
                    if status[p] & bit:
                        errors[p] = cls(msg)
                        break
                e = int(err_frame[p])
                means[p, e:] = np.nan
                sigmas[p, e:] = np.nan
                if covariances is not None:
                    covariances[p, e:] = np.nan
                if return_particles:
                    out_particles[p, e:] = np.nan
                    out_weights[p, e:] = np.nan
        if raise_errors and errors[0] is not None:
            raise errors[0]
        # single-track state, like the reference leaves it after the last track
        self._particles = self._weights = None
        self._last_state = (ctx, ntracks - 1)  # fetched when `particles` / `weights` are first read
        warnings = [None] * ntracks
        for p_, w_ in warn_log.items():
            warnings[p_] = tuple(w_)
        kwargs = dict(time_unit=time_unit, datetimes=datetimes, means=means,
                      sigmas=None if return_covariances else sigmas, covariances=covariances,
                      particles=None if reduce_particles else out_particles,
                      weights=None if reduce_particles else out_weights, tracker=self, images=matching,
                      params=params, errors=errors,
                      warnings=warnings)
        tracks = Tracks(**kwargs)
        if reduce_particles:
            tracks.reduced = [reduce_particles(out_particles[p], out_weights[p]) for p in range(ntracks)]
        return tracks

    def _attempt(self, run, rng, state0, ntracks, n, first, last, systematic, motion_models):
        """One pass over the sequence: `run(None)` on the device RNG, or the replay loop on the np.random stream."""
        if rng == "philox":
            return run(None)
        # The reference stops drawing for a track at the frame where it fails, which shifts
        # the stream of the tracks after it: replay until the assumed consumption is consistent.
        stops = np.stack((last, last, last), axis=1)
        for _ in range(ntracks + 1):
            np.random.set_state(state0)
            draws = self._draw_numpy(ntracks, n, first, last, stops, per_particle_u=not systematic,
                                     models=motion_models)
            out = run(draws)
            status, err_frame = out[2], out[3]
            new_stops = np.stack((last, last, last), axis=1)
            for p in np.nonzero(status)[0]:
                e = int(err_frame[p])
                # a dem / dem_sigma raster that does not cover the initial positions raises inside
                # initialize_particles right after randn(n, 2) (motion.py:158): nothing else is drawn
                gridded = isinstance(motion_models[p].dem, Raster) or isinstance(motion_models[p].dem_sigma, Raster)
                if e == first[p] and status[p] & _lib.PT_RASTER_OOB and gridded:
                    new_stops[p, 2] = -1
                new_stops[p, 0] = e
                new_stops[p, 1] = e if (status[p] & _lib.PT_RESAMPLE_CLAMP and not status[p] & 0x77) else e - 1
            if (new_stops == stops).all():
                break
            stops = new_stops
        return out

    # ---- parallel=N: N worker processes, one GPU each (the reference's process pool, tracker.py:381-387) ----------
    @staticmethod
    def _parse_parallel(parallel, ntracks):
        """helpers._parse_parallel (helpers.py:2008-2017) with GPUs for CPUs: True = one worker per visible GPU, an
        int = that many workers (they share GPUs round-robin when there are fewer), False / 0 / 1 = this process."""
        if parallel is True:
            n = _lib.device_count()
        elif not parallel:
            n = 0
        else:
            n = int(parallel)
        return max(0, min(n, ntracks))

    def _track_parallel(self, workers, motion_models, params, observer_mask=None, rng="numpy", seed=0, point_offset=0,
                        **kw):
        """Tracks are independent (the reference maps `process` over them, tracker.py:381-387): contiguous blocks of
        tracks go to `workers` PERSISTENT processes (glimpse_amd/parallel.py: started at the first parallel call, one
        context each on GPU (worker mod device count), reused by later calls), each with `point_offset` = its first
        track, so a device-RNG run draws exactly what the single-process run draws.  The frames reach the workers once,
        through shared memory (image.py:209 `sharedmem.copy` in the reference), never through pickles.  The posterior
        history is collected on worker 0 by ONE RCCL exchange (`glh_gather_moments`) when the workers can make the
        communicator, through host memory otherwise -- `Tracks.transport` says which.  With rng="numpy" every worker gets
        its own np.random seed (drawn here from the global stream): like the reference's pool, a parallel run is not
        stream-compatible with a serial one."""
        import time

        from . import parallel, sharding

        t_start = time.perf_counter()
        ntracks = len(motion_models)
        ndev = max(1, _lib.device_count())
        # (objects defined in the caller's main script can only be unpickled by workers that run that script again -- which
        # then needs its `if __name__ == "__main__":` guard; everything else starts workers that do not)
        from_main = any(getattr(type(m), "__module__", "") == "__main__" for m in motion_models) or \
            getattr(kw.get("reduce_particles"), "__module__", "") == "__main__"
        pool = getattr(self, "_pool", None)
        if pool is None or pool.n != workers or not pool.alive() or (from_main and not pool.import_main):
            if pool is not None:
                pool.close()
            pool = self._pool = parallel.WorkerPool(workers, [w % ndev for w in range(workers)], import_main=from_main)
        shared = pool.share(self.observers)
        mask = None if observer_mask is None else np.asarray(observer_mask, dtype=bool)
        seeds = np.random.randint(0, 2 ** 31 - 1, size=workers) if rng == "numpy" else [None] * workers
        bounds = [sharding.shard_range(ntracks, workers, w) for w in range(workers)]  # (workers <= ntracks: none empty)
        sizes = [b - a for a, b in bounds]
        # one device batch per block (the usual case): the block's history lies in ONE context and can be gathered there;
        # blocks that run as several batches (ragged particle counts, user-defined models, the serial residual stream)
        # hand their host results over instead
        serial = self.resample_method == "residual" and rng == "numpy"
        gather = not serial and all(_batches(motion_models[a:b]) == [0] and _on_device(motion_models[a])
                                    for a, b in bounds)
        settings = dict(viewshed=self.viewshed, resample_method=self.resample_method, highpass=self.highpass,
                        interpolation=self.interpolation, max_search_dim=self.max_search_dim)
        # where the history goes: one shared-memory block (tracks, times, 12) the workers write their rows into -- when the
        # number of time steps is known here (datetimes given or the observers' own: what the workers will find too)
        result = None
        if gather:
            try:
                dts = self.datetimes if kw.get("datetimes") is None else \
                    self.parse_datetimes(datetimes=kw["datetimes"], maxdt=kw.get("maxdt", datetime.timedelta(0)))
                result = pool.result_block((ntracks, len(dts), 12))
            except Exception:  # noqa: BLE001  (the workers raise it properly)
                result = None
        # what a worker gets of its tracks: ONE parameter table when the block is one device batch (`gather`: every block
        # is) -- thousands of model objects cost more to pickle and unpickle than their tracks take to run --, the model
        # objects otherwise
        host_only = "" if workers <= ndev else f"{workers} workers share {ndev} GPU{'s' if ndev > 1 else ''}"
        jobs = []
        for w, (a, b) in enumerate(bounds):
            models = ModelBlock.from_models(motion_models[a:b]) if gather else motion_models[a:b]
            jobs.append(dict(tracker=settings, models=models, np_seed=seeds[w], catch=ntracks >= 2,
                             gather=gather, sizes=sizes, call=pool.calls, want_last=w == workers - 1, result=result,
                             host_only=host_only,
                             rows=(a, b),
                             kw=dict(kw, observer_mask=None if mask is None else mask[a:b], rng=rng, seed=seed,
                                     point_offset=point_offset + a)))
        t_ready = time.perf_counter()
        # (the arrays of the Rasters the models and the viewshed bring travel through shared memory, once: while the jobs
        # are pickled the Rasters hold references instead)
        with pool.rasters.lent(parallel.rasters_of(motion_models, self.viewshed)):
            replies = pool.call("track", jobs)
        t_replied = time.perf_counter()
        parts = [{k: [parallel._import(x) for x in v] if isinstance(v, list) else parallel._import(v)
                  for k, v in part.items()} for part in replies]
        t_imported = time.perf_counter()

        def cat(name):
            values = [part[name] for part in parts]
            if values[0] is None:
                return None
            if all(isinstance(v, np.ndarray) for v in values) and len({v.shape[1:] for v in values}) == 1:
                return np.concatenate(values, axis=0)
            return [row for v in values for row in v]

        errors = cat("errors")
        transport = parts[0]["transport"] if gather else "host"
        in_block = [part.get("in_block") for part in parts]
        want_sigmas = not kw.get("return_covariances")
        # (a view of the shared block: means / sigmas below are the copies that leave this function)
        full = pool.result_view(result[1]) if result is not None and any(in_block) else None  # (tracks, times, 12)
        if transport == "rccl":
            # worker 0 received every worker's history (T, sum P, 12) and wrote it into the block (or sent it, when there was
            # no block); rows of a failed track are NaN from the frame where it failed -- what the workers' own copies hold
            # from their status words is repeated here
            if in_block[0] != "all":
                full = np.ascontiguousarray(np.transpose(parts[0]["gathered"], (1, 0, 2)))
            means = np.ascontiguousarray(full[:, :, 0:6])
            sigmas = np.ascontiguousarray(full[:, :, 6:12]) if want_sigmas else None
            lo = 0
            for part, n in zip(parts, sizes):
                for p, e in part.get("nan_from", ()):
                    means[lo + p, e:] = np.nan
                    if sigmas is not None:
                        sigmas[lo + p, e:] = np.nan
                lo += n
        elif full is not None and all(tag == "rows" for tag in in_block):
            # every worker wrote its rows into the block: one copy out of it per array
            means = np.ascontiguousarray(full[:, :, 0:6])
            sigmas = np.ascontiguousarray(full[:, :, 6:12]) if want_sigmas else None
        else:
            mparts, sparts = [], []
            for part, (a, b) in zip(parts, bounds):
                if part.get("in_block") == "rows":
                    mparts.append(np.ascontiguousarray(full[a:b, :, 0:6]))
                    sparts.append(np.ascontiguousarray(full[a:b, :, 6:12]) if want_sigmas else None)
                else:
                    mparts.append(part["means"])
                    sparts.append(part["sigmas"])

            def join(values):
                if values[0] is None:
                    return None
                if all(isinstance(v, np.ndarray) for v in values) and len({v.shape[1:] for v in values}) == 1:
                    return np.concatenate(values, axis=0)
                return [row for v in values for row in v]

            means, sigmas = join(mparts), join(sparts)
        if gather and transport != "rccl":
            why = next((part["why_host"] for part in parts if part.get("why_host")), "")
            parallel.log.warning("Tracker.track(parallel=%d): no RCCL communicator (%s); the posterior history was "
                                 "collected through host memory", workers, why or "unavailable")
        if ntracks < 2 and errors[0] is not None:
            raise errors[0]
        self.particles, self.weights = parts[-1]["last_particles"], parts[-1]["last_weights"]
        tracks = Tracks(datetimes=parts[0]["datetimes"], time_unit=parts[0]["time_unit"], means=means,
                        sigmas=sigmas, covariances=cat("covariances"), particles=cat("particles"),
                        weights=cat("weights"), tracker=self, images=parts[0]["images"], params=params, errors=errors,
                        warnings=cat("warnings"))
        tracks.transport = transport
        tracks.parallel_info = dict(workers=workers, transport=transport, frames_shared_now=shared,
                                    shared_frame_bytes=pool.frames.nbytes() if pool.frames else 0,
                                    contexts_made=[bool(part["context_made"]) for part in parts],
                                    worker_seconds=[part["seconds"] for part in parts],
                                    worker_track_seconds=[part["track_seconds"] for part in parts],
                                    call_seconds=time.perf_counter() - t_start,
                                    parent_seconds=dict(prepare=t_ready - t_start, workers=t_replied - t_ready,
                                                        import_arrays=t_imported - t_replied,
                                                        assemble=time.perf_counter() - t_imported))
        if kw.get("reduce_particles"):
            tracks.reduced = [r for part in parts for r in part["reduced"]]
        return tracks

    def forget_frames(self):
        """The next run reads and uploads every frame again (the files or arrays behind the images have changed); the
        next parallel run shares them with its workers again."""
        self._uploaded = set()
        pool = getattr(self, "_pool", None)
        if pool is not None and pool.frames is not None:
            pool.frames.key = None

    def close(self):
        """Release the device contexts and the worker processes of `track(parallel=N)` (they are kept between calls)."""
        if getattr(self, "_feed", None) is not None:
            self._feed.close()
            self._feed = None
        if getattr(self, "_decoders", None) is not None:
            self._decoders.close()
            self._decoders = None
        pool, self._pool = getattr(self, "_pool", None), None
        if pool is not None:
            pool.close()
        for name in ("_ctx", "_sctx"):
            ctx = getattr(self, name, None)
            if ctx is not None:
                setattr(self, name, None)
                ctx.close()
        self._ctx_key = None
        self._sctx_key = None
        self._last_state = None

    def __del__(self):
        try:
            for name in ("_pool", "_decoders"):
                pool = getattr(self, name, None)
                if pool is not None:
                    pool.close()
        except Exception:  # noqa: BLE001
            pass

    def _track_runs(self, motion_models, params, observer_mask=None, reduce_particles=None, point_offset=0,
                    serial=False, **kw):
        """Tracks that cannot share one batch (`_batches`): consecutive compatible device models form a batch, every
        user-defined model is a run of its own (`_track_custom`); the runs go in track order, so np.random is consumed
        like the reference consumes it (one track after another), and are merged.  `serial`: every track is a run of
        its own through the per-track loop (residual resampling on the np.random stream)."""
        ntracks = len(motion_models)
        if observer_mask is not None:
            observer_mask = np.asarray(observer_mask, dtype=bool)
        bounds = list(range(ntracks + 1)) if serial else _batches(motion_models) + [ntracks]
        parts = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            mask = None if observer_mask is None else observer_mask[a:b]
            if serial or not _on_device(motion_models[a]):
                parts.append(self._track_custom(motion_models[a], None if mask is None else mask[0],
                                                reduce_particles=reduce_particles, catch_errors=ntracks >= 2, **kw))
            else:
                parts.append(self.track(motion_models[a:b], observer_mask=mask, reduce_particles=reduce_particles,
                                        point_offset=point_offset + a, _catch_errors=ntracks >= 2, **kw))

        def cat(name):
            values = [getattr(part, name) for part in parts]
            return None if values[0] is None else [row for v in values for row in v]

        first = parts[0]
        tracks = Tracks(datetimes=first.datetimes, time_unit=first.time_unit, means=cat("means"), sigmas=cat("sigmas"),
                        covariances=cat("covariances"), particles=cat("particles"), weights=cat("weights"),
                        tracker=self, images=first.images, params=params, errors=cat("errors"),
                        warnings=cat("warnings"))
        if reduce_particles:
            tracks.reduced = [r for part in parts for r in part.reduced]
        return tracks

    def _track_custom(self, model, mask, datetimes=None, maxdt=datetime.timedelta(0), tile_size=(15, 15),
                      return_covariances=False, return_particles=False, reduce_particles=None, catch_errors=False,
                      rng="numpy", seed=0):
        """One track of a user-defined motion model: the reference's per-track loop (`process`, tracker.py:305-374)
        with the model's own initialize_particles / evolve_particles / compute_log_likelihoods on the host -- they are
        the user's code -- and this class's step methods (templates, observer likelihoods, weights, resampling,
        moments: device kernels) in between.  np.random is consumed exactly as the reference consumes it."""
        if reduce_particles:
            return_particles = True
        datetimes = self.datetimes if datetimes is None else self.parse_datetimes(datetimes=datetimes, maxdt=maxdt)
        nobs = len(self.observers)
        mask = np.ones(nobs, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        matching = self.match_datetimes(datetimes=datetimes, maxdt=maxdt)
        has = np.not_equal(matching, None)
        template_indices = has.argmax(axis=0)
        ntimes = len(datetimes)
        dts = np.diff(datetimes)
        n = int(model.n)
        means = np.full((ntimes, 6), np.nan)
        sigmas = np.full((ntimes, 6, 6) if return_covariances else (ntimes, 6), np.nan)
        particles = np.full((ntimes, n, 6), np.nan) if return_particles else None
        weights = np.full((ntimes, n), np.nan) if return_particles else None
        error, caught = None, []
        self.reset()
        self._single_tile = tuple(int(v) for v in tile_size)
        try:
            with _warnings.catch_warnings(record=True) as caught:
                _warnings.simplefilter("always")
                observed = has[:, mask].any(axis=1)
                first = int(np.argmax(observed))
                last = len(observed) - 1 - int(np.argmax(observed[::-1]))
                for i in range(first, last + 1):
                    if i == first:
                        self.particles = np.array(model.initialize_particles(), dtype=float)
                        self.test_particles()
                        self.initialize_weights()
                    else:
                        model.evolve_particles(self.particles, dt=dts[i - 1])
                        self.test_particles()
                    for obs in np.nonzero(mask & (template_indices == i))[0]:
                        self.initialize_template(obs=int(obs), img=matching[i][obs], tile_size=tile_size)
                    if i > first:
                        imgs = [img if m else None for img, m in zip(matching[i], mask)]
                        self.update_weights(imgs=imgs, motion_model=model)
                        self.resample_particles()
                    means[i] = self.particle_mean
                    sigmas[i] = self.particle_covariance if return_covariances else self.compute_particle_sigma()
                    if return_particles:
                        particles[i], weights[i] = self.particles, self.weights
        except Exception as e:  # noqa: BLE001  (tracker.py:360-368: captured for >= 2 tracks, re-raised for one)
            if not catch_errors:
                raise
            error = e
            first_bad = int(np.argmax(np.isnan(means[:, 0]))) if np.isnan(means[:, 0]).any() else ntimes
            means[first_bad:] = np.nan
            sigmas[first_bad:] = np.nan
        tracks = Tracks(datetimes=datetimes, time_unit=model.time_unit, means=[means],
                        sigmas=None if return_covariances else [sigmas], covariances=[sigmas] if return_covariances else None,
                        particles=None if (reduce_particles or not return_particles) else [particles],
                        weights=None if (reduce_particles or not return_particles) else [weights], tracker=self,
                        images=matching, errors=[error], warnings=[tuple(caught) if caught else None])
        if reduce_particles:
            tracks.reduced = [reduce_particles(particles, weights)]
        return tracks

    @staticmethod
    def _draw_numpy(ntracks, n, first, last, stops, per_particle_u=False, models=None):
        """Consume the legacy global stream exactly like the reference (one track after another):
        randn(n,2), randn(n), randn(n,3), then per step randn(n,3) and random() -- or random(n) for
        stratified resampling (tracker.py:182) and for np.random.choice, whose n uniforms come from the
        same global stream (tracker.py:209).  `stops[p]` = (last frame with an evolve draw, last frame
        with a resample draw) of track p."""
        T = int(max(last.max() + 1, 1))
        nbytes = ntracks * T * n * 3 * 8
        if nbytes > _MAX_HOST_DRAWS_BYTES:
            raise MemoryError(f"rng='numpy' would stage {nbytes / 2**30:.1f} GiB of host draws; use rng='philox'")
        init = np.zeros((ntracks, n, 6))
        evolve = np.zeros((T, ntracks, n, 3))
        u = np.zeros((T, ntracks, n)) if per_particle_u else np.zeros((T, ntracks))
        for p in range(ntracks):
            if last[p] < first[p]:
                continue
            # tangent models draw randn(n,2) velocities and, per step, randn(n,2) then randn(n)
            # (motion.py:393, :404-409); the others randn(n,3) (motion.py:161, :176)
            tangent = models is not None and models[p].TANGENT
            init[p, :, 0:2] = np.random.randn(n, 2)
            if stops.shape[1] > 2 and stops[p, 2] < 0:
                continue  # initialize_particles raised after its first draw (see track())
            init[p, :, 2] = np.random.randn(n)
            if tangent:
                init[p, :, 3:5] = np.random.randn(n, 2)
            else:
                init[p, :, 3:6] = np.random.randn(n, 3)
            for i in range(first[p] + 1, last[p] + 1):
                if i <= stops[p, 0]:
                    if tangent:
                        evolve[i, p, :, 0:2] = np.random.randn(n, 2)
                        evolve[i, p, :, 2] = np.random.randn(n)
                    else:
                        evolve[i, p] = np.random.randn(n, 3)
                if i <= stops[p, 1]:
                    u[i, p] = np.random.random(n) if per_particle_u else np.random.random()
        return {"init": init, "evolve": evolve, "u": u}

    # ---- single-track step methods of the reference's public API ---------------------------------
    # They operate on `self.particles` (n, 6) / `self.weights` (n,) through a one-point device context.
    def _single(self, tile_size=None):
        n = len(self.particles)
        tile = tile_size or getattr(self, "_single_tile", (15, 15))
        key = (n, tuple(tile))
        if getattr(self, "_sctx_key", None) != key:
            if getattr(self, "_sctx", None) is not None:
                self._sctx.close()
            O = len(self.observers)
            ctx = _lib.Context(1, n, O, device_id=self.device, max_tile=max(31, max(tile)),
                               max_search_dim=self.max_search_dim or 320, max_frames=2)
            for o, obs in enumerate(self.observers):
                a0 = obs.images[0].read()
                if a0.dtype not in (np.uint8, np.uint16, np.float32, np.float64) or \
                        (a0.ndim == 3 and a0.shape[2] not in (1, 3)):
                    raise NotImplementedError("frames must be uint8, uint16, float32 or float64 with one or three channels "
                                              f"on the GPU path, not {a0.dtype} {a0.shape}")
                ctx.observer_init(o, len(obs.images), a0.shape[1], a0.shape[0], 1 if a0.ndim == 2 else a0.shape[2],
                                  obs.sigma)
                ctx.observer_set_depth(o, a0.dtype)
                ctx.observer_set_cameras(o, np.stack([_vector24(img) for img in obs.images]))
            ctx.begin_sequence(1, n, tile)
            ctx.set_motion_cartesian(np.zeros((1, _lib.MOTION_LEN)))
            ctx.set_highpass(self._highpass_size, self._highpass_mode)
            ctx.set_interpolation(*self._orders)
            self._sctx, self._sctx_key, self._single_tile, self._s_uploaded = ctx, key, tuple(tile), set()
        return self._sctx

    def _single_upload(self, ctx, obs, img):
        if (obs, img) not in self._s_uploaded:
            ctx.observer_upload_frame(obs, int(img), self.observers[obs].images[img].read())
            self._s_uploaded.add((obs, img))

    def _push(self, ctx):
        ctx.set_particles(np.asarray(self.particles, dtype=float)[None])
        ctx.set_weights(np.asarray(self.weights, dtype=float)[None])

    @property
    def particle_mean(self):
        """tracker.py:72-76 (device reduction)."""
        ctx = self._single()
        self._push(ctx)
        ctx.record_moments(0)
        return ctx.get_moments(0, 1)[0, 0, 0:6]

    def compute_particle_sigma(self, mean=None):
        """tracker.py:89-104."""
        ctx = self._single()
        self._push(ctx)
        ctx.record_moments(0)
        return ctx.get_moments(0, 1)[0, 0, 6:12]

    @property
    def particle_covariance(self):
        """tracker.py:78-82: np.cov(particles.T, aweights=weights, ddof=0) (device reduction)."""
        ctx = self._single()
        self._push(ctx)
        ctx.record_covariances(0)
        return ctx.get_covariances(0, 1)[0, 0]

    def initialize_weights(self):
        """tracker.py:121-124."""
        self.weights = np.ones(len(self.particles))

    def test_particles(self):
        """tracker.py:106-119 (NaN test; no viewshed)."""
        if np.isnan(self.particles).any():
            raise ValueError("Some particles have missing (NaN) values")

    def extract_tile(self, obs, img, box, histogram=None, return_histogram=False):
        """tracker.py:494-534: the grayscale, normalised, (optionally histogram-matched) and median high-passed tile
        of one image box, computed by the same device tile-prep code the frame step uses.  Without `histogram` the
        result is float64 (the template path, bit-equal to the reference's); with one it is the float32 search tile
        the SSD consumes, widened to float64.  8-bit frames only: 16-bit frames are prepared inside the context."""
        frame = self.observers[obs].images[img].read()
        if np.asarray(frame).dtype != np.uint8:
            raise NotImplementedError("extract_tile: standalone tile prep takes uint8 frames; 16-bit frames are "
                                      "prepared on the device inside track()")
        box = np.asarray(box).astype(int)
        if histogram is None:
            tile, own = _lib.stage_template(frame, box, device_id=self.device, highpass=self._highpass_size,
                                            mode=self._highpass_mode)
            return (tile, own) if return_histogram else tile
        if return_histogram:
            raise NotImplementedError("extract_tile: histogram= together with return_histogram=True (no caller on "
                                      "the tracking path asks for the histogram of a matched tile)")
        return _lib.stage_search_tile(frame, box, histogram, device_id=self.device, highpass=self._highpass_size,
                                      mode=self._highpass_mode).astype(float)

    def initialize_template(self, obs, img, tile_size):
        """tracker.py:536-561."""
        ctx = self._single(tuple(int(v) for v in tile_size))
        self._single_upload(ctx, obs, img)
        self._push(ctx)
        ctx.init_templates(obs, img)
        if ctx.point_status()[0] & _lib.PT_TEMPLATE_OOB:
            ctx.begin_sequence(1, len(self.particles), self._single_tile)
            ctx.set_motion_cartesian(np.zeros((1, _lib.MOTION_LEN)))
            raise IndexError("Box extends beyond grid bounds")
        if self.templates is None:
            self.templates = [None] * len(self.observers)
        t = ctx.get_template(obs, 0)
        self.templates[obs] = {"obs": obs, "img": img, **t}

    def update_weights(self, imgs, motion_model=None):
        """tracker.py:126-149.  A device motion model contributes its built-in term (the DEM likelihood of the
        Cartesian / Cylindrical models, none for the tangent ones); a user-defined model's
        compute_log_likelihoods(particles) is evaluated on the host and appended by the device (None = no term)."""
        ctx = self._single()
        params = np.zeros((1, _lib.MOTION_FULL_LEN))
        extra = None
        if motion_model is not None and _on_device(motion_model):
            params[0] = motion_model.params_full()
        elif motion_model is not None:
            params[0, 18] = _lib.MOTION_KINDS["external"]
            # (called unconditionally, like tracker.py:143: a model without the method fails the way it does there)
            extra = motion_model.compute_log_likelihoods(self.particles)
        else:
            params[0, 18] = _lib.MOTION_KINDS["external"]  # no motion model: no term (not an array of zeros)
        ctx.set_motion(params)
        ctx.set_extra_log_likelihoods(None if extra is None else np.asarray(extra, dtype=float)[None])
        for o, img in enumerate(imgs):
            if img is not None:
                self._single_upload(ctx, o, img)
        self._push(ctx)
        ctx.update_weights([-1 if i is None else i for i in imgs])
        st = ctx.observer_status()[:, 0]
        if any(s == _lib.OBS_OUT_OF_BOUNDS for s in st):
            _warnings.warn(_OOB_WARNING)
        if ctx.point_status()[0] & _lib.PT_SAMPLE_OUTSIDE:
            raise ValueError("Some sampling points are outside box")
        has_term = (motion_model is not None and _on_device(motion_model) and not motion_model.TANGENT) or extra is not None
        if has_term or any(s == _lib.OBS_OK for s in st):
            self.weights = ctx.get_weights()[0]

    def compute_observer_log_likelihoods(self, obs, img):
        """tracker.py:563-625: (n,) log likelihoods of `self.particles` for one observer, or None."""
        if img is None:
            return None
        ctx = self._single()
        ctx.set_debug(True)
        ctx.set_motion_cartesian(np.zeros((1, _lib.MOTION_LEN)))
        ctx.set_extra_log_likelihoods(None)
        self._single_upload(ctx, obs, img)
        self._push(ctx)
        imgs = [-1] * len(self.observers)
        imgs[obs] = img
        ctx.update_weights(imgs)
        st = ctx.observer_status()[obs, 0]
        if st == _lib.OBS_OUT_OF_BOUNDS:
            _warnings.warn(_OOB_WARNING)
            return None
        if st != _lib.OBS_OK:
            return None
        if ctx.point_status()[0] & _lib.PT_SAMPLE_OUTSIDE:
            raise ValueError("Some sampling points are outside box")
        return ctx.log_likelihoods(obs)[0]

    def resample_particles(self, method=None):
        """tracker.py:151-223 (systematic, stratified, residual, choice), drawing from the legacy global stream
        like the reference: random() / random(n) / random(n - sum(repetitions)) / the n uniforms of
        np.random.choice."""
        method = method or self.resample_method
        if method not in _lib.RESAMPLE:
            raise ValueError(f"resampling method {method!r}: expected one of {sorted(_lib.RESAMPLE)}")
        ctx = self._single()
        self._push(ctx)
        if method == "systematic":
            ctx.resample(u=np.array([np.random.random()]))
        elif method == "residual":
            # the number of uniforms consumed is known only afterwards: draw n, then rewind and re-draw that many
            state = np.random.get_state()
            ctx.resample(u=np.random.random(len(self.particles))[None], method=method)
            np.random.set_state(state)
            np.random.random(int(ctx.residual_draws()[0]))
        else:
            ctx.resample(u=np.random.random(len(self.particles))[None], method=method)
        self.particles = ctx.get_particles()[0]
        self.weights = ctx.get_weights()[0]
