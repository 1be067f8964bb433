"""ctypes binding of libglimpse_hip.so (include/glimpse_hip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded, every
entry point raises.  Build it with `python -m glimpse_amd.build` (hipcc, gfx950).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLH_LIB") or os.path.join(HERE, "lib", "libglimpse_hip.so")  # GLH_LIB: experimental builds

CAM_LEN = 24
MOTION_LEN = 18
MOTION_FULL_LEN = 24
MOTION_KINDS = {"cartesian": 0, "cylindrical": 1, "tangent_cartesian": 2, "tangent_cylindrical": 3, "external": 4}
RNG_HOST, RNG_PHILOX = 0, 1
MATH_EXACT, MATH_FAST = 0, 1
RESAMPLE = {"systematic": 0, "stratified": 1, "choice": 2, "residual": 3}
OK = 0
PT_NAN, PT_TEMPLATE_OOB, PT_SAMPLE_OUTSIDE, PT_RESAMPLE_CLAMP, PT_CONST_TILE = 1, 2, 4, 8, 16
PT_RASTER_OOB, PT_NOT_VISIBLE = 32, 64
RASTER_DEM, RASTER_DEM_SIGMA, RASTER_VIEWSHED = 0, 1, 2
OBS_OK, OBS_SKIPPED, OBS_OUT_OF_BOUNDS, OBS_TILE_TOO_LARGE, OBS_NO_TEMPLATE = 0, 1, 2, 3, 4
NO_ERROR_FRAME = 0x7F7F7F7F
COMM_ID_BYTES = 128


class GlhError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libglimpse_hip error {code}: {message}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("device_id", C.c_int32),
        ("max_points", C.c_int32),
        ("max_particles", C.c_int32),
        ("n_observers", C.c_int32),
        ("max_tile", C.c_int32),
        ("max_search_dim", C.c_int32),
        ("max_frames", C.c_int32),
        ("reserved", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int
_D = C.c_double
_U64 = C.c_uint64

# name -> (restype, argtypes); mirrors include/glimpse_hip.h one to one
SIGNATURES = {
    "glh_version": (_I, []),
    "glh_last_error": (C.c_char_p, []),
    "glh_device_count": (_I, [_P]),
    "glh_device_compute_units": (_I, [_I, _P]),
    "glh_device_memory": (_I, [_I, _P, _P]),
    "glh_create": (_I, [_P, _P]),
    "glh_destroy": (_I, [_P]),
    "glh_sync": (_I, [_P]),
    "glh_get_stream": (_I, [_P, _P]),
    "glh_observer_init": (_I, [_P, _I, _I, _I, _I, _I, _D]),
    "glh_observer_set_cameras": (_I, [_P, _I, _I, _I, _P]),
    "glh_observer_set_depth": (_I, [_P, _I, _I]),
    "glh_observer_upload_frame": (_I, [_P, _I, _I, _P]),
    "glh_observer_upload_frame_async": (_I, [_P, _I, _I, _P]),
    "glh_host_register": (_I, [_P, _U64]),
    "glh_host_unregister": (_I, [_P]),
    "glh_observer_upload_frame_pinned": (_I, [_P, _I, _I, _P, _P]),
    "glh_upload_done": (_I, [_P, C.c_int64, _I, _P]),
    "glh_observer_set_frame_device": (_I, [_P, _I, _I, _P]),
    "glh_begin_sequence": (_I, [_P, _I, _I, _I, _I]),
    "glh_set_motion_cartesian": (_I, [_P, _P]),
    "glh_set_motion": (_I, [_P, _P]),
    "glh_set_raster": (_I, [_P, _I, _P, _I, _I, _P, _P, _I, _I, _D, _D, _D, _D]),
    "glh_set_point_offset": (_I, [_P, _I]),
    "glh_set_observer_mask": (_I, [_P, _P]),
    "glh_set_active": (_I, [_P, _P]),
    "glh_set_particles": (_I, [_P, _P]),
    "glh_get_particles": (_I, [_P, _P]),
    "glh_set_weights": (_I, [_P, _P]),
    "glh_get_weights": (_I, [_P, _P]),
    "glh_set_extra_log_likelihoods": (_I, [_P, _P]),
    "glh_get_point_status": (_I, [_P, _P]),
    "glh_get_point_error_frame": (_I, [_P, _P]),
    "glh_get_observer_status": (_I, [_P, _P]),
    "glh_get_search_boxes": (_I, [_P, _P]),
    "glh_get_observer_status_frames": (_I, [_P, _I, _I, _P]),
    "glh_get_point_state": (_I, [_P, _I, _P, _P]),
    "glh_set_frame": (_I, [_P, _I]),
    "glh_init_particles": (_I, [_P, _I, _P, _U64]),
    "glh_evolve": (_I, [_P, _D, _I, _P, _U64, _U64]),
    "glh_init_templates": (_I, [_P, _I, _I]),
    "glh_update_weights": (_I, [_P, _P]),
    "glh_resample": (_I, [_P, _I, _P, _U64, _U64]),
    "glh_resample_method": (_I, [_P, _I, _I, _P, _U64, _U64]),
    "glh_get_residual_draws": (_I, [_P, _P]),
    "glh_record_covariances": (_I, [_P, _I]),
    "glh_get_covariances": (_I, [_P, _I, _I, _P]),
    "glh_record_moments": (_I, [_P, _I]),
    "glh_step": (_I, [_P, _I, _D, _P, _I, _P, _P, _U64]),
    "glh_track": (_I, [_P, _I, _P, _P, _P, _U64]),
    "glh_track_covariances": (_I, [_P, _I]),
    "glh_set_track_streams": (_I, [_P, _I]),
    "glh_set_fused": (_I, [_P, _I]),
    "glh_set_math": (_I, [_P, _I]),
    "glh_set_highpass": (_I, [_P, _I, _I]),
    "glh_set_highpass_mode": (_I, [_P, _I]),
    "glh_set_interpolation": (_I, [_P, _I, _I]),
    "glh_debug_phase_stamps": (_I, [_P, _P]),
    "glh_debug_draws": (_I, [_P, _I, _U64, _U64, _P]),
    "glh_debug_last_variant": (_I, [_P, _P]),
    "glh_debug_last_track_streams": (_I, [_P, _P]),
    "glh_get_moments": (_I, [_P, _I, _I, _P]),
    "glh_get_tracks": (_I, [_P, _I, _I, _P, _P]),
    "glh_get_moments_device": (_I, [_P, _P, _P]),
    "glh_get_template": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "glh_get_likelihood_debug": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "glh_set_debug": (_I, [_P, _I]),
    "glh_get_resample_indices": (_I, [_P, _P]),
    "glh_get_log_likelihoods": (_I, [_P, _I, _P]),
    "glh_profile_enable": (_I, [_P, _I]),
    "glh_profile_reset": (_I, [_P]),
    "glh_stage_count": (_I, []),
    "glh_stage_name": (C.c_char_p, [_I]),
    "glh_profile_get": (_I, [_P, _P, _P]),
    "glh_profile_get_launches": (_I, [_P, _I, _P, _I, _P]),
    "glh_profile_get_span": (_I, [_P, _I, _P]),
    "glh_comm_unique_id": (_I, [_P]),
    "glh_comm_init": (_I, [_P, _P, _I, _I]),
    "glh_comm_destroy": (_I, [_P]),
    "glh_comm_barrier": (_I, [_P]),
    "glh_comm_max_f64": (_I, [_P, _P]),
    "glh_gather_moments": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "glh_get_gathered": (_I, [_P, _P, _P]),
    "glh_measure_copy_bandwidth": (_I, [_P, _U64, _I, _P]),
    "glh_stage_project": (_I, [_I, _P, _P, _I, _P]),
    "glh_stage_project_directions": (_I, [_I, _P, _P, _I, _P]),
    "glh_stage_project_depth": (_I, [_I, _P, _P, _I, _I, _P, _P]),
    "glh_stage_unproject": (_I, [_I, _P, _P, _I, _P, _I, _I, _P]),
    "glh_stage_template": (_I, [_I, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "glh_stage_search_tile": (_I, [_I, _P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "glh_stage_template_highpass": (_I, [_I, _P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P]),
    "glh_stage_search_tile_highpass": (_I, [_I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "glh_stage_ssd": (_I, [_I, _P, _I, _I, _P, _I, _I, _P]),
    "glh_stage_sample": (_I, [_I, _P, _I, _I, _P, _P, _I, _P, _P]),
    "glh_stage_sample_orders": (_I, [_I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    "glh_stage_resample": (_I, [_I, _P, _I, _D, _P]),
    "glh_stage_raster_sample": (_I, [_I, _P, _I, _I, _P, _P, _I, _I, _D, _D, _D, _D, _P, _I, _I, _P, _P]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GlhError(
            -100,
            f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
            "Build it with `python -m glimpse_amd.build`.",
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table ever diverge
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        raise GlhError(rc, load().glh_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def host_register(address, nbytes):
    """Page-lock `nbytes` of host memory at `address` for the device (hipHostRegister): uploads from it need no staging
    copy (Context.observer_upload_frame_pinned)."""
    check(load().glh_host_register(C.c_void_p(int(address)), int(nbytes)))


def host_unregister(address):
    check(load().glh_host_unregister(C.c_void_p(int(address))))


def device_count():
    n = C.c_int(0)
    check(load().glh_device_count(C.byref(n)))
    return n.value


def device_compute_units(device_id=0):
    n = C.c_int(0)
    check(load().glh_device_compute_units(int(device_id), C.byref(n)))
    return n.value


def device_memory(device_id=0):
    """(free, total) bytes of a device (hipMemGetInfo)."""
    free, total = C.c_uint64(0), C.c_uint64(0)
    check(load().glh_device_memory(int(device_id), C.byref(free), C.byref(total)))
    return free.value, total.value


def comm_unique_id():
    """128-byte RCCL communicator id (ncclGetUniqueId): made by one rank, handed to the others."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(load().glh_comm_unique_id(buf))
    return buf.raw


def stage_names():
    lib = load()
    return [lib.glh_stage_name(i).decode() for i in range(lib.glh_stage_count())]


class Context:
    """One device context = one GPU, one HIP stream, resident frames and particle state."""

    def __init__(self, max_points, max_particles, n_observers=1, device_id=0, max_tile=31,
                 max_search_dim=320, max_frames=128):
        self.lib = load()
        self.cfg = Config(device_id, max_points, max_particles, n_observers, max_tile, max_search_dim,
                          max_frames, 0)
        self.handle = C.c_void_p()
        check(self.lib.glh_create(C.byref(self.cfg), C.byref(self.handle)))
        self.O = n_observers
        self.P = self.N = 0
        self.tile = (0, 0)
        self.rank, self.world = 0, 1
        self._keep = []  # device-borrowed frame owners
        self._frame_shape = {}  # observer -> (height, width, channels) the library copies per frame
        self._frame_dtype = {}  # observer -> sample dtype (uint8, or uint16 after observer_set_depth)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.glh_destroy(self.handle)
            self.handle = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- observers
    def observer_init(self, obs, n_images, width, height, channels, sigma):
        check(self.lib.glh_observer_init(self.handle, obs, n_images, width, height, channels, float(sigma)))
        self._frame_shape[obs] = (int(height), int(width), int(channels))
        self._frame_dtype[obs] = np.dtype(np.uint8)

    def set_interpolation(self, kx=3, ky=3):
        """Orders of the surface-sampling spline (rows axis, columns axis), each 1 .. 5; (3, 3) by default."""
        check(self.lib.glh_set_interpolation(self.handle, int(kx), int(ky)))

    def observer_set_depth(self, obs, dtype):
        """Sample type of the observer's frames: uint8 (default), uint16, float32 or float64 (one or three channels);
        before the first upload."""
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.uint8), np.dtype(np.uint16), np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError(f"frames are uint8, uint16, float32 or float64 (got {dtype})")
        check(self.lib.glh_observer_set_depth(self.handle, obs, 8 * dtype.itemsize))
        self._frame_dtype[obs] = dtype

    def _frame(self, obs, pixels):
        """The C side copies width * height * channels bytes: refuse anything that is not exactly that."""
        a = np.asarray(pixels)
        if a.dtype != self._frame_dtype[obs]:
            raise TypeError(f"observer {obs}: frames are {self._frame_dtype[obs]} (got {a.dtype}); the library never "
                            "casts pixel data")
        h, w, ch = self._frame_shape[obs]
        want = (h, w) if ch == 1 else (h, w, ch)
        if a.shape != want and not (ch == 1 and a.shape == (h, w, 1)):
            raise ValueError(f"observer {obs}: frame shape {a.shape} != {want} declared by observer_init")
        return np.ascontiguousarray(a)

    def observer_set_cameras(self, obs, cams, first=0):
        cams = _arr(cams, np.float64)
        assert cams.ndim == 2 and cams.shape[1] == CAM_LEN
        check(self.lib.glh_observer_set_cameras(self.handle, obs, first, len(cams), _ptr(cams)))

    def observer_upload_frame(self, obs, image, pixels):
        pixels = self._frame(obs, pixels)
        check(self.lib.glh_observer_upload_frame(self.handle, obs, image, _ptr(pixels)))

    def observer_upload_frame_async(self, obs, image, pixels):
        """Upload without waiting for the device; `pixels` may be reused as soon as the call returns."""
        pixels = self._frame(obs, pixels)
        check(self.lib.glh_observer_upload_frame_async(self.handle, obs, image, _ptr(pixels)))

    def observer_upload_frame_pinned(self, obs, image, pixels):
        """Upload straight from REGISTERED host memory (`host_register`): no staging copy.  Returns the ticket of the copy;
        `pixels` must stay untouched until `upload_done(ticket)`."""
        pixels = self._frame(obs, pixels)
        ticket = C.c_int64(0)
        check(self.lib.glh_observer_upload_frame_pinned(self.handle, obs, image, _ptr(pixels), C.byref(ticket)))
        return ticket.value

    def upload_done(self, ticket, wait=False):
        done = C.c_int(0)
        check(self.lib.glh_upload_done(self.handle, int(ticket), 1 if wait else 0, C.byref(done)))
        return bool(done.value)

    def observer_set_frame_device(self, obs, image, dev_ptr, owner=None):
        if owner is not None:
            self._keep.append(owner)
        check(self.lib.glh_observer_set_frame_device(self.handle, obs, image, C.c_void_p(dev_ptr)))

    # ---- sequence
    def begin_sequence(self, n_points, n_particles, tile_size):
        tw, th = int(tile_size[0]), int(tile_size[1])
        check(self.lib.glh_begin_sequence(self.handle, n_points, n_particles, tw, th))
        self.P, self.N, self.tile = n_points, n_particles, (tw, th)

    def set_motion_cartesian(self, params):
        params = _arr(params, np.float64, (self.P, MOTION_LEN))
        check(self.lib.glh_set_motion_cartesian(self.handle, _ptr(params)))

    def set_motion(self, params):
        """Any mix of motion models: [P][MOTION_FULL_LEN] (include/glimpse_hip.h)."""
        params = _arr(params, np.float64, (self.P, MOTION_FULL_LEN))
        check(self.lib.glh_set_motion(self.handle, _ptr(params)))

    def set_raster(self, which, raster):
        """Upload a glimpse_amd.Raster as the context's dem (0), dem_sigma (1) or viewshed (2); None removes it."""
        if raster is None:
            check(self.lib.glh_set_raster(self.handle, int(which), None, 0, 0, None, None, 1, 1, 0.0, 0.0, 0.0, 0.0))
            return
        z, nx, ny, gx, gy, sx, sy, x0, x1, y0, y1 = raster.device_args()
        check(self.lib.glh_set_raster(self.handle, int(which), _ptr(z), nx, ny, _ptr(gx), _ptr(gy), sx, sy, x0, x1,
                                      y0, y1))

    def set_observer_mask(self, mask):
        m = None if mask is None else _arr(mask, np.uint8, (self.P, self.O))
        check(self.lib.glh_set_observer_mask(self.handle, _ptr(m)))

    def set_active(self, active):
        a = None if active is None else _arr(active, np.uint8, (self.P,))
        check(self.lib.glh_set_active(self.handle, _ptr(a)))

    def set_particles(self, p):
        p = _arr(p, np.float64, (self.P, self.N, 6))
        check(self.lib.glh_set_particles(self.handle, _ptr(p)))

    def get_particles(self):
        out = np.empty((self.P, self.N, 6))
        check(self.lib.glh_get_particles(self.handle, _ptr(out)))
        return out

    def set_weights(self, w):
        w = _arr(w, np.float64, (self.P, self.N))
        check(self.lib.glh_set_weights(self.handle, _ptr(w)))

    def set_extra_log_likelihoods(self, ll):
        """A caller-computed log-likelihood term [P][N] for the next update_weights calls (None removes it)."""
        a = None if ll is None else _arr(ll, np.float64, (self.P, self.N))
        check(self.lib.glh_set_extra_log_likelihoods(self.handle, _ptr(a)))

    def get_weights(self):
        out = np.empty((self.P, self.N))
        check(self.lib.glh_get_weights(self.handle, _ptr(out)))
        return out

    def point_status(self):
        out = np.empty(self.P, dtype=np.uint32)
        check(self.lib.glh_get_point_status(self.handle, _ptr(out)))
        return out

    def point_error_frame(self):
        out = np.empty(self.P, dtype=np.int32)
        check(self.lib.glh_get_point_error_frame(self.handle, _ptr(out)))
        return out

    def observer_status(self):
        out = np.empty((self.O, self.P), dtype=np.int32)
        check(self.lib.glh_get_observer_status(self.handle, _ptr(out)))
        return out

    def observer_status_frames(self, frame0, n_frames):
        out = np.empty((n_frames, self.O, self.P), dtype=np.int32)
        check(self.lib.glh_get_observer_status_frames(self.handle, int(frame0), int(n_frames), _ptr(out)))
        return out

    def get_point_state(self, point):
        """(particles (N, 6), weights (N,)) of one point."""
        p, w = np.empty((self.N, 6)), np.empty(self.N)
        check(self.lib.glh_get_point_state(self.handle, int(point), _ptr(p), _ptr(w)))
        return p, w

    def search_boxes(self):
        out = np.empty((self.O, self.P, 4), dtype=np.int32)
        check(self.lib.glh_get_search_boxes(self.handle, _ptr(out)))
        return out

    # ---- stages
    def set_frame(self, frame):
        check(self.lib.glh_set_frame(self.handle, int(frame)))

    def init_particles(self, normals=None, seed=0):
        if normals is None:
            check(self.lib.glh_init_particles(self.handle, RNG_PHILOX, None, seed))
        else:
            n = _arr(normals, np.float64, (self.P, self.N, 6))
            check(self.lib.glh_init_particles(self.handle, RNG_HOST, _ptr(n), 0))

    def evolve(self, tau, normals=None, seed=0, step=0):
        if normals is None:
            check(self.lib.glh_evolve(self.handle, float(tau), RNG_PHILOX, None, seed, step))
        else:
            n = _arr(normals, np.float64, (self.P, self.N, 3))
            check(self.lib.glh_evolve(self.handle, float(tau), RNG_HOST, _ptr(n), 0, step))

    def init_templates(self, obs, image):
        check(self.lib.glh_init_templates(self.handle, obs, int(image)))

    def _images(self, images):
        im = np.array([-1 if i is None else int(i) for i in images], dtype=np.int32)
        assert im.shape == (self.O,)
        return im

    def update_weights(self, images):
        im = self._images(images)
        check(self.lib.glh_update_weights(self.handle, _ptr(im)))

    def resample(self, u=None, seed=0, step=0, method="systematic"):
        m = RESAMPLE[method]
        if u is None:
            check(self.lib.glh_resample_method(self.handle, m, RNG_PHILOX, None, seed, step))
        else:
            uu = _arr(u, np.float64, (self.P,) if m == 0 else (self.P, self.N))
            check(self.lib.glh_resample_method(self.handle, m, RNG_HOST, _ptr(uu), 0, step))

    def residual_draws(self):
        """Uniforms the last residual resampling consumed per point: n - sum(repetitions) (tracker.py:199-201)."""
        out = np.empty(self.P, dtype=np.int32)
        check(self.lib.glh_get_residual_draws(self.handle, _ptr(out)))
        return out

    def record_covariances(self, frame):
        check(self.lib.glh_record_covariances(self.handle, int(frame)))

    def get_covariances(self, frame0, n_frames):
        out = np.empty((n_frames, self.P, 6, 6))
        check(self.lib.glh_get_covariances(self.handle, frame0, n_frames, _ptr(out)))
        return out

    def record_moments(self, frame):
        check(self.lib.glh_record_moments(self.handle, int(frame)))

    def step(self, frame, tau, images, normals=None, u=None, seed=0):
        im = self._images(images)
        if normals is None:
            check(self.lib.glh_step(self.handle, int(frame), float(tau), _ptr(im), RNG_PHILOX, None, None, seed))
        else:
            n = _arr(normals, np.float64, (self.P, self.N, 3))
            uu = _arr(u, np.float64, (self.P,))
            check(self.lib.glh_step(self.handle, int(frame), float(tau), _ptr(im), RNG_HOST, _ptr(n), _ptr(uu), 0))

    def track(self, frames, taus, images, seed=0):
        """len(frames) consecutive `step` updates (device RNG) in one call: frames (T,), taus (T,), images (T, O)
        with -1 / None for "no image" (tracker.py:326-357)."""
        fr = np.ascontiguousarray(frames, dtype=np.int32).reshape(-1)
        ta = np.ascontiguousarray(taus, dtype=np.float64).reshape(-1)
        im = np.array([[-1 if v is None else int(v) for v in row] for row in images], dtype=np.int32)
        if ta.shape != fr.shape or im.shape != (len(fr), self.O):
            raise ValueError("frames (T,), taus (T,) and images (T, O) do not agree")
        check(self.lib.glh_track(self.handle, len(fr), _ptr(fr), _ptr(ta), _ptr(np.ascontiguousarray(im)), seed))

    def track_covariances(self, on=True):
        """`track` also records the covariances of every frame it runs (glh_track_covariances)."""
        check(self.lib.glh_track_covariances(self.handle, int(bool(on))))

    def set_track_streams(self, n):
        """Streams of `track`'s frame loop: 0 automatic, 1 one, 2 two (glh_set_track_streams)."""
        check(self.lib.glh_set_track_streams(self.handle, int(n)))

    def last_track_streams(self):
        n = C.c_int(0)
        check(self.lib.glh_debug_last_track_streams(self.handle, C.byref(n)))
        return n.value

    def set_fused(self, mode=1):
        """0 staged kernels, 1 fused per-point kernel (default), 2 fused with tiles forced to HBM (test)."""
        check(self.lib.glh_set_fused(self.handle, int(mode)))

    def set_math(self, mode="exact"):
        """"exact" (NumPy rounding; default) or "fast" (FMA / reciprocal arithmetic for device-RNG runs)."""
        check(self.lib.glh_set_math(self.handle, {"exact": MATH_EXACT, "fast": MATH_FAST}[mode]))

    def set_highpass(self, size=(5, 5), mode="reflect"):
        """Window of the median high-pass filter, scipy order (rows, columns); odd sizes up to 7.  `mode`: scipy.ndimage's
        boundary mode -- "reflect" (its default), "nearest", "mirror" or "wrap"."""
        sy, sx = (int(size), int(size)) if np.isscalar(size) else (int(size[0]), int(size[1]))
        check(self.lib.glh_set_highpass(self.handle, sx, sy))
        check(self.lib.glh_set_highpass_mode(self.handle, HIGHPASS_MODES[mode]))

    def set_point_offset(self, offset):
        check(self.lib.glh_set_point_offset(self.handle, int(offset)))

    def phase_stamps(self):
        """Diagnostic: s_memtime stamps (P, 24) of the fused kernel's phase boundaries (first call arms)."""
        out = np.zeros((self.P, 24), dtype=np.uint64)
        check(self.lib.glh_debug_phase_stamps(self.handle, _ptr(out)))
        return out

    def debug_draws(self, kind, seed, step=0):
        """The device stream's numbers: kind "init" (P, N, 6), "evolve" (P, N, 3) or "u" (P,) of frame `step`."""
        k = {"init": 0, "evolve": 1, "u": 2}[kind]
        shape = {0: (self.P, self.N, 6), 1: (self.P, self.N, 3), 2: (self.P,)}[k]
        out = np.empty(shape, dtype=np.float64)
        check(self.lib.glh_debug_draws(self.handle, k, int(seed), int(step), _ptr(out)))
        return out

    def last_variant(self):
        """Diagnostic: (threads, particles in registers, observers, flags) of the fused kernel instantiation that took
        the last fused step; flags bit 0 = fast arithmetic, bit 1 = the general code, bit 2 = the compile-time contract."""
        out = np.zeros(4, dtype=np.int32)
        check(self.lib.glh_debug_last_variant(self.handle, _ptr(out)))
        return tuple(int(v) for v in out)

    def sync(self):
        check(self.lib.glh_sync(self.handle))

    def stream(self):
        s = C.c_void_p()
        check(self.lib.glh_get_stream(self.handle, C.byref(s)))
        return s.value

    # ---- results
    def get_moments(self, frame0, n_frames):
        out = np.empty((n_frames, self.P, 12))
        check(self.lib.glh_get_moments(self.handle, frame0, n_frames, _ptr(out)))
        return out

    def get_tracks(self, frame0, n_frames):
        """The posterior history as Tracks holds it: means (P, n_frames, 6), sigmas (P, n_frames, 6)."""
        means, sigmas = np.empty((self.P, n_frames, 6)), np.empty((self.P, n_frames, 6))
        check(self.lib.glh_get_tracks(self.handle, frame0, n_frames, _ptr(means), _ptr(sigmas)))
        return means, sigmas

    def moments_device(self):
        p, n = C.c_void_p(), C.c_uint64()
        check(self.lib.glh_get_moments_device(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def get_template(self, obs, point):
        tw, th = self.tile
        box = np.empty(4, dtype=np.int32)
        duv = np.empty(2)
        tile = np.empty((th, tw))
        hv = np.empty(tw * th)
        hq = np.empty(tw * th)
        hn = C.c_int32()
        check(self.lib.glh_get_template(self.handle, obs, point, _ptr(box), _ptr(duv), _ptr(tile), _ptr(hv),
                                        _ptr(hq), C.byref(hn)))
        return {"box": box, "duv": duv, "tile": tile, "histogram": (hv[: hn.value].copy(), hq[: hn.value].copy())}

    def set_debug(self, keep=True):
        """True / 1: SSE surfaces, log likelihoods and resample indices are kept (staged kernels); 2: indices only."""
        check(self.lib.glh_set_debug(self.handle, int(keep)))

    def likelihood_debug(self, obs, point, want_sse=True):
        """uv (N,2), box (4,) or None, search float32 (Hs,Ws), sse float64 (Ho,Wo)."""
        tw, th = self.tile
        uv = np.empty((self.N, 2))
        box = np.empty(4, dtype=np.int32)
        check(self.lib.glh_get_likelihood_debug(self.handle, obs, point, _ptr(uv), _ptr(box), None, None))
        if box[0] < 0:
            return {"uv": uv, "box": None}
        ws, hs = int(box[2] - box[0]), int(box[3] - box[1])
        search = np.empty((hs, ws), dtype=np.float32)
        sse = np.empty((hs - th + 1, ws - tw + 1)) if want_sse else None
        check(self.lib.glh_get_likelihood_debug(self.handle, obs, point, None, None, _ptr(search), _ptr(sse)))
        return {"uv": uv, "box": box, "search": search, "sse": sse}

    def log_likelihoods(self, obs):
        out = np.empty((self.P, self.N))
        check(self.lib.glh_get_log_likelihoods(self.handle, obs, _ptr(out)))
        return out

    def resample_indices(self):
        out = np.empty((self.P, self.N), dtype=np.int32)
        check(self.lib.glh_get_resample_indices(self.handle, _ptr(out)))
        return out

    def copy_bandwidth(self, nbytes=1 << 30, iters=10):
        """Measured device-to-device copy rate (read + write bytes per second, GB/s)."""
        out = C.c_double(0.0)
        check(self.lib.glh_measure_copy_bandwidth(self.handle, int(nbytes), int(iters), C.byref(out)))
        return out.value

    def profile_enable(self, on=True):
        check(self.lib.glh_profile_enable(self.handle, int(bool(on))))

    def profile_reset(self):
        check(self.lib.glh_profile_reset(self.handle))

    def profile_get(self):
        n = self.lib.glh_stage_count()
        ms = np.zeros(n)
        launches = np.zeros(n, dtype=np.int64)
        check(self.lib.glh_profile_get(self.handle, _ptr(ms), _ptr(launches)))
        return {name: (float(ms[i]), int(launches[i])) for i, name in enumerate(stage_names())}

    def profile_launches(self, stage):
        """Duration (ms) of every timed launch of `stage` (a name of stage_names()) since the last reset."""
        k = stage_names().index(stage)
        n = C.c_int(0)
        check(self.lib.glh_profile_get_launches(self.handle, k, None, 0, C.byref(n)))
        out = np.zeros(n.value)
        if n.value:
            check(self.lib.glh_profile_get_launches(self.handle, k, _ptr(out), n.value, C.byref(n)))
        return out

    def profile_span(self, stage):
        """GPU time (ms) from the start of the first timed launch of `stage` to the end of its last one since the last
        reset (launches on two streams overlap: their durations do not add up to it)."""
        ms = C.c_double(0.0)
        check(self.lib.glh_profile_get_span(self.handle, stage_names().index(stage), C.byref(ms)))
        return ms.value

    # ---- multi-GPU (RCCL behind the C ABI; glimpse_amd/sharding.py drives it)
    def comm_init(self, comm_id, rank, world):
        """Join the communicator made by `comm_unique_id()` on one rank (collective)."""
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError(f"comm_id must be {COMM_ID_BYTES} bytes")
        buf = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        check(self.lib.glh_comm_init(self.handle, buf, int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)

    def comm_destroy(self):
        check(self.lib.glh_comm_destroy(self.handle))

    def comm_barrier(self):
        check(self.lib.glh_comm_barrier(self.handle))

    def comm_max(self, value):
        v = C.c_double(float(value))
        check(self.lib.glh_comm_max_f64(self.handle, C.byref(v)))
        return v.value

    def gather_moments(self, frame0, n_frames, points_per_rank, root=0, download=True):
        """One RCCL exchange: on `root` returns (moments (n_frames, sum P, 12) in rank order, status (sum P,));
        None elsewhere.  download=False leaves the blocks on the root's device (returns None everywhere); `gathered()`
        fetches them later."""
        ppr = np.ascontiguousarray(points_per_rank, dtype=np.int32)
        if ppr.shape != (self.world,):
            raise ValueError("points_per_rank must have one entry per rank")
        self._gather_shape = (int(n_frames), [int(v) for v in ppr])
        check(self.lib.glh_gather_moments(self.handle, root, frame0, n_frames, _ptr(ppr), None, None))
        if self.rank != root or not download:
            return None
        return self.gathered()

    def gathered(self):
        """(moments (n_frames, sum P, 12), status (sum P,)) of the last gather_moments on the root."""
        n_frames, ppr = self._gather_shape
        total = sum(ppr)
        flat = np.empty(total * n_frames * 12)
        status = np.empty(total, dtype=np.uint32)
        check(self.lib.glh_get_gathered(self.handle, _ptr(flat), _ptr(status)))
        blocks, at = [], 0
        for pr in ppr:
            blocks.append(flat[at: at + pr * n_frames * 12].reshape(n_frames, pr, 12))
            at += pr * n_frames * 12
        return np.concatenate(blocks, axis=1), status

# ---- stateless stage hooks (parity tests) -----------------------------------------------
def stage_project(cam, xyz, device_id=0, directions=False):
    cam = _arr(cam, np.float64, (CAM_LEN,))
    xyz = _arr(xyz, np.float64)
    uv = np.empty((len(xyz), 2))
    fn = load().glh_stage_project_directions if directions else load().glh_stage_project
    check(fn(device_id, _ptr(cam), _ptr(xyz), len(xyz), _ptr(uv)))
    return uv


def stage_project_depth(cam, xyz, directions=False, device_id=0):
    cam = _arr(cam, np.float64, (CAM_LEN,))
    xyz = _arr(xyz, np.float64)
    uv = np.empty((len(xyz), 2))
    depth = np.empty(len(xyz))
    check(load().glh_stage_project_depth(device_id, _ptr(cam), _ptr(xyz), len(xyz), int(bool(directions)), _ptr(uv),
                                         _ptr(depth)))
    return uv, depth


def stage_unproject(cam, uv, depth=None, directions=True, device_id=0):
    cam = _arr(cam, np.float64, (CAM_LEN,))
    uv = _arr(uv, np.float64)
    d = None if depth is None else _arr(np.atleast_1d(depth), np.float64)
    xyz = np.empty((len(uv), 3))
    check(load().glh_stage_unproject(device_id, _ptr(cam), _ptr(uv), len(uv), _ptr(d), 0 if d is None else len(d),
                                     int(bool(directions)), _ptr(xyz)))
    return xyz


def _frame_dims(frame):
    frame = _arr(frame, np.uint8)
    h, w = frame.shape[:2]
    ch = 1 if frame.ndim == 2 else frame.shape[2]
    return frame, w, h, ch


# scipy.ndimage boundary modes of the median high-pass the device implements (glh_set_highpass_mode); the grid-* names are
# scipy's aliases
HIGHPASS_MODES = {"reflect": 0, "grid-mirror": 0, "nearest": 1, "mirror": 2, "wrap": 3, "grid-wrap": 3}


def _highpass_xy(size):
    return (int(size), int(size)) if np.isscalar(size) else (int(size[1]), int(size[0]))  # scipy: (rows, columns)


def stage_template(frame, box, device_id=0, highpass=(5, 5), mode="reflect"):
    frame, w, h, ch = _frame_dims(frame)
    sx, sy = _highpass_xy(highpass)
    box = _arr(box, np.int32, (4,))
    tw, th = int(box[2] - box[0]), int(box[3] - box[1])
    tile = np.empty((th, tw))
    hv = np.empty(tw * th)
    hq = np.empty(tw * th)
    hn = C.c_int32()
    check(load().glh_stage_template_highpass(device_id, _ptr(frame), w, h, ch, _ptr(box), sx, sy, HIGHPASS_MODES[mode],
                                             _ptr(tile), _ptr(hv), _ptr(hq), C.byref(hn)))
    return tile, (hv[: hn.value].copy(), hq[: hn.value].copy())


def stage_search_tile(frame, box, histogram, device_id=0, highpass=(5, 5), mode="reflect"):
    frame, w, h, ch = _frame_dims(frame)
    sx, sy = _highpass_xy(highpass)
    box = _arr(box, np.int32, (4,))
    hv = _arr(histogram[0], np.float64)
    hq = _arr(histogram[1], np.float64)
    out = np.empty((int(box[3] - box[1]), int(box[2] - box[0])), dtype=np.float32)
    check(load().glh_stage_search_tile_highpass(device_id, _ptr(frame), w, h, ch, _ptr(box), _ptr(hv), _ptr(hq),
                                                len(hv), sx, sy, HIGHPASS_MODES[mode], _ptr(out)))
    return out


def stage_ssd(search, templ, device_id=0):
    search = _arr(search, np.float32)
    templ = _arr(templ, np.float32)
    hs, ws = search.shape
    th, tw = templ.shape
    out = np.empty((hs - th + 1, ws - tw + 1), dtype=np.float32)
    check(load().glh_stage_ssd(device_id, _ptr(search), hs, ws, _ptr(templ), th, tw, _ptr(out)))
    return out


def stage_sample(sse, box, uv, device_id=0, orders=None):
    """Observer.sample_tile (observer.py:178-214) on the device: (values, outside).  `orders` = (kx, ky) of
    RectBivariateSpline (rows axis, columns axis); None: the bicubic default."""
    sse = _arr(sse, np.float32)
    box = _arr(box, np.float64, (4,))
    uv = _arr(uv, np.float64)
    values = np.empty(len(uv))
    outside = np.empty(len(uv), dtype=np.uint8)
    if orders is None:
        check(load().glh_stage_sample(device_id, _ptr(sse), sse.shape[0], sse.shape[1], _ptr(box), _ptr(uv), len(uv),
                                      _ptr(values), _ptr(outside)))
    else:
        check(load().glh_stage_sample_orders(device_id, _ptr(sse), sse.shape[0], sse.shape[1], int(orders[0]),
                                             int(orders[1]), _ptr(box), _ptr(uv), len(uv), _ptr(values), _ptr(outside)))
    return values, outside.astype(bool)


def stage_resample(weights, u, device_id=0):
    w = _arr(weights, np.float64)
    idx = np.empty(len(w), dtype=np.int64)
    check(load().glh_stage_resample(device_id, _ptr(w), len(w), float(u), _ptr(idx)))
    return idx


def stage_raster_sample(raster, xy, order=1, device_id=0):
    z, nx, ny, gx, gy, sx, sy, x0, x1, y0, y1 = raster.device_args()
    xy = _arr(xy, np.float64)
    vals = np.empty(len(xy))
    oob = np.empty(len(xy), dtype=np.uint8)
    check(load().glh_stage_raster_sample(device_id, _ptr(z), nx, ny, _ptr(gx), _ptr(gy), sx, sy, x0, x1, y0, y1,
                                         _ptr(xy), len(xy), int(order), _ptr(vals), _ptr(oob)))
    return vals, oob.astype(bool)
