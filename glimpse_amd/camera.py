"""`Camera`: host-side mirror of `glimpse.Camera` for the tracking path.

Same constructor and attributes as the reference (/root/reference/src/glimpse/camera.py:77-123,
state vector `_vector[20]` :101, :128-198).  `xyz_to_uv` (camera.py:591-628) -- the projection
half that sits on the Tracker hot path -- runs on the GPU through libglimpse_hip.so; there is no
CPU fallback for it.  The inverse projection (`uv_to_xyz`, camera.py:630-663, with the k1 closed form
and the Oulu undistortion, :1198-1337) runs on the GPU as well (`glh_stage_unproject`); it is not on
the per-frame path.  Calibration / rendering methods are out of scope.
"""
import numpy as np

from . import _lib, synth


def _fmt(value, length, default=None, dtype=float):
    """helpers.format_list (helpers.py:27-84) for the cases the Camera constructor uses."""
    if value is None:
        return None
    v = np.atleast_1d(np.asarray(value, dtype=dtype)).ravel()
    if len(v) == length:
        return v
    if len(v) == 1 and default is None:
        return np.repeat(v, length)
    if len(v) < length:
        fill = v[-1] if default is None else default
        return np.concatenate((v, np.full(length - len(v), fill, dtype=dtype)))
    return v[:length]


class Camera:
    def __init__(self, imgsz, f=None, c=None, sensorsz=None, fmm=None, cmm=None, k=(0, 0, 0, 0, 0, 0),
                 p=(0, 0), xyz=(0, 0, 0), viewdir=(0, 0, 0), correction=False):
        if (fmm is not None or cmm is not None) and sensorsz is None:
            raise ValueError("Attributes in mm (fmm, cmm) provided without sensor size")
        if f is not None and fmm is not None:
            raise ValueError("Focal length provided in both pixels and mm (f, fmm)")
        if c is not None and cmm is not None:
            raise ValueError("Principal point offset provided in both pixels and mm (c, cmm)")
        if imgsz is None:
            raise ValueError("Image size (imgsz) cannot be None")
        self._vector = np.full(20, np.nan, dtype=float)
        self.xyz = xyz
        self.viewdir = viewdir
        self.imgsz = imgsz
        self.sensorsz = sensorsz
        if fmm is not None:
            f = _fmt(fmm, 2) * self.imgsz / self.sensorsz
        if f is None:
            raise ValueError("Focal length (f or fmm) is missing")
        self.f = f
        if cmm is not None:
            c = _fmt(cmm, 2) * self.imgsz / self.sensorsz
        if c is None:
            c = (0, 0)
        self.c = c
        self.k = k
        self.p = p
        if correction is True:
            correction = {}
        if isinstance(correction, dict):
            correction = {"radius": 6.3781e6, "refraction": 0.13, **correction}
        self.correction = correction
        self._original_vector = self._vector.copy()

    # ---- properties (camera.py:128-236)
    xyz = property(lambda s: s._vector[0:3], lambda s, v: s._vector.__setitem__(slice(0, 3), _fmt(v, 3, 0)))
    viewdir = property(lambda s: s._vector[3:6], lambda s, v: s._vector.__setitem__(slice(3, 6), _fmt(v, 3, 0)))
    f = property(lambda s: s._vector[8:10], lambda s, v: s._vector.__setitem__(slice(8, 10), _fmt(v, 2)))
    c = property(lambda s: s._vector[10:12], lambda s, v: s._vector.__setitem__(slice(10, 12), _fmt(v, 2, 0)))
    k = property(lambda s: s._vector[12:18], lambda s, v: s._vector.__setitem__(slice(12, 18), _fmt(v, 6, 0)))
    p = property(lambda s: s._vector[18:20], lambda s, v: s._vector.__setitem__(slice(18, 20), _fmt(v, 2, 0)))

    @property
    def imgsz(self):
        return self._vector[6:8].astype(int)

    @imgsz.setter
    def imgsz(self, value):
        as_float = _fmt(value, 2)
        as_int = as_float.astype(int)
        if np.any(as_int != as_float):
            raise ValueError("Image size is not integer")
        self._vector[6:8] = as_int

    @property
    def sensorsz(self):
        return self._sensorsz

    @sensorsz.setter
    def sensorsz(self, value):
        self._sensorsz = None if value is None else np.array(_fmt(value, 2), dtype=float)

    @property
    def fmm(self):
        return None if self.sensorsz is None else self.f * self.sensorsz / self.imgsz

    @property
    def cmm(self):
        return None if self.sensorsz is None else self.c * self.sensorsz / self.imgsz

    @property
    def R(self):
        """camera.py:239-280."""
        return synth.rotation_matrix(self.viewdir)

    @property
    def vector24(self):
        """The 24-double layout of include/glimpse_hip.h (GLH_CAM_LEN)."""
        v = np.zeros(_lib.CAM_LEN)
        v[:20] = self._vector
        if isinstance(self.correction, dict):
            v[20], v[21], v[22] = 1.0, self.correction["radius"], self.correction["refraction"]
        return v

    def copy(self):
        cam = Camera(imgsz=self.imgsz, f=self.f, c=self.c, sensorsz=self.sensorsz, k=self.k, p=self.p,
                     xyz=self.xyz, viewdir=self.viewdir,
                     correction=dict(self.correction) if isinstance(self.correction, dict) else self.correction)
        return cam

    # ---- projection (hot path: GPU)
    def xyz_to_uv(self, xyz, directions=False, return_depth=False):
        """camera.py:591-628, evaluated by the projection kernel (`directions=True`: xyz are rays, :1448)."""
        xyz = np.atleast_2d(np.asarray(xyz, dtype=float))
        if return_depth:  # (uv, distance along the optical axis), camera.py:1468-1469
            return _lib.stage_project_depth(self.vector24, xyz, directions=directions)
        return _lib.stage_project(self.vector24, xyz, directions=directions)

    def inframe(self, uv):
        """camera.py:700-718."""
        uv = np.asarray(uv)
        with np.errstate(invalid="ignore"):
            return np.all((uv >= 0) & (uv <= self.imgsz), axis=1)

    def uv_to_xyz(self, uv, directions=True, depth=1):
        """camera.py:630-663 on the device (`glh_stage_unproject`): closed form for k1 alone, else the Oulu
        fixed point, 20 iterations (camera.py:1198-1337)."""
        uv = np.atleast_2d(np.asarray(uv, dtype=float))
        d = None if (isinstance(depth, (int, float)) and depth == 1) else depth
        return _lib.stage_unproject(self.vector24, uv, depth=d, directions=directions)
