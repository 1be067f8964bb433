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

    # ---- the formats either side of the path: camera models live in JSON files (camera.py:334-509) --------------------
    _FIELDS = ("xyz", "viewdir", "imgsz", "f", "c", "k", "p", "correction")

    @classmethod
    def from_json(cls, path, **kwargs):
        """camera.py:334-357: the constructor arguments stored by `to_json`; entries that are null (all NaN once read as
        numbers) count as absent; `kwargs` override the file."""
        import json

        with open(path) as fp:
            stored = json.load(fp)
        args = {}
        for key, value in stored.items():
            if isinstance(value, dict) or isinstance(value, bool):  # (correction: a dict of constants, or a flag)
                args[key] = value
                continue
            numbers = np.array(value, dtype=float)
            args[key] = None if np.isnan(numbers).all() else numbers
        args.update(kwargs)
        return cls(**args)

    def to_array(self):
        """camera.py:412-429: xyz | viewdir | imgsz | f | c | k | p as one vector of 20."""
        return self._vector.copy()

    def to_dict(self, attributes=_FIELDS):
        """camera.py:431-460: attribute name -> plain Python lists / numbers."""
        return {key: getattr(getattr(self, key), "tolist", lambda key=key: getattr(self, key))() for key in attributes}

    def to_json(self, path=None, attributes=_FIELDS, **kwargs):
        """camera.py:462-509: the dictionary of `to_dict` as JSON text, returned or written to `path`."""
        import json

        text = json.dumps(self.to_dict(attributes=attributes), **kwargs)
        if path is None:
            return text
        with open(path, "w") as fp:
            fp.write(text)
        return None

    def reset(self):
        """camera.py:399-410: back to the state the camera was constructed (or copied) with."""
        self._vector = self._original_vector.copy()

    def idealize(self):
        """camera.py:511-530: no distortion, no principal point offset."""
        self.k, self.p, self.c = np.zeros(6), np.zeros(2), np.zeros(2)

    def resize(self, size=1, force=False):
        """camera.py:532-589: scale imgsz, f and c to a target image size (nx, ny) or by a factor of the ORIGINAL size.
        A target size must be reachable by one factor for both axes (round(factor * original) == target) unless
        `force`."""
        target = np.atleast_1d(np.asarray(size, dtype=float))
        original = self._original_vector[6:8]
        if len(target) > 1 and force:
            new_size = target
        else:
            if len(target) > 1:
                # the factors s with round(s * original) == target on an axis form an interval; one factor must serve both
                lo, hi = np.max((target - 0.5) / original), np.min((target + 0.5) / original)
                exact = target / original
                if np.all(exact == exact[0]):
                    scale = exact[0]
                elif lo < hi:
                    scale = 0.5 * (lo + hi)
                else:
                    raise ValueError("Target image size does not preserve the original aspect ratio")
            else:
                scale = target[0]
            new_size = np.floor(scale * original + 0.5)
        ratio = new_size / self.imgsz
        self.imgsz = np.round(new_size)
        self.f = self.f * ratio
        self.c = self.c * ratio

    def infront(self, xyz, directions=False):
        """camera.py:665-683: which points (or ray directions) lie in front of the camera, i.e. project at all."""
        xyz = np.atleast_2d(np.asarray(xyz, dtype=float))
        rel = xyz if directions else xyz - self.xyz
        return (rel @ self.R.T)[:, 2] > 0

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
