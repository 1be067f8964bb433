"""`Raster`: the in-memory part of `glimpse.Raster` that the tracking path samples
(/root/reference/src/glimpse/raster.py): a gridded surface (DEM, DEM uncertainty, viewshed) defined by
its array and outer x / y limits, sampled at points.

Only what `Tracker` / the motion models use is mirrored: the constructor from an array and limits
(raster.py:652-694 with Grid, :25-98), the cell-centre coordinates (`x`, `y`, raster.py:131-165), the
bounds test (`inbounds_xy`, :313-337) and `sample(xy, order in {0, 1})` at points (:913-1027).  Sampling
runs on the GPU (`glh_stage_raster_sample`; inside a tracking run the kernels sample the uploaded
raster themselves).  File I/O (GDAL), resampling, terrain analysis are out of scope.
"""
import numpy as np

from . import _lib


class Raster:
    def __init__(self, array, x=None, y=None, datetime=None):
        self.array = np.atleast_2d(np.asarray(array))
        ny, nx = self.array.shape[:2]
        self.size = np.array((nx, ny))
        self.xlim = self._limits(x, nx)
        self.ylim = self._limits(y, ny)
        self.datetime = datetime

    @staticmethod
    def _limits(value, n):
        """Grid._parse_xy (raster.py:245-272): outer limits (2,), or cell-centre coordinates (n > 2)."""
        if value is None:
            value = (0, n)
        value = np.atleast_1d(np.asarray(value, dtype=float))
        if value.ndim == 1 and len(value) > 2:
            d = value[1] - value[0]
            value = np.array((value[0] - d / 2, value[-1] + d / 2))
        if value.shape != (2,):
            raise ValueError("Could not parse limits from x, y inputs")
        if value[0] == value[1]:
            raise ValueError("Grid limits cannot be equal")
        return value

    @property
    def d(self):
        """Cell size (dx, dy); negative where the coordinate decreases along the array (raster.py:115-118)."""
        return np.hstack((np.diff(self.xlim), np.diff(self.ylim))) / self.size

    @property
    def min(self):
        return np.array((min(self.xlim), min(self.ylim)))

    @property
    def max(self):
        return np.array((max(self.xlim), max(self.ylim)))

    def _centres(self, dim):
        d = abs(self.d[dim])
        value = np.linspace(start=self.min[dim] + d / 2, stop=self.max[dim] - d / 2, num=self.size[dim])
        return value[::-1] if self.d[dim] < 0 else value

    @property
    def x(self):
        """Cell-centre x from the first to the last column (raster.py:131-147)."""
        return self._centres(0)

    @property
    def y(self):
        """Cell-centre y from the first to the last row (raster.py:149-165)."""
        return self._centres(1)

    # ---- a Raster as an Observer image (orthophoto tracking; observer.py:26, :113, :129, :144)
    @property
    def vector24(self):
        """GLH_CAM_LEN layout in its georeferenced-raster form (include/glimpse_hip.h, entry [23] = 1)."""
        v = np.zeros(_lib.CAM_LEN)
        v[0], v[1] = self.xlim[0], self.ylim[0]
        v[6:8] = self.size
        v[8:10] = self.d
        v[23] = 1.0
        return v

    def xyz_to_uv(self, xyz):
        """Grid.xyz_to_uv (raster.py:423-445), evaluated by the projection kernel."""
        xyz = np.atleast_2d(np.asarray(xyz, dtype=float))
        if xyz.shape[1] == 2:
            xyz = np.column_stack((xyz, np.zeros(len(xyz))))
        return _lib.stage_project(self.vector24, xyz)

    def inbounds(self, uv):
        """Grid.inbounds (raster.py:339-341)."""
        uv = np.atleast_2d(np.asarray(uv, dtype=float))
        return np.all((uv >= 0) & (uv <= self.size), axis=1)

    def read(self, box=None, cache=True):
        """Raster.read of an in-memory array (raster.py:763-837): the window [top:bottom, left:right]."""
        if box is None:
            return self.array
        return self.array[box[1]:box[3], box[0]:box[2]]

    def inbounds_xy(self, xy):
        """raster.py:313-337 (points)."""
        xy = np.atleast_2d(np.asarray(xy, dtype=float))
        return np.all((xy >= self.min[0:2]) & (xy <= self.max[0:2]), axis=1)

    def device_args(self):
        """Arguments of glh_set_raster / glh_stage_raster_sample after `which` / `device_id`."""
        sx, sy = (1 if v > 0 else -1 for v in self.d)
        gx, gy = np.ascontiguousarray(self.x[::sx]), np.ascontiguousarray(self.y[::sy])
        z = np.ascontiguousarray(self.array, dtype=np.float64)
        nx, ny = (int(v) for v in self.size)
        return z, nx, ny, gx, gy, sx, sy, float(self.min[0]), float(self.max[0]), float(self.min[1]), float(self.max[1])

    def sample(self, xy, grid=False, order=1, bounds_error=True, fill_value=np.nan):
        """Values at points (n, 2): bilinear (order 1) or nearest cell (order 0) (raster.py:913-1027)."""
        if grid or order not in (0, 1):
            raise NotImplementedError("only point sampling with order 0 or 1 is built")
        xy = np.atleast_2d(np.asarray(xy, dtype=float))
        values, oob = _lib.stage_raster_sample(self, xy, order)
        if oob.any():
            if bounds_error:
                raise ValueError("Some of the sampling coordinates are out of bounds")
            values[oob] = fill_value
        return values
