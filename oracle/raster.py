"""Oracle: in-memory Raster sampling (test infrastructure only).

Follows /root/reference/src/glimpse/raster.py: Grid limits / cell centres (:100-165), `inbounds_xy`
(:313-337), the cached `RegularGridInterpolator` over ascending cell centres (:891-900) and
`Raster.sample` at points for order 0 / 1 (:913-1027).  SciPy is the reference's own dependency here.
"""
import numpy as np
import scipy.interpolate


class Raster:
    def __init__(self, array, x=None, y=None):
        self.array = np.atleast_2d(np.asarray(array, dtype=float))
        ny, nx = self.array.shape
        self.size = np.array((nx, ny))
        self.xlim = np.asarray((0, nx) if x is None else x, dtype=float)
        self.ylim = np.asarray((0, ny) if y is None else y, dtype=float)
        self.d = np.hstack((np.diff(self.xlim), np.diff(self.ylim))) / self.size
        self.min = np.array((min(self.xlim), min(self.ylim)))
        self.max = np.array((max(self.xlim), max(self.ylim)))
        self._zf = None

    def _centres(self, dim):
        d = abs(self.d[dim])
        v = np.linspace(self.min[dim] + d / 2, self.max[dim] - d / 2, self.size[dim])
        return v[::-1] if self.d[dim] < 0 else v

    def sample(self, xy, order=1):
        xy = np.atleast_2d(np.asarray(xy, dtype=float))
        if not np.all((xy >= self.min) & (xy <= self.max)):
            raise ValueError("Some of the sampling coordinates are out of bounds")
        if self._zf is None:
            sign = np.sign(self.d).astype(int)
            self._zf = scipy.interpolate.RegularGridInterpolator(
                (self._centres(0)[:: sign[0]], self._centres(1)[:: sign[1]]),
                self.array.T[:: sign[0], :: sign[1]].copy(), bounds_error=False, fill_value=None)
        return self._zf(xy, method=("nearest", "linear")[order])


def sample(surface, xy):
    """A Raster at points, or a constant (the reference's infinite 1 x 1 raster, raster.py:1021-1026)."""
    if isinstance(surface, Raster):
        return surface.sample(xy)
    return np.full(len(xy), float(surface))
