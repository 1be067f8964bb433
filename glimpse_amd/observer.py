"""`Observer`: a time-ordered image list at one camera station
(/root/reference/src/glimpse/track/observer.py:16-214; compute methods only, plotting out of scope)."""
import datetime

import numpy as np

from . import _lib
from .timeutil import select_datetimes


class Observer:
    def __init__(self, images, sigma=0.3, cache=True):
        """observer.py:50-69."""
        if len(images) < 2:
            raise ValueError("Images are not two or greater")
        datetimes = []
        for i, img in enumerate(images):
            if img.datetime is None:
                raise ValueError(f"Image {i} is missing datetime")
            datetimes.append(img.datetime)
        time_deltas = np.array([dt.total_seconds() for dt in np.diff(datetimes)])
        if any(time_deltas <= 0):
            raise ValueError("Image datetimes are not stricly increasing")
        self.images = list(images)
        self.datetimes = np.array(datetimes)
        self.sigma = sigma
        self.cache = cache

    def index(self, value, maxdt=datetime.timedelta(0)):
        """observer.py:71-100: position of an image object, or of the image nearest to a datetime (ValueError when
        that image is further than |maxdt| away; maxdt=None accepts any distance)."""
        if not isinstance(value, datetime.datetime):
            return self.images.index(value)
        gaps = [abs(value - dt) for dt in self.datetimes]
        nearest = min(range(len(gaps)), key=gaps.__getitem__)  # first of equally near images
        if maxdt is not None and gaps[nearest] > abs(maxdt):
            raise ValueError("Nearest image out of range by " + str(gaps[nearest] - abs(maxdt)))
        return nearest

    def xyz_to_uv(self, xyz, img):
        """observer.py:102-113 (GPU projection)."""
        return self.images[img].xyz_to_uv(xyz)

    def tile_box(self, uv, size, img):
        """observer.py:115-130 with Grid.snap_box (raster.py:390-421): integer pixel-edge box."""
        uv = np.asarray(uv, dtype=float)
        halfsize = np.multiply(size, 0.5)
        xy_box = np.vstack((uv - halfsize, uv + halfsize))
        imgsz = np.asarray(self.images[img].size, dtype=float)
        if any(~np.all((xy_box >= 0) & (xy_box <= imgsz), axis=1)):
            raise IndexError("Box extends beyond grid bounds")
        return np.floor(xy_box + 0.5).flatten().astype(int)

    def extract_tile(self, box, img):
        """observer.py:132-144."""
        return self.images[img].read(box=box, cache=self.cache)

    def sample_tile(self, uv, tile, box, grid=False, **kwargs):
        """observer.py:178-214: the interpolating `RectBivariateSpline` of the tile sampled on the GPU -- at points
        (n, [u, v]), or with `grid=True` at every pair of the coordinate vectors `uv = (u values, v values)`, returned
        like scipy returns it, (len(v), len(u)).  `kx` / `ky` in 1..5 (the spline's degree along the rows / columns);
        a smoothing factor or a bounding box (another spline than the interpolating one) is not served."""
        extra = set(kwargs) - {"kx", "ky"}
        if extra and not (kwargs.get("s", 0) == 0 and kwargs.get("bbox", [None] * 4) == [None] * 4 and extra <= {"s", "bbox"}):
            raise NotImplementedError(f"RectBivariateSpline arguments {sorted(extra)}: only the interpolating spline "
                                      "(s = 0, default bbox) is on the tracking path")
        kx, ky = int(kwargs.get("kx", 3)), int(kwargs.get("ky", 3))
        if not (1 <= kx <= 5 and 1 <= ky <= 5):
            raise ValueError("kx, ky must be in 1..5")  # (what FITPACK accepts)
        if grid:
            u, v = (np.asarray(a, dtype=float).ravel() for a in uv)
            if (np.diff(u) < 0).any() or (np.diff(v) < 0).any():
                raise ValueError("x and y must be sorted to increasing order")  # (scipy's own check for grid=True)
            points = np.stack(np.meshgrid(u, v), axis=-1).reshape(-1, 2)  # rows: v, columns: u
        else:
            points = np.asarray(uv, dtype=float)
        values, outside = _lib.stage_sample(np.asarray(tile, dtype=np.float32), np.asarray(box, dtype=float), points,
                                            orders=None if (kx, ky) == (3, 3) else (kx, ky))
        if outside.any():
            raise ValueError("Some sampling points are outside box")
        return values.reshape(len(v), len(u)) if grid else values

    def shift_tile(self, tile, duv, **kwargs):
        """observer.py:146-176: resample a tile at pixel centres moved by (du, dv), each at most half a pixel
        (host spline: a plotting / alignment helper, not on the tracking path)."""
        import scipy.interpolate
        du, dv = (float(v) for v in duv)
        if max(abs(du), abs(dv)) > 0.5:
            raise ValueError("Shift larger than 0.5 pixels")
        tile = np.asarray(tile)
        rows, cols = np.arange(tile.shape[0]) + 0.5, np.arange(tile.shape[1]) + 0.5
        bands = tile[:, :, None] if tile.ndim == 2 else tile
        for b in range(bands.shape[2]):  # in place, like the reference
            bands[:, :, b] = scipy.interpolate.RectBivariateSpline(rows, cols, bands[:, :, b], **kwargs)(
                rows + dv, cols + du, grid=True)
        return tile

    def _pick(self, index):
        return np.asarray(self.images, dtype=object)[index].tolist() if not isinstance(index, slice) else \
            self.images[index]

    def cache_images(self, index=slice(None)):
        """observer.py:256-264."""
        for img in self._pick(index):
            img.read(cache=True)

    def clear_images(self, index=slice(None)):
        """observer.py:266-274."""
        for img in self._pick(index):
            img.array = None

    def subset(self, **kwargs):
        """observer.py:455-464: a new Observer over the images select_datetimes(**kwargs) keeps."""
        keep = select_datetimes(self.datetimes, **kwargs)
        return type(self)([img for img, k in zip(self.images, keep) if k], sigma=self.sigma, cache=self.cache)

    def split(self, n, overlap=1):
        """observer.py:466-493: `n` equal time spans (int) or the spans between datetime breaks (iterable); each
        following Observer starts `overlap` images before the end of the previous one."""
        first, last = self.datetimes[0], self.datetimes[-1]
        if np.iterable(n):
            breaks = sorted(set(n) | {first, last})
        else:
            span = (last - first) / n
            breaks = [first + k * span for k in range((last - first) // span + 1)]
        parts, start = [], breaks[0]
        for stop in breaks[1:]:
            part = self.subset(start=start, end=stop)
            parts.append(part)
            if overlap:
                start = part.datetimes[-min(overlap, len(part.datetimes))]
            else:
                start = part.datetimes[-1] + datetime.timedelta(microseconds=1)
        return parts
