"""`Observer`: a time-ordered image list at one camera station
(/root/reference/src/glimpse/track/observer.py:16-214; compute methods only, plotting out of scope)."""
import datetime

import numpy as np

from . import _lib


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
        """observer.py:178-214 for grid=False, kx=ky=3: bicubic spline sampling on the GPU."""
        if grid or kwargs.get("kx", 3) != 3 or kwargs.get("ky", 3) != 3:
            raise NotImplementedError("only pointwise bicubic sampling (kx=ky=3) is on the tracking path")
        values, outside = _lib.stage_sample(np.asarray(tile, dtype=np.float32), np.asarray(box, dtype=float),
                                            np.asarray(uv, dtype=float))
        if outside.any():
            raise ValueError("Some sampling points are outside box")
        return values
