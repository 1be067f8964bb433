"""`Image`: `glimpse.Image` (/root/reference/src/glimpse/image.py) for the tracking path.

Holds a `Camera`, a capture datetime and the pixels: an in-memory array (the reference's cached read,
image.py:180-186, :211-213) or a file that is decoded on first use (the reference reads it band by band
through GDAL, image.py:187-210; here Pillow decodes it, which for the 8-bit JPEG / PNG / TIFF files of a
time-lapse camera gives the same samples).  A read at another size than the file's (`cam.resize(...)`, image.py:188-193)
is GDAL's default RasterIO resampling -- nearest neighbour, destination pixel i from source column
floor((i + 0.5) * src / dst) -- on the decoded array; GDAL itself is absent here, and for JPEG files it would first pick
one of the decoder's built-in 1/2, 1/4, 1/8 overviews, which this does not reproduce (parity unpinned for resized
reads).  EXIF parsing, `project`, `write` and `plot` are out of scope.
"""
import numpy as np

from .camera import Camera


class Image:
    def __init__(self, path=None, cam=None, datetime=None, exif=None, array=None):
        self.path = None if path is None else str(path)
        if isinstance(cam, dict):
            cam = Camera(**cam)
        if cam is None:
            raise ValueError("cam is required (EXIF / file metadata are out of scope here)")
        if path is None and array is None:
            raise ValueError("either path or array is required")
        self.cam = cam
        if not datetime:
            raise ValueError("datetime is required (EXIF parsing is out of scope here)")
        self.datetime = datetime
        self.exif = exif
        self.array = None if array is None else np.asarray(array)

    @property
    def size(self):
        """image.py:121-124."""
        return self.cam.imgsz

    def _decode(self):
        """The whole file as (h, w) or (h, w, bands), like np.dstack of GDAL's bands (image.py:200-206)."""
        if self.path is None:
            raise ValueError("the image has neither an array nor a path")
        try:
            from PIL import Image as _PILImage
        except ImportError as e:  # pragma: no cover
            raise NotImplementedError("reading image files needs Pillow; assign Image.array instead") from e
        with _PILImage.open(self.path) as im:
            if im.mode == "P":
                im = im.convert("RGB")
            a = np.asarray(im)
        if a.ndim == 3 and a.shape[2] == 1:
            a = a[:, :, 0]
        return a

    @staticmethod
    def _nearest(n_src, n_dst, offset=0.0, count=None):
        """Source index of every destination pixel under GDAL's default (nearest neighbour) RasterIO resampling of a
        window of `count` source pixels starting at `offset` into `n_dst` pixels."""
        count = n_src if count is None else count
        idx = np.floor(offset + (np.arange(n_dst) + 0.5) * (count / n_dst) + 1e-10).astype(np.int64)
        return np.clip(idx, 0, n_src - 1)

    def read(self, box=None, cache=True):
        """image.py:137-214: the cached array, or the file (decoded once and kept when `cache`), at the camera's image
        size: an array of another size is resampled to it (and the resampled array is what is cached, like the
        reference caches what GDAL returned; for an image without a file the caller's array is kept and the resampled
        copy cached beside it).  `box` = (left, top, right, bottom) in camera image coordinates; with `cache=False`
        the box is read straight from the file's own pixels and resampled to the BOX's size -- the reference hands GDAL
        buf_xsize / buf_ysize = the camera's whole image size for that window (image.py:192-201), so its output has
        another shape there; tracking never reads that way (Observer.extract_tile passes the observer's cache flag,
        and a cached image is sliced)."""
        cw, ch = (int(v) for v in self.cam.imgsz)
        array = self.array
        kept = getattr(self, "_resized", None)
        if kept is not None and self.path is None and array is not None and kept[0] == (cw, ch) \
                and array.shape[1::-1] != (cw, ch):
            array = kept[1]  # (the copy of an in-memory image resampled to this camera size)
        if array is not None and array.shape[1::-1] != (cw, ch) and self.path is not None:
            array = None  # (a cached read of another size is not reused: image.py:183-187)
        from_file = array is None
        if from_file:
            array = self._decode()
        h, w = array.shape[:2]
        if (w, h) != (cw, ch):
            if box is not None and not cache and from_file:
                # the window of the file that the box covers, resampled to the box's size (image.py:196-201)
                xs, ys = w / cw, h / ch
                x0, y0 = int(round(box[0] * xs)), int(round(box[1] * ys))
                nx, ny = int(round((box[2] - box[0]) * xs)), int(round((box[3] - box[1]) * ys))
                cols = self._nearest(w, box[2] - box[0], x0, nx)
                rows = self._nearest(h, box[3] - box[1], y0, ny)
                return array[rows][:, cols]
            array = array[self._nearest(h, ch)][:, self._nearest(w, cw)]
        if from_file and cache:
            self.array = array
        elif not from_file and array is not self.array and cache and self.path is not None:
            self.array = array  # (image.py:207-209: a cached array of another size is replaced by the resized read)
        elif not from_file and array is not self.array and cache:
            # an in-memory image (no file to go back to: the reference would fail on gdal.Open(None) here): the caller's
            # pixels stay, the resampled copy is kept beside them for this camera size -- a later cam.resize(1) reads the
            # original again instead of upsampling a decimated copy
            self._resized = ((cw, ch), array)
        if box is not None:
            return array[box[1]:box[3], box[0]:box[2]]
        return array

    def xyz_to_uv(self, xyz, **kwargs):
        """image.py:279-285."""
        return self.cam.xyz_to_uv(xyz, **kwargs)

    def uv_to_xyz(self, uv, directions=False, **kwargs):
        return self.cam.uv_to_xyz(uv, directions=directions, **kwargs)

    def inbounds(self, uv):
        """image.py:297-299."""
        return self.cam.inframe(uv)
