"""`Image`: `glimpse.Image` (/root/reference/src/glimpse/image.py) for the tracking path.

Holds a `Camera`, a capture datetime and the pixels: an in-memory array (the reference's cached read,
image.py:180-186, :211-213) or a file that is decoded on first use (the reference reads it band by band
through GDAL, image.py:187-210; here Pillow decodes it, which for the 8-bit JPEG / PNG / TIFF files of a
time-lapse camera gives the same samples).  EXIF parsing, resized reads, `project`, `write` and `plot` are
out of scope.
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

    def read(self, box=None, cache=True):
        """image.py:137-214: the cached array, or the file (decoded once and kept when `cache`)."""
        array = self.array
        if array is None:
            array = self._decode()
            if cache:
                self.array = array
        h, w = array.shape[:2]
        if (w, h) != tuple(self.cam.imgsz):
            raise NotImplementedError("resized reads (cam.imgsz != array size) are out of scope")
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
