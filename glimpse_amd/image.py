"""`Image`: the cached-array path of `glimpse.Image` (/root/reference/src/glimpse/image.py).

Holds a `Camera`, a capture datetime and the pixel array.  Only the in-memory (cached) read
path of `Image.read` (image.py:180-186, :211-213) is mirrored; GDAL file I/O, EXIF parsing,
`project`, `write` and `plot` are out of scope.
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

    def read(self, box=None, cache=True):
        """image.py:137-214, cached-array path only."""
        if self.array is None:
            raise NotImplementedError("reading image files (GDAL) is out of scope: assign Image.array")
        h, w = self.array.shape[:2]
        if (w, h) != tuple(self.cam.imgsz):
            raise NotImplementedError("resized reads (cam.imgsz != array size) are out of scope")
        if box is not None:
            return self.array[box[1]:box[3], box[0]:box[2]]
        return self.array

    def xyz_to_uv(self, xyz, **kwargs):
        """image.py:279-285."""
        return self.cam.xyz_to_uv(xyz, **kwargs)

    def uv_to_xyz(self, uv, directions=False, **kwargs):
        return self.cam.uv_to_xyz(uv, directions=directions, **kwargs)

    def inbounds(self, uv):
        """image.py:297-299."""
        return self.cam.inframe(uv)
