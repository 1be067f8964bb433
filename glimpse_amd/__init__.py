"""glimpse_amd: the `glimpse.Tracker` particle-filter hot path on AMD MI355X (gfx950).

Drop-in names for that path only (see DESIGN.md for scope):

    from glimpse_amd import Camera, Image, Observer, CartesianMotion, Tracker, Tracks
    (+ CylindricalMotion, TangentCartesianMotion, TangentCylindricalMotion)

The compute runs in hand-written HIP kernels behind a C ABI (include/glimpse_hip.h,
glimpse_amd/lib/libglimpse_hip.so, bound with ctypes in glimpse_amd._lib).  There is no CPU
fallback: build the library with `python -m glimpse_amd.build`.
"""
from .camera import Camera
from .image import Image
from .motion import (CartesianMotion, CylindricalMotion, Motion, TangentCartesianMotion,
                     TangentCylindricalMotion)
from .observer import Observer
from .raster import Raster
from .tracker import Tracker
from .tracks import Tracks

__all__ = ["Camera", "Image", "Observer", "Motion", "CartesianMotion", "CylindricalMotion",
           "TangentCartesianMotion", "TangentCylindricalMotion", "Raster", "Tracker", "Tracks"]
__version__ = "0.1.0"
