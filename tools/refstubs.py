"""Stub modules that let the *reference* package import in this container.

Test/fixture infrastructure only (used by tools/make_golden.py and
tools/check_oracle_vs_reference.py).  Nothing here is shipped, and nothing here
can run on the GPU box: /root/reference does not exist there.

The reference (`/root/reference/src/glimpse/__init__.py:2`) imports `optimize`
(cv2, lmfit), `config` (sharedmem), `helpers` (osgeo, progress) and `exif`
(piexif) at package import.  None of those are installed here.  Only one stub
carries arithmetic: `cv2.matchTemplate(..., TM_SQDIFF)`, restated from OpenCV's
documented formula  R(x,y) = sum_{x',y'} (T(x',y') - I(x+x',y+y'))^2  with a
float64 accumulator and a float32 result (third-party dependency
opencv-python-headless 4.4.0.46, `poetry.lock:641-643`; parity is unpinned at
this boundary).
"""
import sys
import types

import numpy as np

REFERENCE_SRC = "/root/reference/src"


class _Anything:
    """Class whose attributes/calls all return itself (for annotations/defaults)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        return _Anything()


class _PermissiveModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


def match_template_sqdiff(image, templ):
    """OpenCV TM_SQDIFF by its documented formula (float64 accumulate -> float32)."""
    image = np.asarray(image)
    templ = np.asarray(templ)
    assert image.dtype == np.float32 and templ.dtype == np.float32
    th, tw = templ.shape
    win = np.lib.stride_tricks.sliding_window_view(image, (th, tw))
    ho, wo = win.shape[:2]
    out = np.empty((ho, wo), dtype=np.float64)
    t64 = templ.astype(np.float64)
    for r in range(ho):
        d = win[r].astype(np.float64) - t64
        out[r] = np.einsum("cij,cij->c", d, d)
    return out.astype(np.float32)


def install():
    """Insert the stub modules into sys.modules and put the reference on sys.path."""
    if "glimpse" in sys.modules:
        return

    class _MapReduce:
        def __init__(self, np=None):
            self.np = np

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def map(self, func, sequence, reduce=None, star=False):
            results = []
            for item in sequence:
                r = func(*item) if star else func(item)
                if reduce is not None:
                    r = reduce(r)
                results.append(r)
            return results

    sharedmem = types.ModuleType("sharedmem")
    sharedmem.MapReduce = _MapReduce
    sharedmem.MapReduceByThread = _MapReduce
    sharedmem.copy = np.array
    sys.modules["sharedmem"] = sharedmem

    progress = types.ModuleType("progress")
    progress_bar = types.ModuleType("progress.bar")

    class Bar:
        def __init__(self, *a, **k):
            pass

        def next(self, *a, **k):
            pass

        def finish(self):
            pass

    progress_bar.Bar = Bar
    progress.bar = progress_bar
    sys.modules["progress"] = progress
    sys.modules["progress.bar"] = progress_bar

    for name in (
        "osgeo",
        "osgeo.gdal",
        "osgeo.gdal_array",
        "osgeo.ogr",
        "osgeo.osr",
        "piexif",
        "lmfit",
        "lmfit.parameter",
    ):
        sys.modules[name] = _PermissiveModule(name)
    sys.modules["osgeo"].gdal = sys.modules["osgeo.gdal"]
    sys.modules["osgeo"].gdal_array = sys.modules["osgeo.gdal_array"]
    sys.modules["osgeo"].ogr = sys.modules["osgeo.ogr"]
    sys.modules["osgeo"].osr = sys.modules["osgeo.osr"]
    sys.modules["lmfit"].parameter = sys.modules["lmfit.parameter"]

    cv2 = _PermissiveModule("cv2")
    cv2.TM_SQDIFF = 0

    def matchTemplate(image, templ, method=0, **kw):
        assert method == 0
        return match_template_sqdiff(image, templ)

    cv2.matchTemplate = matchTemplate
    sys.modules["cv2"] = cv2

    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)


def import_reference():
    install()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import glimpse  # noqa: F401

    return sys.modules["glimpse"]
