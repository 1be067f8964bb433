"""Oracle: area sum of squared differences (test infrastructure only).

Restates `cv2.matchTemplate(search.astype(f32), templ.astype(f32), TM_SQDIFF)`
(/root/reference/src/glimpse/track/tracker.py:609-613) from OpenCV's documented
formula; float64 accumulate, one rounding to float32 (`accumulate="f64"`, the default: what the golden fixtures'
stand-in computes), or -- the same formula with OpenCV-style float32 accumulation -- float32 fused multiply-adds along
every template row and a float64 sum over the rows (`accumulate="row_f32"`: the summation of the HIP kernels,
oracle/ssd.c).  Third-party dependency
(opencv-python-headless 4.4.0.46, poetry.lock:641-643), absent here: PARITY
UNPINNED at this boundary.  Uses oracle/_build/liboracle_ssd.so (oracle/ssd.c)
when built, else an equivalent NumPy loop.
"""
import ctypes
import os

import numpy as np

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(__file__), "_build", "liboracle_ssd.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.oracle_ssd_f32.restype = ctypes.c_int
            for fn in (lib.oracle_ssd_f32, lib.oracle_ssd_f32_rows):
                fn.restype = ctypes.c_int
                fn.argtypes = [
                    ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                    ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                ]
            _LIB = lib
        else:
            _LIB = False
    return _LIB


def match_template_sqdiff_numpy(image, templ):
    th, tw = templ.shape
    win = np.lib.stride_tricks.sliding_window_view(image, (th, tw))
    ho, wo = win.shape[:2]
    out = np.empty((ho, wo), dtype=np.float64)
    t64 = templ.astype(np.float64)
    for r in range(ho):
        d = win[r].astype(np.float64) - t64
        out[r] = np.einsum("cij,cij->c", d, d)
    return out.astype(np.float32)


def match_template_sqdiff(image, templ, accumulate="f64"):
    """float32 (Hs,Ws), float32 (th,tw) -> float32 (Hs-th+1, Ws-tw+1)."""
    image = np.ascontiguousarray(image, dtype=np.float32)
    templ = np.ascontiguousarray(templ, dtype=np.float32)
    lib = _lib()
    if accumulate == "row_f32":
        if not lib:
            raise RuntimeError("accumulate='row_f32' needs oracle/_build/liboracle_ssd.so (make -C oracle)")
        hs, ws = image.shape
        th, tw = templ.shape
        out = np.empty((hs - th + 1, ws - tw + 1), dtype=np.float32)
        if lib.oracle_ssd_f32_rows(image.ctypes.data, hs, ws, templ.ctypes.data, th, tw, out.ctypes.data) != 0:
            raise ValueError("template larger than image")
        return out
    if accumulate != "f64":
        raise ValueError(f"accumulate={accumulate!r}")
    if not lib:
        return match_template_sqdiff_numpy(image, templ)
    hs, ws = image.shape
    th, tw = templ.shape
    out = np.empty((hs - th + 1, ws - tw + 1), dtype=np.float32)
    rc = lib.oracle_ssd_f32(
        image.ctypes.data, hs, ws, templ.ctypes.data, th, tw, out.ctypes.data
    )
    if rc != 0:
        raise ValueError("template larger than image")
    return out
