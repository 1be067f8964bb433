"""CPU oracle for the glimpse Tracker hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy (+ one small C file) restatement of the per-frame
particle-filter step of `glimpse.Tracker` (reference: /root/reference, cited
file:line in every function).  It exists to CHECK the HIP path:

* only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
  may import it;
* nothing under `glimpse_amd/` imports it, and the product path never falls
  back to it (the product raises if `libglimpse_hip.so` is missing).

Pinning (how far the oracle itself is trusted)
----------------------------------------------
* Rows 1-9, 11-17 of SURVEY.md section 8(a) (projection, search box, tile prep,
  median high-pass, template init, spline sampling, weights, resampling,
  motion model, moments, driver) are pinned against the *reference itself*
  imported in the build container: `tools/make_golden.py` runs
  `glimpse.Tracker` under `tools/refstubs.py` and writes `tests/golden/*.npz`;
  `tests/test_oracle_golden.py` checks every oracle stage and the end-to-end
  tracks against those files.  Projection is additionally pinned by the
  reference's own known-answer doctests (`camera.py:615-620`, `:683-694`).
* Row 10, `cv2.matchTemplate(TM_SQDIFF)` (`tracker.py:609-613`), is a
  third-party dependency that is absent from /root/reference and from this
  image (opencv-python-headless 4.4.0.46, `poetry.lock:641-643`).  The oracle
  restates OpenCV's published formula
  R(x,y) = sum (T(x',y') - I(x+x',y+y'))^2 with a float64 accumulator rounded
  once to float32.  PARITY IS UNPINNED at that single boundary: no reference
  test or fixture holds an OpenCV output for this path.
"""
from . import camera, motion, raster, resample, spline, ssd, tiles, tracker  # noqa: F401
