/* Oracle (test infrastructure only): cv2.matchTemplate(..., TM_SQDIFF) by
 * OpenCV's documented formula
 *     R(r,c) = sum_{i,j} (T(i,j) - I(r+i,c+j))^2
 * (call site /root/reference/src/glimpse/track/tracker.py:609-613; third-party
 * opencv-python-headless 4.4.0.46, poetry.lock:641-643 -- absent from the
 * reference tree, parity unpinned at this boundary).  Inputs are float32, the
 * accumulator is float64, the result is rounded once to float32.
 * Built by oracle/Makefile into oracle/_build/liboracle_ssd.so.
 */
#include <stddef.h>

int oracle_ssd_f32(const float *img, int hs, int ws, const float *tpl, int th,
                   int tw, float *out) {
  int ho = hs - th + 1, wo = ws - tw + 1;
  if (ho <= 0 || wo <= 0) return -1;
  for (int r = 0; r < ho; ++r) {
    for (int c = 0; c < wo; ++c) {
      double acc = 0.0;
      for (int i = 0; i < th; ++i) {
        const float *s = img + (size_t)(r + i) * ws + c;
        const float *t = tpl + (size_t)i * tw;
        for (int j = 0; j < tw; ++j) {
          double d = (double)s[j] - (double)t[j];
          acc += d * d;
        }
      }
      out[(size_t)r * wo + c] = (float)acc;
    }
  }
  return 0;
}
