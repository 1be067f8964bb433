/* Oracle (test infrastructure only): cv2.matchTemplate(..., TM_SQDIFF) by
 * OpenCV's documented formula
 *     R(r,c) = sum_{i,j} (T(i,j) - I(r+i,c+j))^2
 * (call site /root/reference/src/glimpse/track/tracker.py:609-613; third-party
 * opencv-python-headless 4.4.0.46, poetry.lock:641-643 -- absent from the
 * reference tree, parity unpinned at this boundary).  Inputs are float32, the
 * accumulator is float64, the result is rounded once to float32.
 *
 * OpenCV does not specify the precision of the accumulation (its CV_32F path works in float32), so the
 * same formula is restated a second time with a float32 accumulator along every template row -- the
 * difference rounded to float32, then fused multiply-adds in row order, j = 0 .. tw-1 -- and a float64
 * sum over the rows (exact: at most 127 float32 terms), rounded once to float32:
 * oracle_ssd_f32_rows.  This is the summation the HIP kernels use (glh_kernels.h: ssd_strip_rows); the
 * two restatements differ by at most a few units in the last place of float32, which is enough to
 * move a resampling index once in ~10-100 particle-filter steps, so index-for-index comparisons over
 * long sequences are made against this one.
 * Built by oracle/Makefile into oracle/_build/liboracle_ssd.so.
 */
#include <math.h>
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

int oracle_ssd_f32_rows(const float *img, int hs, int ws, const float *tpl, int th,
                        int tw, float *out) {
  int ho = hs - th + 1, wo = ws - tw + 1;
  if (ho <= 0 || wo <= 0) return -1;
  for (int r = 0; r < ho; ++r) {
    for (int c = 0; c < wo; ++c) {
      double acc = 0.0;
      for (int i = 0; i < th; ++i) {
        const float *s = img + (size_t)(r + i) * ws + c;
        const float *t = tpl + (size_t)i * tw;
        float row = 0.0f;
        for (int j = 0; j < tw; ++j) {
          float d = s[j] - t[j];
          row = fmaf(d, d, row); /* one rounding per term */
        }
        acc += (double)row;
      }
      out[(size_t)r * wo + c] = (float)acc;
    }
  }
  return 0;
}
