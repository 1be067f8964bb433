// glh_median.h -- rank-12-of-25 selection network for the 5x5 median high-pass
// (scipy.ndimage.median_filter(size=(5,5)), /root/reference/src/glimpse/track/tracker.py:530).
//
// 99 compare-exchanges with compile-time indices only (runtime-indexed register arrays
// would spill to scratch on gfx950).  The list below is an X-macro so that
// tests/test_median_network.py can parse it and prove it with the 0/1 principle over all
// 2^25 binary inputs (a selection network is correct for every input iff it is correct
// for every 0/1 input).
#pragma once
#include "glh_math.h"

// clang-format off
#define GLH_MED25_NETWORK(X) \
  X(0,1) X(3,4) X(2,4) X(2,3) X(6,7) X(5,7) X(5,6) X(9,10) X(8,10) X(8,9) \
  X(12,13) X(11,13) X(11,12) X(15,16) X(14,16) X(14,15) X(18,19) X(17,19) X(17,18) X(21,22) \
  X(20,22) X(20,21) X(23,24) X(2,5) X(3,6) X(0,6) X(0,3) X(4,7) X(1,7) X(1,4) \
  X(11,14) X(8,14) X(8,11) X(12,15) X(9,15) X(9,12) X(13,16) X(10,16) X(10,13) X(20,23) \
  X(17,23) X(17,20) X(21,24) X(18,24) X(18,21) X(19,22) X(8,17) X(9,18) X(0,18) X(0,9) \
  X(10,19) X(1,19) X(1,10) X(11,20) X(2,20) X(2,11) X(12,21) X(3,21) X(3,12) X(13,22) \
  X(4,22) X(4,13) X(14,23) X(5,23) X(5,14) X(15,24) X(6,24) X(6,15) X(7,16) X(7,19) \
  X(13,21) X(15,23) X(7,13) X(7,15) X(1,9) X(3,11) X(5,17) X(11,17) X(9,17) X(4,10) \
  X(6,12) X(7,14) X(4,6) X(4,7) X(12,14) X(10,14) X(6,7) X(10,12) X(6,10) X(6,17) \
  X(12,17) X(7,17) X(7,10) X(12,18) X(7,12) X(10,18) X(12,20) X(10,20) X(10,12)
// clang-format on

namespace glh {

template <typename T>
GLH_HD T median25(T* v) {
#define GLH_CE(a, b)                   \
  {                                    \
    T lo_ = v[a] < v[b] ? v[a] : v[b]; \
    T hi_ = v[a] < v[b] ? v[b] : v[a]; \
    v[a] = lo_;                        \
    v[b] = hi_;                        \
  }
  GLH_MED25_NETWORK(GLH_CE)
#undef GLH_CE
  return v[12];
}

#if defined(__HIPCC__)
// Two medians at once on packed 16-bit keys (v_pk_min_u16 / v_pk_max_u16): the same network
// applied lane-wise to the low and high halves, i.e. to two neighbouring pixels.
typedef unsigned short glh_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ glh_us2 median25_pk(glh_us2* v) {
#define GLH_CE(a, b)                                          \
  {                                                           \
    glh_us2 lo_ = __builtin_elementwise_min(v[a], v[b]);      \
    glh_us2 hi_ = __builtin_elementwise_max(v[a], v[b]);      \
    v[a] = lo_;                                               \
    v[b] = hi_;                                               \
  }
  GLH_MED25_NETWORK(GLH_CE)
#undef GLH_CE
  return v[12];
}
#endif

}  // namespace glh
