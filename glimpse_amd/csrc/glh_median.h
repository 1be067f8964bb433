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

// ---- Round 4: the median of a RUN of vertically adjacent 5 x 5 windows from SORTED ROWS -------------------------------
// A thread that produces outputs (r, c) .. (r + 3, c) touches window rows r - 2 .. r + 5; every row's five keys are
// sorted once (9 exchanges) and reused by the up to five windows that contain the row.  Outputs j and j + 1 share the
// four rows j + 1 .. j + 4: those are merged pairwise (two sorted 5-lists -> a sorted 10-list: 13 exchanges, the optimum)
// and of the two 10-lists only ranks 7 .. 12 of their union are formed (the 7 below are below the median of every
// window that contains the four rows whatever the fifth row holds, the 7 above are above it) -- 42 operations; the
// median of such a window is then the element of rank 5 among those six (x) and the fifth row's sorted five (y):
//     max(x0, min(x1, y4), min(x2, y3), min(x3, y2), min(x4, y1), min(x5, y0))        (10 operations).
// Four outputs from eight rows: 8 x 18 + 3 x 26 + 2 x 42 + 4 x 10 = 346 min / max operations, 87 per output against
// the 198 of the 99-exchange network.  tools/median_search.py found the merge and the rank-7..12 selection (simulated
// annealing over comparator lists, cost = operations left after dead-code elimination); tests/test_median_network.py
// proves every piece and the composition with the 0/1 principle over all inputs whose rows are sorted.
// A comparator X(a, b) leaves the minimum on wire a and the maximum on wire b (a > b occurs); outputs that nothing
// reads are dropped by the compiler.
// clang-format off
#define GLH_SORT5_NETWORK(X) X(0,1) X(3,4) X(2,4) X(2,3) X(1,4) X(0,3) X(0,2) X(1,3) X(1,2)
// wires 0..4 and 5..9 sorted ascending -> rank k on wire k
#define GLH_MERGE55_NETWORK(X) \
  X(0,5) X(4,9) X(1,6) X(1,5) X(4,5) X(3,8) X(5,8) X(2,7) X(3,6) X(2,4) X(5,7) X(3,4) X(5,6)
// wires 0..9 and 10..19 sorted ascending -> ranks 7..12 of the twenty on wires 7..12
#define GLH_MID6_NETWORK(X) \
  X(4,3) X(8,10) X(0,16) X(1,15) X(9,11) X(2,12) X(3,12) X(12,18) X(4,19) X(14,3) \
  X(4,13) X(5,17) X(7,15) X(7,3) X(10,3) X(6,8) X(5,12) X(10,12) X(8,16) X(5,9) \
  X(7,9) X(8,13) X(9,10) X(11,13) X(7,8) X(10,11) X(11,12) X(8,9)
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

// The shared-row form on packed 16-bit lanes (see GLH_SORT5_NETWORK above).  V: any type with min / max.
#define GLH_PK_CE(a, b)                                     \
  {                                                         \
    glh_us2 lo_ = __builtin_elementwise_min(v[a], v[b]);    \
    glh_us2 hi_ = __builtin_elementwise_max(v[a], v[b]);    \
    v[a] = lo_;                                             \
    v[b] = hi_;                                             \
  }
__device__ __forceinline__ void med_sort5_pk(glh_us2* v) { GLH_SORT5_NETWORK(GLH_PK_CE) }
// m[0..9] = the merge of the sorted a[0..4] and b[0..4]
__device__ __forceinline__ void med_merge55_pk(const glh_us2* a, const glh_us2* b, glh_us2* v) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    v[k] = a[k];
    v[5 + k] = b[k];
  }
  GLH_MERGE55_NETWORK(GLH_PK_CE)
}
// x[0..5] = ranks 7..12 of the union of the sorted p[0..9] and q[0..9]
__device__ __forceinline__ void med_mid6_pk(const glh_us2* p, const glh_us2* q, glh_us2* x) {
  glh_us2 v[20];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    v[k] = p[k];
    v[10 + k] = q[k];
  }
  GLH_MID6_NETWORK(GLH_PK_CE)
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = v[7 + k];
}
// the element of rank 5 among the sorted x[0..5] and the sorted y[0..4]
__device__ __forceinline__ glh_us2 med_fin_pk(const glh_us2* x, const glh_us2* y) {
  glh_us2 m = x[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) m = __builtin_elementwise_max(m, __builtin_elementwise_min(x[k], y[5 - k]));
  return m;
}
#undef GLH_PK_CE
#endif

}  // namespace glh
