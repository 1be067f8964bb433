// glh_kernels.h -- HIP kernels of the glimpse.Tracker hot path for gfx950 (MI355X, wave64).
//
// One batch = all tracked points of one frame.  Work decomposition:
//   * per-particle kernels  (grid = [ceil(N/256), P]): evolve+project+bbox, sample+weight
//   * per-point kernels     (grid = [P] or [G, P]):     tile prep, SSD, spline fit, resample,
//                                                        moments, template init
// Float64 everywhere the reference is float64; float32 exactly where the reference casts
// (SSD inputs/outputs, tracker.py:609-614).  No MFMA: the SSD is (s-t)^2, not a contraction.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/glimpse_hip.h"
#include "glh_math.h"
#include "glh_median.h"

namespace glh {

constexpr int BLK = 256;
constexpr int WAVE = 64;
constexpr int NWAVES = BLK / WAVE;
constexpr int MAX_OBS = 4;
constexpr int NBINS = 768;  // >= 766 = 3 * 255 + 1 channel-sum keys (RGB); 256 for gray
constexpr int BAND_H = 16;  // rows per median band in k_tileprep
constexpr int GLH_NSTAMP = 24;  // s_memtime stamps per workgroup of the fused kernel (diagnostic: glh_debug_phase_stamps)

struct ObsFrame {
  const CamDev* cam;     // camera of the image matched to this frame
  const uint8_t* frame;  // uint8 [H][W][C], or uint16 [H][W][C] when bits == 16
  int32_t on;            // images[o] >= 0
  int32_t width, height, channels;
  int32_t bits;          // 8 or 16 (unsigned integer samples), 32 (float32 samples) or 64 (float64 samples): glh_observer_set_depth
  uint32_t* bins;        // 16-bit frames: [P][bins16_count(channels)] zeroed key histogram workspace (staged kernels)
  double* fwork;         // float64 frames: [P][fwork_cap] workspace (normalised / matched values of a tile)
  int64_t fwork_cap;
};

// Division of small non-negative integers by a divisor that is the same for the whole workgroup: n / d as
// umulhi(n, ceil(2^32 / d)), exact whenever n * d < 2^32 (here n < 2^17 pixels or outputs, d < 2^9).  The
// compiler's general 32-bit division is ~40 instructions per use; this is one, after one float64 division per
// divisor.
struct UDiv {
  uint32_t m, d;
};
__device__ __forceinline__ UDiv udiv_make(int d) {
  UDiv r;
  r.d = (uint32_t)d;
  r.m = d > 1 ? (uint32_t)(4294967296.0 / (double)d) + 1u : 0u;
  return r;
}
__device__ __forceinline__ int udiv(const UDiv& u, int n) { return u.d > 1 ? (int)__umulhi((uint32_t)n, u.m) : n; }

// ------------------------------------------------------------------------------------------
// wave / block reductions (deterministic order)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, WAVE));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, WAVE));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, WAVE));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, WAVE));
  return v;
}

// Sum over the block; result valid in every thread.  `red` holds >= NWAVES doubles.
template <int NT = BLK>
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = v;
  __syncthreads();
  double t = red[0];
#pragma unroll
  for (int w = 1; w < NT / WAVE; ++w) t += red[w];
  return t;
}

__device__ __forceinline__ void flag_point(uint32_t* pt_status, int32_t* pt_err_frame, int pt,
                                           uint32_t bit, int frame) {
  atomicOr(&pt_status[pt], bit);
  atomicMin(&pt_err_frame[pt], frame);
}

// ------------------------------------------------------------------------------------------
// normals: host-fed (parity with np.random) or Philox + Box-Muller
// ------------------------------------------------------------------------------------------
// Standard normals from one Philox4x32 block (glh_math.h: 7 rounds).  Bench-mode process noise does not need
// float64 transcendentals: the Box-Muller radius/angle run on the float32 hardware units
// (v_log_f32, v_sqrt_f32, v_sin/cos_f32 in revolutions), |z| <= 6.6, then widen to float64.
__device__ __forceinline__ void box_muller_f32(uint32_t ra, uint32_t rb, double& z0, double& z1) {
  const float u1 = ((float)ra + 0.5f) * 2.3283064365386963e-10f;   // (0, 1]
  const float u2 = (float)(rb >> 8) * 5.9604644775390625e-08f;      // [0, 1) revolutions
  // raw v_sqrt_f32 (1 ulp): an IEEE-exact square root would cost ~12 more instructions per normal pair
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1)
  z0 = (double)(rad * __builtin_amdgcn_cosf(u2));
  z1 = (double)(rad * __builtin_amdgcn_sinf(u2));
}
__device__ __forceinline__ void philox_normals2(uint64_t seed, uint32_t c0, uint32_t c1,
                                                uint32_t c2, uint32_t c3, double& z0, double& z1) {
  uint32_t r[4];
  philox4x32(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  box_muller_f32(r[0], r[1], z0, z1);
}
// three normals from ONE block (all four output words are used)
__device__ __forceinline__ void philox_normals3(uint64_t seed, uint32_t c0, uint32_t c1,
                                                uint32_t c2, uint32_t c3, double* z) {
  uint32_t r[4];
  philox4x32(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  double dump;
  box_muller_f32(r[0], r[1], z[0], z[1]);
  box_muller_f32(r[2], r[3], z[2], dump);
}

// The three evolve normals of particle i of point pt at frame `step`: host-fed (parity with
// np.random, motion.py:176) or counter-based, hence recomputable wherever the particle is
// needed again (the fused step re-evolves the resampled sources instead of storing them).
// `third` (uniform): false where the caller knows the third normal is multiplied by a zero sigma -- it is then not
// computed (the second Box-Muller pair of the block) and reads 0.
__device__ __forceinline__ void evolve_noise(int rng_mode, const double* normals, uint64_t seed, uint64_t step,
                                             int pt, int pt_base, int i, int N, double* n, bool third = true) {
  if (rng_mode == GLH_RNG_HOST) {
    const double* src = normals + ((size_t)pt * N + i) * 3;
    n[0] = src[0]; n[1] = src[1]; n[2] = src[2];
  } else if (third) {
    philox_normals3(seed, (uint32_t)i, (uint32_t)(pt + pt_base), (uint32_t)step, 0x45564f4cu, n);
  } else {
    philox_normals2(seed, (uint32_t)i, (uint32_t)(pt + pt_base), (uint32_t)step, 0x45564f4cu, n[0], n[1]);
    n[2] = 0.0;
  }
}

// Motion.initialize_particles for one particle from its six normals n = randn(n,2) | randn(n) |
// randn(n,3) (the tangent models draw randn(n,2) last and leave vz = 0):
// motion.py:149-163, :262-286, :382-394, :470-488.  m = [GLH_MOTION_FULL_LEN] parameters.
__device__ __forceinline__ void init_particle(const double* m, const double* n, double* p, const Surfaces& surf,
                                              bool* oob) {
  const int kind = (int)m[18];
  p[0] = m[0] + m[2] * n[0];
  p[1] = m[1] + m[3] * n[1];
  const double zd = dem_at(m, surf, p[0], p[1], oob), zs = dem_sigma_at(m, surf, p[0], p[1], oob);
  if (kind == GLH_MOTION_CARTESIAN || kind == GLH_MOTION_CYLINDRICAL) {
    double z = zd;
    z += zs * n[2];
    p[2] = z;
  } else {
    const double z_off = zs * n[2];
    p[2] = zd + z_off;
  }
  if (kind == GLH_MOTION_CARTESIAN) {
    p[3] = m[4] + m[7] * n[3];
    p[4] = m[5] + m[8] * n[4];
    p[5] = m[6] + m[9] * n[5];
  } else if (kind == GLH_MOTION_CYLINDRICAL) {
    const double vr = m[4] + m[7] * n[3], th = m[5] + m[8] * n[4];
    p[3] = vr * cos(th);
    p[4] = vr * sin(th);
    p[5] = m[6] + m[9] * n[5];
  } else if (kind == GLH_MOTION_TANGENT_CARTESIAN) {
    p[3] = m[4] + m[7] * n[3];
    p[4] = m[5] + m[8] * n[4];
    p[5] = 0.0;
  } else {
    const double vr = m[4] + m[7] * n[3], th = m[5] + m[8] * n[4];
    p[3] = vr * cos(th);
    p[4] = vr * sin(th);
    p[5] = 0.0;
  }
}

// Tracker.test_particles beyond the NaN test (tracker.py:114-117): every particle must sit on a
// visible viewshed cell (nearest-cell lookup).  Returns the status bits to raise.
__device__ __forceinline__ uint32_t viewshed_bits(const Surfaces& surf, double x, double y) {
  if (!surf.viewshed.z) return 0u;
  bool oob = false;
  const double vis = raster_sample(surf.viewshed, x, y, 0, &oob);
  if (oob) return GLH_PT_RASTER_OOB;
  return vis != 0.0 ? 0u : GLH_PT_NOT_VISIBLE;
}

// CartesianMotion.compute_log_likelihoods (motion.py:181-204) for one (evolved) particle
// FAST: the surface samples in fast arithmetic (glh_math.h: raster_bilinear_fast), the scale by a Newton reciprocal.
template <bool FAST = false>
__device__ __forceinline__ double dem_log_likelihood(const double* m, const Surfaces& surf, double x, double y,
                                                     double z, bool* oob, const RasterPatch* patches = nullptr,
                                                     const RasterWin* wins = nullptr) {
  double zd, zs;
  // (both surfaces rasters on one grid, the sample inside their windows: the cell and the weights once for both)
  if (!(patches && m[20] != 0.0 && m[21] != 0.0 &&
        raster_sample_pair<FAST>(surf.dem, surf.dem_sigma, surf.same_grid != 0, patches, patches + 1, x, y, zd, zs, wins))) {
    zd = dem_at<FAST>(m, surf, x, y, oob, patches, wins);
    zs = dem_sigma_at<FAST>(m, surf, x, y, oob, patches, wins);
  }
  if (zs != 0.0) {
    const double d = zd - z;
    if constexpr (FAST) return rcp_nr(2.0 * (zs * zs)) * (d * d);
    return (1.0 / (2.0 * (zs * zs))) * (d * d);
  }
  return 0.0;
}

// Wave64 reductions and inclusive scans on the DPP data path (VALU moves; the canonical row_shr / row_bcast ladder)
// instead of ds_bpermute shuffles through the LDS crossbar: lane i ends with OP over lanes 0 .. i (lane 63: the total).
template <int CTRL, int ROW_MASK, int BANK_MASK, bool ZERO_FILL>
__device__ __forceinline__ double pt_dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // ZERO_FILL: lanes without a source read 0 (identity of +); otherwise they keep their own value
  // (identity of min / max)
  lo = __builtin_amdgcn_update_dpp(ZERO_FILL ? 0 : lo, lo, CTRL, ROW_MASK, BANK_MASK, ZERO_FILL);
  hi = __builtin_amdgcn_update_dpp(ZERO_FILL ? 0 : hi, hi, CTRL, ROW_MASK, BANK_MASK, ZERO_FILL);
  return __hiloint2double(hi, lo);
}
// row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
#define PT_DPP_LADDER(OP, ZF)                          \
  v = OP(v, (pt_dpp_mov<0x111, 0xf, 0xf, ZF>(v)));     \
  v = OP(v, (pt_dpp_mov<0x112, 0xf, 0xf, ZF>(v)));     \
  v = OP(v, (pt_dpp_mov<0x114, 0xf, 0xe, ZF>(v)));     \
  v = OP(v, (pt_dpp_mov<0x118, 0xf, 0xc, ZF>(v)));     \
  v = OP(v, (pt_dpp_mov<0x142, 0xa, 0xf, ZF>(v)));     \
  v = OP(v, (pt_dpp_mov<0x143, 0xc, 0xf, ZF>(v)));
__device__ __forceinline__ double pt_add(double a, double b) { return a + b; }
__device__ __forceinline__ double pt_wave_sum63(double v) {
  PT_DPP_LADDER(pt_add, true)
  return v;
}
__device__ __forceinline__ double pt_wave_min63(double v) {
  PT_DPP_LADDER(fmin, false)
  return v;
}
__device__ __forceinline__ double pt_wave_max63(double v) {
  PT_DPP_LADDER(fmax, false)
  return v;
}

// inclusive scans of 32-bit values (lanes without a source read 0: the identity of + and of an unsigned max)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t pt_dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, true);
}
#define PT_DPP_LADDER_U32(OP)                        \
  v = OP(v, (pt_dpp_u32<0x111, 0xf, 0xf>(v)));       \
  v = OP(v, (pt_dpp_u32<0x112, 0xf, 0xf>(v)));       \
  v = OP(v, (pt_dpp_u32<0x114, 0xf, 0xe>(v)));       \
  v = OP(v, (pt_dpp_u32<0x118, 0xf, 0xc>(v)));       \
  v = OP(v, (pt_dpp_u32<0x142, 0xa, 0xf>(v)));       \
  v = OP(v, (pt_dpp_u32<0x143, 0xc, 0xf>(v)));
__device__ __forceinline__ uint32_t pt_add_u32(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t pt_max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
  PT_DPP_LADDER_U32(pt_add_u32)
  return v;
}
__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v) {
  PT_DPP_LADDER_U32(pt_max_u32)
  return v;
}
__device__ __forceinline__ double wave_scan_add_f64(double v) { return pt_wave_sum63(v); }
// lane i takes lane i - 1's value, lane 0 takes 0 (wave_shr:1 = 0x138)
__device__ __forceinline__ uint32_t wave_shr1_u32(uint32_t v) { return pt_dpp_u32<0x138, 0xf, 0xf>(v); }
__device__ __forceinline__ double wave_shr1_f64(double v) { return pt_dpp_mov<0x138, 0xf, 0xf, true>(v); }

// Sum over groups of G = 2, 4, 8 or 16 adjacent lanes, every lane ending with its group's total: the butterfly
// v += shfl_xor(v, 1), 2, 4, 8 on the DPP path.  quad_perm [1,0,3,2] / [2,3,0,1] are the xor-1 / xor-2 exchanges; for
// the third and fourth step row_half_mirror (i <-> 7 - i) and row_mirror (i <-> 15 - i) pair every lane with one of
// the OTHER half, which by then holds that half's total in all its lanes -- the same two operands as the xor
// exchange, and a + b == b + a, so the float64 result is bit for bit the shuffle butterfly's.
__device__ __forceinline__ double group_sum_dpp(double v, int G) {
  if (G > 1) v += pt_dpp_mov<0xb1, 0xf, 0xf, false>(v);   // quad_perm:[1,0,3,2]
  if (G > 2) v += pt_dpp_mov<0x4e, 0xf, 0xf, false>(v);   // quad_perm:[2,3,0,1]
  if (G > 4) v += pt_dpp_mov<0x141, 0xf, 0xf, false>(v);  // row_half_mirror
  if (G > 8) v += pt_dpp_mov<0x140, 0xf, 0xf, false>(v);  // row_mirror
  return v;
}

// Motion.evolve_particles for one particle p[6], tau2 = tau * tau, n = the step's three normals
// (randn(n,3); the tangent models draw randn(n,2) then randn(n)):
// motion.py:165-179, :288-311, :396-412, :490-522.
__device__ __forceinline__ void evolve_cartesian(double* p, const double* m, const double* n, double tau,
                                                 double tau2) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double acc = m[10 + k] + m[13 + k] * n[k];
    p[k] += tau * p[3 + k] + 0.5 * acc * tau2;
    p[3 + k] += tau * acc;
  }
}
// the same step in fast arithmetic (glh_math.h): 4 fused multiply-adds per axis instead of 8 operations
__device__ __forceinline__ void evolve_cartesian_fast(double* p, const double* m, const double* n, double tau,
                                                      double tau2) {
  const double h = 0.5 * tau2;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double acc = glh_fma(m[13 + k], n[k], m[10 + k]);
    p[k] = glh_fma(h, acc, glh_fma(tau, p[3 + k], p[k]));
    p[3 + k] = glh_fma(tau, acc, p[3 + k]);
  }
}
template <bool FAST>
__device__ __forceinline__ void evolve_cartesian_m(double* p, const double* m, const double* n, double tau,
                                                   double tau2) {
  if (FAST)
    evolve_cartesian_fast(p, m, n, tau, tau2);
  else
    evolve_cartesian(p, m, n, tau, tau2);
}
// GRID = false: the caller knows that this point's surfaces are constants (m[20] = m[21] = 0) -- no raster code is compiled
// into that copy (the fused kernel's phase A keeps one loop for each: with the raster samples in the body, the loop of a
// run over constant surfaces paid for their registers)
// ZKNOWN (tangent models only): the caller has the evolved height (the fused kernel parks it in phase A for the gather's
// re-evolution) -- the two surface samples, the square root and the third normal are not needed again.
template <bool FAST = false, bool GRID = true, bool ZKNOWN = false>
__device__ __forceinline__ void evolve_particle(double* p, const double* m, const double* n, double tau,
                                                double tau2, const Surfaces& surf, bool* oob,
                                                const RasterPatch* patches = nullptr, double z_known = 0.0,
                                                const RasterWin* wins = nullptr) {
  const int kind = (int)m[18];
  if (kind == GLH_MOTION_CARTESIAN) {
    evolve_cartesian_m<FAST>(p, m, n, tau, tau2);
    return;
  }
  if (kind == GLH_MOTION_EXTERNAL) return;  // evolved by the caller (a user-defined Motion): nothing to do here
  const bool cyl = kind == GLH_MOTION_CYLINDRICAL || kind == GLH_MOTION_TANGENT_CYLINDRICAL;
  const bool tangent = kind >= GLH_MOTION_TANGENT_CARTESIAN;
  double a[3];
  if constexpr (FAST) {
    a[0] = glh_fma(m[13], n[0], m[10]);
    a[1] = glh_fma(m[14], n[1], m[11]);
    a[2] = tangent ? 0.0 : glh_fma(m[15], n[2], m[12]);
  } else {
    a[0] = m[10] + m[13] * n[0];
    a[1] = m[11] + m[14] * n[1];
    a[2] = tangent ? 0.0 : m[12] + m[15] * n[2];
  }
  if (cyl) {
    // (r'', theta') -> (x'', y''): r'' * cos(th) - r' * sin(th) * th',  r'' * sin(th) + r' * cos(th) * th'
    const double vx = p[3], vy = p[4];
    const double vr = sqrt(vx * vx + vy * vy);
    const double ar = a[0], ath = a[1];
    a[0] = ar * (vx / vr) - vy * ath;
    a[1] = ar * (vy / vr) + vx * ath;
  }
  if (!tangent) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p[k] += tau * p[3 + k] + 0.5 * a[k] * tau2;
      p[3 + k] += tau * a[k];
    }
    return;
  }
  // tangent models: the height follows the surface plus a random walk of the offset
  // (FAST: fused multiply-adds, the two surface samples in fast arithmetic -- raster_bilinear_fast --, a Newton square root)
  const double dx = FAST ? glh_fma(0.5 * tau2, a[0], tau * p[3]) : tau * p[3] + 0.5 * a[0] * tau2;
  const double dy = FAST ? glh_fma(0.5 * tau2, a[1], tau * p[4]) : tau * p[4] + 0.5 * a[1] * tau2;
  if constexpr (ZKNOWN) {
    p[0] += dx;
    p[1] += dy;
    p[2] = z_known;
    if constexpr (FAST) {
      p[3] = glh_fma(tau, a[0], p[3]);
      p[4] = glh_fma(tau, a[1], p[4]);
    } else {
      p[3] += tau * a[0];
      p[4] += tau * a[1];
    }
    return;
  }
  double z_old, z_new;
  bool both = false;
  if constexpr (FAST && GRID) {
    // (fast arithmetic, the fused kernel's window: both samples at once -- the second usually in the first one's cell)
    if (patches && m[20] != 0.0 && patches->full)
      both = raster_sample_window2(patches, p[0], p[1], p[0] + dx, p[1] + dy, z_old, z_new, wins);
  }
  if (!both) z_old = GRID ? dem_at<FAST>(m, surf, p[0], p[1], oob, patches, wins) : m[16];
  double z_off = p[2] - z_old;
  if constexpr (FAST)
    z_off = glh_fma(m[19] * n[2], sqrt_nr(glh_fma(dx, dx, dy * dy)), z_off);
  else
    z_off += m[19] * n[2] * sqrt(dx * dx + dy * dy);
  p[0] += dx;
  p[1] += dy;
  if (!both) z_new = GRID ? dem_at<FAST>(m, surf, p[0], p[1], oob, patches, wins) : m[16];
  p[2] = z_new + z_off;
  if constexpr (FAST) {
    p[3] = glh_fma(tau, a[0], p[3]);
    p[4] = glh_fma(tau, a[1], p[4]);
  } else {
    p[3] += tau * a[0];
    p[4] += tau * a[1];
  }
}

// ------------------------------------------------------------------------------------------
// K0  CartesianMotion.initialize_particles (motion.py:149-163) + initialize_weights
// ------------------------------------------------------------------------------------------
struct InitArgs {
  double* particles;  // [P][N][6]
  double* weights;    // [P][N]
  const double* motion;
  const uint8_t* active;
  const double* normals;  // [P][N][6] or null
  uint64_t seed;
  int32_t rng_mode, N, pt_base;  // pt_base: global index of point 0 (sharding-invariant Philox streams)
  int32_t frame;
  uint32_t* pt_status;
  int32_t* pt_err_frame;
  Surfaces surf;
};

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_init_particles(InitArgs a) {
  const int pt = blockIdx.y;
  if (a.active && !a.active[pt]) return;
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= a.N) return;
  const double* m = a.motion + (size_t)pt * GLH_MOTION_FULL_LEN;
  double n[6];
  if (a.rng_mode == GLH_RNG_HOST) {
    const double* src = a.normals + ((size_t)pt * a.N + i) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) n[k] = src[k];
  } else {
    const uint32_t gp = (uint32_t)(pt + a.pt_base);
    philox_normals2(a.seed, i, gp, 0u, 0x494e4954u, n[0], n[1]);
    philox_normals2(a.seed, i, gp, 1u, 0x494e4954u, n[2], n[3]);
    philox_normals2(a.seed, i, gp, 2u, 0x494e4954u, n[4], n[5]);
  }
  double x[6];
  bool oob = false;
  init_particle(m, n, x, a.surf, &oob);
  uint32_t bits = viewshed_bits(a.surf, x[0], x[1]);
  if (oob) bits |= GLH_PT_RASTER_OOB;
  if (bits) flag_point(a.pt_status, a.pt_err_frame, pt, bits, a.frame);
  double* p = a.particles + ((size_t)pt * a.N + i) * 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) p[k] = x[k];
  a.weights[(size_t)pt * a.N + i] = 1.0;
}
#endif

// ------------------------------------------------------------------------------------------
// K1  evolve (motion.py:165-179) + NaN test (tracker.py:118) + project (camera.py:591)
//     + per-block uv bounding box partials (tracker.py:583).
// ------------------------------------------------------------------------------------------
struct EvolveArgs {
  double* particles;  // [P][N][6] current buffer (updated in place when `store`)
  const double* motion;
  const uint8_t* active;
  const uint8_t* obs_mask;  // [P][O] or null
  const double* normals;    // [P][N][3] or null
  double* uv;               // [O][P][N][2]
  double* bbox_part;        // [O][P][NB][5]
  uint32_t* pt_status;
  int32_t* pt_err_frame;
  uint64_t seed, step;
  double tau;
  int32_t do_evolve, store, rng_mode, N, P, O, NB, frame, pt_base;
  int32_t fast;  // GLH_MATH_FAST: fast arithmetic (glh_math.h), like the fused kernel's FAST instantiation
  Surfaces surf;
  ObsFrame obs[MAX_OBS];
};

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_evolve_project(EvolveArgs a) {
  __shared__ double red[NWAVES][5];
  const int pt = blockIdx.y;
  if (a.active && !a.active[pt]) return;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * BLK + tid;
  const bool valid = i < a.N;
  double p[6] = {0, 0, 0, 0, 0, 0};
  double* pp = a.particles + ((size_t)pt * a.N + (valid ? i : 0)) * 6;
  if (valid) {
    const double2* src = reinterpret_cast<const double2*>(pp);
    double2 v0 = src[0], v1 = src[1], v2 = src[2];
    p[0] = v0.x; p[1] = v0.y; p[2] = v1.x; p[3] = v1.y; p[4] = v2.x; p[5] = v2.y;
    if (a.do_evolve) {
      const double* m = a.motion + (size_t)pt * GLH_MOTION_FULL_LEN;
      double n[3];
      evolve_noise(a.rng_mode, a.normals, a.seed, a.step, pt, a.pt_base, i, a.N, n);
      bool oob = false;
      if (a.fast)
        evolve_particle<true>(p, m, n, a.tau, a.tau * a.tau, a.surf, &oob);
      else
        evolve_particle<false>(p, m, n, a.tau, a.tau * a.tau, a.surf, &oob);
      uint32_t bits = viewshed_bits(a.surf, p[0], p[1]);
      if (oob) bits |= GLH_PT_RASTER_OOB;
      if (bits) flag_point(a.pt_status, a.pt_err_frame, pt, bits, a.frame);
      if (a.store) {
        double2* dst = reinterpret_cast<double2*>(pp);
        dst[0] = make_double2(p[0], p[1]);
        dst[1] = make_double2(p[2], p[3]);
        dst[2] = make_double2(p[4], p[5]);
      }
    }
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) bad |= isnan(p[k]);
    if (bad) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_NAN, a.frame);
  }
  for (int o = 0; o < a.O; ++o) {
    if (!a.obs[o].on) continue;
    if (a.obs_mask && !a.obs_mask[(size_t)pt * a.O + o]) continue;
    double u = 0.0, v = 0.0;
    double mnu = INFINITY, mnv = INFINITY, mxu = -INFINITY, mxv = -INFINITY, nanf = 0.0;
    if (valid) {
      if (a.fast)
        project_fast(*a.obs[o].cam, cam_flags(*a.obs[o].cam), p[0], p[1], p[2], u, v);
      else
        project(*a.obs[o].cam, p[0], p[1], p[2], u, v);
      reinterpret_cast<double2*>(a.uv)[((size_t)o * a.P + pt) * a.N + i] = make_double2(u, v);
      if (isnan(u) || isnan(v)) {
        nanf = 1.0;
      } else {
        mnu = mxu = u;
        mnv = mxv = v;
      }
    }
    mnu = wave_min(mnu); mnv = wave_min(mnv);
    mxu = wave_max(mxu); mxv = wave_max(mxv);
    nanf = wave_max(nanf);
    __syncthreads();
    if ((tid & (WAVE - 1)) == 0) {
      double* r = red[tid / WAVE];
      r[0] = mnu; r[1] = mnv; r[2] = mxu; r[3] = mxv; r[4] = nanf;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NWAVES; ++w) {
        mnu = fmin(mnu, red[w][0]); mnv = fmin(mnv, red[w][1]);
        mxu = fmax(mxu, red[w][2]); mxv = fmax(mxv, red[w][3]);
        nanf = fmax(nanf, red[w][4]);
      }
      double* out = a.bbox_part + (((size_t)o * a.P + pt) * a.NB + blockIdx.x) * 5;
      out[0] = mnu; out[1] = mnv; out[2] = mxu; out[3] = mxv; out[4] = nanf;
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------
// K7  particle_mean / compute_particle_sigma (tracker.py:72-76, :89-104)
//     mean = sum(p*w)/sum(w);  sigma = sqrt(sum((p-mean)^2 * w)/sum(w)).  One block per point.
// ------------------------------------------------------------------------------------------
struct MomentsArgs {
  const double* particles;
  const double* weights;
  const uint8_t* active;
  double* out;         // row stride `ld` doubles per point: mean(6) [| sigma(6)]
  int32_t N, ld, with_sigma;
};

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_moments(MomentsArgs a) {
  __shared__ double red[NWAVES];
  const int pt = blockIdx.x;
  if (a.active && !a.active[pt]) return;
  const int tid = threadIdx.x;
  const double* P0 = a.particles + (size_t)pt * a.N * 6;
  const double* W0 = a.weights + (size_t)pt * a.N;
  double sw = 0.0, s[6] = {0, 0, 0, 0, 0, 0};
  for (int i = tid; i < a.N; i += BLK) {
    double w = W0[i];
    const double2* src = reinterpret_cast<const double2*>(P0 + (size_t)i * 6);
    double2 v0 = src[0], v1 = src[1], v2 = src[2];
    sw += w;
    s[0] += v0.x * w; s[1] += v0.y * w; s[2] += v1.x * w;
    s[3] += v1.y * w; s[4] += v2.x * w; s[5] += v2.y * w;
  }
  sw = block_sum(sw, red);
  double mean[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) mean[k] = block_sum(s[k], red) / sw;
  double* out = a.out + (size_t)pt * a.ld;
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] = mean[k];
  }
  if (!a.with_sigma) return;
  double q[6] = {0, 0, 0, 0, 0, 0};
  for (int i = tid; i < a.N; i += BLK) {
    double w = W0[i];
    const double2* src = reinterpret_cast<const double2*>(P0 + (size_t)i * 6);
    double2 v0 = src[0], v1 = src[1], v2 = src[2];
    double d;
    d = v0.x - mean[0]; q[0] += d * d * w;
    d = v0.y - mean[1]; q[1] += d * d * w;
    d = v1.x - mean[2]; q[2] += d * d * w;
    d = v1.y - mean[3]; q[3] += d * d * w;
    d = v2.x - mean[4]; q[4] += d * d * w;
    d = v2.y - mean[5]; q[5] += d * d * w;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double var = block_sum(q[k], red) / sw;
    if (tid == 0) out[6 + k] = sqrt(var);
  }
}
#endif

// ------------------------------------------------------------------------------------------
// Tracker.particle_covariance (tracker.py:78-82): np.cov(particles.T, aweights=weights, ddof=0)
//   avg = sum(w x)/sum(w);  cov[i][j] = sum(w (x_i - avg_i)(x_j - avg_j)) / sum(w).  One block per point.
// ------------------------------------------------------------------------------------------
struct CovArgs {
  const double* particles;
  const double* weights;
  const uint16_t* uidx;  // [P][N] or null: the state is run-length compact (planar records, uidx[j] = record of particle j)
  const uint8_t* active;
  double* out;  // [P][36]
  int32_t N;
};

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_covariance(CovArgs a) {
  __shared__ double red[NWAVES];
  const int pt = blockIdx.x;
  if (a.active && !a.active[pt]) return;
  const int tid = threadIdx.x;
  const double* P0 = a.particles + (size_t)pt * a.N * 6;
  const double* W0 = a.weights + (size_t)pt * a.N;
  // particle i of the point: its record, expanded or compact -- the same values in the same order either way, so the
  // sums (and the covariance) are bit for bit those of the expanded state
  const uint16_t* U0 = a.uidx ? a.uidx + (size_t)pt * a.N : nullptr;
  auto load = [&](int i, double2& v0, double2& v1, double2& v2) -> double {
    if (U0) {
      const int r = U0[i];
      const double2* src = reinterpret_cast<const double2*>(P0) + r;  // planar: chunk c of record r at c N + r
      v0 = src[0]; v1 = src[a.N]; v2 = src[2 * (size_t)a.N];
      return W0[r];
    }
    const double2* src = reinterpret_cast<const double2*>(P0 + (size_t)i * 6);
    v0 = src[0]; v1 = src[1]; v2 = src[2];
    return W0[i];
  };
  double sw = 0.0, s[6] = {0, 0, 0, 0, 0, 0};
  for (int i = tid; i < a.N; i += BLK) {
    double2 v0, v1, v2;
    const double w = load(i, v0, v1, v2);
    sw += w;
    s[0] += v0.x * w; s[1] += v0.y * w; s[2] += v1.x * w;
    s[3] += v1.y * w; s[4] += v2.x * w; s[5] += v2.y * w;
  }
  sw = block_sum(sw, red);
  double mean[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) mean[k] = block_sum(s[k], red) / sw;
  double q[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) q[k] = 0.0;
  for (int i = tid; i < a.N; i += BLK) {
    double2 v0, v1, v2;
    const double w = load(i, v0, v1, v2);
    const double d[6] = {v0.x - mean[0], v0.y - mean[1], v1.x - mean[2], v1.y - mean[3], v2.x - mean[4], v2.y - mean[5]};
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double wr = w * d[r];
#pragma unroll
      for (int cidx = r; cidx < 6; ++cidx) q[k++] += wr * d[cidx];
    }
  }
  double* out = a.out + (size_t)pt * 36;
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int cidx = r; cidx < 6; ++cidx) {
      const double v = block_sum(q[k++], red) * (1.0 / sw);
      if (tid == 0) {
        out[r * 6 + cidx] = v;
        out[cidx * 6 + r] = v;
      }
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------
// pixel keys: gray value, or channel sum for RGB (tile.mean(axis=2) is sum/3, tracker.py:524)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int pixel_key(const uint8_t* frame, int width, int channels, int row,
                                         int col) {
  const uint8_t* px = frame + ((size_t)row * width + col) * channels;
  if (channels == 1) return px[0];
  int s = 0;
  for (int c = 0; c < channels; ++c) s += px[c];
  return s;
}
__device__ __forceinline__ double key_value(int key, int channels) {
  return channels == 1 ? (double)key : (double)key / (double)channels;
}

// ------------------------------------------------------------------------------------------
// Template tile from an integer box: Tracker.extract_tile(return_histogram=True)
// (tracker.py:494-534): normalize (helpers.py:344) -> CDF (helpers.py:458-464) ->
// tile - median5x5(tile).  normalize is affine increasing, so the median is taken on the
// integer keys.  Whole block cooperates; keys/hist live in LDS.
// ------------------------------------------------------------------------------------------
struct TemplateOut {
  double* tile64;  // [th*tw]
  float* tile32;   // [th*tw]
  double* hist_v;  // [th*tw]
  double* hist_q;  // [th*tw]
  int32_t* hist_n;
};

// Median of the (2 ry + 1) x (2 rx + 1) window of raw keys around (r, c) of a w x h key tile with row stride `ld`
// whose row 0 sits at tile row `row0` (scipy.ndimage.median_filter(size=(2 ry + 1, 2 rx + 1)), mode 'reflect':
// Tracker(highpass={"size": ...}), tracker.py:59, :530).  Any odd size up to 7 x 7: the smallest key with at least
// half the window at or below it, by bisection over the key range (the 5 x 5 default has its own network).
__device__ __forceinline__ int median_window(const uint16_t* keys, int ld, int row0, int w, int h, int r, int c,
                                             int rx, int ry_mode, int max_key) {
  const int ry = GLH_HP_RY(ry_mode), mode = GLH_HP_MODE(ry_mode);  // (the boundary mode rides in the half height)
  const int need = ((2 * rx + 1) * (2 * ry + 1) + 1) / 2;
  int lo = 0, hi = max_key;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    int cnt = 0;
    for (int dr = -ry; dr <= ry; ++dr) {
      const uint16_t* row = keys + (border_index(r + dr, h, mode) - row0) * ld;
      for (int dc = -rx; dc <= rx; ++dc) cnt += row[border_index(c + dc, w, mode)] <= mid;
    }
    if (cnt >= need)
      hi = mid;
    else
      lo = mid + 1;
  }
  return lo;
}

__device__ void template_from_box(const uint8_t* frame, int width, int channels, const int* box,
                                  uint16_t* keys, uint32_t* hist, double* red, TemplateOut out,
                                  bool* const_tile, int hp_rx = 2, int hp_ry = 2) {
  const int tid = threadIdx.x;
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  for (int b = tid; b < NBINS; b += BLK) hist[b] = 0;
  __syncthreads();
  double sx = 0.0;
  for (int idx = tid; idx < n; idx += BLK) {
    int r = idx / w, c = idx - r * w;
    int key = pixel_key(frame, width, channels, box[1] + r, box[0] + c);
    keys[idx] = (uint16_t)key;
    atomicAdd(&hist[key], 1u);
    sx += key_value(key, channels);
  }
  const double mean = block_sum(sx, red) / (double)n;  // a.mean()
  double sq = 0.0;
  for (int idx = tid; idx < n; idx += BLK) {
    double d = key_value(keys[idx], channels) - mean;
    sq += d * d;
  }
  const double var = block_sum(sq, red) / (double)n;  // a.var()
  const double inv_std = 1.0 / sqrt(var);             // 1 / a.std()
  if (tid == 0) {
    *const_tile = !(var > 0.0);
    // np.unique + cumsum(counts) / size (helpers.py:459-461), in increasing key order
    const int nb = channels == 1 ? 256 : 255 * channels + 1;
    uint32_t cum = 0;
    int k = 0;
    for (int b = 0; b < nb && b < NBINS; ++b) {
      uint32_t cnt = hist[b];
      if (cnt) {
        cum += cnt;
        out.hist_v[k] = (key_value(b, channels) - mean) * inv_std;
        out.hist_q[k] = (double)cum / (double)n;
        ++k;
      }
    }
    *out.hist_n = k;
  }
  __syncthreads();
  for (int idx = tid; idx < n; idx += BLK) {
    int r = idx / w, c = idx - r * w;
    int v[25];
#pragma unroll
    for (int dr = -2; dr <= 2; ++dr) {
      int rr = reflect_index(r + dr, h);
#pragma unroll
      for (int dc = -2; dc <= 2; ++dc) {
        int cc = reflect_index(c + dc, w);
        v[(dr + 2) * 5 + (dc + 2)] = keys[rr * w + cc];
      }
    }
    int med = median25(v);
    if (hp_rx != 2 || hp_ry != 2)
      med = median_window(keys, w, 0, w, h, r, c, hp_rx, hp_ry, channels == 1 ? 255 : 255 * channels);
    double x = (key_value(keys[idx], channels) - mean) * inv_std;
    double xm = (key_value(med, channels) - mean) * inv_std;
    double t = x - xm;
    out.tile64[idx] = t;
    out.tile32[idx] = (float)t;
  }
}

// ------------------------------------------------------------------------------------------
// 16-bit frames (uint16 gray or RGB; tracker.py:494-534 works on any dtype).  Same algorithms as above on wider
// keys: a key is the pixel value (gray) or the channel sum (RGB, <= 3 * 65535), the distinct values of a tile and
// their cumulative counts come from a per-point histogram in HBM (`bins`, zeroed by the host before the launch)
// instead of 256 / 766 bins in LDS, and the CDF-matched value of a key is interpolated where it is needed instead of
// being tabulated.  Staged kernels only.
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int bins16_count(int channels) { return 65535 * channels + 1; }

__device__ __forceinline__ int pixel_key16(const uint8_t* frame, int width, int channels, int row, int col) {
  const uint16_t* px = reinterpret_cast<const uint16_t*>(frame) + ((size_t)row * width + col) * channels;
  if (channels == 1) return px[0];
  int s = 0;
  for (int c = 0; c < channels; ++c) s += px[c];
  return s;
}

// exclusive prefix of `v` over the block (all threads call); `tmp` holds NWAVES words; *total = block sum
template <int NT = BLK>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* tmp, uint32_t* total) {
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    uint32_t t = __shfl_up(incl, off, WAVE);
    if (lane >= off) incl += t;
  }
  __syncthreads();
  if (lane == WAVE - 1) tmp[tid / WAVE] = incl;
  __syncthreads();
  uint32_t base = 0, all = 0;
  for (int w = 0; w < NT / WAVE; ++w) {
    if (w < tid / WAVE) base += tmp[w];
    all += tmp[w];
  }
  *total = all;
  return base + incl - v;
}

// The same median by bisection over the key range [0, key_max]: no private array (median_window32 below gathers the
// window into 49 ints with dynamic indices -- scratch memory, which the fused kernel must not reserve)
__device__ __forceinline__ int median_window32_bisect(const uint32_t* keys, int ld, int w, int h, int r, int c, int rx,
                                                      int ry_mode, int key_max) {
  const int ry = GLH_HP_RY(ry_mode), mode = GLH_HP_MODE(ry_mode);
  const int need = ((2 * rx + 1) * (2 * ry + 1) + 1) / 2;
  int lo = 0, hi = key_max;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    int cnt = 0;
    for (int dr = -ry; dr <= ry; ++dr) {
      const uint32_t* row = keys + (size_t)border_index(r + dr, h, mode) * ld;
      for (int dc = -rx; dc <= rx; ++dc) cnt += (int)row[border_index(c + dc, w, mode)] <= mid;
    }
    if (cnt >= need) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// median of the window around (r, c) of a w x h tile of 32-bit keys (row stride ld, tile row `row0` first)
__device__ __forceinline__ int median_window32(const uint32_t* keys, int ld, int row0, int w, int h, int r, int c,
                                               int rx, int ry_mode) {
  const int ry = GLH_HP_RY(ry_mode), mode = GLH_HP_MODE(ry_mode);
  int v[49];
  const int nx = 2 * rx + 1, n = nx * (2 * ry + 1);
  for (int dr = -ry; dr <= ry; ++dr) {
    const uint32_t* row = keys + (border_index(r + dr, h, mode) - row0) * ld;
    for (int dc = -rx; dc <= rx; ++dc) v[(dr + ry) * nx + (dc + rx)] = (int)row[border_index(c + dc, w, mode)];
  }
  if (rx == 2 && ry == 2) return median25(v);
  // any other odd window: the smallest element with at least half the window at or below it
  const int need = (n + 1) / 2;
  int best = 0x7fffffff;
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    for (int j = 0; j < n; ++j) cnt += v[j] <= v[i];
    if (cnt >= need && v[i] < best) best = v[i];
  }
  return best;
}

__device__ void template_from_box16(const uint8_t* frame, int width, int channels, const int* box, uint32_t* keys,
                                    uint32_t* bins, uint32_t* scan_tmp, double* red, TemplateOut out,
                                    bool* const_tile, int hp_rx, int hp_ry) {
  const int tid = threadIdx.x;
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  const int nb = bins16_count(channels);
  double sx = 0.0;
  for (int idx = tid; idx < n; idx += BLK) {
    int r = idx / w, c = idx - r * w;
    int key = pixel_key16(frame, width, channels, box[1] + r, box[0] + c);
    keys[idx] = (uint32_t)key;
    atomicAdd(&bins[key], 1u);
    sx += key_value(key, channels);
  }
  const double mean = block_sum(sx, red) / (double)n;
  double sq = 0.0;
  for (int idx = tid; idx < n; idx += BLK) {
    double d = key_value((int)keys[idx], channels) - mean;
    sq += d * d;
  }
  const double var = block_sum(sq, red) / (double)n;
  const double inv_std = 1.0 / sqrt(var);
  if (tid == 0) *const_tile = !(var > 0.0);
  __syncthreads();
  // np.unique + cumsum(counts) / size in increasing key order: every thread owns a run of bins
  const int chunk = (nb + BLK - 1) / BLK;
  const int b0 = min(tid * chunk, nb), b1 = min(b0 + chunk, nb);
  uint32_t ne = 0, cnt = 0;
  for (int b = b0; b < b1; ++b) {
    const uint32_t v = bins[b];
    ne += v != 0;
    cnt += v;
  }
  uint32_t total_ne, total_cnt;
  uint32_t k = block_excl_scan_u32(ne, scan_tmp, &total_ne);
  uint32_t cum = block_excl_scan_u32(cnt, scan_tmp, &total_cnt);
  for (int b = b0; b < b1; ++b) {
    const uint32_t v = bins[b];
    if (v) {
      cum += v;
      out.hist_v[k] = (key_value(b, channels) - mean) * inv_std;
      out.hist_q[k] = (double)cum / (double)n;
      ++k;
    }
  }
  if (tid == 0) *out.hist_n = (int)total_ne;
  for (int idx = tid; idx < n; idx += BLK) {
    int r = idx / w, c = idx - r * w;
    const int med = median_window32(keys, w, 0, w, h, r, c, hp_rx, hp_ry);
    double x = (key_value((int)keys[idx], channels) - mean) * inv_std;
    double xm = (key_value(med, channels) - mean) * inv_std;
    double t = x - xm;
    out.tile64[idx] = t;
    out.tile32[idx] = (float)t;
  }
}

// ------------------------------------------------------------------------------------------
// Float64 frames (one channel): Tracker.extract_tile works on any dtype (tracker.py:522-534).  The values of a
// tile are arbitrary doubles, so np.unique is not a histogram over keys: what the reference needs from it is, per
// pixel, the number of pixels at or below its (normalised) value -- cumsum(counts)[inverse] -- and, for a template,
// the sorted distinct values.  Both come from counting over the tile (every thread runs over all pixels; they all
// read the same element at a time: a broadcast), O(n^2 / BLK) per thread: a functional path for modest tiles, staged
// kernels only.  The median high-pass runs on the values themselves (the CDF match is monotone: the median of the
// matched window is the matched median).
// ------------------------------------------------------------------------------------------
// The gray value of a float pixel, in the frame's own arithmetic: one channel as it is; three channels their mean as
// tile.mean(axis=2) computes it (tracker.py:523-524) -- the sum ((a0 + a1) + a2) and the division by 3 in the dtype of
// the frame (a float32 frame gives a float32 mean).  bits: 32 (float32) or 64 (float64).
__device__ __forceinline__ double pixel_float(const uint8_t* frame, int width, int channels, int bits, int row, int col) {
  const size_t at = ((size_t)row * width + col) * channels;
  if (bits == 32) {
    const float* p = reinterpret_cast<const float*>(frame) + at;
    return channels == 1 ? (double)p[0] : (double)(((p[0] + p[1]) + p[2]) / 3.0f);
  }
  const double* p = reinterpret_cast<const double*>(frame) + at;
  return channels == 1 ? p[0] : ((p[0] + p[1]) + p[2]) / 3.0;
}
// np.add.reduce over n contiguous float32 items as NumPy's inner loop sums them (FLOAT_pairwise_sum: fewer than 8 items
// one after another; up to 128 with 8 interleaved accumulators; longer ranges split at n / 2 rounded down to a multiple
// of 8) -- what decides the last bit of a float32 mean / std (helpers.normalize on a float32 tile, helpers.py:344).
__device__ __forceinline__ float np_pairwise_leaf_f32(const float* a, int n) {  // n <= 128
  if (n < 8) {
    float res = 0.0f;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  }
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}
// The recursion pairwise(n) = pairwise(n2) + pairwise(n - n2), n2 = n / 2 rounded down to a multiple of 8, down to leaves of
// at most 128 items, on an explicit stack (depth <= log2(n / 128) + 1): `leaf(off, len)` is called for every leaf in order
// and its results are added as NumPy adds them.  ONE thread of the block runs this; the stack lives in LDS -- as private
// arrays with dynamic indices it was 400 bytes of scratch memory, which every kernel that can reach this code reserves.
// `k`: the ordinal of the next leaf, counted on (leaf(off, len, ordinal)).
struct NpWalk {
  float sum;
  int k;
};
struct NpStack {  // (a walk never exceeds 8192 items -- NumPy's buffer -- or a tile's row: depth <= 8)
  int off[16], len[16], state[16];
  float left[16];
};
__device__ __forceinline__ NpStack* np_stack() {  // (ONE stack for every instantiation of the walk)
  __shared__ NpStack s;
  return &s;
}
template <typename LEAF>
__device__ __forceinline__ NpWalk np_pairwise_walk(int off0, int n, int k, LEAF leaf) {
  NpStack* st = np_stack();
  int *off = st->off, *len = st->len, *state = st->state;
  float* left = st->left;
  int sp = 0;
  off[0] = off0; len[0] = n; state[0] = 0; left[0] = 0.0f;
  float ret = 0.0f;
  while (sp >= 0) {
    if (len[sp] <= 128) {
      ret = leaf(off[sp], len[sp], k);
      ++k;
      --sp;
      continue;
    }
    int n2 = len[sp] / 2;
    n2 -= n2 % 8;
    if (state[sp] == 0) {
      state[sp] = 1;
      off[sp + 1] = off[sp]; len[sp + 1] = n2; state[sp + 1] = 0;
      ++sp;
    } else if (state[sp] == 1) {
      left[sp] = ret;
      state[sp] = 2;
      off[sp + 1] = off[sp] + n2; len[sp + 1] = len[sp] - n2; state[sp + 1] = 0;
      ++sp;
    } else {
      ret = left[sp] + ret;
      --sp;
    }
  }
  return NpWalk{ret, k};
}
__device__ __forceinline__ float np_pairwise_f32(const float* a, int n) {
  return np_pairwise_walk(0, n, 0, [&](int o, int l, int) { return np_pairwise_leaf_f32(a + o, l); }).sum;
}
// ... over a whole contiguous array: the reduction hands the inner loop chunks of 8192 items (np.getbufsize()) and adds
// their sums to the running total in order (oracle/resample.py: numpy_pairwise_sum states the same for float64)
__device__ __forceinline__ float np_sum_flat_f32(const float* a, int n) {
  float acc = 0.0f;
  for (int s0 = 0; s0 < n; s0 += 8192) acc += np_pairwise_f32(a + s0, min(8192, n - s0));
  return acc;
}

// The same sums by the whole block (round 4; the serial form was a third of a float32 frame's tile stage): the LEAVES of
// the recursion are independent, and so are the 8 interleaved accumulators inside a leaf.  Thread 0 walks the recursion
// once to list the leaves (no data touched), 8 lanes per leaf then form its sum exactly as np_pairwise_leaf_f32 does --
// lane j the accumulator r[j], ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) by three exchange-adds (float addition
// commutes, so both lanes of a pair hold the same bits), the tail by lane 0 -- and thread 0 walks the recursion again,
// taking the leaf sums in order.  row_len > 0: the sum of a 2-D strided view as np.add.reduce forms it, out += pairwise(row)
// row by row; row_len == 0: a flat array in chunks of 8192.  `list`: `cap` words of LDS (a leaf = offset | length << 16:
// offsets below 65 536).  Returns false (nothing done) when the leaves may not fit -- the caller sums serially.
template <int NT>
__device__ __forceinline__ bool np_sum_f32_block(const float* a, int n, int row_len, uint32_t* list, int cap, float* result) {
  __shared__ int s_nleaf;
  const int tid = threadIdx.x, sub = tid & 7;
  const int unit = row_len > 0 ? row_len : 8192;
  const int units = (n + unit - 1) / unit;
  if (!list || n >= 65536 || (long long)units * (unit <= 128 ? 1 : (min(unit, n) + 63) / 64) > cap) return false;  // (uniform)
  if (tid == 0) {
    int k = 0;
    for (int o = 0; o < n; o += unit)
      k = np_pairwise_walk(o, min(unit, n - o), k, [list](int off, int len, int at) {
            list[at] = (uint32_t)off | ((uint32_t)len << 16);
            return 0.0f;
          }).k;
    s_nleaf = k;
  }
  __syncthreads();
  const int nleaf = s_nleaf;
  for (int k0 = 0; k0 < nleaf; k0 += NT / 8) {  // (whole groups of 8 lanes stay together: the exchanges below need them)
    const int k = k0 + (tid >> 3);
    const bool live = k < nleaf;
    const uint32_t e = live ? list[k] : 0u;
    const int off = (int)(e & 0xffffu), len = (int)(e >> 16);
    float res = 0.0f;
    if (len >= 8) {
      const int body = len - (len % 8);
      float r = a[off + sub];
      for (int i = 8; i < body; i += 8) r += a[off + i + sub];
      r += __shfl_xor(r, 1, WAVE);
      r += __shfl_xor(r, 2, WAVE);
      r += __shfl_xor(r, 4, WAVE);
      res = r;
      if (sub == 0)
        for (int i = body; i < len; ++i) res += a[off + i];
    } else if (sub == 0) {
      for (int i = 0; i < len; ++i) res += a[off + i];
    }
    if (live && sub == 0) list[k] = __float_as_uint(res);
  }
  __syncthreads();
  if (tid == 0) {
    int k = 0;
    float acc = 0.0f;
    for (int o = 0; o < n; o += unit) {
      const NpWalk part = np_pairwise_walk(o, min(unit, n - o), k, [list](int, int, int at) { return __uint_as_float(list[at]); });
      acc += part.sum;
      k = part.k;
    }
    *result = acc;
  }
  __syncthreads();
  return true;
}

// Round 4b: the same sums without a serial walk.  ONE call of NumPy's pairwise sum (n <= 8192 items) is a binary tree whose
// shape depends on n alone: a range longer than 128 splits at n2 = n / 2 rounded down to a multiple of 8 into (off, n2) and
// (off + n2, n - n2); from 8192 the right halves run 4096 .. 8191 -> 4103 -> 2055 -> 1031 -> 519 -> 263 -> 135 -> 71: no
// leaf lies deeper than level 7.  Numbered as a heap (root 1, children 2 i and 2 i + 1) the tree fits 256 slots, and thread
// `id` finds its node by following the bits of `id` down from the root -- registers only, no stack.  Leaves are summed by 8
// lanes each, exactly as np_pairwise_leaf_f32 does (lane j the accumulator r[j]; float addition commutes, so the three
// exchange-adds give every lane ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))); the inner nodes are added level by
// level by one wave (LDS operations of a wave complete in order: a level reads what the level below wrote).
// `load(i)`: item i of the array (the squares of the normalisation are formed here, not stored); `node`: 256 words,
// `val`: 256 floats of LDS.  All threads; barriers inside.  (The serial walk above: 35 k and 55 k cycles of a float32 frame's
// 217 k-cycle tile stage, tools/experiments/r04_jobs/j31_float_split.sh.)
constexpr uint32_t NP_NODE_INNER = 0xffffffffu;
template <typename LOAD>
__device__ __forceinline__ float np_leaf_8lanes(int off, int len, int sub, LOAD load) {
  float res = 0.0f;
  if (len >= 8) {
    const int body = len - (len % 8);
    float r = load(off + sub);
    for (int i = 8; i < body; i += 8) r += load(off + i + sub);
    r += __shfl_xor(r, 1, WAVE);
    r += __shfl_xor(r, 2, WAVE);
    r += __shfl_xor(r, 4, WAVE);
    res = r;
    if (sub == 0)
      for (int i = body; i < len; ++i) res += load(off + i);
  } else if (sub == 0) {
    for (int i = 0; i < len; ++i) res += load(off + i);
  }
  return res;  // (lane 0 of the group holds the leaf's sum)
}
template <int NT, typename LOAD>
__device__ __forceinline__ float np_pairwise_block(int off0, int n, uint32_t* node, float* val, LOAD load) {
  static_assert(NT >= 256, "one thread per tree slot");
  const int tid = threadIdx.x, sub = tid & 7;
  if (tid >= 1 && tid < 256) {
    int off = off0, len = n;
    bool exists = true;
    for (int b = 30 - __clz(tid); b >= 0; --b) {  // (the bits of tid below its leading one, from the top)
      if (len <= 128) {
        exists = false;  // an ancestor is a leaf
        break;
      }
      int n2 = len / 2;
      n2 -= n2 % 8;
      if ((tid >> b) & 1) {
        off += n2;
        len -= n2;
      } else {
        len = n2;
      }
    }
    node[tid] = !exists ? 0u : len <= 128 ? ((uint32_t)off | ((uint32_t)len << 16)) : NP_NODE_INNER;  // (len >= 1, off < 65536)
  }
  __syncthreads();
  for (int id0 = 0; id0 < 256; id0 += NT / 8) {
    const int id = id0 + (tid >> 3);
    const uint32_t e = id >= 1 && id < 256 ? node[id] : 0u;
    if (e != 0u && e != NP_NODE_INNER) {  // (the 8 lanes of a group agree)
      const float res = np_leaf_8lanes((int)(e & 0xffffu), (int)(e >> 16), sub, load);
      if (sub == 0) val[id] = res;
    }
  }
  __syncthreads();
  if (tid < WAVE) {
#pragma unroll
    for (int d = 6; d >= 0; --d) {
      asm volatile("" ::: "memory");  // (program order = LDS order inside a wave)
      const int id = (1 << d) + tid;
      if (tid < (1 << d) && node[id] == NP_NODE_INNER) val[id] = val[2 * id] + val[2 * id + 1];
    }
  }
  __syncthreads();
  return val[1];
}
// np.add.reduce of a flat float32 array: chunks of 8192 items (np.getbufsize()), their pairwise sums added in order
template <int NT, typename LOAD>
__device__ __forceinline__ float np_sum_flat_block(int n, uint32_t* node, float* val, LOAD load) {
  float acc = 0.0f;
  for (int s0 = 0; s0 < n; s0 += 8192) {
    acc += np_pairwise_block<NT>(s0, min(8192, n - s0), node, val, load);
    __syncthreads();  // (val[1] is read by every thread before the next chunk's tree is written)
  }
  return acc;
}
// ... of an h x w strided view, w <= 128: out += pairwise(row), row by row, and a row is one leaf.  The rows' sums by 8
// lanes each into val[h]; their ordered sum by wave 0, the values passed from lane to lane through v_readlane (64 dependent
// additions, no memory in the chain).  val: >= h floats of LDS; *out: LDS.  All threads; ends with a barrier.
template <int NT, typename LOAD>
__device__ __forceinline__ void np_sum_rows_block(int h, int w, float* val, float* out, LOAD load) {
  const int tid = threadIdx.x, sub = tid & 7;
  for (int r0 = 0; r0 < h; r0 += NT / 8) {
    const int r = r0 + (tid >> 3);
    if (r < h) {
      const float res = np_leaf_8lanes(r * w, w, sub, load);
      if (sub == 0) val[r] = res;
    }
  }
  __syncthreads();
  if (tid < WAVE) {
    float acc = 0.0f;
    for (int c0 = 0; c0 < h; c0 += WAVE) {
      const int v = c0 + tid < h ? __float_as_int(val[c0 + tid]) : 0;
#pragma unroll
      for (int k = 0; k < WAVE; ++k) {
        if (c0 + k >= h) break;  // (uniform)
        acc += __int_as_float(__builtin_amdgcn_readlane(v, k));
      }
    }
    if (tid == 0) *out = acc;
  }
  __syncthreads();
}

// helpers.normalize (helpers.py:344) of a float32 box by the whole block with the tile in LDS (the fused kernel's float
// branch): g[n] receives the gray values, then -- in place -- the normalised ones, (g - mean) * (1 / std) in float32, the two
// sums in NumPy's order as above.  `words`: 512 words of LDS (tree slots / row sums).  Same bits as normalize_box_float
// below (the staged kernels' form: values as doubles in memory), which it replaces for tiles that fit.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void normalize_box_f32_block(const uint8_t* frame, int width, int channels, const int* box, float* g,
                                                        uint32_t* words, unsigned long long* stp = nullptr) {
  __shared__ float s_rows;
  const int tid = threadIdx.x;
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  uint32_t* node = words;
  float* val = reinterpret_cast<float*>(words + 256);
  const UDiv by_w = udiv_make(w);
  for (int base = 0; base < n; base += 4 * NT) {  // (four pixel loads in flight per thread)
    float v[4];
    // (the channel count is decided outside the four fetches: behind a per-pixel branch every load was waited for in its
    // own block; the arithmetic is pixel_float's)
    const float* px[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = min(base + q * NT + tid, n - 1);
      const int r = udiv(by_w, idx), c = idx - r * w;
      px[q] = reinterpret_cast<const float*>(frame) + ((size_t)(box[1] + r) * width + (box[0] + c)) * channels;
    }
    if (channels == 1) {  // uniform
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = px[q][0];
    } else {
      float ch[4][3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ch[q][0] = px[q][0]; ch[q][1] = px[q][1]; ch[q][2] = px[q][2];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = ((ch[q][0] + ch[q][1]) + ch[q][2]) / 3.0f;
    }
    asm volatile("" ::: "memory");  // (the loads stay above the stores)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (base + q * NT + tid < n) g[base + q * NT + tid] = v[q];
  }
  __syncthreads();
  if (stp && tid == 0) stp[(size_t)blockIdx.x * GLH_NSTAMP + 20] = __builtin_amdgcn_s_memtime();
  auto plain = [g](int i) { return g[i]; };
  float total;
  if (channels == 1 && w <= 128 && h <= 256) {  // a strided view of the frame: row by row
    np_sum_rows_block<NT>(h, w, val, &s_rows, plain);
    total = s_rows;
  } else if (channels == 1) {  // (rows that split: one tree per row, in order)
    total = 0.0f;
    for (int r = 0; r < h; ++r) {
      total += np_pairwise_block<NT>(r * w, w, node, val, plain);
      __syncthreads();
    }
  } else {  // the channel mean is a new contiguous array
    total = np_sum_flat_block<NT>(n, node, val, plain);
  }
  const float mean = total / (float)n;
  if (stp && tid == 0) stp[(size_t)blockIdx.x * GLH_NSTAMP + 21] = __builtin_amdgcn_s_memtime();
  const float sq = np_sum_flat_block<NT>(n, node, val, [g, mean](int i) {
    const float d = g[i] - mean;
    return d * d;  // (-ffp-contract=off: the product is rounded before it is added)
  });
  const float var = sq / (float)n;
  const float inv = 1.0f / sqrtf(var);
  for (int idx = tid; idx < n; idx += NT) g[idx] = (g[idx] - mean) * inv;
  __syncthreads();
  if (stp && tid == 0) stp[(size_t)blockIdx.x * GLH_NSTAMP + 22] = __builtin_amdgcn_s_memtime();
}

// normalize (helpers.py:344) of the box into y[n]: (a - a.mean()) * (1 / a.std()); all threads; ends with a barrier.
// float64 frames: block reductions (the last bits of a float64 mean decide nothing).  float32 frames: float32 arithmetic
// with NumPy's own summation order, on thread 0 -- a one-channel tile is a strided view of the frame, which NumPy sums
// row by row (out += pairwise(row)); the channel mean of a three-channel tile is a new contiguous array, summed flat;
// (a - mean)^2 is contiguous either way.  `tmp`: 2 n floats of scratch that may overlap y (not each other).
template <int NT = BLK>
__device__ __forceinline__ void normalize_box_float(const uint8_t* frame, int width, int channels, int bits, const int* box,
                                                    double* y, float* tmp_g, float* tmp_x2, double* red, bool* const_tile,
                                                    uint32_t* list = nullptr, int list_cap = 0) {
  const int tid = threadIdx.x;
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  if (bits == 32) {
    __shared__ float s_mean, s_inv, s_sum;
    for (int idx = tid; idx < n; idx += NT) {
      const int r = idx / w, c = idx - r * w;
      tmp_g[idx] = (float)pixel_float(frame, width, channels, 32, box[1] + r, box[0] + c);
    }
    __syncthreads();
    // (the sums by the whole block when the leaves of NumPy's recursion fit `list`, by thread 0 otherwise: same bits)
    const bool par_g = np_sum_f32_block<NT>(tmp_g, n, channels == 1 ? w : 0, list, list_cap, &s_sum);
    if (tid == 0) {
      float acc = s_sum;
      if (!par_g) {
        acc = 0.0f;
        if (channels == 1)
          for (int r = 0; r < h; ++r) acc += np_pairwise_f32(tmp_g + (size_t)r * w, w);
        else
          acc = np_sum_flat_f32(tmp_g, n);
      }
      s_mean = acc / (float)n;
    }
    __syncthreads();
    const float mean = s_mean;
    for (int idx = tid; idx < n; idx += NT) {
      const float d = tmp_g[idx] - mean;
      tmp_x2[idx] = d * d;
    }
    __syncthreads();
    const bool par_x = np_sum_f32_block<NT>(tmp_x2, n, 0, list, list_cap, &s_sum);
    if (tid == 0) {
      const float var = (par_x ? s_sum : np_sum_flat_f32(tmp_x2, n)) / (float)n;
      s_inv = 1.0f / sqrtf(var);
      if (const_tile) *const_tile = !(var > 0.0f);
    }
    __syncthreads();
    const float inv = s_inv;
    // (y may overlap tmp_x2: every thread reads its g before anybody writes y -- g and y do not overlap)
    for (int idx = tid; idx < n; idx += NT) y[idx] = (double)((tmp_g[idx] - mean) * inv);
    __syncthreads();
    return;
  }
  double sx = 0.0;
  for (int idx = tid; idx < n; idx += NT) {
    const int r = idx / w, c = idx - r * w;
    const double x = pixel_float(frame, width, channels, 64, box[1] + r, box[0] + c);
    y[idx] = x;
    sx += x;
  }
  const double mean = block_sum<NT>(sx, red) / (double)n;
  double sq = 0.0;
  for (int idx = tid; idx < n; idx += NT) {
    const double d = y[idx] - mean;
    sq += d * d;
  }
  const double var = block_sum<NT>(sq, red) / (double)n;
  const double inv_std = 1.0 / sqrt(var);
  if (tid == 0 && const_tile) *const_tile = !(var > 0.0);
  for (int idx = tid; idx < n; idx += NT) y[idx] = (y[idx] - mean) * inv_std;
  __syncthreads();
}

// scipy.ndimage.median_filter(size = (2 ry + 1, 2 rx + 1), mode = 'reflect') of a w x h array of doubles at (r, c)
__device__ __forceinline__ double median_window_f64(const double* a, int w, int h, int r, int c, int rx, int ry_mode) {
  const int ry = GLH_HP_RY(ry_mode), mode = GLH_HP_MODE(ry_mode);
  double v[49];
  const int nx = 2 * rx + 1, n = nx * (2 * ry + 1);
  for (int dr = -ry; dr <= ry; ++dr) {
    const double* row = a + (size_t)border_index(r + dr, h, mode) * w;
    for (int dc = -rx; dc <= rx; ++dc) v[(dr + ry) * nx + (dc + rx)] = row[border_index(c + dc, w, mode)];
  }
  const int need = (n + 1) / 2;  // the smallest element with at least half the window at or below it
  double best = INFINITY;
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    for (int j = 0; j < n; ++j) cnt += v[j] <= v[i];
    if (cnt >= need && v[i] < best) best = v[i];
  }
  return best;
}
// template: y [n] and cnt [n] (uint32) in shared memory (n = tw * th)
__device__ void template_from_boxf(const uint8_t* frame, int width, int channels, int bits, const int* box, double* y,
                                   uint32_t* cnt, double* red, TemplateOut out, bool* const_tile, int hp_rx, int hp_ry) {
  const int tid = threadIdx.x;
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  // (float32 scratch: the gray values over cnt, the squares over the first half of y -- both dead before they are reused)
  normalize_box_float(frame, width, channels, bits, box, y, reinterpret_cast<float*>(cnt), reinterpret_cast<float*>(y), red,
                      const_tile);
  // pass 1: pixels at or below this one (the cumulative count of its value), and whether an equal one precedes it
  for (int idx = tid; idx < n; idx += BLK) {
    const double yi = y[idx];
    uint32_t leq = 0, eq_before = 0;
    for (int j = 0; j < n; ++j) {
      const double yj = y[j];
      leq += yj <= yi;
      eq_before += (j < idx) & (yj == yi);
    }
    cnt[idx] = (leq << 1) | (eq_before == 0 ? 1u : 0u);  // (n < 2^31)
  }
  __syncthreads();
  // pass 2: the first occurrence of every distinct value takes its place in the sorted unique arrays
  double firsts = 0.0;
  for (int idx = tid; idx < n; idx += BLK) {
    if (!(cnt[idx] & 1u)) continue;
    const double yi = y[idx];
    uint32_t rank = 0;
    for (int j = 0; j < n; ++j) rank += (cnt[j] & 1u) & (uint32_t)(y[j] < yi);
    out.hist_v[rank] = yi;
    out.hist_q[rank] = (double)(cnt[idx] >> 1) / (double)n;
    firsts += 1.0;
  }
  const double total = block_sum(firsts, red);
  if (tid == 0) *out.hist_n = (int)total;
  for (int idx = tid; idx < n; idx += BLK) {
    const int r = idx / w, c = idx - r * w;
    const double t = y[idx] - median_window_f64(y, w, h, r, c, hp_rx, hp_ry);
    out.tile64[idx] = bits == 32 ? (double)(float)t : t;  // (a float32 frame has a float32 template tile)
    out.tile32[idx] = (float)t;
  }
}

// np.unique's cumsum(counts)[inverse] for n values (any doubles): `emit(idx, count)` receives, for every value, the number
// of values at or below it.  Two-level ranking: NBK buckets LINEAR in the value over the values' own range (a
// floating-point map that is monotone non-decreasing: x <= y implies bucket(x) <= bucket(y), which is all the ranking
// needs; the bit pattern of a double would be logarithmic in it -- nearly all of a normalised tile in half a dozen
// buckets), a block scan for the bucket offsets, the values scattered into bucket order (`sorted`, n doubles), a value's
// count = its bucket's offset + the members of its bucket at or below it.  `tab`: NBK words of LDS, `scan_tmp`: NT / WAVE
// words.  All threads; ends with a barrier.
// (T: double -- values in memory, the staged kernels -- or float: a float32 tile in LDS, the fused kernel; the counts do
// not depend on the bucket map, only on its being monotone, which the float32 form of it is as well)
template <int NT, int NBK, typename T, typename EMIT>
__device__ __forceinline__ void rank_values(const T* work, T* sorted, int n, uint32_t* tab, uint32_t* scan_tmp, EMIT emit) {
  __shared__ T s_mm[NT / WAVE][2];
  const int tid = threadIdx.x;
  static_assert(NBK % NT == 0, "whole buckets per thread");
  for (int b = tid; b < NBK; b += NT) tab[b] = 0;
  T xmin = (T)INFINITY, xmax = (T)-INFINITY;
  for (int idx = tid; idx < n; idx += NT) {
    const T x = work[idx];
    xmin = fmin(xmin, x);
    xmax = fmax(xmax, x);
  }
  xmin = wave_min(xmin);
  xmax = wave_max(xmax);
  if ((tid & (WAVE - 1)) == 0) {
    s_mm[tid / WAVE][0] = xmin;
    s_mm[tid / WAVE][1] = xmax;
  }
  __syncthreads();
  xmin = s_mm[0][0];
  xmax = s_mm[0][1];
  for (int wv = 1; wv < NT / WAVE; ++wv) {
    xmin = fmin(xmin, s_mm[wv][0]);
    xmax = fmax(xmax, s_mm[wv][1]);
  }
  const T scale = xmax > xmin ? (T)(NBK - 1) / (xmax - xmin) : (T)0;
  auto bucket = [&](T x) -> int { return min(NBK - 1, max(0, (int)((x - xmin) * scale))); };
  for (int idx = tid; idx < n; idx += NT) atomicAdd(&tab[bucket(work[idx])], 1u);
  __syncthreads();
  {
    constexpr int PER = NBK / NT;
    uint32_t cnt[PER], local = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      cnt[k] = tab[PER * tid + k];
      local += cnt[k];
    }
    uint32_t total;
    uint32_t run = block_excl_scan_u32<NT>(local, scan_tmp, &total);  // (barriers inside)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      tab[PER * tid + k] = run;
      run += cnt[k];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < n; idx += NT) {
    const T x = work[idx];
    sorted[atomicAdd(&tab[bucket(x)], 1u)] = x;
  }
  __syncthreads();  // (tab[b] is now the END of bucket b; the scattered values are visible to the block)
  for (int idx = tid; idx < n; idx += NT) {
    const T x = work[idx];
    const int b = bucket(x);
    const uint32_t lo = b ? tab[b - 1] : 0u, hi = tab[b];
    uint32_t c = lo;
    for (uint32_t j = lo; j < hi; ++j) c += sorted[j] <= x;
    emit(idx, c);
  }
  __syncthreads();
}

// search tile from values: work [2 n] doubles of memory (the values -- any increasing function of the pixel --, behind them
// the matched ones); the template CDF
// `tab`: >= NBINS words of LDS, `scan_tmp`: NWAVES words.  The pixels at or below every pixel (np.unique's
// cumsum(counts)[inverse]) by the two-level ranking of the fused kernel's 16-bit path (glh_point.h: pt_tile_prep_wide) on
// the normalised values: buckets over the tile's own value range, a block scan for their offsets, the values scattered
// into bucket order (over the matched-value array, which is written afterwards), a pixel's count = its bucket's offset
// + the members of its bucket at or below it.  (Rounds 2-3a counted over the whole tile for
// every pixel: O(n^2 / BLK) per thread.)
// NT: threads of the block; NBK: buckets (>= NT, a multiple of it; `tab` holds NBK words); `out_ld`: row stride of the
// search tile (0: dense, w) -- the fused kernel's tiles carry padding columns, which are zeroed.
template <int NT = BLK, int NBK = NBINS>
__device__ __forceinline__ void search_tile_from_values(int w, int h, const double* hist_v, const double* hist_q, int hist_n, double* work,
                                        float* out, int hp_rx, int hp_ry, uint32_t* tab, uint32_t* scan_tmp,
                                        unsigned char* lds, int lds_bytes, int out_ld = 0) {
  const int tid = threadIdx.x;
  const int n = w * h;
  if (2 * hist_n * (int)sizeof(double) <= lds_bytes) {
    // the template CDF into LDS (np.interp searches it twice per pixel)
    double* cdf_lds = reinterpret_cast<double*>(lds);
    for (int k = tid; k < hist_n; k += NT) {
      cdf_lds[k] = hist_q[k];
      cdf_lds[hist_n + k] = hist_v[k];
    }
    hist_q = cdf_lds;
    hist_v = cdf_lds + hist_n;
    // (visible after the barriers of the ranking below)
  }
  double* matched = work + n;
  double* sorted = matched;                          // values in bucket order, until `matched` is made
  uint32_t* leq = reinterpret_cast<uint32_t*>(out);  // counts, until `out` is made
  rank_values<NT, NBK>(work, sorted, n, tab, scan_tmp, [&](int idx, uint32_t c) { leq[idx] = c; });
  // helpers.match_cdf: quantile of every pixel = (pixels at or below it) / size, through np.interp of the template CDF.
  // The counts move over the normalised values (dead): the high-pass takes its median on them -- the match is monotone in
  // the count, so the median of the matched window is the matched value of the median count (integers in registers for
  // the 5 x 5 window, where a window of doubles lived in scratch memory: 6.7 -> ms per frame at 1 024 points).
  uint32_t* rank = reinterpret_cast<uint32_t*>(work);
  for (int idx = tid; idx < n; idx += NT) {
    matched[idx] = np_interp((double)leq[idx] / (double)n, hist_q, hist_v, hist_n);
    rank[idx] = leq[idx];
  }
  __syncthreads();
  const int ld = out_ld ? out_ld : w;
  for (int idx = tid; idx < n; idx += NT) {
    const int r = idx / w, c = idx - r * w;
    // (NT != BLK: inside the fused kernel -- no private arrays there; the counts are in [1, n])
    const int med = NT == BLK ? median_window32(rank, w, 0, w, h, r, c, hp_rx, hp_ry)
                              : median_window32_bisect(rank, w, w, h, r, c, hp_rx, hp_ry, n);
    out[(size_t)r * ld + c] = (float)(matched[idx] - np_interp((double)med / (double)n, hist_q, hist_v, hist_n));
  }
  if (ld > w) {  // (padding columns are only read for outputs that are discarded; keep them finite)
    __syncthreads();  // (with a stride, `out` held the counts `leq` at other places than the values now written)
    for (int idx = tid; idx < h * (ld - w); idx += NT) {
      const int r = idx / (ld - w), c = w + idx - r * (ld - w);
      out[(size_t)r * ld + c] = 0.0f;
    }
  }
}

// float frames: the tile normalised in the frame's dtype (helpers.normalize), then the ranking above
template <int NT = BLK, int NBK = NBINS>
__device__ __forceinline__ void search_tile_from_boxf(const uint8_t* frame, int width, int channels, int bits, const int* box,
                                      const double* hist_v, const double* hist_q, int hist_n, double* work, double* red,
                                      float* out, int hp_rx, int hp_ry, uint32_t* tab, uint32_t* scan_tmp,
                                      unsigned char* lds, int lds_bytes, int out_ld = 0) {
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  // the float32 scratch of the normalisation (2 n floats; a float32 frame is summed in NumPy's order by ONE thread, whose
  // 4 n dependent loads should not be cache misses): LDS when it fits, else behind the values (the matched ones go there)
  float* scratch = 2 * n * (int)sizeof(float) <= lds_bytes ? reinterpret_cast<float*>(lds) : reinterpret_cast<float*>(work + n);
  normalize_box_float<NT>(frame, width, channels, bits, box, work, scratch, scratch + n, red, nullptr);
  search_tile_from_values<NT, NBK>(w, h, hist_v, hist_q, hist_n, work, out, hp_rx, hp_ry, tab, scan_tmp, lds, lds_bytes, out_ld);
}

// 16-bit frames: the keys themselves are the values (any increasing function of the key ranks the same; the template's
// normalisation is in its CDF) -- the ranking replaces the per-point histogram over all 65 536 / 196 606 keys that rounds
// 2-3a kept in memory (zeroed and scanned for every tile: 11.4 ms per frame at C3)
__device__ void search_tile_from_box16_ranked(const uint8_t* frame, int width, int channels, const int* box,
                                              const double* hist_v, const double* hist_q, int hist_n, double* work,
                                              float* out, int hp_rx, int hp_ry, uint32_t* tab, uint32_t* scan_tmp,
                                              unsigned char* lds, int lds_bytes) {
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  for (int idx = threadIdx.x; idx < n; idx += BLK) {
    const int r = idx / w, c = idx - r * w;
    work[idx] = (double)pixel_key16(frame, width, channels, box[1] + r, box[0] + c);
  }
  __syncthreads();
  search_tile_from_values(w, h, hist_v, hist_q, hist_n, work, out, hp_rx, hp_ry, tab, scan_tmp, lds, lds_bytes);
}

// ------------------------------------------------------------------------------------------
// K2a  Tracker.initialize_template (tracker.py:536-561) for one observer, one block per point
// ------------------------------------------------------------------------------------------
struct TemplateArgs {
  const double* mean6;  // [P][6] weighted particle mean (tracker.py:548)
  const uint8_t* active;
  const uint8_t* obs_mask;
  ObsFrame obs;
  int32_t o, O, P, tw, th, tile_cap, frame;  // tile_cap = max_tile^2 (per-template stride)
  int32_t hp_rx, hp_ry;                      // half sizes of the median high-pass window (2, 2 = the 5 x 5 default)
  int32_t* tmpl_box;    // [O][P][4]
  double* tmpl_duv;     // [O][P][2]
  double* tmpl_tile64;  // [O][P][tile_cap]
  float* tmpl_tile32;
  double* tmpl_hist_v;
  double* tmpl_hist_q;
  int32_t* tmpl_hist_n;  // [O][P]
  int32_t* tmpl_valid;   // [O][P]
  uint32_t* pt_status;
  int32_t* pt_err_frame;
};

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_template_init(TemplateArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ uint32_t hist[NBINS];
  __shared__ double red[NWAVES];
  __shared__ int s_box[4];
  __shared__ int s_ok;
  __shared__ bool s_const;
  uint16_t* keys = reinterpret_cast<uint16_t*>(smem);
  const int pt = blockIdx.x;
  if (a.active && !a.active[pt]) return;
  if (a.obs_mask && !a.obs_mask[(size_t)pt * a.O + a.o]) return;
  const size_t slot = (size_t)a.o * a.P + pt;
  if (threadIdx.x == 0) {
    const double* m = a.mean6 + (size_t)pt * 6;
    double u, v, duv[2];
    project(*a.obs.cam, m[0], m[1], m[2], u, v);
    int bad = template_box(u, v, a.tw, a.th, a.obs.cam->imgsz[0], a.obs.cam->imgsz[1], s_box, duv);
    if (!bad && (s_box[0] < 0 || s_box[1] < 0 || s_box[2] > a.obs.width || s_box[3] > a.obs.height))
      bad = 1;
    s_ok = !bad;
    if (bad) {
      a.tmpl_valid[slot] = 0;
      flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_TEMPLATE_OOB, a.frame);
    } else {
      for (int k = 0; k < 4; ++k) a.tmpl_box[slot * 4 + k] = s_box[k];
      a.tmpl_duv[slot * 2 + 0] = duv[0];
      a.tmpl_duv[slot * 2 + 1] = duv[1];
    }
  }
  __syncthreads();
  if (!s_ok) return;
  TemplateOut out;
  out.tile64 = a.tmpl_tile64 + slot * a.tile_cap;
  out.tile32 = a.tmpl_tile32 + slot * a.tile_cap;
  out.hist_v = a.tmpl_hist_v + slot * a.tile_cap;
  out.hist_q = a.tmpl_hist_q + slot * a.tile_cap;
  out.hist_n = a.tmpl_hist_n + slot;
  if (a.obs.bits >= 32)
    template_from_boxf(a.obs.frame, a.obs.width, a.obs.channels, a.obs.bits, s_box, reinterpret_cast<double*>(smem),
                       reinterpret_cast<uint32_t*>(smem + (size_t)a.tw * a.th * 8), red, out, &s_const, a.hp_rx, a.hp_ry);
  else if (a.obs.bits == 16)
    template_from_box16(a.obs.frame, a.obs.width, a.obs.channels, s_box, reinterpret_cast<uint32_t*>(smem),
                        a.obs.bins + (size_t)pt * bins16_count(a.obs.channels), hist, red, out, &s_const, a.hp_rx,
                        a.hp_ry);
  else
    template_from_box(a.obs.frame, a.obs.width, a.obs.channels, s_box, keys, hist, red, out, &s_const, a.hp_rx, a.hp_ry);
  __syncthreads();
  if (threadIdx.x == 0) {
    a.tmpl_valid[slot] = 1;
    if (s_const) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_CONST_TILE, a.frame);
  }
}
#endif

// Test hook: template from an explicit box (glh_stage_template).
struct TemplateBoxArgs {
  const uint8_t* frame;
  int32_t width, channels;
  int32_t box[4];
  int32_t hp_rx, hp_ry;
  TemplateOut out;
};
#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_template_from_box(TemplateBoxArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ uint32_t hist[NBINS];
  __shared__ double red[NWAVES];
  __shared__ bool s_const;
  __shared__ int s_box[4];
  if (threadIdx.x < 4) s_box[threadIdx.x] = a.box[threadIdx.x];
  __syncthreads();
  template_from_box(a.frame, a.width, a.channels, s_box, reinterpret_cast<uint16_t*>(smem), hist,
                    red, a.out, &s_const, a.hp_rx, a.hp_ry);
}
#endif

// ------------------------------------------------------------------------------------------
// Search tile from an integer box: extract_tile(histogram=template CDF) (tracker.py:605-607):
// match_cdf (helpers.py:489-493; normalize before it is a no-op) via a per-key LUT
//   LUT[key] = np.interp(cumcount[key]/size, template_q, template_v)
// then tile - median5x5(tile) == LUT[key] - LUT[median5x5(key)] (LUT is monotone, window odd).
// Output is the float32 cast of tracker.py:610.  Median runs on LDS row bands of BAND_H rows.
// ------------------------------------------------------------------------------------------
__device__ void search_tile_from_box(const uint8_t* frame, int width, int channels, const int* box,
                                     const double* hist_v, const double* hist_q, int hist_n,
                                     uint32_t* hist, uint32_t* cum, double* lut, uint16_t* band,
                                     uint32_t* scan_tmp, float* out, int hp_rx = 2, int hp_ry = 2) {
  const int tid = threadIdx.x;
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  for (int b = tid; b < NBINS; b += BLK) hist[b] = 0;
  __syncthreads();
  for (int idx = tid; idx < n; idx += BLK) {
    int r = idx / w, c = idx - r * w;
    atomicAdd(&hist[pixel_key(frame, width, channels, box[1] + r, box[0] + c)], 1u);
  }
  __syncthreads();
  // inclusive scan of the 768 bins: 3 bins per thread + block scan of the per-thread sums
  {
    uint32_t h0 = hist[3 * tid], h1 = hist[3 * tid + 1], h2 = hist[3 * tid + 2];
    uint32_t local = h0 + h1 + h2;
    uint32_t incl = local;
    const int lane = tid & (WAVE - 1);
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      uint32_t t = __shfl_up(incl, off, WAVE);
      if (lane >= off) incl += t;
    }
    if (lane == WAVE - 1) scan_tmp[tid / WAVE] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int wv = 0; wv < tid / WAVE; ++wv) base += scan_tmp[wv];
    uint32_t excl = base + incl - local;
    cum[3 * tid] = excl + h0;
    cum[3 * tid + 1] = excl + h0 + h1;
    cum[3 * tid + 2] = excl + local;
  }
  __syncthreads();
  for (int b = tid; b < NBINS; b += BLK) {
    if (hist[b]) {
      double q = (double)cum[b] / (double)n;  // np.cumsum(counts) / a.size
      lut[b] = np_interp(q, hist_q, hist_v, hist_n);
    }
  }
  __syncthreads();
  if (hp_rx != 2 || hp_ry != 2) {
    // any other odd window up to 7 x 7 (or another boundary mode: it rides in hp_ry): bands with a halo of ry rows,
    // median by bisection
    const int hp_mode = GLH_HP_MODE(hp_ry);
    hp_ry = GLH_HP_RY(hp_ry);
    const int max_key = channels == 1 ? 255 : 255 * channels;
    for (int r0 = 0; r0 < h; r0 += BAND_H) {
      const int rows = min(BAND_H, h - r0);
      for (int idx = tid; idx < (rows + 2 * hp_ry) * w; idx += BLK) {
        int br = idx / w, c = idx - br * w;
        int rr = border_index(r0 + br - hp_ry, h, hp_mode);
        band[idx] = (uint16_t)pixel_key(frame, width, channels, box[1] + rr, box[0] + c);
      }
      __syncthreads();
      for (int idx = tid; idx < rows * w; idx += BLK) {
        int br = idx / w, c = idx - br * w;
        // (the band holds reflected rows r0 - ry .. r0 + rows + ry - 1 in order: row index = br + dr + ry)
        const int need = ((2 * hp_rx + 1) * (2 * hp_ry + 1) + 1) / 2;
        int lo = 0, hi = max_key;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          int cnt = 0;
          for (int dr = 0; dr <= 2 * hp_ry; ++dr)
            for (int dc = -hp_rx; dc <= hp_rx; ++dc) cnt += band[(br + dr) * w + border_index(c + dc, w, hp_mode)] <= mid;
          if (cnt >= need) hi = mid; else lo = mid + 1;
        }
        const int key = band[(br + hp_ry) * w + c];
        out[(size_t)(r0 + br) * w + c] = (float)(lut[key] - lut[lo]);
      }
      __syncthreads();
    }
    return;
  }
  for (int r0 = 0; r0 < h; r0 += BAND_H) {
    const int rows = min(BAND_H, h - r0);
    // stage rows r0-2 .. r0+rows+1 (reflected at the tile's top/bottom) as keys
    for (int idx = tid; idx < (rows + 4) * w; idx += BLK) {
      int br = idx / w, c = idx - br * w;
      int rr = reflect_index(r0 + br - 2, h);
      band[idx] = (uint16_t)pixel_key(frame, width, channels, box[1] + rr, box[0] + c);
    }
    __syncthreads();
    for (int idx = tid; idx < rows * w; idx += BLK) {
      int br = idx / w, c = idx - br * w;
      int v[25];
#pragma unroll
      for (int dr = 0; dr < 5; ++dr) {
#pragma unroll
        for (int dc = -2; dc <= 2; ++dc) {
          int cc = reflect_index(c + dc, w);
          v[dr * 5 + (dc + 2)] = band[(br + dr) * w + cc];
        }
      }
      int key = v[12];
      int med = median25(v);
      out[(size_t)(r0 + br) * w + c] = (float)(lut[key] - lut[med]);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// K2b  search box (tracker.py:580-603) + search tile (tracker.py:605-607), one block/point
// ------------------------------------------------------------------------------------------
struct TilePrepArgs {
  const uint8_t* active;
  const uint8_t* obs_mask;
  ObsFrame obs;
  int32_t o, O, P, NB, tw, th, tile_cap, search_cap, max_dim;
  int32_t hp_rx, hp_ry;     // half sizes of the median high-pass window
  int32_t kcols, krows;     // interpolation orders that set the least surface size (search_box)
  int32_t lds_bytes;        // dynamic LDS of the launch
  const double* bbox_part;  // [O][P][NB][5]
  const int32_t* tmpl_valid;
  const double* tmpl_hist_v;
  const double* tmpl_hist_q;
  const int32_t* tmpl_hist_n;
  int32_t* box;         // [O][P][4]
  int32_t* obs_status;  // [O][P]
  float* search;        // [O][P][search_cap]
};

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_tileprep(TilePrepArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ uint32_t hist[NBINS];
  __shared__ uint32_t cum[NBINS];
  __shared__ double lut[NBINS];
  __shared__ uint32_t scan_tmp[NWAVES];
  __shared__ double red[NWAVES][5];
  __shared__ int s_box[4];
  __shared__ int s_status;
  const int pt = blockIdx.x, tid = threadIdx.x;
  const size_t slot = (size_t)a.o * a.P + pt;
  if ((a.active && !a.active[pt]) || (a.obs_mask && !a.obs_mask[(size_t)pt * a.O + a.o])) {
    if (tid == 0) a.obs_status[slot] = GLH_OBS_SKIPPED;
    return;
  }
  if (!a.tmpl_valid[slot]) {
    if (tid == 0) a.obs_status[slot] = GLH_OBS_NO_TEMPLATE;
    return;
  }
  double mnu = INFINITY, mnv = INFINITY, mxu = -INFINITY, mxv = -INFINITY, nanf = 0.0;
  for (int b = tid; b < a.NB; b += BLK) {
    const double* p = a.bbox_part + (slot * a.NB + b) * 5;
    mnu = fmin(mnu, p[0]); mnv = fmin(mnv, p[1]);
    mxu = fmax(mxu, p[2]); mxv = fmax(mxv, p[3]);
    nanf = fmax(nanf, p[4]);
  }
  mnu = wave_min(mnu); mnv = wave_min(mnv); mxu = wave_max(mxu); mxv = wave_max(mxv);
  nanf = wave_max(nanf);
  if ((tid & (WAVE - 1)) == 0) {
    double* r = red[tid / WAVE];
    r[0] = mnu; r[1] = mnv; r[2] = mxu; r[3] = mxv; r[4] = nanf;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < NWAVES; ++w) {
      mnu = fmin(mnu, red[w][0]); mnv = fmin(mnv, red[w][1]);
      mxu = fmax(mxu, red[w][2]); mxv = fmax(mxv, red[w][3]);
      nanf = fmax(nanf, red[w][4]);
    }
    int st = GLH_OBS_OK;
    if (search_box(mnu, mnv, mxu, mxv, nanf != 0.0, a.tw, a.th, a.obs.cam->imgsz[0],
                   a.obs.cam->imgsz[1], s_box, a.kcols, a.krows))
      st = GLH_OBS_OUT_OF_BOUNDS;
    else if (s_box[2] > a.obs.width || s_box[3] > a.obs.height)
      st = GLH_OBS_OUT_OF_BOUNDS;
    else {
      int w = s_box[2] - s_box[0], h = s_box[3] - s_box[1];
      if (w > a.max_dim || h > a.max_dim || (long long)w * h > a.search_cap)
        st = GLH_OBS_TILE_TOO_LARGE;
    }
    s_status = st;
    a.obs_status[slot] = st;
    if (st == GLH_OBS_OK)
      for (int k = 0; k < 4; ++k) a.box[slot * 4 + k] = s_box[k];
  }
  __syncthreads();
  if (s_status != GLH_OBS_OK) return;
  if (a.obs.bits >= 32)
    search_tile_from_boxf(a.obs.frame, a.obs.width, a.obs.channels, a.obs.bits, s_box, a.tmpl_hist_v + slot * a.tile_cap,
                          a.tmpl_hist_q + slot * a.tile_cap, a.tmpl_hist_n[slot], a.obs.fwork + (size_t)pt * a.obs.fwork_cap,
                          &red[0][0], a.search + slot * (size_t)a.search_cap, a.hp_rx, a.hp_ry, hist, scan_tmp, smem,
                          a.lds_bytes);
  else if (a.obs.bits == 16)
    search_tile_from_box16_ranked(a.obs.frame, a.obs.width, a.obs.channels, s_box, a.tmpl_hist_v + slot * a.tile_cap,
                                  a.tmpl_hist_q + slot * a.tile_cap, a.tmpl_hist_n[slot],
                                  a.obs.fwork + (size_t)pt * a.obs.fwork_cap, a.search + slot * (size_t)a.search_cap, a.hp_rx,
                                  a.hp_ry, hist, scan_tmp, smem, a.lds_bytes);
  else
    search_tile_from_box(a.obs.frame, a.obs.width, a.obs.channels, s_box,
                         a.tmpl_hist_v + slot * a.tile_cap, a.tmpl_hist_q + slot * a.tile_cap,
                         a.tmpl_hist_n[slot], hist, cum, lut, reinterpret_cast<uint16_t*>(smem),
                         scan_tmp, a.search + slot * (size_t)a.search_cap, a.hp_rx, a.hp_ry);
}
#endif

// Test hook: search tile from an explicit box (glh_stage_search_tile).
struct SearchBoxArgs {
  const uint8_t* frame;
  int32_t width, channels;
  int32_t box[4];
  const double* hist_v;
  const double* hist_q;
  int32_t hist_n;
  int32_t hp_rx, hp_ry;
  float* out;
};
#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_search_from_box(SearchBoxArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ uint32_t hist[NBINS];
  __shared__ uint32_t cum[NBINS];
  __shared__ double lut[NBINS];
  __shared__ uint32_t scan_tmp[NWAVES];
  __shared__ int s_box[4];
  if (threadIdx.x < 4) s_box[threadIdx.x] = a.box[threadIdx.x];
  __syncthreads();
  search_tile_from_box(a.frame, a.width, a.channels, s_box, a.hist_v, a.hist_q, a.hist_n, hist, cum,
                       lut, reinterpret_cast<uint16_t*>(smem), scan_tmp, a.out, a.hp_rx, a.hp_ry);
}
#endif

// ------------------------------------------------------------------------------------------
// K3  area-averaged SSD surface (tracker.py:609-614):
//       sse[r][c] = float32( sum_ij (S[r+i][c+j] - T[i][j])^2 ) * 1/(tw*th)
//     float32 sub + FMA inside a template row, float64 across rows, one rounding to float32,
//     then the reference's  float32(float64(sse) * float64(1/(tw*th))).  Stored widened to
//     float64 because the spline fit that follows works in float64 (observer.py:210).
//
//     Work decomposition: a block owns output tiles of SSD_TOH x SSD_TOW cells and stages the
//     matching (TOH+th-1) x (TOW+tw-1) search window in LDS (fixed footprint, independent of
//     the search-tile size).  A thread owns a 4-wide strip of one output row; when a tile has
//     fewer strips than threads (small surfaces are the common case: clouds of ~2 px sigma
//     give 13x13 .. 20x20 surfaces) the template rows are split G ways across adjacent lanes
//     and the float64 partials are combined with wave shuffles.
// ------------------------------------------------------------------------------------------
constexpr int SSD_W = 4;     // outputs per thread
constexpr int SSD_TOW = 32;  // output tile width
constexpr int SSD_TOH = 16;  // output tile height

struct SsdArgs {
  int32_t o, P, tw, th, tile_cap, search_cap, sse_cap, reserved;
  const int32_t* box;         // [O][P][4]
  const int32_t* obs_status;  // [O][P]
  const float* search;        // [O][P][search_cap]
  const float* tmpl;          // [O][P][tile_cap]
  double* sse;                // [O][P][sse_cap]
};

__host__ __device__ __forceinline__ int ssd_twp(int tw) { return (tw + 7) & ~7; }
// LDS row stride of the search window: >= TOW + twp + 4 and == 8 (mod 64) floats so that the
// G row-split lanes of neighbouring strips land on distinct banks
__host__ __device__ __forceinline__ int ssd_ld(int tw) {
  int need = SSD_TOW + ssd_twp(tw) + 4;
  int ld = ((need - 8 + 63) / 64) * 64 + 8;
  return ld;
}
__host__ __device__ __forceinline__ size_t ssd_lds_bytes(int tw, int th) {
  return (size_t)(th * ssd_twp(tw) + (SSD_TOH + th - 1) * ssd_ld(tw)) * sizeof(float);
}

// One strip of SSD_W outputs: template rows i = g, g + G, ... of the (s - t)^2 sum; float32
// FMA along a template row, float64 across rows.  S row stride ld (floats, multiple of 4,
// readable up to column cc + twp + 3), T [th][twp] zero padded.
//
// Round 4: written on explicit float pairs.  The arithmetic is the one of rounds 1-3 -- per output and tap
// d = s - t, acc = fma(d, d, acc), taps in order j = 0 .. tw - 1 (oracle/ssd.c: oracle_ssd_f32_rows) -- but the
// instructions around it are chosen here instead of left to the vectoriser: (1) the template value of an odd tap is
// the HIGH half of its register pair and is selected by the packed subtract itself (op_sel), where the compiler built
// a (t, t) pair with a move per odd tap; (2) the partial last block of a row (tw = 31: its fourth) ends at a uniform
// branch after its last tap, where the compiler computed all eight taps and kept or dropped each with four selects;
// (3) two blocks per loop iteration, so the sliding window changes registers instead of being moved.  Per 8 taps x 4
// outputs: 32 packed operations + 16 others before, + 9 now; the partial block 75 -> ~40.
typedef float glh_f2 __attribute__((ext_vector_type(2)));
// (w.x - t.x, w.y - t.x) and (w.x - t.y, w.y - t.y): one v_pk_add_f32 each, IEEE subtraction
__device__ __forceinline__ glh_f2 ssd_sub_lo(glh_f2 w, glh_f2 t) {
#if defined(__HIP_DEVICE_COMPILE__)
  glh_f2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(w), "v"(t));
  return d;
#else
  return glh_f2{w.x - t.x, w.y - t.x};  // (the host pass only parses this)
#endif
}
__device__ __forceinline__ glh_f2 ssd_sub_hi(glh_f2 w, glh_f2 t) {
#if defined(__HIP_DEVICE_COMPILE__)
  glh_f2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(w), "v"(t));
  return d;
#else
  return glh_f2{w.x - t.y, w.y - t.y};
#endif
}

// Taps 0 .. NV-1 (NV = 8: a whole block; otherwise `nv` of them, uniform) of one 8-tap block on the four outputs.
// W[0..5]: the window s[0..11] as aligned pairs; odd[0]: (s[1], s[2]) (carried over from the previous block);
// T[0..3]: the block's template values.  Leaves odd[0] = (s[9], s[10]) for the next block.
template <bool WHOLE>
__device__ __forceinline__ void ssd_block8(const glh_f2* W, const glh_f2* T, glh_f2& odd0, glh_f2& a01, glh_f2& a23, int nv) {
  glh_f2 od[5];
  od[0] = odd0;
#pragma unroll
  for (int m = 1; m < 5; ++m) od[m] = glh_f2{W[m].y, W[m + 1].x};  // (s[2m + 1], s[2m + 2])
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (!WHOLE && j >= nv) break;  // uniform: a scalar branch (the inline assembly below is not speculated)
    const glh_f2 w01 = (j & 1) ? od[j >> 1] : W[j >> 1];
    const glh_f2 w23 = (j & 1) ? od[(j >> 1) + 1] : W[(j >> 1) + 1];
    const glh_f2 d01 = (j & 1) ? ssd_sub_hi(w01, T[j >> 1]) : ssd_sub_lo(w01, T[j >> 1]);
    const glh_f2 d23 = (j & 1) ? ssd_sub_hi(w23, T[j >> 1]) : ssd_sub_lo(w23, T[j >> 1]);
    a01 = __builtin_elementwise_fma(d01, d01, a01);
    a23 = __builtin_elementwise_fma(d23, d23, a23);
  }
  odd0 = od[4];
}

__device__ __forceinline__ void ssd_strip_rows(const float* S, int ld, const float* T, int tw, int th, int twp,
                                               int rr, int cc, int g, int G, double* acc64) {
  static_assert(SSD_W == 4, "the strip code is written for four outputs");
  for (int i = g; i < th; i += G) {
    const float* rowp = S + (rr + i) * ld + cc;
    const float* trow = T + i * twp;
    glh_f2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
    glh_f2 p0, p1;  // s[0..3] of the next block
    {
      const float4 a0 = *reinterpret_cast<const float4*>(rowp);
      p0 = glh_f2{a0.x, a0.y};
      p1 = glh_f2{a0.z, a0.w};
    }
    glh_f2 odd0 = glh_f2{p0.y, p1.x};
    // one 8-tap block at taps jj ..: `nv` < 8 of them when !WHOLE
    auto block = [&](int jj, auto whole_tag, int nv) {
      constexpr bool WHOLE = decltype(whole_tag)::value;
      glh_f2 W[6], Tv[4];
      const float4 b0 = *reinterpret_cast<const float4*>(rowp + jj + 4);
      const float4 b1 = *reinterpret_cast<const float4*>(rowp + jj + 8);
      const float4 t0 = *reinterpret_cast<const float4*>(trow + jj);
      const float4 t1 = *reinterpret_cast<const float4*>(trow + jj + 4);
      W[0] = p0; W[1] = p1;
      W[2] = glh_f2{b0.x, b0.y}; W[3] = glh_f2{b0.z, b0.w}; W[4] = glh_f2{b1.x, b1.y}; W[5] = glh_f2{b1.z, b1.w};
      Tv[0] = glh_f2{t0.x, t0.y}; Tv[1] = glh_f2{t0.z, t0.w}; Tv[2] = glh_f2{t1.x, t1.y}; Tv[3] = glh_f2{t1.z, t1.w};
      ssd_block8<WHOLE>(W, Tv, odd0, a01, a23, nv);
      p0 = W[4];
      p1 = W[5];
    };
    int jj = 0;
    for (; jj + 16 <= tw; jj += 16) {  // two blocks per iteration: the window changes registers, not places
      block(jj, std::true_type{}, 8);
      // (the second block's loads stay behind the first block's arithmetic: hoisted above it they cost 16 more live
      // registers, which the 1 024-thread and the two-observer instantiations do not have)
      asm volatile("" ::: "memory");
      block(jj + 8, std::true_type{}, 8);
    }
    if (jj + 8 <= tw) {  // uniform
      block(jj, std::true_type{}, 8);
      jj += 8;
    }
    if (jj < tw) block(jj, std::false_type{}, tw - jj);  // uniform
    acc64[0] += (double)a01.x;
    acc64[1] += (double)a01.y;
    acc64[2] += (double)a23.x;
    acc64[3] += (double)a23.y;
  }
}

// Template-row split of a whole wo x ho surface: the largest power of two G <= 8 such that
// (strips of the surface) * G <= 512.  Shared by the staged and the fused kernels so that
// their float64 row partials combine in the same order.
__host__ __device__ __forceinline__ int ssd_row_split(int wo, int ho) {
  const int nstrips = ((wo + SSD_W - 1) / SSD_W) * ho;
  int G = 1;
  while (G < 8 && nstrips * (G * 2) <= 512) G *= 2;
  return G;
}

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_ssd(SsdArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int pt = blockIdx.y, tid = threadIdx.x;
  const size_t slot = (size_t)a.o * a.P + pt;
  if (a.obs_status[slot] != GLH_OBS_OK) return;
  const int* box = a.box + slot * 4;
  const int ws = box[2] - box[0], hs = box[3] - box[1];
  const int tw = a.tw, th = a.th;
  const int wo = ws - tw + 1, ho = hs - th + 1;
  const int tiles_x = (wo + SSD_TOW - 1) / SSD_TOW, tiles_y = (ho + SSD_TOH - 1) / SSD_TOH;
  const int ntiles = tiles_x * tiles_y;
  if ((int)blockIdx.x >= ntiles) return;
  const int twp = ssd_twp(tw);
  const int ld = ssd_ld(tw);
  float* T = reinterpret_cast<float*>(smem);  // [th][twp]
  float* S = T + th * twp;                    // [TOH + th - 1][ld]
  const float* tg = a.tmpl + slot * a.tile_cap;
  for (int idx = tid; idx < th * twp; idx += BLK) {
    int i = idx / twp, j = idx - i * twp;
    T[idx] = j < tw ? tg[i * tw + j] : 0.0f;
  }
  const float* sg = a.search + slot * (size_t)a.search_cap;
  double* outg = a.sse + slot * (size_t)a.sse_cap;
  const double inv_area = 1.0 / (double)(tw * th);
  const int G = ssd_row_split(wo, ho);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * SSD_TOH, x0 = tx * SSD_TOW;
    const int oh = min(SSD_TOH, ho - y0), ow = min(SSD_TOW, wo - x0);
    const int rows = oh + th - 1, cols = min(ld, ws - x0);
    __syncthreads();
    for (int idx = tid; idx < rows * ld; idx += BLK) {
      int rr = idx / ld, c = idx - rr * ld;
      S[idx] = c < cols ? sg[(size_t)(y0 + rr) * ws + x0 + c] : 0.0f;
    }
    __syncthreads();
    const int spr = (ow + SSD_W - 1) / SSD_W;
    const int nstrips = spr * oh;  // <= 8 * 16 = 128
    for (int s0 = 0; s0 < nstrips; s0 += BLK / G) {
      const int strip = s0 + tid / G, g = tid % G;
      const bool live = strip < nstrips;
      const int rr = live ? strip / spr : 0;
      const int cc = live ? (strip - rr * spr) * SSD_W : 0;
      double acc64[SSD_W];
#pragma unroll
      for (int k = 0; k < SSD_W; ++k) acc64[k] = 0.0;
      if (live) ssd_strip_rows(S, ld, T, tw, th, twp, rr, cc, g, G, acc64);
      // combine the G row-split partials (adjacent lanes of one wave; G divides 64)
      for (int off = 1; off < G; off <<= 1) {
#pragma unroll
        for (int k = 0; k < SSD_W; ++k) acc64[k] += __shfl_xor(acc64[k], off, WAVE);
      }
      if (live && g == 0) {
        const int r = y0 + rr;
#pragma unroll
        for (int k = 0; k < SSD_W; ++k) {
          if (cc + k < ow) {
            float raw = (float)acc64[k];
            float val = (float)((double)raw * inv_area);  // sse *= 1/(tw*th) (tracker.py:614)
            outg[(size_t)r * wo + x0 + cc + k] = (double)val;
          }
        }
      }
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------
// K4  spline coefficients: two passes of banded (|i-j| <= 2) not-a-knot collocation solves
//     with host-precomputed LU factors (they depend only on n).  In place on the SSE surface.
// ------------------------------------------------------------------------------------------
struct SplineFitArgs {
  int32_t o, P, tw, th, sse_cap, max_n;
  int32_t linear, reserved;  // linear: interpolation order 1 -- the surface values are the coefficients, nothing to fit
  int32_t kx, ky;            // orders along the rows / columns axis (3, 3 unless Tracker(interpolation=...) says otherwise)
  const double* glu_v;       // general orders: factors of spline_lu_general for degree kx, by size at glu_v_off[n]
  const int64_t* glu_v_off;
  const double* glu_u;       // ... and for degree ky
  const int64_t* glu_u_off;
  const int32_t* box;
  const int32_t* obs_status;
  const double* lu;          // packed factors: for n, 5 arrays of n at lu_off[n]
  const int64_t* lu_off;     // [max_n + 1]
  const double* inv;         // explicit inverses for sides 4 .. GLH_SPL_DENSE_MAX at spline_inverse_off(n)
  double* sse;               // in: SSE surface, out: B-spline coefficients
  double* sse_copy;          // optional debug copy of the surface (or null)
};

// The fit of a small surface as two dense products with the explicit inverses, C = Ih . Z . Iw^T: one
// independent dot product per coefficient (k ascending, separate multiply and add) instead of two serial chains
// per line.  Z [ho][wo] in place, Z1 [ho * wo] scratch; whole block, NT threads.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void spline_fit_dense(double* Z, double* Z1, int wo, int ho, const double* Ih,
                                                 const double* Iw) {
  const int tid = threadIdx.x;
  const UDiv by_wo = udiv_make(wo);
  for (int idx = tid; idx < ho * wo; idx += NT) {
    const int r = udiv(by_wo, idx), c = idx - r * wo;
    const double* row = Ih + (size_t)r * ho;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < ho; ++k) acc += row[k] * Z[(size_t)k * wo + c];
    Z1[idx] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < ho * wo; idx += NT) {
    const int r = udiv(by_wo, idx), c = idx - r * wo;
    const double* row = Iw + (size_t)c * wo;
    const double* zr = Z1 + (size_t)r * wo;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < wo; ++k) acc += zr[k] * row[k];
    Z[idx] = acc;
  }
  __syncthreads();
}

__device__ __forceinline__ void solve_line(double* x, int stride, int n, const double* f) {
  const double *l1 = f, *l2 = f + n, *u0i = f + 2 * n, *u1 = f + 3 * n, *u2 = f + 4 * n;
  double ym1 = x[0], ym2 = 0.0;
  for (int i = 1; i < n; ++i) {
    double y = x[(size_t)i * stride] - l1[i] * ym1;
    if (i >= 2) y -= l2[i] * ym2;
    x[(size_t)i * stride] = y;
    ym2 = ym1;
    ym1 = y;
  }
  double xp1 = 0.0, xp2 = 0.0;
  for (int i = n - 1; i >= 0; --i) {
    double acc = x[(size_t)i * stride];
    if (i + 1 < n) acc -= u1[i] * xp1;
    if (i + 2 < n) acc -= u2[i] * xp2;
    acc *= u0i[i];
    x[(size_t)i * stride] = acc;
    xp2 = xp1;
    xp1 = acc;
  }
}

// one line of the degree-k fit: forward / backward substitution with the factors of spline_lu_general (glh_host.h)
__device__ __forceinline__ void solve_line_general(double* x, int stride, int n, int k, const double* f) {
  const double *L = f, *u0inv = f + (size_t)k * n, *U = u0inv + n;
  for (int i = 1; i < n; ++i) {
    double y = x[(size_t)i * stride];
    for (int d = 1; d <= k && d <= i; ++d) y = y - L[(size_t)(d - 1) * n + i] * x[(size_t)(i - d) * stride];
    x[(size_t)i * stride] = y;
  }
  for (int i = n - 1; i >= 0; --i) {
    double acc = x[(size_t)i * stride];
    for (int d = 1; d <= k; ++d)
      if (i + d < n) acc = acc - U[(size_t)(d - 1) * n + i] * x[(size_t)(i + d) * stride];
    x[(size_t)i * stride] = acc * u0inv[i];
  }
}

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_spline_fit(SplineFitArgs a) {
  const int pt = blockIdx.x, tid = threadIdx.x;
  const size_t slot = (size_t)a.o * a.P + pt;
  if (a.obs_status[slot] != GLH_OBS_OK) return;
  const int* box = a.box + slot * 4;
  const int wo = box[2] - box[0] - a.tw + 1, ho = box[3] - box[1] - a.th + 1;
  double* z = a.sse + slot * (size_t)a.sse_cap;
  if (a.sse_copy) {
    double* cp = a.sse_copy + slot * (size_t)a.sse_cap;
    for (int idx = tid; idx < wo * ho; idx += BLK) cp[idx] = z[idx];
    __syncthreads();
  }
  if (a.linear) return;
  if (a.glu_v) {  // any other orders than (3, 3) and (1, 1): banded solves of bandwidth k, columns then rows
    const double* fv = a.glu_v + a.glu_v_off[ho];
    const double* fu = a.glu_u + a.glu_u_off[wo];
    for (int c = tid; c < wo; c += BLK) solve_line_general(z + c, wo, ho, a.kx, fv);
    __syncthreads();
    for (int r = tid; r < ho; r += BLK) solve_line_general(z + (size_t)r * wo, 1, wo, a.ky, fu);
    return;
  }
  if (spline_dense(ho, wo)) {
    __shared__ double z1[GLH_SPL_DENSE_MAX * GLH_SPL_DENSE_MAX];
    spline_fit_dense<BLK>(z, z1, wo, ho, a.inv + spline_inverse_off(ho), a.inv + spline_inverse_off(wo));
    return;
  }
  const double* fh = a.lu + a.lu_off[ho];
  const double* fw = a.lu + a.lu_off[wo];
  for (int c = tid; c < wo; c += BLK) solve_line(z + c, wo, ho, fh);
  __syncthreads();
  for (int r = tid; r < ho; r += BLK) solve_line(z + (size_t)r * wo, 1, wo, fw);
}
#endif

// ------------------------------------------------------------------------------------------
// K5  sample the spline at every particle (observer.py:178-214), scale by 1/(2 sigma^2)
//     (tracker.py:625), add the DEM term (motion.py:181-204), w = exp(-ll) + 1e-300
//     (tracker.py:146-149).
// ------------------------------------------------------------------------------------------
struct WeightArgs {
  const double* particles;
  double* weights;
  const double* motion;
  const uint8_t* active;
  const double* uv;
  const int32_t* box;
  const int32_t* obs_status;
  const double* tmpl_duv;
  const double* coef;
  const double* poly;  // [GLH_NPOLY][16] basis polynomials (glh_host.h)
  double* ll_out;      // optional [O][P][N] per-observer log likelihoods (debug), NaN = skipped
  const double* extra; // optional [P][N] log-likelihood term computed by the caller (a user-defined Motion's
                       // compute_log_likelihoods, motion.py:74-89), appended like the built-in one (tracker.py:143)
  uint32_t* pt_status;
  int32_t* pt_err_frame;
  double inv2s2[MAX_OBS];  // 1 / (2 sigma^2)
  int32_t on[MAX_OBS];
  int32_t N, P, O, tw, th, sse_cap, frame;
  int32_t fast;  // GLH_MATH_FAST
  int32_t cell_cap;  // fast: surfaces of up to this many cells are evaluated in per-cell form (the fused kernel's bound)
  int32_t linear;    // Tracker(interpolation={"kx": 1, "ky": 1}): `coef` is the surface itself, sampled bilinearly
  int32_t kx, ky;    // other orders than (3, 3) / (1, 1) when general != 0 (rows axis, columns axis)
  int32_t general;
  Surfaces surf;
};

__device__ __forceinline__ void sse_box_of(const int* box, const double* duv, int tw, int th,
                                           double* sb) {
  // tracker.py:617-620
  double beu = tw * 0.5 - 0.5, bev = th * 0.5 - 0.5;
  sb[0] = ((double)box[0] + beu) + duv[0];
  sb[1] = ((double)box[1] + bev) + duv[1];
  sb[2] = ((double)box[2] + -beu) + duv[0];
  sb[3] = ((double)box[3] + -bev) + duv[1];
}

constexpr int WEIGHTS_PER_THREAD = 4;

// 2^(j / 32), j < 32: the table of exp_fast (glh_math.h), made by the first 32 threads of a block
__device__ __forceinline__ void exp_table_fill(double* tab32) {
  if (threadIdx.x < GLH_EXP_TAB) tab32[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / GLH_EXP_TAB));
}

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_weights(WeightArgs a) {
  __shared__ double tab[16 * GLH_NPOLY];
  __shared__ double tab32[GLH_EXP_TAB];
  const int pt = blockIdx.y;
  if (a.active && !a.active[pt]) return;
  for (int k = threadIdx.x; k < 16 * GLH_NPOLY; k += BLK) tab[k] = a.poly[k];
  exp_table_fill(tab32);
  __syncthreads();
  const double* m = a.motion + (size_t)pt * GLH_MOTION_FULL_LEN;
  const bool has_motion_term = (int)m[18] <= GLH_MOTION_CYLINDRICAL;  // tangent models return None
  const bool gridded = m[20] != 0.0 || m[21] != 0.0;
  const double zs = has_motion_term ? m[17] : 0.0;
  const double dem_scale = zs != 0.0 ? 1.0 / (2.0 * (zs * zs)) : 0.0;
  bool oob = false;
  bool any_obs = false;  // uniform across the block
  for (int o = 0; o < a.O; ++o)
    any_obs |= a.on[o] && a.obs_status[(size_t)o * a.P + pt] == GLH_OBS_OK;
  const bool has_extra = a.extra != nullptr;
  if (!has_motion_term && !has_extra && !any_obs && !a.ll_out) return;  // every term is None: weights unchanged (tracker.py:146)
  for (int it = 0; it < WEIGHTS_PER_THREAD; ++it) {
    const int i = (blockIdx.x * WEIGHTS_PER_THREAD + it) * BLK + threadIdx.x;
    if (i >= a.N) break;
    double ll = 0.0;
    for (int o = 0; o < a.O; ++o) {
      const size_t slot = (size_t)o * a.P + pt;
      if (a.ll_out) a.ll_out[slot * a.N + i] = NAN;
      if (!a.on[o]) continue;
      if (a.obs_status[slot] != GLH_OBS_OK) continue;
      const int* box = a.box + slot * 4;
      const int wo = box[2] - box[0] - a.tw + 1, ho = box[3] - box[1] - a.th + 1;
      double sb[4];
      sse_box_of(box, a.tmpl_duv + slot * 2, a.tw, a.th, sb);
      double2 q = reinterpret_cast<const double2*>(a.uv)[slot * a.N + i];
      if (!(q.x >= sb[0] && q.x <= sb[2] && q.y >= sb[1] && q.y <= sb[3]))
        flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_SAMPLE_OUTSIDE, a.frame);
      double cu0 = cell_origin(sb[0], sb[2], wo), cv0 = cell_origin(sb[1], sb[3], ho);
      const double* coef = a.coef + slot * (size_t)a.sse_cap;
      // fast arithmetic: surfaces the fused kernel holds in per-cell form are evaluated by that formula here too
      const bool by_cell = !a.linear && !a.general && a.fast && spline_cells(ho) * spline_cells(wo) <= a.cell_cap;
      double val = a.general ? spline_eval_general(coef, wo, ho, wo, a.kx, a.ky, cv0, cu0, q.x, q.y)
                   : a.linear ? spline_eval_linear(coef, wo, ho, wo, cv0, cu0, q.x, q.y)
                   : by_cell ? spline_eval_cell_direct(tab, coef, wo, ho, wo, cv0, cu0, q.x, q.y)
                   : a.fast ? spline_eval_poly_fast(tab, coef, wo, ho, wo, cv0, cu0, q.x, q.y)
                            : spline_eval_poly(tab, coef, wo, ho, wo, cv0, cu0, q.x, q.y);
      if (a.ll_out) a.ll_out[slot * a.N + i] = val * a.inv2s2[o];
      ll += val * a.inv2s2[o];
    }
    if (has_motion_term && gridded) {
      const double* q = a.particles + ((size_t)pt * a.N + i) * 6;
      ll += a.fast ? dem_log_likelihood<true>(m, a.surf, q[0], q[1], q[2], &oob)
                   : dem_log_likelihood<false>(m, a.surf, q[0], q[1], q[2], &oob);
    } else if (zs != 0.0) {
      double z = a.particles[((size_t)pt * a.N + i) * 6 + 2];
      double d = m[16] - z;
      ll += dem_scale * (d * d);
    }
    if (has_extra) ll += a.extra[(size_t)pt * a.N + i];
    if (has_motion_term || has_extra || any_obs)
      a.weights[(size_t)pt * a.N + i] = a.fast ? weight_of<true>(ll, tab32) : weight_of<false>(ll, tab32);
  }
  if (oob) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_RASTER_OOB, a.frame);
}
#endif

// The posterior history [T][P][12] (mean | sigma per frame and point) as the two arrays a caller of Tracker.track
// receives, means [P][T][6] and sigmas [P][T][6] (tracks.py:52-88): one thread per (point, frame, component).
#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_tracks_layout(const double* moments, int T, int P, double* means, double* sigmas) {
  const size_t i = (size_t)blockIdx.x * BLK + threadIdx.x;  // index into [P][T][6]
  if (i >= (size_t)P * T * 6) return;
  const int k = (int)(i % 6);
  const size_t pt = i / 6;
  const int t = (int)(pt % T), p = (int)(pt / T);
  const double* m = moments + ((size_t)t * P + p) * 12;
  means[i] = m[k];
  sigmas[i] = m[6 + k];
}
#endif

// Test hook: sample a fitted surface (glh_stage_sample); one "point".
struct SampleArgs {
  const double* coef;
  int32_t ho, wo, n;
  int32_t kx, ky;  // 0, 0: the bicubic default (spline_eval); else spline_eval_general
  double sb[4];
  const double* uv;
  double* values;
  uint8_t* outside;
};
#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_sample(SampleArgs a) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= a.n) return;
  double u = a.uv[2 * i], v = a.uv[2 * i + 1];
  a.outside[i] = !(u >= a.sb[0] && u <= a.sb[2] && v >= a.sb[1] && v <= a.sb[3]);
  double cu0 = cell_origin(a.sb[0], a.sb[2], a.wo), cv0 = cell_origin(a.sb[1], a.sb[3], a.ho);
  a.values[i] = a.kx ? spline_eval_general(a.coef, a.wo, a.ho, a.wo, a.kx, a.ky, cv0, cu0, u, v)
                     : spline_eval(a.coef, a.wo, a.ho, a.wo, cv0, cu0, u, v);
}
#endif

// ------------------------------------------------------------------------------------------
// K6  systematic resampling (tracker.py:168-176, :222-223) + posterior moments
//     (tracker.py:72-76, :89-104), one block per point:
//       wn = w / w.sum()            -- w.sum() reproduces NumPy's pairwise tree bit for bit
//       c  = cumsum(wn)             -- float64 block scan in LDS
//       idx_j = #{k : c_k < (j+u)/n} -- binary search (np.searchsorted, side='left')
//       particles, weights = particles[idx], weights[idx]
//       mean = sum(w p)/sum(w), sigma = sqrt(sum(w (p-mean)^2)/sum(w)) of the gathered set,
//       accumulated in one pass around a pivot particle K (shifted moments: no cancellation
//       at UTM-scale coordinates).
// ------------------------------------------------------------------------------------------
// #{j in [0, n): j + u <= ck * scale}, scale = n / total: the number of systematic positions (j + u) / n at or
// below the cumulative weight ck / total, in fast arithmetic (no verification against the reference's rounding of
// the positions, which only the host-RNG parity mode needs).  NaN -> 0.
__device__ __forceinline__ int count_le_fast(double ck, double scale, double u, int n) {
  const double g = floor(glh_fma(ck, scale, -u)) + 1.0;
  return (int)min_nn(max_nn(g, 0.0), (double)n);  // (max returns its number operand: NaN -> 0)
}

struct ResampleArgs {
  const double* particles_in;
  const double* weights_in;
  double* particles_out;
  double* weights_out;
  const uint8_t* active;
  const double* u;   // [P] (systematic, host mode) or [P][N] (stratified / choice, host mode) or null
  int32_t* idx_out;  // [P][N] or null
  int32_t* n_draws;  // [P] residual: uniforms consumed, n - sum(repetitions); or null
  double* moments;   // [P][12] mean | sigma of the resampled set, or null
  uint32_t* pt_status;
  int32_t* pt_err_frame;
  const int32_t* leaf_off;  // NumPy pairwise-sum plan (depends only on N), see glh_host.h
  const int32_t* leaf_len;
  const int32_t* ops;        // (dst, a, b) per internal node, sorted by level
  const int32_t* level_off;  // [nlevels + 1]
  const int32_t* roots;      // chunk roots, summed left to right
  uint64_t seed, step;
  int32_t N, nleaves, nnodes, nlevels, nroots, rng_mode, frame, pt_base;
  int32_t method;  // GLH_RESAMPLE_*
  int32_t fast;    // GLH_MATH_FAST (systematic resampling only): see the fast branch in k_resample
};

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_resample(ResampleArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double wave_tot[NWAVES];
  __shared__ double red[NWAVES];
  double* c = reinterpret_cast<double*>(smem);  // [N] cumulative weights
  double* node = c + a.N;                       // [nnodes] pairwise-sum tree
  const int pt = blockIdx.x, tid = threadIdx.x;
  if (a.active && !a.active[pt]) return;
  const int N = a.N;
  const double* W = a.weights_in + (size_t)pt * N;
  // --- stage the weights in LDS with coalesced loads; everything below reads LDS
  for (int k = tid; k < N; k += BLK) c[k] = W[k];
  __syncthreads();
  // --- c.sum() as NumPy's pairwise tree: leaves of <= 128 items, 8 interleaved accumulators each (8 lanes per
  //     leaf), then the tree level by level.  Starts from data visible to the block, ends with a barrier.
  auto pairwise_total = [&]() -> double {
    const int sub = tid & 7;
    for (int L = tid >> 3; L < a.nleaves; L += BLK / 8) {
      const int off = a.leaf_off[L], len = a.leaf_len[L];
      double res;
      if (len < 8) {
        res = 0.0;
        if (sub == 0)
          for (int i = 0; i < len; ++i) res += c[off + i];
      } else {
        double r = c[off + sub];
        const int body = len - (len & 7);
        for (int i = 8; i < body; i += 8) r += c[off + i + sub];
        // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); float add is commutative, so a butterfly
        // gives every lane the same bits
        r += __shfl_xor(r, 1, WAVE);
        r += __shfl_xor(r, 2, WAVE);
        r += __shfl_xor(r, 4, WAVE);
        res = r;
        if (sub == 0)
          for (int i = body; i < len; ++i) res += c[off + i];
      }
      if (sub == 0) node[L] = res;
    }
    __syncthreads();
    for (int l = 0; l < a.nlevels; ++l) {
      for (int k = a.level_off[l] + tid; k < a.level_off[l + 1]; k += BLK) {
        const int32_t* op = a.ops + 3 * k;
        node[op[0]] = node[op[1]] + node[op[2]];
      }
      __syncthreads();
    }
    double t = node[a.roots[0]];
    for (int r = 1; r < a.nroots; ++r) t += node[a.roots[r]];
    __syncthreads();  // node[] may be rewritten by the next call
    return t;
  };
  // fast arithmetic, systematic resampling: no normalisation pass at all -- the raw weights are scanned and the
  // positions are scaled instead, pos_j <= c_k / total  <=>  j <= c_k * (N / total) - u (glh_point.h does the same)
  const bool fast_sys = a.fast && a.method == GLH_RESAMPLE_SYSTEMATIC;
  const double total = fast_sys ? 1.0 : pairwise_total();
  const int seg = (N + BLK - 1) / BLK;
  const int k0 = min(tid * seg, N), k1 = min(k0 + seg, N);
  const int lane = tid & (WAVE - 1);
  // inclusive cumsum of c[] in place: contiguous segment per thread, block scan of the segment sums; `quot`
  // divides every element by it first (cumsum(w / total)).  Ends with a barrier.
  auto cumsum_inplace = [&](bool divide, double quot) {
    double run = 0.0;
    for (int k = k0; k < k1; ++k) {
      run += divide ? c[k] / quot : c[k];
      c[k] = run;
    }
    double incl = run, prev;
    if (fast_sys) {  // (the fused kernel's fast arithmetic scans with the DPP ladder: same association here)
      incl = wave_scan_add_f64(run);
      prev = wave_shr1_f64(incl);
    } else {
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) {
        double t = __shfl_up(incl, off, WAVE);
        if (lane >= off) incl += t;
      }
      prev = __shfl_up(incl, 1, WAVE);  // exclusive prefix inside the wave (no subtraction)
      if (lane == 0) prev = 0.0;
    }
    if (lane == WAVE - 1) wave_tot[tid / WAVE] = incl;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < tid / WAVE; ++w) base += wave_tot[w];
    const double excl = base + prev;
    if (tid > 0)
      for (int k = k0; k < k1; ++k) c[k] = excl + c[k];
    __syncthreads();
  };
  uint16_t* sidx = reinterpret_cast<uint16_t*>(node + a.nnodes);  // [N], N < 65536
  if (a.method == GLH_RESAMPLE_RESIDUAL) {
    // tracker.py:188-203, arithmetic as written there: the integer repetition counts are subtracted from the
    // NORMALISED weights, so the scaled residuals change sign and their cumulative sum is not monotone;
    // np.searchsorted then returns whatever its bisection -- which narrows its range with the previous key's
    // result -- arrives at.  That search is one serial chain over the keys (thread 0).
    uint16_t* reps = sidx + N;                 // [N] repetitions = (n * weights).astype(int)
    __shared__ int s_R;
    int local = 0;
    for (int k = k0; k < k1; ++k) {
      const double wn = c[k] / total;          // weights / weights.sum()
      const int r = (int)((double)N * wn);     // astype(int): truncation
      reps[k] = (uint16_t)r;
      c[k] = wn - (double)r;                   // residuals = weights - repetitions
      local += r;
    }
    // exclusive scan of the per-thread repetition totals -> where each thread's copies start
    int incl_i = local;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const int t = __shfl_up(incl_i, off, WAVE);
      if (lane >= off) incl_i += t;
    }
    __shared__ int wave_rep[NWAVES];
    if (lane == WAVE - 1) wave_rep[tid / WAVE] = incl_i;
    __syncthreads();
    int start = incl_i - local;
    for (int w = 0; w < tid / WAVE; ++w) start += wave_rep[w];
    if (tid == BLK - 1) s_R = start + local;
    // initial_indexes = np.repeat(np.arange(n), repetitions)
    for (int k = k0; k < k1; ++k) {
      const int r = reps[k];
      for (int j = 0; j < r && start + j < N; ++j) sidx[start + j] = (uint16_t)k;
      start += r;
    }
    __syncthreads();
    const int R = min(s_R, N);
    if (tid == 0 && a.n_draws) a.n_draws[pt] = N - R;
    const double S = pairwise_total();         // residuals.sum()
    const double scale = 1.0 / S;              // residuals *= 1 / residuals.sum()
    for (int k = k0; k < k1; ++k) c[k] = c[k] * scale;
    __syncthreads();
    cumsum_inplace(false, 1.0);                // cumulative_sum = np.cumsum(residuals)
    if (tid == 0) {
      c[N - 1] = 1.0;                          // cumulative_sum[-1] = 1.0
      // np.searchsorted(cumulative_sum, np.random.random(n - len(initial_indexes))), side='left': NumPy's
      // npy_binsearch keeps [min_idx, max_idx) from one key to the next -- only one end is reset, depending on
      // whether the key grew (numpy/_core/src/npysort/binsearch.cpp)
      const int m = N - R;
      int min_idx = 0, max_idx = N;
      double last_key = 0.0;
      bool clamp = false;
      for (int j = 0; j < m; ++j) {
        double key;
        if (a.rng_mode == GLH_RNG_HOST) {
          key = a.u[(size_t)pt * N + j];
        } else {
          uint32_t r[4];
          philox4x32((uint32_t)j, (uint32_t)(pt + a.pt_base), (uint32_t)a.step, 0x52455344u, (uint32_t)a.seed,
                        (uint32_t)(a.seed >> 32), r);
          key = u01_halfopen(r[0], r[1]);
        }
        if (j == 0) last_key = key;
        if (last_key < key) {
          max_idx = N;
        } else {
          min_idx = 0;
          max_idx = max_idx < N ? max_idx + 1 : N;
        }
        last_key = key;
        while (min_idx < max_idx) {
          const int mid = min_idx + ((max_idx - min_idx) >> 1);
          if (c[mid] < key) min_idx = mid + 1; else max_idx = mid;
        }
        int res = min_idx;
        if (res >= N) {  // IndexError in the reference
          res = N - 1;
          clamp = true;
        }
        sidx[R + j] = (uint16_t)res;
      }
      if (clamp) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_RESAMPLE_CLAMP, a.frame);
    }
  } else {
  cumsum_inplace(!fast_sys, total);
  const double inv_n = 1.0 / (double)N;
  if (a.method == GLH_RESAMPLE_SYSTEMATIC) {
    // --- positions (tracker.py:173): pos_j = (j + u) * (1 / n)
    double u;
    if (a.rng_mode == GLH_RNG_HOST) {
      u = a.u[pt];
    } else {
      uint32_t r[4];
      philox4x32((uint32_t)(pt + a.pt_base), 0u, (uint32_t)a.step, 0x52455341u, (uint32_t)a.seed,
                    (uint32_t)(a.seed >> 32), r);
      u = u01_halfopen(r[0], r[1]);
    }
    // --- np.searchsorted(c, pos) by its inverse: source k serves the positions with
    //     c[k-1] < pos_j <= c[k], i.e. j in [f(k-1), f(k)) with f(k) = #{j : pos_j <= c[k]}.
    //     f is guessed arithmetically and fixed up with the exact float comparison, so the
    //     indices are exactly searchsorted's; each thread scatters the runs of its own k range.
    const double scale = fast_sys ? (double)N / c[N - 1] : 0.0;
    auto count_le = [&](double ck) -> int {
      if (fast_sys) return count_le_fast(ck, scale, u, N);
      double g = floor(ck * (double)N - u) + 1.0;
      int f = g < 0.0 ? 0 : (g > (double)N ? N : (int)g);
      while (f < N && ((double)f + u) * inv_n <= ck) ++f;
      while (f > 0 && ((double)(f - 1) + u) * inv_n > ck) --f;
      return f;
    };
    int f_prev = k0 > 0 ? count_le(c[k0 - 1]) : 0;
    for (int k = k0; k < k1; ++k) {
      int f = count_le(c[k]);
      for (int j = f_prev; j < f; ++j) sidx[j] = (uint16_t)k;
      f_prev = f;
    }
    if (k1 == N && k0 < N) {
      // positions beyond c[N-1] (searchsorted == N: IndexError in the reference): clamp + flag
      if (f_prev < N) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_RESAMPLE_CLAMP, a.frame);
      for (int j = f_prev; j < N; ++j) sidx[j] = (uint16_t)(N - 1);
    }
  } else {
    // --- one uniform per output (tracker.py:178-186 stratified, :205-209 np.random.choice): plain binary
    //     search of every position in the LDS-resident cumulative weights
    if (a.method == GLH_RESAMPLE_CHOICE) {
      // RandomState.choice: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(cdf, uniform, side='right')
      const double last = c[N - 1];
      __syncthreads();
      for (int k = tid; k < N; k += BLK) c[k] = c[k] / last;
      __syncthreads();
    }
    bool clamp = false;
    for (int j = tid; j < N; j += BLK) {
      double uj;
      if (a.rng_mode == GLH_RNG_HOST) {
        uj = a.u[(size_t)pt * N + j];
      } else {
        uint32_t r[4];
        philox4x32((uint32_t)j, (uint32_t)(pt + a.pt_base), (uint32_t)a.step, 0x53545241u, (uint32_t)a.seed,
                      (uint32_t)(a.seed >> 32), r);
        uj = u01_halfopen(r[0], r[1]);
      }
      int lo = 0, hi = N;
      if (a.method == GLH_RESAMPLE_STRATIFIED) {
        const double pos = ((double)j + uj) * inv_n;  // positions = (arange(n) + random(n)) * (1 / n)
        while (lo < hi) {                              // side='left': #{k : c[k] < pos}
          const int mid = (lo + hi) >> 1;
          if (c[mid] < pos) lo = mid + 1; else hi = mid;
        }
      } else {
        while (lo < hi) {                              // side='right': #{k : c[k] <= u}
          const int mid = (lo + hi) >> 1;
          if (c[mid] <= uj) lo = mid + 1; else hi = mid;
        }
      }
      if (lo >= N) {
        lo = N - 1;
        clamp = true;
      }
      sidx[j] = (uint16_t)lo;
    }
    if (clamp) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_RESAMPLE_CLAMP, a.frame);
  }
  }  // methods that search the cumulative weights
  __syncthreads();
  // --- gather + moments
  const double* Pin = a.particles_in + (size_t)pt * N * 6;
  double* Pout = a.particles_out + (size_t)pt * N * 6;
  double* Wout = a.weights_out + (size_t)pt * N;
  double K[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) K[k] = Pin[k];  // pivot: the point's first particle
  double s0 = 0.0, s1[6] = {0, 0, 0, 0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
  constexpr int GU = 4;  // independent gathers in flight per thread
  for (int j0 = tid; j0 < N; j0 += GU * BLK) {
    int lo[GU];
    double2 v0[GU], v1[GU], v2[GU];
    double w[GU];
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const int j = j0 + g * BLK;
      lo[g] = j < N ? sidx[j] : 0;
    }
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const double2* src = reinterpret_cast<const double2*>(Pin + (size_t)lo[g] * 6);
      v0[g] = src[0]; v1[g] = src[1]; v2[g] = src[2];
      w[g] = W[lo[g]];
    }
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const int j = j0 + g * BLK;
      if (j < N) {
        double2* dst = reinterpret_cast<double2*>(Pout + (size_t)j * 6);
        dst[0] = v0[g]; dst[1] = v1[g]; dst[2] = v2[g];
        Wout[j] = w[g];
        if (a.idx_out) a.idx_out[(size_t)pt * N + j] = lo[g];
        const double x[6] = {v0[g].x, v0[g].y, v1[g].x, v1[g].y, v2[g].x, v2[g].y};
        s0 += w[g];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double d = x[k] - K[k];
          double wd = w[g] * d;
          s1[k] += wd;
          s2[k] += wd * d;
        }
      }
    }
  }
  if (a.moments) {
    s0 = block_sum(s0, red);
    double* out = a.moments + (size_t)pt * 12;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double m1 = block_sum(s1[k], red) / s0;
      double m2 = block_sum(s2[k], red) / s0;
      if (tid == 0) {
        double var = m2 - m1 * m1;
        out[k] = K[k] + m1;
        out[6 + k] = sqrt(var > 0.0 ? var : 0.0);
      }
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------
// Run-length compact state (written by the fused step, glh_point.h: planar, chunk c of record r at c N + r)
// -> one record per particle: out[j] = in[uidx[j]] for the particles and the weights.  grid (ceil(N / BLK), P).
// ------------------------------------------------------------------------------------------
#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_expand_state(const double* pin, const double* win, const uint16_t* uidx,
                                                      double* pout, double* wout, int N) {
  const int pt = blockIdx.y, j = blockIdx.x * BLK + threadIdx.x;
  if (j >= N) return;
  const size_t base = (size_t)pt * N;
  const int r = uidx[base + j];
  const double2* src = reinterpret_cast<const double2*>(pin + base * 6) + r;
  double2* dst = reinterpret_cast<double2*>(pout + (base + j) * 6);
  const double2 v0 = src[0], v1 = src[N], v2 = src[2 * (size_t)N];
  dst[0] = v0; dst[1] = v1; dst[2] = v2;
  wout[base + j] = win[base + r];
}
#endif

// ------------------------------------------------------------------------------------------
// Test hook: Camera.xyz_to_uv on explicit points
// ------------------------------------------------------------------------------------------
#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_project_points(const CamDev* cam, const double* xyz, int n,
                                                        double* uv, int directions, double* depth) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  double u, v, d = NAN;
  project_f(*cam, cam_flags(*cam) | (directions ? CAM_F_DIRECTIONS : 0u), xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2],
            u, v, depth ? &d : nullptr);
  uv[2 * i] = u;
  uv[2 * i + 1] = v;
  if (depth) depth[i] = d;
}
#endif

// Camera.uv_to_xyz on explicit points; depth null = 1, else [n] (or [1], broadcast)
#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_unproject_points(const CamDev* cam, const double* uv, int n,
                                                          const double* depth, int n_depth, int directions,
                                                          double* xyz) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const double d = depth ? depth[n_depth == 1 ? 0 : i] : 1.0;
  double out[3];
  unproject(*cam, cam_flags(*cam), uv[2 * i], uv[2 * i + 1], d, directions, out);
  xyz[3 * i] = out[0];
  xyz[3 * i + 1] = out[1];
  xyz[3 * i + 2] = out[2];
}
#endif

// Test hook: Raster.sample at explicit points
#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ __launch_bounds__(BLK) void k_raster_sample(RasterDev r, const double* xy, int n, int order,
                                                       double* values, uint8_t* oob) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  bool out = false;
  values[i] = raster_sample(r, xy[2 * i], xy[2 * i + 1], order, &out);
  oob[i] = out;
}
#endif

#ifndef GLH_POINT_TU  // (the fused kernel's own translation units carry none of the staged kernels)
__global__ void k_fill_f64(double* p, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
#endif

#ifndef GLH_POINT_TU
// Diagnostic (glh_debug_draws): the numbers of the device streams, written out so that a test can hand the oracle the
// very draws a GLH_RNG_PHILOX run consumed.  kind 0: the six initialisation normals [P][N][6] (k_init_particles);
// kind 1: the three evolve normals of frame `step` [P][N][3] (evolve_noise); kind 2: the systematic resampling offset
// of frame `step` [P] (k_resample / k_point_step).
struct DrawsArgs {
  double* out;
  uint64_t seed, step;
  int32_t kind, N, P, pt_base;
};
__global__ __launch_bounds__(BLK) void k_debug_draws(DrawsArgs a) {
  const int pt = blockIdx.y;
  const int i = blockIdx.x * BLK + threadIdx.x;
  const uint32_t gp = (uint32_t)(pt + a.pt_base);
  if (a.kind == 2) {
    if (i == 0) {
      uint32_t r[4];
      philox4x32(gp, 0u, (uint32_t)a.step, 0x52455341u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
      a.out[pt] = u01_halfopen(r[0], r[1]);
    }
    return;
  }
  if (i >= a.N) return;
  if (a.kind == 0) {
    double n[6];
    philox_normals2(a.seed, i, gp, 0u, 0x494e4954u, n[0], n[1]);
    philox_normals2(a.seed, i, gp, 1u, 0x494e4954u, n[2], n[3]);
    philox_normals2(a.seed, i, gp, 2u, 0x494e4954u, n[4], n[5]);
    double* o = a.out + ((size_t)pt * a.N + i) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = n[k];
  } else {
    double n[3];
    evolve_noise(GLH_RNG_PHILOX, nullptr, a.seed, a.step, pt, a.pt_base, i, a.N, n);
    double* o = a.out + ((size_t)pt * a.N + i) * 3;
    o[0] = n[0]; o[1] = n[1]; o[2] = n[2];
  }
}
#endif

}  // namespace glh
