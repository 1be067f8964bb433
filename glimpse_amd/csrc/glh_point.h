// glh_point.h -- the fused per-point frame step for gfx950: one 512-thread workgroup owns one
// tracked point for the whole update (tracker.py:343-354) so that weights, cumulative sums and
// resample indices never leave the CU:
//
//   C  weights      spline coefficients staged in LDS, w = exp(-ll) + 1e-300   (tracker.py:126-149)
//   D  resample     NumPy-exact w.sum(), float64 LDS scan, inverse searchsorted (tracker.py:168-176)
//   E  gather       particles[idx] -- read the PRE-evolve record of the source particle and
//                   re-apply its evolve step (same noise: host normals or counter-based Philox),
//                   so the evolved state is never written to and re-read from HBM
//                                                                              (tracker.py:222-223)
//   F  moments      weighted mean / sigma of the resampled set                 (tracker.py:72-104)
//
// The staged kernels of glh_kernels.h stay as the general path (active masks, debug hooks,
// the reference's public step methods); this kernel is what glh_step runs.
#pragma once
#include "glh_kernels.h"

namespace glh {

constexpr int PT_BLK = 512;
constexpr int PT_WAVES = PT_BLK / WAVE;
constexpr int PT_COEF_CAP = 1600;  // spline coefficients per observer staged in LDS (<= 40 x 40)

__device__ __forceinline__ double pt_block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = v;
  __syncthreads();
  double t = red[0];
#pragma unroll
  for (int w = 1; w < PT_WAVES; ++w) t += red[w];
  return t;
}

struct PointArgs {
  const double* particles_in;  // [P][N][6] state after the previous frame (pre-evolve)
  double* particles_out;       // [P][N][6] evolved + resampled
  double* weights_tmp;         // [P][N] in: DEM log likelihood (if has_dem); out: this frame's weights
  double* weights_out;         // [P][N] weights[idx]
  const double* motion;
  const double* normals;       // [P][N][3] host-fed evolve normals, or null (Philox)
  const double* u;             // [P] host-fed resample offsets, or null (Philox)
  const double* uv;            // [O][P][N][2]
  const int32_t* box;          // [O][P][4]
  const int32_t* obs_status;   // [O][P]
  const double* tmpl_duv;      // [O][P][2]
  const double* coef;          // [O][P][sse_cap] spline coefficients
  const double* poly;          // [GLH_NPOLY][16]
  int32_t* idx_out;            // [P][N] or null
  double* moments;             // [P][12]
  uint32_t* pt_status;
  int32_t* pt_err_frame;
  const int32_t* leaf_off;
  const int32_t* leaf_len;
  const int32_t* ops;
  const int32_t* level_off;
  const int32_t* roots;
  uint64_t seed, step;
  double tau;
  double inv2s2[MAX_OBS];
  int32_t on[MAX_OBS];
  int32_t N, P, O, tw, th, sse_cap, frame, rng_mode, has_dem;
  int32_t nleaves, nnodes, nlevels, nroots;
};

// LDS: c[N] | region2 = max(PT_COEF_CAP coefficients, nnodes tree nodes + N uint16 indices)
__host__ __device__ __forceinline__ size_t pt_lds_bytes(int N, int O, int nnodes) {
  size_t r2a = (size_t)PT_COEF_CAP * sizeof(double);
  size_t r2b = (size_t)nnodes * sizeof(double) + (((size_t)N * sizeof(uint16_t) + 15) & ~(size_t)15);
  return (size_t)N * sizeof(double) + (r2a > r2b ? r2a : r2b);
}

__global__ __launch_bounds__(PT_BLK, 4) void k_point_step(PointArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double tab[16 * GLH_NPOLY];
  __shared__ double wave_tot[PT_WAVES];
  __shared__ double red[PT_WAVES];
  const int pt = blockIdx.x, tid = threadIdx.x;
  const int N = a.N;
  double* c = reinterpret_cast<double*>(smem);  // [N] weights, then cumulative weights
  double* r2 = c + N;
  const double* m = a.motion + (size_t)pt * GLH_MOTION_LEN;
  double* W = a.weights_tmp + (size_t)pt * N;

  // ---------------- C: weights ---------------------------------------------------------------
  for (int k = tid; k < 16 * GLH_NPOLY; k += PT_BLK) tab[k] = a.poly[k];
  // observers outermost: each one's surface is staged in LDS, sampled by every particle and
  // accumulated into c[] in the reference's order (tracker.py:139-146: obs 0, obs 1, ..., motion)
  for (int i = tid; i < N; i += PT_BLK) c[i] = 0.0;
  bool outside = false;
  for (int o = 0; o < a.O; ++o) {
    const size_t slot = (size_t)o * a.P + pt;
    if (!a.on[o] || a.obs_status[slot] != GLH_OBS_OK) continue;  // uniform across the block
    const int* box = a.box + slot * 4;
    const int wo = box[2] - box[0] - a.tw + 1, ho = box[3] - box[1] - a.th + 1;
    double sb[4];
    sse_box_of(box, a.tmpl_duv + slot * 2, a.tw, a.th, sb);
    const double cu0 = cell_origin(sb[0], sb[2], wo), cv0 = cell_origin(sb[1], sb[3], ho);
    const double scale = a.inv2s2[o];
    const double* cg = a.coef + slot * (size_t)a.sse_cap;
    const double2* uvp = reinterpret_cast<const double2*>(a.uv) + slot * N;
    const bool in_lds = wo * ho <= PT_COEF_CAP;
    __syncthreads();  // r2 free (previous observer's samples done), c[] initialised
    if (in_lds) {
      for (int k = tid; k < wo * ho; k += PT_BLK) r2[k] = cg[k];
      __syncthreads();
      for (int i = tid; i < N; i += PT_BLK) {
        const double2 q = uvp[i];
        if (!(q.x >= sb[0] && q.x <= sb[2] && q.y >= sb[1] && q.y <= sb[3])) outside = true;
        c[i] += spline_eval_poly(tab, r2, wo, ho, wo, cv0, cu0, q.x, q.y) * scale;
      }
    } else {
      for (int i = tid; i < N; i += PT_BLK) {
        const double2 q = uvp[i];
        if (!(q.x >= sb[0] && q.x <= sb[2] && q.y >= sb[1] && q.y <= sb[3])) outside = true;
        c[i] += spline_eval_poly(tab, cg, wo, ho, wo, cv0, cu0, q.x, q.y) * scale;
      }
    }
  }
  if (outside) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_SAMPLE_OUTSIDE, a.frame);
  for (int i = tid; i < N; i += PT_BLK) {
    double ll = c[i];
    if (a.has_dem) ll += W[i];  // CartesianMotion.compute_log_likelihoods, appended last (tracker.py:143)
    const double w = exp(-ll) + 1e-300;
    W[i] = w;
    c[i] = w;
  }
  __syncthreads();

  // ---------------- D: w.sum() as NumPy's pairwise tree, cumsum(w / total), searchsorted ------
  double* node = r2;
  {
    const int sub = tid & 7;
    for (int L = tid >> 3; L < a.nleaves; L += PT_BLK / 8) {
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
        r += __shfl_xor(r, 1, WAVE);
        r += __shfl_xor(r, 2, WAVE);
        r += __shfl_xor(r, 4, WAVE);
        res = r;
        if (sub == 0)
          for (int i = body; i < len; ++i) res += c[off + i];
      }
      if (sub == 0) node[L] = res;
    }
  }
  __syncthreads();
  for (int l = 0; l < a.nlevels; ++l) {
    for (int k = a.level_off[l] + tid; k < a.level_off[l + 1]; k += PT_BLK) {
      const int32_t* op = a.ops + 3 * k;
      node[op[0]] = node[op[1]] + node[op[2]];
    }
    __syncthreads();
  }
  double total = node[a.roots[0]];
  for (int r = 1; r < a.nroots; ++r) total += node[a.roots[r]];
  const int seg = (N + PT_BLK - 1) / PT_BLK;
  const int k0 = min(tid * seg, N), k1 = min(k0 + seg, N);
  double run = 0.0;
  for (int k = k0; k < k1; ++k) {
    run += c[k] / total;
    c[k] = run;
  }
  double incl = run;
  const int lane = tid & (WAVE - 1);
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    double t = __shfl_up(incl, off, WAVE);
    if (lane >= off) incl += t;
  }
  if (lane == WAVE - 1) wave_tot[tid / WAVE] = incl;
  double prev = __shfl_up(incl, 1, WAVE);
  if (lane == 0) prev = 0.0;
  __syncthreads();
  double base = 0.0;
  for (int w = 0; w < tid / WAVE; ++w) base += wave_tot[w];
  const double excl = base + prev;
  if (tid > 0)
    for (int k = k0; k < k1; ++k) c[k] = excl + c[k];
  __syncthreads();
  double u;
  if (a.rng_mode == GLH_RNG_HOST) {
    u = a.u[pt];
  } else {
    uint32_t r[4];
    philox4x32_10((uint32_t)pt, 0u, (uint32_t)a.step, 0x52455341u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
    u = u01_halfopen(r[0], r[1]);
  }
  const double inv_n = 1.0 / (double)N;
  uint16_t* sidx = reinterpret_cast<uint16_t*>(node + a.nnodes);
  {
    auto count_le = [&](double ck) -> int {
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
      if (f_prev < N) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_RESAMPLE_CLAMP, a.frame);
      for (int j = f_prev; j < N; ++j) sidx[j] = (uint16_t)(N - 1);
    }
  }
  __syncthreads();

  // ---------------- E + F: gather with re-evolve, moments --------------------------------------
  const double* Pin = a.particles_in + (size_t)pt * N * 6;
  double* Pout = a.particles_out + (size_t)pt * N * 6;
  double* Wout = a.weights_out + (size_t)pt * N;
  const double tau = a.tau, tau2 = a.tau * a.tau;
  auto evolved = [&](int k, double* x) {
    const double2* src = reinterpret_cast<const double2*>(Pin + (size_t)k * 6);
    const double2 v0 = src[0], v1 = src[1], v2 = src[2];
    x[0] = v0.x; x[1] = v0.y; x[2] = v1.x; x[3] = v1.y; x[4] = v2.x; x[5] = v2.y;
    double n[3];
    evolve_noise(a.rng_mode, a.normals, a.seed, a.step, pt, k, N, n);
    evolve_particle(x, m, n, tau, tau2);
  };
  double K[6];
  evolved(0, K);  // pivot of the shifted moments: the point's first evolved particle
  double s0 = 0.0, s1[6] = {0, 0, 0, 0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
  constexpr int GU = 2;
  for (int j0 = tid; j0 < N; j0 += GU * PT_BLK) {
    int lo[GU];
    double x[GU][6], w[GU];
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const int j = j0 + g * PT_BLK;
      lo[g] = j < N ? sidx[j] : 0;
    }
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      evolved(lo[g], x[g]);
      w[g] = W[lo[g]];
    }
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const int j = j0 + g * PT_BLK;
      if (j < N) {
        double2* dst = reinterpret_cast<double2*>(Pout + (size_t)j * 6);
        dst[0] = make_double2(x[g][0], x[g][1]);
        dst[1] = make_double2(x[g][2], x[g][3]);
        dst[2] = make_double2(x[g][4], x[g][5]);
        Wout[j] = w[g];
        if (a.idx_out) a.idx_out[(size_t)pt * N + j] = lo[g];
        s0 += w[g];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double d = x[g][k] - K[k];
          double wd = w[g] * d;
          s1[k] += wd;
          s2[k] += wd * d;
        }
      }
    }
  }
  s0 = pt_block_sum(s0, red);
  double* out = a.moments + (size_t)pt * 12;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double m1 = pt_block_sum(s1[k], red) / s0;
    double m2 = pt_block_sum(s2[k], red) / s0;
    if (tid == 0) {
      double var = m2 - m1 * m1;
      out[k] = K[k] + m1;
      out[6 + k] = sqrt(var > 0.0 ? var : 0.0);
    }
  }
}

}  // namespace glh
