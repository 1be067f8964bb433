// glh_point.h -- the fused frame step for gfx950: ONE workgroup owns ONE tracked point for the
// whole update (tracker.py:331-354), so per frame the particle state is read once from HBM and
// written once, and everything in between lives in registers and LDS:
//
//   A  evolve + NaN test + project every particle; observer 0's u stays in registers (PPT per
//      thread) and its v in c[] (LDS), wave-shuffle / LDS min-max -> integer search box (motion.py:165-179,
//      tracker.py:118, camera.py:591-628, tracker.py:580-603)
//   B  per observer: crop -> histogram -> CDF-match LUT -> 5x5 median high-pass -> float32 search
//      tile -> SSD surface -> not-a-knot spline coefficients, all in LDS  (tracker.py:605-614,
//      observer.py:210); tiles too large for LDS use the HBM workspaces with the same code
//   C  sample the spline at every particle, w = exp(-ll) + 1e-300         (tracker.py:622-625, :126-149)
//   D  NumPy-exact w.sum(), float64 scan (weights stay in LDS), inverse searchsorted; the sources that
//      serve at least one position (the survivors) are ranked                 (tracker.py:168-176)
//   E  particles[idx], run-length compact: for every SURVIVOR re-read its PRE-evolve record (L2 /
//      Infinity-Cache hot: this workgroup streamed it in phase A), re-apply its evolve step with the same
//      noise (host normals or counter-based Philox) -- the evolved state is never stored un-resampled --
//      and store ONE record however many copies it has; every output stores the 2-byte index of its
//      record                                                                   (tracker.py:222-223)
//   F  weighted mean / sigma of the resampled set                          (tracker.py:72-104)
//
// HBM traffic per particle-frame, U/N ~ 0.57 survivors: ~27 B read (A) + ~20 B re-read (E) + ~32 B of
// records and weights + 4 B of record indices.
// The staged kernels of glh_kernels.h remain the general path (active masks, debug capture,
// the reference's public step methods) and are bit-identical on the same inputs.
#pragma once
#include "glh_kernels.h"
#include <type_traits>

namespace glh {

// Two observers, plain code (round 5, experiment -DGLH_PT_RECOMP=1): observer 1's coordinates and the DEM term are not
// parked in memory between phase A and phase C (32 + 16 bytes per particle-frame through the uv scratch and the weights
// scratch) -- phase C re-evolves the particle from its pre-evolve record, like the gather does, and projects it again.
#ifndef GLH_PT_PRIO
#define GLH_PT_PRIO 0
#endif
#ifndef GLH_PT_RECOMP
#define GLH_PT_RECOMP 0
#endif
#ifndef GLH_PT_PREFETCH2
#define GLH_PT_PREFETCH2 0
#endif
// Tangent models over rasters: phase A parks every particle's evolved height for the gather's re-evolution (1) or the
// gather samples the surface again (0: an experiment of round 5, now that a sample from the window is ~45 instructions).
#ifndef GLH_PT_ZPARK
#define GLH_PT_ZPARK 1
#endif
#ifndef GLH_PT_LDS_BARRIERS_A
#define GLH_PT_LDS_BARRIERS_A 0
#endif

constexpr int PT_BLK = 512;    // threads per workgroup (TB) for N <= 5120: two workgroups share a CU
constexpr int PT_BLK_BIG = 1024;  // TB for larger N: c[N] alone is > half the LDS, one 16-wave workgroup per CU
constexpr int PT_MAX_TILE = 63;  // largest template side the fused kernel handles (rows padded to 64 floats)
constexpr int PT_NSTAMP = GLH_NSTAMP;
constexpr int PT_MAX_OBS = 4;    // observers per point in the fused kernel (= MAX_OBS of the library)

template <int TB>
__device__ __forceinline__ double pt_block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = v;
  __syncthreads();
  double t = red[0];
#pragma unroll
  for (int w = 1; w < TB / WAVE; ++w) t += red[w];
  return t;
}

// u of observer 0 lives in a per-thread register array (v is parked in c[i], LDS, which is free until
// phase C writes the log likelihoods); the particle loops stay ROLLED (small code, bounded live
// ranges) and address the array with compare-select chains instead of dynamic indexing (which would
// send it to scratch).
template <int PPT>
__device__ __forceinline__ double pt_pick(const double (&v)[PPT], int r) {
  double q = v[0];
#pragma unroll
  for (int k = 1; k < PPT; ++k) q = r == k ? v[k] : q;
  return q;
}
template <int PPT>
__device__ __forceinline__ void pt_put(double (&v)[PPT], int r, double q) {
#pragma unroll
  for (int k = 0; k < PPT; ++k) v[k] = r == k ? q : v[k];
}

// Windows of the dem and dem_sigma rasters around the point, in LDS (instantiations with the raster samples only)
template <bool GRID>
struct PtPatches {
  __device__ __forceinline__ const RasterPatch* get() const { return nullptr; }
};
template <>
struct PtPatches<true> {
  RasterPatch p[2];
  __device__ __forceinline__ const RasterPatch* get() const { return p; }
};
// Whole block: the window of raster r around (x, y) into *dst (LDS).  No barrier.
template <int TB>
__device__ __forceinline__ void pt_patch_load(const RasterDev& r, double x, double y, bool used, RasterPatch* dst) {
  const int tid = threadIdx.x;
  if (!r.z || !used) {  // (uniform)
    if (tid == 0) dst->w = dst->h = dst->i0 = dst->j0 = dst->full = dst->pair = 0;
    return;
  }
  int i0, j0, w, h;
  raster_patch_origin(r, x, y, i0, j0, w, h);  // (the same in every thread)
  static_assert(TB >= GLH_PATCH_W * GLH_PATCH_W + 2 * GLH_PATCH_W, "one item per thread");
  const int lj = tid / GLH_PATCH_W, li = tid - lj * GLH_PATCH_W;
  if (tid < GLH_PATCH_W * GLH_PATCH_W) {
    if (li < w && lj < h) dst->z[tid] = raster_node(r, i0 + li, j0 + lj);
  } else if (tid < GLH_PATCH_W * GLH_PATCH_W + GLH_PATCH_W) {
    // a coordinate and (fast arithmetic) the reciprocal width of the interval it starts
    const int k = tid - GLH_PATCH_W * GLH_PATCH_W;
    if (k < w) {
      const double g0 = r.gx[i0 + k], g1 = r.gx[k + 1 < w ? i0 + k + 1 : i0 + k];
      dst->ax[2 * k] = g0;
      dst->ax[2 * k + 1] = k + 1 < w ? rcp_nr(g1 - g0) : 0.0;
    }
  } else if (tid < GLH_PATCH_W * GLH_PATCH_W + 2 * GLH_PATCH_W) {
    const int k = tid - GLH_PATCH_W * GLH_PATCH_W - GLH_PATCH_W;
    if (k < h) {
      const double g0 = r.gy[j0 + k], g1 = r.gy[k + 1 < h ? j0 + k + 1 : j0 + k];
      dst->ay[2 * k] = g0;
      dst->ay[2 * k + 1] = k + 1 < h ? rcp_nr(g1 - g0) : 0.0;
    }
  }
  if (tid == 0) {
    dst->i0 = i0; dst->j0 = j0; dst->w = w; dst->h = h;
    dst->fkx = r.kx; dst->fky = r.ky;
    dst->full = w == GLH_PATCH_W && h == GLH_PATCH_W;
    dst->pair = 0;
  }
}

struct PointArgs {
  // The resampled state is stored RUN-LENGTH COMPACT: systematic resampling returns its sources in order, so the
  // copies of a source are adjacent and identical; only the first of each run is written (records 0 .. U-1) and
  // uidx[j] names the record of output j.  Expanded on demand (k_expand_state) for everything but this kernel.
  const double* particles_in;  // [P][N][6] state after the previous frame (pre-evolve), compact if uidx_in
  double* particles_out;       // [P][N][6] evolved + resampled, compact (uidx_out)
  double* weights_tmp;         // [P][N] scratch: the motion model's log-likelihood term (has_dem) from phase A to C
  double* weights_out;         // [P][N] weights[idx]
  const double* motion;
  const uint8_t* obs_mask;     // [P][O] or null
  const double* normals;       // [P][N][3] host-fed evolve normals, or null (Philox)
  const double* u;             // [P] host-fed resample offsets, or null (Philox)
  double* uv;                  // [O][P][N][2] scratch for observers >= 1 (observer 0 stays in registers)
  int32_t* box;                // [O][P][4]
  int32_t* obs_status;         // [O][P]
  const int32_t* tmpl_valid;
  const double* tmpl_duv;
  const float* tmpl_tile32;
  const double* tmpl_hist_v;
  const double* tmpl_hist_q;
  const int32_t* tmpl_hist_n;
  float* ws_search;            // [O][P][search_cap]  HBM workspaces for tiles that do not fit in LDS
  uint16_t* ws_keys;           // [O][P][keys_cap]
  double* ws_sse;              // [O][P][sse_cap]
  const double* lu;            // spline LU factors by size (glh_host.h)
  const int64_t* lu_off;
  const double* inv;           // explicit spline-matrix inverses, sides 4 .. GLH_SPL_DENSE_MAX (glh_host.h)
  const double* poly;          // [GLH_NPOLY][16]
  const uint16_t* uidx_in;     // [P][N] record of particle i in particles_in (run-length compact state), or null:
                               // particle i is record i
  uint16_t* uidx_out;          // [P][N] record of output j in particles_out / weights_out
  int32_t* idx_out;            // [P][N] or null
  unsigned long long* stamps;  // [P][PT_NSTAMP] s_memtime at the phase boundaries (diagnostic), or null
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
  double inv2s2[PT_MAX_OBS];
  ObsFrame obs[PT_MAX_OBS];
  CamDev cam[PT_MAX_OBS];  // by value; copied to LDS once per workgroup
  Surfaces surf;           // gridded dem / dem_sigma / viewshed (null pointers when absent)
  uint32_t cam_flags[PT_MAX_OBS];  // cam_flags(cam[o]): scalar, so the optional projection terms branch uniformly
  int32_t N, P, O, tw, th, tile_cap, search_cap, keys_cap, sse_cap, max_dim, frame, rng_mode, has_dem;
  int32_t hp_rx, hp_ry;  // half sizes of the median high-pass window (2, 2 = the 5 x 5 default; others: general code only)
  int32_t interp_k;      // order of the surface sampling: 3 (bicubic spline) or 1 (bilinear; general code only)
  int32_t cell_cap;  // fast arithmetic: surfaces of up to this many cells are sampled in per-cell form (0: never)
  int32_t r2_bytes;  // bytes of LDS behind c[N] (followed by the pairwise-sum plan, pt_plan_ints() ints)
  int32_t pt_base;   // global index of point 0 (sharding-invariant Philox streams)
  int32_t pt0;       // first point of this launch (glh_track runs the halves of a large batch on two streams)
  int32_t stop_at;   // diagnostic (tools/phase_counts.sh): every workgroup returns at this stamp (-1: never)
  int32_t nleaves, nnodes, nlevels, nroots;
};

__host__ __device__ __forceinline__ int pt_align16(int x) { return (x + 15) & ~15; }
// LDS row stride (floats) of the search tile: >= ws + 11 readable columns and == 8 (mod 32), so the
// row-split lanes of a strip (rows g = 0..7) start in distinct 16-byte bank slots
__host__ __device__ __forceinline__ int pt_search_ld(int ws) { return ((ws + 3 + 31) / 32) * 32 + 8; }
// histogram | LUT of nb bins (round 4: the cumulative counts are not written out any more -- the thread that owns a bin
// makes its LUT entry behind the scan, from registers)
__host__ __device__ __forceinline__ int pt_hcl_bytes(int nb) { return pt_align16(nb * 4) + pt_align16(nb * 8); }
// bytes of the arrays that always live in LDS: template tile + histogram / cumulative counts / LUT
__host__ __device__ __forceinline__ int pt_small_bytes(int tw, int th, int nb) {
  return pt_align16(th * ssd_twp(tw) * 4) + pt_hcl_bytes(nb);
}

// The raw-key tile of a w x h crop is stored with its 'reflect' border already in place (2 rows above and below,
// 2 columns left, 3 right), row stride even: every 5x5 window of a PAIR of pixels is then three aligned 32-bit
// words per row, without index arithmetic.
__host__ __device__ __forceinline__ int pt_keys_stride(int w) { return (w + 6 + 1) & ~1; }
__host__ __device__ __forceinline__ int pt_keys_count(int w, int h) { return (h + 4) * pt_keys_stride(w); }

// dynamic LDS the moment sums of phase F are parked in: [13][512] doubles + their 13 totals
__host__ __device__ __forceinline__ int pt_park_bytes() { return (13 * 512 + 16) * 8; }

// ints of the NumPy pairwise-sum plan staged in LDS: leaf_off | leaf_len | ops | level_off | roots
__host__ __device__ __forceinline__ int pt_plan_ints(int nleaves, int nnodes, int nlevels, int nroots) {
  return 2 * nleaves + 3 * (nnodes - nleaves) + (nlevels + 1) + nroots;
}

// Where one observer's tile arrays live.
struct TileWs {
  float* T;         // [th][twp] template, zero padded                         (always LDS)
  uint32_t* hist;   // [nb]                                                     (always LDS)
  double* lut;      // [nb]
  float* S;         // [hs][ld] search tile
  uint16_t* keys;   // [pt_keys_count(ws, hs)] raw pixel keys with their reflected border
  double* Z;        // [ho * wo] SSD surface -> spline coefficients
  double* Z1;       // [ho * wo] scratch of the dense spline fit (sides <= GLH_SPL_DENSE_MAX), always LDS
  const double* ih;  // explicit inverses of the ho / wo collocation matrices (LDS copy or the global table)
  const double* iw;
  const double* cdf_q;
  const double* cdf_v;
  const double* fh;  // LU factors for ho / wo
  const double* fw;
  int ld;
};

// scipy.ndimage 'reflect' (d c b a | a b c d | d c b a) border of the key tile, columns first: 2 left, 3 right of every
// crop row (tiles are >= 8 pixels on a side, so one reflection is exact) ...
template <int TB>
__device__ __forceinline__ void pt_border_cols(uint16_t* keys, int wp, int w, int h) {
  for (int idx = threadIdx.x; idx < 5 * h; idx += TB) {
    const int r = idx / 5, k = idx - 5 * r;
    const int c = k < 2 ? -1 - k : w + (k - 2);         // -1, -2, w, w + 1, w + 2
    const int src = k < 2 ? k : w - 1 - (k - 2);        //  0,  1, w - 1, w - 2, w - 3
    keys[r * wp + c] = keys[r * wp + src];
  }
}
// ... then (after a barrier) whole padded rows: -1 <- 0, -2 <- 1, h <- h - 1, h + 1 <- h - 2
template <int TB>
__device__ __forceinline__ void pt_border_rows(uint16_t* keys, int wp, int h) {
  for (int idx = threadIdx.x; idx < 4 * wp; idx += TB) {
    const int k = idx / wp, c = idx - k * wp - 2;
    const int r = k < 2 ? -1 - k : h + (k - 2);
    const int src = k < 2 ? k : h - 1 - (k - 2);
    keys[r * wp + c] = keys[src * wp + c];
  }
}

// tile - median_filter(tile) (tracker.py:530-531) of the CDF-matched tile into ws.S, from the bordered key tile: the match
// is monotone in the key, so the median is taken on the keys and `value_of` (key -> matched value, float64) is applied
// to the pixel and to its median.  Ends with a barrier.
// MAP = PtPackPair: the slot of ws.S receives the pixel's key and its median's, packed (key | median << 16), for a caller
// that maps them afterwards with every thread at work (pt_counts_finish).
struct PtPackPair {};
template <int TB, bool GEN, typename MAP>
__device__ __forceinline__ void pt_highpass_write(const TileWs& ws, const uint16_t* keys, int wp, int w, int h, int hp_rx,
                                                  int hp_ry, int key_max, MAP value_of) {
  constexpr bool PACK = std::is_same<MAP, PtPackPair>::value;
  const int tid = threadIdx.x, n = w * h;
  const int ld = ws.ld;
  auto put = [&](int at, int key, int med) {
    if constexpr (PACK)
      reinterpret_cast<uint32_t*>(ws.S)[at] = (uint32_t)key | ((uint32_t)med << 16);
    else
      ws.S[at] = (float)(value_of(key) - value_of(med));
  };
  // pad columns [w, ld) are only read for outputs that are discarded; keep them finite
  for (int idx = tid; idx < h * (ld - w); idx += TB) {
    const int r = idx / (ld - w), c = w + idx - r * (ld - w);
    ws.S[r * ld + c] = 0.0f;
  }
  // 5x5 median of the raw keys around every pixel, TWO horizontally adjacent pixels per thread on
  // packed 16-bit lanes (same selection network, v_pk_min/max_u16).  The window of the pair (r, c0), (r, c0 + 1)
  // is columns c0 - 2 .. c0 + 3 of rows r - 2 .. r + 2 of the bordered tile: three aligned 32-bit words per row
  // (c0 and the row stride are even), i.e. the pairs (k0 k1) (k2 k3) (k4 k5) directly and (k1 k2) (k3 k4) by a
  // funnel shift.
  if (GEN && (hp_rx != 2 || hp_ry != 2)) {
    // another window: the median by bisection over the key range, one pixel per thread, on the same key tile
    // (`reflect` by index arithmetic: the tile's own border is the 5 x 5 window's) -- what k_tileprep does
    const UDiv by_w2 = udiv_make(w);
    for (int idx = tid; idx < n; idx += TB) {
      const int r = udiv(by_w2, idx), c = idx - r * w;
      const int key = keys[r * wp + c];
      const int med = median_window(keys, wp, 0, w, h, r, c, hp_rx, hp_ry, key_max);
      put(r * ld + c, key, med);
    }
    __syncthreads();
    return;
  }
  // Round 4: FOUR vertically adjacent outputs per thread from sorted rows (glh_median.h: GLH_SORT5_NETWORK ...): the
  // eight window rows r0 - 2 .. r0 + 5 are fetched and sorted once, rows 1..4 and 3..6 of them are reduced to the six
  // candidates every window containing them shares, and each output costs ten more operations -- 87 min / max per
  // output instead of 198, and a quarter of the tasks (one pass over a steady-state tile instead of three).
  constexpr int MR = 4;
  const int npc = (w + 1) >> 1, nrb = (h + MR - 1) / MR;
  const UDiv by_npc = udiv_make(npc);
  for (int idx = tid; idx < nrb * npc; idx += TB) {
    const int rb = udiv(by_npc, idx), c0 = 2 * (idx - rb * npc), r0 = rb * MR;
    const uint32_t* win = reinterpret_cast<const uint32_t*>(keys + (r0 - 2) * wp + (c0 - 2));
    const int last = h + 3 - r0;  // the last row of the bordered tile, as a window row of this run (>= 4)
    uint32_t centre[MR];
    // window row dr of the run, sorted: pixel c0 in the low halves, pixel c0 + 1 in the high halves
    auto row = [&](int dr, glh_us2* v) {
      const int q = (dr < last ? dr : last) * (wp / 2);  // (rows past the tile serve outputs that are not stored)
      const uint32_t d0 = win[q], d1 = win[q + 1], d2 = win[q + 2];
      const uint32_t m01 = (d0 >> 16) | (d1 << 16), m12 = (d1 >> 16) | (d2 << 16);
      if (dr >= 2 && dr < 2 + MR) centre[dr - 2] = d1;
      v[0] = __builtin_bit_cast(glh_us2, d0);
      v[1] = __builtin_bit_cast(glh_us2, m01);
      v[2] = __builtin_bit_cast(glh_us2, d1);
      v[3] = __builtin_bit_cast(glh_us2, m12);
      v[4] = __builtin_bit_cast(glh_us2, d2);
      med_sort5_pk(v);
    };
    auto store = [&](int j, glh_us2 med) {
      const int r = r0 + j;
      if (r < h) {
        const glh_us2 key = __builtin_bit_cast(glh_us2, centre[j]);
        put(r * ld + c0, key.x, med.x);
        if (c0 + 1 < w) put(r * ld + c0 + 1, key.y, med.y);
      }
    };
    glh_us2 ra[5], rb5[5], r2[5], m12[10], m34[10], x[6];
    row(1, ra);
    row(2, r2);
    med_merge55_pk(ra, r2, m12);
    row(3, ra);
    row(4, rb5);
    med_merge55_pk(ra, rb5, m34);
    med_mid6_pk(m12, m34, x);
    row(0, ra);
    store(0, med_fin_pk(x, ra));
    row(5, rb5);
    store(1, med_fin_pk(x, rb5));
    row(6, ra);
    med_merge55_pk(rb5, ra, m12);  // (rows 5 and 6)
    med_mid6_pk(m34, m12, x);
    store(2, med_fin_pk(x, r2));
    row(7, ra);
    store(3, med_fin_pk(x, ra));
  }
  __syncthreads();
}

// extract_tile(histogram=template CDF) (tracker.py:605-607) into ws.S; see search_tile_from_box.
// GEN (the general instantiations): any odd median window up to 7 x 7 (`hp_rx`, `hp_ry` half sizes; Tracker(highpass=
// {"size": ...}), tracker.py:59, :530) -- the 5 x 5 default keeps its packed network everywhere.
template <int TB, bool GEN = false>
__device__ __forceinline__ void pt_tile_prep(const ObsFrame& ob, const int* box, int nb, int hist_n, const TileWs& ws,
                                             uint32_t* scan_tmp, unsigned long long* stp = nullptr, int hp_rx = 2,
                                             int hp_ry = 2) {
  // diagnostic s_memtime stamps 13 / 14 inside this stage (tools/phase_probe.py), when armed
#define TP_STAMP(k)                                                                                      \
  do {                                                                                                   \
    if (stp && threadIdx.x == 0) stp[(size_t)blockIdx.x * PT_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
  const int tid = threadIdx.x;
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  const int wp = pt_keys_stride(w);
  uint16_t* keys = ws.keys + 2 * wp + 2;  // element (r, c) of the crop at keys[r * wp + c], r in [-2, h + 2), c in [-2, w + 3)
  for (int b = tid; b < nb; b += TB) ws.hist[b] = 0;
  __syncthreads();
  // four pixel loads in flight per thread (each is a scattered byte fetch with a full memory latency)
  const UDiv by_w = udiv_make(w);
  for (int base = 0; base < n; base += 4 * TB) {
    int key[4], at[4];
    // (a thread past the end fetches pixel 0 again, and the channel count is decided OUTSIDE the four fetches:
    // unconditional loads issued back to back -- behind a guard or a per-pixel channel branch each load sat in its own
    // block and was waited for there, four memory latencies in a row)
    size_t px[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = base + q * TB + tid, idc = idx < n ? idx : 0;
      const int r = udiv(by_w, idc), c = idc - r * w;
      at[q] = r * wp + c;
      px[q] = (size_t)(box[1] + r) * ob.width + (box[0] + c);
    }
    if (ob.channels == 1) {  // uniform
#pragma unroll
      for (int q = 0; q < 4; ++q) key[q] = ob.frame[px[q]];
    } else if (ob.channels == 3) {
      // the three bytes of an RGB pixel as ONE unaligned 4-byte load and a byte sum (v_sad_u8); the frame's last pixel
      // reads the frame's last four bytes and takes the upper three
      const size_t lim = (size_t)ob.width * ob.height * 3 - 4;
      uint32_t wq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t a3 = px[q] * 3;
        __builtin_memcpy(&wq[q], ob.frame + (a3 > lim ? lim : a3), 4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        key[q] = (int)__builtin_amdgcn_sad_u8(px[q] * 3 > lim ? wq[q] >> 8 : wq[q] & 0x00ffffffu, 0u, 0u);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int sum = 0;
        for (int ch = 0; ch < ob.channels; ++ch) sum += ob.frame[px[q] * ob.channels + ch];
        key[q] = sum;
      }
    }
    asm volatile("" ::: "memory");  // (the loads stay above the stores and atomics)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (base + q * TB + tid >= n) key[q] = -1;
      if (key[q] >= 0) {
        keys[at[q]] = (uint16_t)key[q];
        atomicAdd(&ws.hist[key[q]], 1u);
      }
    }
  }
  __syncthreads();
  TP_STAMP(13);
  pt_border_cols<TB>(keys, wp, w, h);
  {
    // inclusive scan of the nb <= 2 * TB bins -- one bin per thread (gray: 256 bins) or two (RGB: 766) + block scan --
    // and, with the counts of the own bins still in registers, their LUT entries at once: np.cumsum(counts) / a.size
    // through np.interp of the template CDF.  (Round 4: the cumulative counts used to be written out, a barrier, and
    // the LUT made by whoever read them back.)
    const bool one = nb <= TB;  // uniform
    const int b0 = one ? tid : 2 * tid, b1 = one ? nb : 2 * tid + 1;
    const uint32_t h0 = b0 < nb ? ws.hist[b0] : 0u, h1 = b1 < nb ? ws.hist[b1] : 0u;
    const uint32_t local = h0 + h1;
    const uint32_t incl = wave_scan_add_u32(local);
    const int lane = tid & (WAVE - 1);
    if (lane == WAVE - 1) scan_tmp[tid / WAVE] = incl;
    __syncthreads();  // (also: the column borders are in place)
    uint32_t base = 0;
    for (int wv = 0; wv < tid / WAVE; ++wv) base += scan_tmp[wv];
    const uint32_t excl = base + incl - local;
    if (h0) ws.lut[b0] = np_interp((double)(excl + h0) / (double)n, ws.cdf_q, ws.cdf_v, hist_n);
    if (h1) ws.lut[b1] = np_interp((double)(excl + local) / (double)n, ws.cdf_q, ws.cdf_v, hist_n);
  }
  pt_border_rows<TB>(keys, wp, h);
  __syncthreads();
  TP_STAMP(14);
  pt_highpass_write<TB, GEN>(ws, keys, wp, w, h, hp_rx, hp_ry, nb - 1, [&](int k) -> double { return ws.lut[k]; });
#undef TP_STAMP
}

// cv2.matchTemplate(TM_SQDIFF) * 1/(tw*th) (tracker.py:609-614) from ws.S / ws.T into ws.Z (widened
// to float64 for the spline fit); arithmetic and summation order of k_ssd.
template <int TB, bool SPLIT16 = false>
__device__ __forceinline__ void pt_ssd(const TileWs& ws, int tw, int th, int wo, int ho, double* park = nullptr,
                                       double park_v = 0.0) {
  const int tid = threadIdx.x;
  const int twp = ssd_twp(tw);
  const int spr = (wo + SSD_W - 1) / SSD_W;
  const int nstrips = spr * ho;
  // 1 024-thread workgroups (one per CU: nothing else runs on the CU meanwhile) split the template rows of a strip over
  // up to 16 lanes while strips x lanes fit the workgroup; the float64 sum of the float32 row sums is exact, so the
  // grouping of the rows does not change a bit of the result
  int G = ssd_row_split(wo, ho);
  // (SPLIT16: the two-observer instantiations do the same at 512 threads -- their small surfaces leave most lanes idle
  // otherwise; C5 -1 %, the one-observer shapes +0.6 .. 0.8 %: profiles/ab_r04/r4j74_ab_g16.txt)
  if (TB >= 1024 || SPLIT16)
    while (G < 16 && nstrips * (G * 2) <= TB) G *= 2;
  const double inv_area = 1.0 / (double)(tw * th);
  const int lgG = __ffs(G) - 1;  // G is a power of two
  const UDiv by_spr = udiv_make(spr);
  for (int s0 = 0; s0 < nstrips; s0 += TB >> lgG) {
    const int strip = s0 + (tid >> lgG), g = tid & (G - 1);
    const bool live = strip < nstrips;
    const int rr = live ? udiv(by_spr, strip) : 0;
    const int cc = live ? (strip - rr * spr) * SSD_W : 0;
    double acc64[SSD_W];
#pragma unroll
    for (int k = 0; k < SSD_W; ++k) acc64[k] = 0.0;
    if (live) ssd_strip_rows(ws.S, ws.ld, ws.T, tw, th, twp, rr, cc, g, G, acc64);
#pragma unroll
    for (int k = 0; k < SSD_W; ++k) acc64[k] = group_sum_dpp(acc64[k], G);  // (G <= 16: inside a DPP row)
    if (live && g == 0) {
#pragma unroll
      for (int k = 0; k < SSD_W; ++k) {
        if (cc + k < wo) {
          const float raw = (float)acc64[k];
          const float val = (float)((double)raw * inv_area);
          ws.Z[(size_t)rr * wo + cc + k] = (double)val;
        }
      }
    }
  }
  if (park) *park = park_v;  // a value its caller fetched from memory before the SSD (LU factors)
  __syncthreads();
}

// solve_line (glh_kernels.h) with the same arithmetic, software-pipelined: the next element and its
// LU factors are loaded BEFORE the current result is stored, so a dependent step costs its float64
// chain instead of a full LDS round trip (the compiler cannot hoist loads above the aliasing store).
__device__ __forceinline__ void pt_solve_line(double* x, int stride, int n, const double* f) {
  const double *l1 = f, *l2 = f + n, *u0i = f + 2 * n, *u1 = f + 3 * n, *u2 = f + 4 * n;
  double ym1 = x[0], ym2 = 0.0;
  double nx = n > 1 ? x[stride] : 0.0, nl1 = n > 1 ? l1[1] : 0.0, nl2 = n > 1 ? l2[1] : 0.0;
  for (int i = 1; i < n; ++i) {
    const double cx = nx, cl1 = nl1, cl2 = nl2;
    if (i + 1 < n) {
      nx = x[(size_t)(i + 1) * stride];
      nl1 = l1[i + 1];
      nl2 = l2[i + 1];
    }
    double y = cx - cl1 * ym1;
    if (i >= 2) y -= cl2 * ym2;
    x[(size_t)i * stride] = y;
    ym2 = ym1;
    ym1 = y;
  }
  double xp1 = 0.0, xp2 = 0.0;
  double na = x[(size_t)(n - 1) * stride], nu0 = u0i[n - 1], nu1 = u1[n - 1], nu2 = u2[n - 1];
  for (int i = n - 1; i >= 0; --i) {
    double acc = na;
    const double c0 = nu0, c1 = nu1, c2 = nu2;
    if (i > 0) {
      na = x[(size_t)(i - 1) * stride];
      nu0 = u0i[i - 1];
      nu1 = u1[i - 1];
      nu2 = u2[i - 1];
    }
    if (i + 1 < n) acc -= c1 * xp1;
    if (i + 2 < n) acc -= c2 * xp2;
    acc *= c0;
    x[(size_t)i * stride] = acc;
    xp2 = xp1;
    xp1 = acc;
  }
}

template <int TB>
__device__ __forceinline__ void pt_spline_fit(const TileWs& ws, int wo, int ho) {
  const int tid = threadIdx.x;
  if (spline_dense(ho, wo)) {
    spline_fit_dense<TB>(ws.Z, ws.Z1, wo, ho, ws.ih, ws.iw);
    return;
  }
  for (int c = tid; c < wo; c += TB) pt_solve_line(ws.Z + c, wo, ho, ws.fh);
  __syncthreads();
  for (int r = tid; r < ho; r += TB) pt_solve_line(ws.Z + (size_t)r * wo, 1, wo, ws.fw);
  __syncthreads();
}

// barrier for phases that exchange data through LDS only: does not wait for this wave's outstanding
// global loads / stores (which __syncthreads() would)
__device__ __forceinline__ void pt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Whole block: n items from memory into LDS with four loads in flight per thread (a plain copy loop waits for
// every load before it issues the next).  No barrier.
template <int TB, typename T>
__device__ __forceinline__ void pt_stage(T* dst, const T* src, int n) {
  const int tid = threadIdx.x;
  for (int base = 0; base < n; base += 4 * TB) {
    const int i0 = base + tid, i1 = i0 + TB, i2 = i1 + TB, i3 = i2 + TB;
    // (four named values: an array here is left in scratch memory by the line below)
    const T v0 = src[i0 < n ? i0 : 0], v1 = src[i1 < n ? i1 : 0], v2 = src[i2 < n ? i2 : 0], v3 = src[i3 < n ? i3 : 0];
    // the loads stay above this line: left alone, the compiler sinks each into its guarded store -- load, wait, store,
    // four times over
    asm volatile("" ::: "memory");
    if (i0 < n) dst[i0] = v0;
    if (i1 < n) dst[i1] = v1;
    if (i2 < n) dst[i2] = v2;
    if (i3 < n) dst[i3] = v3;
  }
}

// The second half of the tile stage when the keys ARE the counts (16-bit and float frames): the key tile's borders, the
// template CDF into LDS, the interval of every occurring count in the CDF (jt), the median high-pass into ws.S.
// tab: the bucket table (dead: reused as the 257-entry index of the quantiles); cdf_lds / jt: see pt_tile_prep_wide.
template <int TB>
__device__ __forceinline__ void pt_counts_finish(const int* box, int hist_n, const TileWs& ws, uint32_t* tab, double* cdf_lds,
                                                 uint16_t* jt, int hp_rx, int hp_ry, unsigned long long* stp = nullptr) {
  const int tid = threadIdx.x;
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  const int wp = pt_keys_stride(w);
  uint16_t* keys = ws.keys + 2 * wp + 2;
  const UDiv by_w = udiv_make(w);
  pt_border_cols<TB>(keys, wp, w, h);
  const double *cq = ws.cdf_q, *cv = ws.cdf_v;
  if (cdf_lds) {  // (raw / low are dead)
    double* lv = cdf_lds + pt_align16(hist_n * 8) / 8;
    // quantiles and values requested together: one memory latency (two staging calls in a row were two)
    for (int base = 0; base < hist_n; base += 2 * TB) {
      const int i0 = base + tid, i1 = i0 + TB;
      const double q0 = ws.cdf_q[i0 < hist_n ? i0 : 0], q1 = ws.cdf_q[i1 < hist_n ? i1 : 0];
      const double v0 = ws.cdf_v[i0 < hist_n ? i0 : 0], v1 = ws.cdf_v[i1 < hist_n ? i1 : 0];
      asm volatile("" ::: "memory");  // (the loads stay above the stores)
      if (i0 < hist_n) { cdf_lds[i0] = q0; lv[i0] = v0; }
      if (i1 < hist_n) { cdf_lds[i1] = q1; lv[i1] = v1; }
    }
    cq = cdf_lds;
    cv = lv;
  }
  __syncthreads();
  pt_border_rows<TB>(keys, wp, h);
  // helpers.match_cdf (helpers.py:489-493) = np.interp(count / n, template quantiles, template values) for a pixel and
  // for its median: the interval search is made once per pixel, into jt[count] (equal counts write equal intervals), and
  // starts from an index of the quantiles by 1/256 steps (acc, over the bucket table: dead)
  // (count / n: np.cumsum(counts) / a.size, the division made once -- glh_math.h: count_fraction gives its quotients)
  const double dn = (double)n, rn = 1.0 / dn;
  uint16_t* acc = reinterpret_cast<uint16_t*>(tab);
  for (int i = tid; i <= 256; i += TB) {
    const int j = np_interp_find((double)i * (1.0 / 256.0), cq, hist_n);
    acc[i] = (uint16_t)(j < 0 ? 0 : j);
  }
  __syncthreads();
  for (int idx = tid; idx < n; idx += TB) {
    const int r = udiv(by_w, idx), c = idx - r * w;
    const int k = keys[r * wp + c];
    const double x = count_fraction(k, dn, rn);  // np.cumsum(counts) / a.size
    const int s256 = min(255, (int)(x * 256.0));
    // xp[acc[s]] <= s / 256 <= x < (s + 1) / 256 < xp[acc[s + 1] + 1]: the interval of x lies between them
    jt[k] = (uint16_t)np_interp_find(x, cq, hist_n, acc[s256], min(hist_n - 1, acc[s256 + 1] + 1));
  }
  __syncthreads();
  if (stp && tid == 0) stp[(size_t)blockIdx.x * PT_NSTAMP + 14] = __builtin_amdgcn_s_memtime();
  // The median on the counts, with every pixel's (count, median count) left in its slot of the tile; then the matched values
  // of the two and their difference by the pixel's own thread: eight evaluations of np.interp (two float64 divisions each)
  // on every thread, where the median's 220 tasks made sixteen each.
  pt_highpass_write<TB, true>(ws, keys, wp, w, h, hp_rx, hp_ry, n, PtPackPair{});
  auto value_of = [&](int k) -> double {
    const int j = jt[k];
    return np_interp_at(j == 0xffff ? NP_INTERP_LEFT : j, count_fraction(k, dn, rn), cq, cv, hist_n);
  };
  const int ld = ws.ld;
  for (int idx = tid; idx < n; idx += TB) {
    const int r = udiv(by_w, idx), c = idx - r * w;
    const uint32_t pair = reinterpret_cast<const uint32_t*>(ws.S)[r * ld + c];
    ws.S[r * ld + c] = (float)(value_of((int)(pair & 0xffffu)) - value_of((int)(pair >> 16)));
  }
  __syncthreads();
}

// 16-bit frames (uint16 gray or RGB; tracker.py:494-534 works on any dtype): a key is the pixel value or the channel sum
// (<= 3 * 65535), far too many for a histogram in LDS.  What extract_tile needs of np.unique is, per pixel, the number of
// pixels at or below its key -- cumsum(counts)[inverse] -- and that count IS a key the rest of the stage can work on: it
// is monotone in the pixel key (equal keys, equal counts), fits 16 bits (tiles of the fused step have < 65 536 pixels),
// so the median network runs on it unchanged, and its matched value is np.interp(count / n, template CDF) -- what the
// 8-bit LUT tabulates -- taken where it is needed.  The counts come from a two-level ranking in LDS: the keys of the
// tile, offset by their minimum, are bucketed by their high bits (PT_WIDE_BUCKETS buckets over the tile's own key range:
// at most 8 low bits remain, none for tiles spanning fewer than 1 024 levels), the bucket offsets are a block scan, and
// a pixel's count is its bucket's offset plus the members of its bucket whose low bits are at or below its own.
// General instantiations only; same results as search_tile_from_box16 (the staged kernel), bit for bit.
constexpr int PT_WIDE_BUCKETS = 1024;
__host__ __device__ __forceinline__ int pt_wide_tab_bytes() { return (PT_WIDE_BUCKETS + 4) * 4; }

// tab: [PT_WIDE_BUCKETS + 4] words in LDS; raw: [n] words, low: [n] bytes (LDS, or a workspace in memory for tiles that do
// not fit); cdf_lds: room for the template CDF (2 x align16(8 hist_n) bytes, may lie over raw / low) or null (the CDF is
// then read from memory); jt: [n + 1] uint16 clear of the CDF copy (may lie over raw / low otherwise); ws: keys (bordered
// tile), S, ld, cdf_q / cdf_v (memory).
template <int TB>
__device__ __forceinline__ void pt_tile_prep_wide(const ObsFrame& ob, const int* box, int hist_n, const TileWs& ws,
                                                  uint32_t* tab, uint32_t* raw, uint8_t* low, double* cdf_lds,
                                                  uint16_t* jt, uint32_t* scan_tmp, int hp_rx, int hp_ry,
                                                  unsigned long long* stp = nullptr) {
  // diagnostic s_memtime stamps 13 / 14 inside this stage (tools/phase_probe.py), when armed
#define TPW_STAMP(k)                                                                                     \
  do {                                                                                                   \
    if (stp && threadIdx.x == 0) stp[(size_t)blockIdx.x * PT_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
  static_assert(PT_WIDE_BUCKETS % TB == 0, "whole buckets per thread");
  constexpr int NBK = PT_WIDE_BUCKETS, PER = NBK / TB;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int w = box[2] - box[0], h = box[3] - box[1], n = w * h;
  const int wp = pt_keys_stride(w);
  uint16_t* keys = ws.keys + 2 * wp + 2;
  for (int b = tid; b < NBK; b += TB) tab[b] = 0;
  if (tid == 0) {
    tab[NBK] = 0xffffffffu;
    tab[NBK + 1] = 0u;
  }
  __syncthreads();
  const UDiv by_w = udiv_make(w);
  uint32_t kmin = 0xffffffffu, kmax = 0u;
  for (int base = 0; base < n; base += 4 * TB) {  // (four pixel loads in flight per thread)
    int key[4];
    // (unconditional loads -- a thread past the end fetches pixel 0 again -- with the channel count decided outside the
    // four fetches: behind a guard or a per-pixel channel branch every load was waited for in its own block)
    const uint16_t* px[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = base + q * TB + tid, idc = idx < n ? idx : 0;
      const int r = udiv(by_w, idc), c = idc - r * w;
      px[q] = reinterpret_cast<const uint16_t*>(ob.frame) + ((size_t)(box[1] + r) * ob.width + (box[0] + c)) * ob.channels;
    }
    if (ob.channels == 1) {  // uniform
#pragma unroll
      for (int q = 0; q < 4; ++q) key[q] = px[q][0];
    } else if (ob.channels == 3) {
      int ch[4][3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ch[q][0] = px[q][0]; ch[q][1] = px[q][1]; ch[q][2] = px[q][2];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) key[q] = ch[q][0] + ch[q][1] + ch[q][2];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int sum = 0;
        for (int k = 0; k < ob.channels; ++k) sum += px[q][k];
        key[q] = sum;
      }
    }
    asm volatile("" ::: "memory");  // (the loads stay above the stores)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (base + q * TB + tid >= n) key[q] = -1;
      if (key[q] >= 0) {
        raw[base + q * TB + tid] = (uint32_t)key[q];
        kmin = min(kmin, (uint32_t)key[q]);
        kmax = max(kmax, (uint32_t)key[q]);
      }
    }
  }
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, WAVE));
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, WAVE));
  }
  if (lane == 0) {
    atomicMin(&tab[NBK], kmin);
    atomicMax(&tab[NBK + 1], kmax);
  }
  __syncthreads();
  const uint32_t k0 = tab[NBK], range = tab[NBK + 1] - k0;
  int shift = 0;  // uniform; <= 8 (range <= 3 * 65535)
  while ((range >> shift) >= (uint32_t)NBK) ++shift;
  const uint32_t lmask = (1u << shift) - 1u;
  for (int idx = tid; idx < n; idx += TB) atomicAdd(&tab[(raw[idx] - k0) >> shift], 1u);
  __syncthreads();
  {
    // exclusive scan of the bucket counts: PER buckets per thread + block scan
    uint32_t cnt[PER], local = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      cnt[k] = tab[PER * tid + k];
      local += cnt[k];
    }
    const uint32_t incl = wave_scan_add_u32(local);
    if (lane == WAVE - 1) scan_tmp[tid / WAVE] = incl;
    __syncthreads();
    uint32_t run = incl - local;
    for (int wv = 0; wv < tid / WAVE; ++wv) run += scan_tmp[wv];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      tab[PER * tid + k] = run;
      run += cnt[k];
    }
  }
  __syncthreads();
  // the low bits of every key into its bucket (any order); afterwards tab[b] is the END of bucket b
  for (int idx = tid; idx < n; idx += TB) {
    const uint32_t d = raw[idx] - k0;
    const uint32_t pos = atomicAdd(&tab[d >> shift], 1u);
    low[pos] = (uint8_t)(d & lmask);
  }
  __syncthreads();
  for (int idx = tid; idx < n; idx += TB) {
    const uint32_t d = raw[idx] - k0, b = d >> shift, mine = d & lmask;
    const uint32_t lo = b ? tab[b - 1] : 0u, hi = tab[b];
    uint32_t cnt = hi;  // (no low bits: a bucket is one key)
    if (shift) {
      cnt = lo;
      for (uint32_t j = lo; j < hi; ++j) cnt += low[j] <= mine;
    }
    const int r = udiv(by_w, idx), c = idx - r * w;
    keys[r * wp + c] = (uint16_t)cnt;  // np.cumsum(counts)[inverse]
  }
  __syncthreads();
  TPW_STAMP(13);
#undef TPW_STAMP
  pt_counts_finish<TB>(box, hist_n, ws, tab, cdf_lds, jt, hp_rx, hp_ry, stp);
}

#define PT_STAMP(k)                                                                      \
  do {                                                                                   \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * PT_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); \
    if (a.stop_at == (k)) return;                                                        \
  } while (0)

// SURF: the general instantiation -- gridded surfaces (dem / dem_sigma / viewshed rasters) and every motion model
// (Cartesian, Cylindrical and the tangent models that follow the surface, motion.py:92-522; the kind is a property of
// the point, hence uniform in its workgroup).  The common instantiation (!SURF) evolves CartesianMotion over constant
// surfaces only and carries none of the other code or its registers.
// FAST: fast arithmetic (GLH_MATH_FAST, glh_math.h): fused multiply-adds, Newton reciprocals, table exp, and a
// resampling that scans the raw weights and scales the positions instead of normalising (no NumPy-exact sum tree).
// SC (the surface / generality code): 0 the plain code, 1 the general code over constant surfaces, 2 the general code with
// the context's rasters (gridded dem / dem_sigma, viewshed).  The raster samples are compiled into code 2 only: in one
// instantiation with the rest they cost every run of the general code registers (spills inside phase A's loop).
template <int TB, int PPT, int MINW, int NOBS, int SC, bool FAST, bool CONTRACT>
__global__ __launch_bounds__(TB, MINW) void k_point_step(PointArgs a) {
  constexpr bool SURF = SC != 0, GRID = SC == 2;
  constexpr int PT_WAVES = TB / WAVE;
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double tab[16 * GLH_NPOLY];
  __shared__ double wave_tot[PT_WAVES];
  __shared__ double bred[NOBS][PT_WAVES][5];
  __shared__ uint32_t scan_tmp[PT_WAVES];
  __shared__ int s_box[NOBS][4];
  __shared__ double s_uvbb[NOBS][4];  // min u, min v, max u, max v of the point's projected particles (phase A)
  __shared__ int s_status[NOBS];
  __shared__ double s_u;     // np.random.random() of this point's systematic resampling (tracker.py:173)
  __shared__ double s_K[6];  // the point's first evolved particle: pivot of the shifted moments (phase A -> F)
  __shared__ PtPatches<GRID> s_patches;  // (code 2: windows of the dem / dem_sigma rasters around the point)
  __shared__ CamDev s_cam[NOBS];           // cameras: LDS broadcast reads instead of ~60 live SGPRs each
  __shared__ double s_m[GLH_MOTION_FULL_LEN];  // this point's motion parameters: the loops below store to global
                                          // memory, so reading them through a global pointer would reload
                                          // (and wait for) them on every iteration
  __shared__ double tab32[GLH_EXP_TAB];   // 2^(j/32) for exp_fast
  __shared__ double s_scale;              // FAST: N / sum of the weights (phase D)
  const int pt = blockIdx.x + a.pt0, tid = threadIdx.x;
  const int lane = tid & (WAVE - 1), wave = tid / WAVE;
  const int N = a.N;
  double* c = reinterpret_cast<double*>(smem);  // [N] log likelihoods -> weights -> cumulative weights
  unsigned char* r2 = smem + pt_align16(N * (int)sizeof(double));
  const double* m = s_m;
  const double* Pin = a.particles_in + (size_t)pt * N * 6;
  double* W = a.weights_tmp + (size_t)pt * N;
  const double tau = a.tau, tau2 = a.tau * a.tau;

  // The common instantiation in fast arithmetic is compiled for the configuration long device-RNG runs have on all but
  // their first frame -- device Philox draws, every observer on and unmasked, compact input records, plain cameras --
  // so that none of those tests is a branch in its particle loops; the host (fused_step) sends every other case to the
  // general instantiation.
  static_assert(!CONTRACT || FAST, "the compile-time contract belongs to the fast arithmetic");
  constexpr bool COMMON = CONTRACT;
  constexpr bool RECOMP = GLH_PT_RECOMP && NOBS == 2 && PPT > 0 && !SURF;
  const int rng_mode = COMMON ? (int)GLH_RNG_PHILOX : a.rng_mode;
  PT_STAMP(0);
  const uint16_t* uin = COMMON || a.uidx_in ? a.uidx_in + (size_t)pt * N : nullptr;
  int rec_first = 0;  // (common instantiation: the record of this thread's first particle, see below)
  if constexpr (COMMON) rec_first = (int)uin[tid < N ? tid : 0];
  // (code 2: the position the raster windows are centred on -- particle 0 before the step -- is requested here, two dependent
  // loads that travel during the prologue; made behind its barrier they were two memory latencies of their own)
  double2 grid_q0 = make_double2(0.0, 0.0);
  if constexpr (GRID)
    grid_q0 = reinterpret_cast<const double2*>(Pin)[(size_t)(uin ? (int)uin[0] : 0) * (COMMON || uin ? 1 : 3)];
  if (tid == 0) {
    if (rng_mode == GLH_RNG_HOST) {
      s_u = a.u[pt];
    } else {
      uint32_t r[4];
      philox4x32((uint32_t)(pt + a.pt_base), 0u, (uint32_t)a.step, 0x52455341u, (uint32_t)a.seed,
                 (uint32_t)(a.seed >> 32), r);
      s_u = u01_halfopen(r[0], r[1]);
    }
  }
  // Round 4: every table of the prologue is REQUESTED before the first of them is stored (poly, motion parameters,
  // cameras, the record indices below).  Written as load-store pairs, each pair waited for its own memory latency before
  // the next load was issued: four latencies in a row at the head of every workgroup.
  static_assert(16 * GLH_NPOLY <= 512, "one entry of the basis table per thread");
  static_assert(sizeof(CamDev) % 8 == 0, "CamDev is copied as doubles");
  constexpr int CW = sizeof(CamDev) / 8;
  static_assert(PT_MAX_OBS * CW <= 512, "one camera word per thread");
  const double pro_poly = tid < 16 * GLH_NPOLY ? a.poly[tid] : 0.0;
  const double pro_motion = tid < GLH_MOTION_FULL_LEN ? a.motion[(size_t)pt * GLH_MOTION_FULL_LEN + tid] : 0.0;
  const double pro_cam = tid < NOBS * CW ? reinterpret_cast<const double*>(&a.cam[0])[tid] : 0.0;
  if (FAST) exp_table_fill(tab32);
  // the pairwise-sum plan (phase D) is read level by level between barriers: from LDS, not from HBM
  int32_t* p_leaf_off = reinterpret_cast<int32_t*>(r2 + a.r2_bytes);
  int32_t* p_leaf_len = p_leaf_off + a.nleaves;
  int32_t* p_ops = p_leaf_len + a.nleaves;
  int32_t* p_level_off = p_ops + 3 * (a.nnodes - a.nleaves);
  int32_t* p_roots = p_level_off + a.nlevels + 1;
  if constexpr (!FAST) {  // (fast arithmetic scans the raw weights: no sum tree)
    for (int k = tid; k < a.nleaves; k += TB) {
      p_leaf_off[k] = a.leaf_off[k];
      p_leaf_len[k] = a.leaf_len[k];
    }
    for (int k = tid; k < 3 * (a.nnodes - a.nleaves); k += TB) p_ops[k] = a.ops[k];
    if (tid <= a.nlevels) p_level_off[tid] = a.level_off[tid];
    if (tid < a.nroots) p_roots[tid] = a.roots[tid];
  }
  // record of every particle (compact input state): staged in region 2, which is free until phase B
  int hist_n0, valid_o;
  uint16_t* s_rec = reinterpret_cast<uint16_t*>(r2);
  {
    // the record indices as 16-byte words (the rows of uidx are N uint16 apart), up to four per thread (N <= 16 384),
    // requested with the tables above; THEN everything is stored
    const bool words = (COMMON || uin) && (N & 7) == 0 && (N >> 3) <= 4 * TB;  // uniform
    const int nw = N >> 3;
    const uint4* uw = reinterpret_cast<const uint4*>(uin);
    uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0, w2 = w0, w3 = w0;
    if (words) {
      w0 = uw[tid < nw ? tid : 0];
      w1 = uw[tid + TB < nw ? tid + TB : 0];
      if (nw > 2 * TB) {  // uniform (N > 8192 at 512 threads)
        w2 = uw[tid + 2 * TB < nw ? tid + 2 * TB : 0];
        w3 = uw[tid + 3 * TB < nw ? tid + 3 * TB : 0];
      }
    }
    // (fetched here, used at the end of phase A: no memory latency between the last particle and the search box -- and
    // behind the loads above: a uniform value is waited for where it is defined)
    valid_o = tid < NOBS ? (int)a.tmpl_valid[(size_t)tid * a.P + pt] : 0;
    hist_n0 = a.tmpl_hist_n[pt];  // (last: uniform, hence waited for at once -- with everything requested above)
    asm volatile("" : "+v"(valid_o));  // (not examined before this point: its test would wait for it on its own)
    asm volatile("" ::: "memory");  // (the loads above stay above the stores below)
    if (tid < 16 * GLH_NPOLY) tab[tid] = pro_poly;
    if (tid < GLH_MOTION_FULL_LEN) s_m[tid] = pro_motion;
    if (tid < NOBS * CW) reinterpret_cast<double*>(&s_cam[0])[tid] = pro_cam;
    if (words) {
      uint4* sw = reinterpret_cast<uint4*>(s_rec);
      if (tid < nw) sw[tid] = w0;
      if (tid + TB < nw) sw[tid + TB] = w1;
      if (nw > 2 * TB) {
        if (tid + 2 * TB < nw) sw[tid + 2 * TB] = w2;
        if (tid + 3 * TB < nw) sw[tid + 3 * TB] = w3;
      }
    } else if (COMMON || uin) {
      pt_stage<TB>(s_rec, uin, N);
    } else {
      for (int k = tid; k < N; k += TB) s_rec[k] = (uint16_t)k;
    }
  }
  bool live[NOBS];  // uniform across the block
#pragma unroll
  for (int o = 0; o < NOBS; ++o) live[o] = COMMON || (a.obs[o].on && (!a.obs_mask || a.obs_mask[(size_t)pt * a.O + o]));
  // Template tile (zero padded rows) and template CDF of one observer into the head of region 2: [T | cq | cv].
  // The usual sizes (two entries per thread): every load of this thread is in flight before the first LDS store (one
  // memory latency instead of four) -- and, in two halves, observer 0's loads are issued at the end of phase A, so
  // that they travel while the search box is being reduced.
  struct TmplRegs {
    float f0, f1;
    double q0, v0, q1, v1;
  };
  auto tmpl_small = [&](int hist_n) { return a.th * ssd_twp(a.tw) <= 2 * TB && hist_n <= 2 * TB; };
  auto tmpl_issue = [&](int o, int hist_n, TmplRegs& t) {
    const size_t slot = (size_t)o * a.P + pt;
    const int tw = a.tw, th = a.th, twp = ssd_twp(tw);
    const float* tg = a.tmpl_tile32 + slot * a.tile_cap;
    const UDiv by_twp = udiv_make(twp);
    const double* hv_g = a.tmpl_hist_v + slot * a.tile_cap;
    const double* hq_g = a.tmpl_hist_q + slot * a.tile_cap;
    const int i0 = tid, i1 = tid + TB;
    const int r0 = udiv(by_twp, i0), j0 = i0 - r0 * twp, r1 = udiv(by_twp, i1), j1 = i1 - r1 * twp;
    const bool t0 = i0 < th * twp && j0 < tw, t1 = i1 < th * twp && j1 < tw;
    t.f0 = t0 ? tg[r0 * tw + j0] : 0.0f;
    t.f1 = t1 ? tg[r1 * tw + j1] : 0.0f;
    const bool h0 = i0 < hist_n, h1 = i1 < hist_n;
    t.q0 = h0 ? hq_g[i0] : 0.0; t.v0 = h0 ? hv_g[i0] : 0.0;
    t.q1 = h1 ? hq_g[i1] : 0.0; t.v1 = h1 ? hv_g[i1] : 0.0;
  };
  auto tmpl_store = [&](int hist_n, const TmplRegs& t) {
    const int th = a.th, twp = ssd_twp(a.tw);
    float* T = reinterpret_cast<float*>(r2);
    double* cq = reinterpret_cast<double*>(r2 + pt_align16(th * twp * 4));
    double* cv = cq + pt_align16(hist_n * 8) / 8;
    const int i0 = tid, i1 = tid + TB;
    if (i0 < th * twp) T[i0] = t.f0;
    if (i1 < th * twp) T[i1] = t.f1;
    if (i0 < hist_n) { cq[i0] = t.q0; cv[i0] = t.v0; }
    if (i1 < hist_n) { cq[i1] = t.q1; cv[i1] = t.v1; }
  };
  auto load_template = [&](int o, bool with_cdf) {
    const size_t slot = (size_t)o * a.P + pt;
    const int tw = a.tw, th = a.th, twp = ssd_twp(tw);
    const int hist_n = with_cdf ? a.tmpl_hist_n[slot] : 0;
    float* T = reinterpret_cast<float*>(r2);
    const float* tg = a.tmpl_tile32 + slot * a.tile_cap;
    const UDiv by_twp = udiv_make(twp);
    double* cq = reinterpret_cast<double*>(r2 + pt_align16(th * twp * 4));
    double* cv = cq + pt_align16(hist_n * 8) / 8;
    const double* hv_g = a.tmpl_hist_v + slot * a.tile_cap;
    const double* hq_g = a.tmpl_hist_q + slot * a.tile_cap;
    if (tmpl_small(hist_n)) {
      TmplRegs t;
      tmpl_issue(o, hist_n, t);
      tmpl_store(hist_n, t);
    } else {
      for (int idx = tid; idx < th * twp; idx += TB) {
        const int i = udiv(by_twp, idx), j = idx - i * twp;
        T[idx] = j < tw ? tg[i * tw + j] : 0.0f;
      }
      for (int k = tid; k < hist_n; k += TB) {
        cq[k] = hq_g[k];
        cv[k] = hv_g[k];
      }
    }
  };
  // Round 4 (common instantiation): the thread's FIRST record is requested before the barrier -- its index was fetched at
  // the top of the kernel, straight from memory, not through the staged copy -- and the barrier waits for the LDS
  // stores only: the kernel starts with ONE memory latency where it had two (tables, then records).
  double2 pf0, pf1, pf2;
  if constexpr (COMMON) {
    const double2* src = reinterpret_cast<const double2*>(Pin) + (size_t)rec_first;
    pf0 = src[0]; pf1 = src[N]; pf2 = src[2 * N];
    pt_lds_barrier();
  } else {
    __syncthreads();
  }
  PT_STAMP(15);

  // (the tangent models have no log-likelihood term: Motion.compute_log_likelihoods returns None, tracker.py:146)
  const bool motion_term = !SURF || (int)s_m[18] <= GLH_MOTION_CYLINDRICAL;  // uniform
  // A record is three 16-byte chunks.  One record per particle (what every other kernel reads and writes): chunk c
  // of record r at 3 r + c.  Compact (this kernel's own output): PLANAR, chunk c of record r at c N + r, so that
  // consecutive lanes store, and mostly load, consecutive 16-byte words.
  const int rec_stride = COMMON || uin ? 1 : 3, chunk_stride = COMMON || uin ? N : 1;
  const double2* Pin2 = reinterpret_cast<const double2*>(Pin);
  if constexpr (GRID) {
    // The point's particles fall into a few cells of its surfaces: a 12 x 12-node window of each raster around particle
    // 0 (before the step) is brought into LDS, and the samples of phases A and E read it -- a sample from memory is two
    // rounds of latency (coordinates, then nodes), and the tangent models take two per particle.  Samples that leave the
    // window read the raster as before.
    const double2 q0 = grid_q0;
    pt_patch_load<TB>(a.surf.dem, q0.x, q0.y, m[20] != 0.0, &s_patches.p[0]);
    pt_patch_load<TB>(a.surf.dem_sigma, q0.x, q0.y, m[21] != 0.0, &s_patches.p[1]);
    if (tid == 0) {  // (its own stores: both windows whole, on one grid -- the host compared the coordinates --, at one origin)
      RasterPatch* p = s_patches.p;
      p[0].pair = a.surf.same_grid && p[0].full && p[1].full && p[0].i0 == p[1].i0 && p[0].j0 == p[1].j0;
    }
    __syncthreads();
  }
  // (code 2, fast arithmetic) the windows' origins and cell sizes in scalar registers: glh_math.h, RasterWin
  RasterWin wins_v[2] = {};
  const RasterWin* wins = nullptr;
  if constexpr (GRID && FAST) {
    auto uni = [](double v) -> double {
      const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
      return __hiloint2double(hi, lo);
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const RasterPatch& p = s_patches.p[q];
      wins_v[q].x0 = uni(p.ax[0]); wins_v[q].kx = uni(p.fkx);
      wins_v[q].y0 = uni(p.ay[0]); wins_v[q].ky = uni(p.fky);
    }
    wins = wins_v;
  }
  // the evolve step of particle k re-applied to its pre-evolve record x (phase E)
  // CartesianMotion with axyz_sigma[2] == 0 (uniform): the third normal only ever meets that zero
  const bool third = SURF || m[15] != 0.0;
  // The tangent models (uniform per point): phase A parks every particle's evolved height in the second half of the point's
  // observer-0 slot of the uv scratch (the first half is observer 0's v where PPT = 0; nothing else lives there), and
  // the gather's re-evolution takes it from there instead of sampling the surface twice more, with a square root and a
  // third normal, per survivor.
  // (Instantiations with the raster samples only: over constant surfaces the re-evolution is cheap, and the extra copy of
  // the evolve step cost the general code 2-3 % in spills.)
  const bool tangent_pt = GRID && (int)m[18] >= GLH_MOTION_TANGENT_CARTESIAN;
  double* ZP = a.uv + (size_t)pt * N * 2 + N;
  auto evolve_loaded = [&](int k, double* x, double z_parked) {
    double n[3];
    if constexpr (SURF) {
      bool oob = false;  // (flagged by phase A, which evolved the same particle)
      if (GRID && tangent_pt) {
#if GLH_PT_ZPARK
        evolve_noise(rng_mode, a.normals, a.seed, a.step, pt, a.pt_base, k, N, n, false);
        evolve_particle<FAST, GRID, true>(x, m, n, tau, tau2, a.surf, &oob, nullptr, z_parked);
#else
        evolve_noise(rng_mode, a.normals, a.seed, a.step, pt, a.pt_base, k, N, n, true);
        evolve_particle<FAST, GRID, false>(x, m, n, tau, tau2, a.surf, &oob, s_patches.get(), 0.0, wins);
#endif
      } else {
        // (with rasters the tangent models took the branch above, and the other models' step reads no surface: the copy
        // without raster code serves -- the gather of the raster instantiations carries no sampler at all)
        evolve_noise(rng_mode, a.normals, a.seed, a.step, pt, a.pt_base, k, N, n, third);
        evolve_particle<FAST, false>(x, m, n, tau, tau2, a.surf, &oob);
      }
    } else {
      evolve_noise(rng_mode, a.normals, a.seed, a.step, pt, a.pt_base, k, N, n, third);
      evolve_cartesian_m<FAST>(x, m, n, tau, tau2);
    }
  };

  // ---------------- A: evolve, NaN test, project, bounding boxes -------------------------------
  // Observer 0's uv.  PPT > 0: u in PPT registers per thread, v parked in c[i] (LDS, free until phase C).
  // PPT == 0: u parked in c[i] and v in the first N doubles of the observer-0 slot of the uv scratch
  // (L2 / Infinity Cache).
  constexpr int NREG = PPT > 0 ? PPT : 1;
  TmplRegs tmpl0{};
  // (16-bit frames: the template CDF has up to tw x th entries; the tile stage places it itself)
  const bool tmpl_early = live[0] && tmpl_small(hist_n0) && !(SURF && a.obs[0].bits >= 16);  // uniform
  const int rounds = (N + TB - 1) / TB;  // <= PPT when PPT > 0 (the host picks the variant)
  double* V0 = a.uv + (size_t)pt * N * 2;
  double u0[NREG];
  {
    double mn[NOBS][2], mx[NOBS][2];
    bool nanf[NOBS];  // some particle of this thread projects to NaN (a lane mask: no vector registers)
#pragma unroll
    for (int o = 0; o < NOBS; ++o) {
      mn[o][0] = mn[o][1] = INFINITY;
      mx[o][0] = mx[o][1] = -INFINITY;
      nanf[o] = false;
    }
    bool bad = false, raster_oob = false;
    uint32_t view_bits = 0u;
    const double zs = motion_term ? m[17] : 0.0;
    const bool gridded = GRID && motion_term && (m[20] != 0.0 || m[21] != 0.0);  // uniform: this point's surfaces are rasters
    // (kernel arguments the loop tests for every particle, read once: where scalar registers are short the loop re-loaded
    // them, and a scalar load is waited for with the counter the LDS reads share)
    const bool with_viewshed = GRID && a.surf.viewshed.z != nullptr;
    const bool with_term = a.has_dem != 0 && motion_term;
#pragma unroll
    for (int r = 0; r < NREG; ++r) u0[r] = 0.0;
    // software pipeline: the next particle's record is in flight while this one is evolved / projected
    double2 nx0, nx1, nx2;
    if constexpr (COMMON) {
      nx0 = pf0; nx1 = pf1; nx2 = pf2;  // (requested before the prologue's barrier)
    } else {
      const double2* src = Pin2 + (size_t)s_rec[tid < N ? tid : 0] * rec_stride;
      nx0 = src[0]; nx1 = src[chunk_stride]; nx2 = src[2 * chunk_stride];
    }
    PT_STAMP(16);
    auto a_iter = [&](int r) {
      // compiler barrier: camera / motion constants are re-read from LDS (broadcast) every iteration
      // instead of being hoisted into ~100 registers that would spill
      asm volatile("" ::: "memory");
      const int i = r * TB + tid;
      double x[6] = {nx0.x, nx0.y, nx1.x, nx1.y, nx2.x, nx2.y};
      {
        const int inext = i + TB;
        const double2* src = Pin2 + (size_t)s_rec[inext < N ? inext : 0] * rec_stride;
        nx0 = src[0]; nx1 = src[chunk_stride]; nx2 = src[2 * chunk_stride];
      }
      if (i < N) {
        double n[3];
#ifdef GLH_PAD_VALU  // sensitivity probe: this many extra vector instructions per particle (tools/ab.sh pad.so)
        {
          uint32_t padv = (uint32_t)i;
#pragma unroll
          for (int q = 0; q < GLH_PAD_VALU; ++q) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(padv) : "v"(tid));
          asm volatile("" ::"v"(padv));
        }
#endif
        evolve_noise(rng_mode, a.normals, a.seed, a.step, pt, a.pt_base, i, N, n, third);
        if constexpr (SURF)
          evolve_particle<FAST, GRID>(x, m, n, tau, tau2, a.surf, &raster_oob, s_patches.get(), 0.0, wins);
        else
          evolve_cartesian_m<FAST>(x, m, n, tau, tau2);
        if (i == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) s_K[k] = x[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) bad |= isnan(x[k]);
#ifndef GLH_ABLATE_ZP
        if (GLH_PT_ZPARK && GRID && tangent_pt) ZP[i] = x[2];  // (for the gather's re-evolution: evolve_loaded)
#endif
        if (with_term) {
          // CartesianMotion.compute_log_likelihoods (motion.py:181-204) of the evolved particle
          double ll = 0.0;
          if (GRID && gridded) {
            ll = dem_log_likelihood<FAST>(m, a.surf, x[0], x[1], x[2], &raster_oob, s_patches.get(), wins);
          } else if (zs != 0.0) {
            const double d = m[16] - x[2];
            ll = (1.0 / (2.0 * (zs * zs))) * (d * d);
          }
          if constexpr (!RECOMP) W[i] = ll;  // (RECOMP: phase C makes the term again from the re-evolved height)
        }
        if (with_viewshed) view_bits |= viewshed_bits(a.surf, x[0], x[1]);
#pragma unroll
        for (int o = 0; o < NOBS; ++o) {
          if (!live[o]) continue;
          double u, v;
          if constexpr (COMMON)
            project_simple_fast(s_cam[o], x[0], x[1], x[2], u, v);  // (the host sends other cameras to the general instantiation)
          else
            project_m<FAST>(s_cam[o], a.cam_flags[o], x[0], x[1], x[2], u, v);
          if (o == 0) {
            if constexpr (PPT > 0) {
              pt_put<NREG>(u0, r, u);
              c[i] = v;
            } else {
              c[i] = u;
              V0[i] = v;
            }
          } else if constexpr (!RECOMP)
            reinterpret_cast<double2*>(a.uv)[((size_t)o * a.P + pt) * N + i] = make_double2(u, v);
          if (isnan(u) || isnan(v)) {
            nanf[o] = true;
          } else {
            mn[o][0] = min_nn(mn[o][0], u); mx[o][0] = max_nn(mx[o][0], u);  // (u, v are not NaN here)
            mn[o][1] = min_nn(mn[o][1], v); mx[o][1] = max_nn(mx[o][1], v);
          }
        }
      }
    };
    if constexpr (PPT > 0 && NOBS <= 2) {
      // unrolled over the per-thread particles (u0[r] = u with a static index instead of a compare-select chain
      // over the array).  Every instantiation that keeps observer 0's coordinates in registers, the general ones included
      // (round 4: TangentCartesianMotion, uint16 and float frames -2 %; over rasters -2 % with one observer, -6 .. -9 % with
      // two, although the unrolled loop is 120 .. 150 KB of code there)
#pragma unroll
      for (int r = 0; r < NREG; ++r)
        if (r < rounds) a_iter(r);
    } else {
#pragma unroll 1
      for (int r = 0; r < rounds; ++r) a_iter(r);
    }
    PT_STAMP(17);
    if (tmpl_early) tmpl_issue(0, hist_n0, tmpl0);
    if (bad) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_NAN, a.frame);
    if (raster_oob) view_bits |= GLH_PT_RASTER_OOB;
    if (view_bits) flag_point(a.pt_status, a.pt_err_frame, pt, view_bits, a.frame);
#pragma unroll
    for (int o = 0; o < NOBS; ++o) {
      if (!live[o]) continue;
      const double r0 = pt_wave_min63(mn[o][0]), r1 = pt_wave_min63(mn[o][1]);
      const double r2m = pt_wave_max63(mx[o][0]), r3 = pt_wave_max63(mx[o][1]);
      const double r4 = __any(nanf[o]) ? 1.0 : 0.0;
      if (lane == WAVE - 1) {
        double* b = bred[o][wave];
        b[0] = r0; b[1] = r1; b[2] = r2m; b[3] = r3; b[4] = r4;
      }
    }
#if GLH_PT_LDS_BARRIERS_A
    // (round 5) LDS only: __syncthreads() also waits for this wave's outstanding memory operations -- here the template
    // loads issued a few lines up precisely so that they travel WHILE the box is reduced, and the stores of phase A
    pt_lds_barrier();
#else
    __syncthreads();
#endif
    PT_STAMP(18);
    if (tid < NOBS) {
      const int o = tid;
      const size_t slot = (size_t)o * a.P + pt;
      int st;
      if (!live[o]) {
        st = GLH_OBS_SKIPPED;
      } else if (!valid_o) {
        st = GLH_OBS_NO_TEMPLATE;
      } else {
        double mnu = bred[o][0][0], mnv = bred[o][0][1], mxu = bred[o][0][2], mxv = bred[o][0][3],
               nf = bred[o][0][4];
        for (int w = 1; w < PT_WAVES; ++w) {
          mnu = fmin(mnu, bred[o][w][0]); mnv = fmin(mnv, bred[o][w][1]);
          mxu = fmax(mxu, bred[o][w][2]); mxv = fmax(mxv, bred[o][w][3]);
          nf = fmax(nf, bred[o][w][4]);
        }
        st = GLH_OBS_OK;
        s_uvbb[o][0] = mnu; s_uvbb[o][1] = mnv; s_uvbb[o][2] = mxu; s_uvbb[o][3] = mxv;
        const ObsFrame& ob = a.obs[o];
        const int korder = SURF ? a.interp_k : 3;  // (the order also sets the least size of the surface, tracker.py:585-590)
        if (search_box(mnu, mnv, mxu, mxv, nf != 0.0, a.tw, a.th, a.cam[o].imgsz[0], a.cam[o].imgsz[1], s_box[o], korder,
                       korder))
          st = GLH_OBS_OUT_OF_BOUNDS;
        else if (s_box[o][2] > ob.width || s_box[o][3] > ob.height)
          st = GLH_OBS_OUT_OF_BOUNDS;
        else {
          const int w = s_box[o][2] - s_box[o][0], h = s_box[o][3] - s_box[o][1];
          if (w > a.max_dim || h > a.max_dim || (long long)pt_keys_count(w, h) > a.keys_cap ||
              (long long)h * ((w + 14) & ~3) > a.search_cap)
            st = GLH_OBS_TILE_TOO_LARGE;
        }
        if (st == GLH_OBS_OK)
          for (int k = 0; k < 4; ++k) a.box[slot * 4 + k] = s_box[o][k];
      }
      s_status[o] = st;
      a.obs_status[slot] = st;
    }
    PT_STAMP(19);
#if GLH_PT_LDS_BARRIERS_A
    pt_lds_barrier();  // (the box and the status words just stored to memory are for the host: nobody here waits for them)
#else
    __syncthreads();
#endif
  }

  PT_STAMP(1);
#if GLH_PT_PRIO
  {
    // experiment (round 5): a workgroup whose search tiles are large -- the slow point a launch of one round ends with --
    // takes issue priority over its neighbour on the compute unit for the rest of its life
    int area = 0, nok = 0;
#pragma unroll
    for (int o = 0; o < NOBS; ++o)
      if (s_status[o] == GLH_OBS_OK) {
        area += (s_box[o][2] - s_box[o][0]) * (s_box[o][3] - s_box[o][1]);
        ++nok;
      }
    area = __builtin_amdgcn_readfirstlane(area);
    const int base = __builtin_amdgcn_readfirstlane(nok) * (a.tw + 8) * (a.th + 8);
    if (area * 2 > base * 4) __builtin_amdgcn_s_setprio(3);
    else if (area * 2 > base * 3) __builtin_amdgcn_s_setprio(2);
    else if (area * 4 > base * 5) __builtin_amdgcn_s_setprio(1);
  }
#endif
  // ---------------- B + C per observer, in the reference's order (tracker.py:139-146) ----------
  bool outside = false;
  const bool w_here = NOBS == 1 && !a.has_dem;  // uniform: phase C of observer 0 writes weights, not log likelihoods
  bool w_done = false;
  bool c_ready = false;  // uniform: c[] holds log likelihoods (not observer 0's parked coordinates)
  // With two observers the loop is unrolled: observer 0's coordinates live in registers (u) and c[] (v), and in a rolled
  // loop those registers stay live through every iteration -- the two-observer instantiation spilled them (round 3:
  // 144 bytes of scratch); unrolled, they are dead while the second observer's tile pipeline runs.
  constexpr int OBS_UNROLL = NOBS <= 2 ? NOBS : 1;
#pragma unroll OBS_UNROLL
  for (int o = 0; o < NOBS; ++o) {
    if (s_status[o] != GLH_OBS_OK) continue;  // uniform
    const size_t slot = (size_t)o * a.P + pt;
    const ObsFrame& ob = a.obs[o];
    const int* box = s_box[o];
    const int tw = a.tw, th = a.th;
    const int ws_ = box[2] - box[0], hs = box[3] - box[1];
    const int wo = ws_ - tw + 1, ho = hs - th + 1;
    const int nb = ob.channels == 1 ? 256 : 255 * ob.channels + 1;
    const int hist_n = o == 0 ? hist_n0 : a.tmpl_hist_n[slot];
    const int twp = ssd_twp(tw);
    // ---- LDS carve: [T | S | X] with X = max(hist + cum + lut + keys, Z + LU).  The template CDF (cq | cv) lies at the
    //      head of S: it is read while the LUT is made, the search tile is written after that -- 4 KB that decide whether
    //      a gray tile fits, 12 KB for the 766 bins of RGB frames (which used to send every RGB tile to the workspaces).
    TileWs ws;
    const int offT = pt_align16(th * twp * 4), cdfb = pt_align16(hist_n * 8);
    ws.T = reinterpret_cast<float*>(r2);
    const int off = offT;
    const int ld_lds = pt_search_ld(ws_);
    const int s_bytes = max(pt_align16(hs * ld_lds * 4), 2 * cdfb);
    const int hcl = pt_hcl_bytes(nb);
    const int l1 = hcl + pt_align16(pt_keys_count(ws_, hs) * 2);
    // Tracker(interpolation={"kx": 1, "ky": 1}) (general code): the surface values are the coefficients -- no fit -- and
    // the sampling is their bilinear interpolant
    const bool linear = SURF && a.interp_k == 1;          // uniform
    const bool dense = !linear && spline_dense(ho, wo);  // spline fit by explicit inverses
    const int zb = pt_align16(ho * wo * 8);
    // small inverses (one entry per thread) are fetched before the SSD and parked in LDS after it, like the LU
    // factors of the larger surfaces: no memory latency inside the fit
    const int ninv = ho * ho + wo * wo;
    const bool inv_lds = dense && ninv <= GLH_SPL_DENSE_NINV;  // (<= TB: one entry per thread; larger ones stay in memory)
    const int l2 = dense ? 2 * zb + (inv_lds ? pt_align16(ninv * 8) : 0) : zb + pt_align16(5 * (ho + wo) * 8);
    const bool wide = SURF && ob.bits == 16;  // uniform: 16-bit frames (pt_tile_prep_wide), through the workspace branch
    const bool flt = SURF && ob.bits >= 32;   // uniform: float32 / float64 frames (glh_kernels.h: search_tile_from_boxf), likewise
    const bool fits = !wide && !flt && off + s_bytes + (l1 > l2 ? l1 : l2) <= a.r2_bytes;
    const double* hv_g = a.tmpl_hist_v + slot * a.tile_cap;
    const double* hq_g = a.tmpl_hist_q + slot * a.tile_cap;
    // (the offsets of the banded LU factors are uniform loads from memory, waited for where they are made: only the
    // surfaces that are fitted by banded solves fetch them -- made here for all, they were a memory latency at the head
    // of every tile pipeline)
    const double *fh_g = nullptr, *fw_g = nullptr;
    const bool need_lu = !(SURF && a.interp_k == 1) && !spline_dense(ho, wo);  // uniform: a banded fit
    if (need_lu) {
      fh_g = a.lu + a.lu_off[ho];
      fw_g = a.lu + a.lu_off[wo];
    }
    if (o == 0 && tmpl_early)
      tmpl_store(hist_n0, tmpl0);  // (issued at the end of phase A)
    else
      load_template(o, !wide && !flt);
    // ---- C: sample at every particle's uv (observer.py:178-214), scaled by 1/(2 sigma^2); called
    //      once per branch below so that the coefficient loads keep their address space
    auto sample_all = [&](const double* Z, auto cells_tag) {
      constexpr bool CELLS = decltype(cells_tag)::value;  // Z = the per-cell power form, not the coefficients
      // geometry is derived here, not before the tile stages: nothing extra stays live across them
      double sb[4];
      // (one 16-byte load: as two uniform words each was waited for on its own)
      const double2 dv = *reinterpret_cast<const double2*>(a.tmpl_duv + slot * 2);
      const double duv[2] = {dv.x, dv.y};
      sse_box_of(box, duv, tw, th, sb);
      const double cu0 = cell_origin(sb[0], sb[2], wo), cv0 = cell_origin(sb[1], sb[3], ho);
      const double scale = a.inv2s2[o];
      const double2* uvp = reinterpret_cast<const double2*>(a.uv) + slot * N;
      auto eval = [&](double u, double v) -> double {
        if constexpr (CELLS) return spline_eval_cell(Z, ho, wo, cv0, cu0, u, v);
        else {
          if constexpr (SURF)
            if (linear) return spline_eval_linear(Z, wo, ho, wo, cv0, cu0, u, v);
          return spline_eval_poly_m<FAST>(tab, Z, wo, ho, wo, cv0, cu0, u, v);
        }
      };
      // "Some sampling points are outside box" (observer.py:201-202): some particle's uv is outside the box exactly
      // when the bounding box of all of them is (NaNs never get here: they skip the observer) -- one test per
      // point instead of four comparisons per particle
      if (!(s_uvbb[o][0] >= sb[0] && s_uvbb[o][2] <= sb[2] && s_uvbb[o][1] >= sb[1] && s_uvbb[o][3] <= sb[3]))
        outside = true;
      if (o == 0) {
        auto sample_one = [&](double2 q) -> double {
          const double term = eval(q.x, q.y) * scale;
          const double ll = 0.0 + term;  // same rounding as accumulating into a zeroed c[i]
          // a single observer and no motion-model term: this IS the log likelihood, the weight follows at once
          return w_here ? weight_of<FAST>(ll, tab32) : ll;
        };
        if constexpr (PPT > 0) {
          // fully unrolled: u0[r] is a register with a static index (a rolled loop sends the array to scratch:
          // 80 B per thread written and re-read through memory).  The results replace the u's in their registers
          // and are stored after the loop: with no LDS store between them, the table / coefficient loads of one
          // particle can be issued while the previous one is still being summed.
#pragma unroll
          for (int r = 0; r < NREG; ++r) {
            if (r < rounds) {  // uniform
              const int i = r * TB + tid;
              if constexpr (FAST && !CELLS) {
                // (the coefficient form in fast arithmetic is the fallback for surfaces beyond the cell table, the
                // first frames after a wide prior: its results are stored at once -- held back like the others they
                // made this loop the register peak of the whole kernel)
                if (i < N) c[i] = sample_one(make_double2(u0[r], c[i]));
              } else {
                if (i < N) u0[r] = sample_one(make_double2(u0[r], c[i]));
              }
            }
          }
          if constexpr (!(FAST && !CELLS)) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
              if (r < rounds) {
                const int i = r * TB + tid;
                if (i < N) c[i] = u0[r];
              }
            }
          }
        } else {
          // (the next particle's v is on its way from the scratch while this one is sampled)
          double vn = V0[tid < N ? tid : 0];
#pragma unroll 1
          for (int r = 0; r < rounds; ++r) {
            const int i = r * TB + tid;
            const double v = vn;
            vn = V0[i + TB < N ? i + TB : 0];
            if (i < N) c[i] = sample_one(make_double2(c[i], v));
          }
        }
        c_ready = true;
        w_done = w_here;
      } else if constexpr (RECOMP) {
        if (!c_ready) {  // observer 0 was skipped: c[] still holds its parked coordinates
          for (int i = tid; i < N; i += TB) c[i] = 0.0;
          c_ready = true;
        }
        // The particle again: its pre-evolve record (L2 / Infinity-Cache hot: phase A streamed it), the same noise, the
        // same step, the same projection as phase A -- the same bits --, and with the evolved height at hand the motion
        // model's term (tracker.py:143: appended last) and the weight in the same pass.
        const bool mterm = a.has_dem && motion_term;  // uniform
        const double zs = m[17];
        auto rec_of = [&](int i) -> int { return COMMON || uin ? (int)uin[i < N ? i : 0] : (i < N ? i : 0); };
        int rn = rec_of(tid + TB);
        double2 nx0, nx1, nx2;
        {
          const double2* src = Pin2 + (size_t)rec_of(tid) * rec_stride;
          nx0 = src[0]; nx1 = src[chunk_stride]; nx2 = src[2 * chunk_stride];
        }
#pragma unroll 1
        for (int i = tid; i < N; i += TB) {
          asm volatile("" ::: "memory");  // (camera / motion constants from LDS every iteration, as in phase A)
          double x[6] = {nx0.x, nx0.y, nx1.x, nx1.y, nx2.x, nx2.y};
          {
            const double2* src = Pin2 + (size_t)rn * rec_stride;
            nx0 = src[0]; nx1 = src[chunk_stride]; nx2 = src[2 * chunk_stride];
            rn = rec_of(i + 2 * TB);
          }
          double n[3];
          evolve_noise(rng_mode, a.normals, a.seed, a.step, pt, a.pt_base, i, N, n, third);
          evolve_cartesian_m<FAST>(x, m, n, tau, tau2);
          double u, v;
          if constexpr (COMMON)
            project_simple_fast(s_cam[o], x[0], x[1], x[2], u, v);
          else
            project_m<FAST>(s_cam[o], a.cam_flags[o], x[0], x[1], x[2], u, v);
          double ll = c[i];
          ll += eval(u, v) * scale;
          if (mterm) {
            double t = 0.0;
            if (zs != 0.0) {
              const double d = m[16] - x[2];
              t = (1.0 / (2.0 * (zs * zs))) * (d * d);
            }
            ll += t;
          }
          c[i] = weight_of<FAST>(ll, tab32);
        }
        w_done = true;
      } else if constexpr (NOBS > 1) {
        if (!c_ready) {  // observer 0 was skipped: c[] still holds its parked coordinates
          for (int i = tid; i < N; i += TB) c[i] = 0.0;
          c_ready = true;
        }
        // (the next particle's uv is on its way from the scratch while this one is sampled)
        double2 qn = uvp[tid < N ? tid : 0];
        if (o == NOBS - 1) {
          // the last observer: the motion model's term is appended and the weight taken in the same pass (same
          // operations in the same order as the separate loop below, which then has nothing left to do)
          const bool mterm = a.has_dem && motion_term;  // uniform
          double wn = mterm ? W[tid < N ? tid : 0] : 0.0;
#if GLH_PT_PREFETCH2
          // (round 5 experiment: the coordinates and the term of the particle after next are requested as well -- what
          // the scratch returns comes from memory, a microsecond away under load, and one particle's sampling is less)
          double2 qn2 = uvp[tid + TB < N ? tid + TB : 0];
          double wn2 = mterm ? W[tid + TB < N ? tid + TB : 0] : 0.0;
          for (int i = tid; i < N; i += TB) {
            const double2 q = qn;
            const double wi = wn;
            qn = qn2;
            wn = wn2;
            qn2 = uvp[i + 2 * TB < N ? i + 2 * TB : 0];
            if (mterm) wn2 = W[i + 2 * TB < N ? i + 2 * TB : 0];
            double ll = c[i];
            ll += eval(q.x, q.y) * scale;
            if (mterm) ll += wi;  // (tracker.py:143: appended last)
            c[i] = weight_of<FAST>(ll, tab32);
          }
#else
          for (int i = tid; i < N; i += TB) {
            const double2 q = qn;
            const double wi = wn;
            qn = uvp[i + TB < N ? i + TB : 0];
            if (mterm) wn = W[i + TB < N ? i + TB : 0];
            double ll = c[i];
            ll += eval(q.x, q.y) * scale;
            if (mterm) ll += wi;  // (tracker.py:143: appended last)
            c[i] = weight_of<FAST>(ll, tab32);
          }
#endif
          w_done = true;
        } else {
          for (int i = tid; i < N; i += TB) {
            const double2 q = qn;
            qn = uvp[i + TB < N ? i + TB : 0];
            c[i] += eval(q.x, q.y) * scale;
          }
        }
      }
    };
    // Fast arithmetic: the fitted surface goes into per-cell power form (glh_math.h) at the start of region 2 --
    // everything else there is dead once the fit is done -- when its cells fit (a.cell_cap: the same bound the
    // staged kernels apply, so that both evaluate a given surface by the same formula).  Rows are computed into
    // registers first: the table may overlay the coefficients it is made from.
    bool cells = false;  // uniform
    auto to_cells = [&](const double* Z) -> bool {
      const int ncu = spline_cells(wo), nrows = 4 * spline_cells(ho) * ncu;
      if (!FAST || linear || nrows > 4 * a.cell_cap) return false;
      double row[2][4];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = k * TB + tid;
        if (t < nrows) {
          const int cell = t >> 2, qv = cell / ncu;
          spline_cell_row(tab, Z, wo, ho, wo, qv, cell - qv * ncu, t & 3, row[k]);
        }
      }
      __syncthreads();
      double* PC = reinterpret_cast<double*>(r2);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = k * TB + tid;
        if (t < nrows) {
          double2* d = reinterpret_cast<double2*>(PC + (size_t)(t >> 2) * GLH_CELL_LD + 4 * (t & 3));
          d[0] = make_double2(row[k][0], row[k][1]);
          d[1] = make_double2(row[k][2], row[k][3]);
        }
      }
      __syncthreads();
      cells = true;
      return true;
    };
    if (fits) {
      ws.ld = ld_lds;
      ws.S = reinterpret_cast<float*>(r2 + off);
      unsigned char* X = r2 + off + s_bytes;
      ws.hist = reinterpret_cast<uint32_t*>(X);
      ws.lut = reinterpret_cast<double*>(X + pt_align16(nb * 4));
      ws.keys = reinterpret_cast<uint16_t*>(X + hcl);
      ws.cdf_q = reinterpret_cast<const double*>(r2 + offT);
      ws.cdf_v = ws.cdf_q + cdfb / 8;
      pt_tile_prep<TB, SURF>(ob, box, nb, hist_n, ws, scan_tmp, (unsigned long long*)a.stamps, a.hp_rx, a.hp_ry);  // starts with a barrier: T, cdf visible
      PT_STAMP(2);
      ws.Z = reinterpret_cast<double*>(X);
      ws.Z1 = ws.Z + zb / 8;
      double* fl = ws.Z + zb / 8;  // LU factors (larger surfaces only): same place as Z1
      double* invl = ws.Z1 + zb / 8;
      ws.ih = a.inv + spline_inverse_off(ho);
      ws.iw = a.inv + spline_inverse_off(wo);
      // X is reused: the histogram / keys are dead once the search tile is written.  The LU factors are
      // fetched before the SSD and parked in LDS after it, so their memory latency hides behind it.
      const int nfl = need_lu ? 5 * (ho + wo) : 0;
      double fl_v = 0.0;
      double* park = nullptr;
      if (inv_lds) {
        if (tid < ho * ho) fl_v = ws.ih[tid];
        else if (tid < ninv) fl_v = ws.iw[tid - ho * ho];
        if (tid < ninv) park = invl + tid;
      } else if (!need_lu) {
      } else if (nfl <= TB) {
        if (tid < nfl) park = fl + tid;
        if (tid < 5 * ho) fl_v = fh_g[tid];
        else if (tid < nfl) fl_v = fw_g[tid - 5 * ho];
      } else {
        for (int k = tid; k < 5 * ho; k += TB) fl[k] = fh_g[k];
        for (int k = tid; k < 5 * wo; k += TB) fl[5 * ho + k] = fw_g[k];
      }
      ws.fh = fl;
      ws.fw = fl + 5 * ho;
      pt_ssd<TB, NOBS == 2>(ws, tw, th, wo, ho, park, fl_v);
      PT_STAMP(3);
      if (inv_lds) {  // (its own call: the inverses' address space stays known)
        ws.ih = invl;
        ws.iw = invl + ho * ho;
        if (!linear) pt_spline_fit<TB>(ws, wo, ho);
      } else {
        if (!linear) pt_spline_fit<TB>(ws, wo, ho);
      }
      PT_STAMP(4);
      if (!to_cells(ws.Z)) sample_all(ws.Z, std::false_type{});
    } else {
      // big tile: search / keys / surface in the HBM workspaces, histogram + LUT stay in LDS (over the LDS
      // copy of the template CDF: this path reads the CDF from memory)
      unsigned char* X = r2 + offT;
      __syncthreads();  // the CDF copy has landed before the histogram is zeroed over it
      ws.hist = reinterpret_cast<uint32_t*>(X);
      ws.lut = reinterpret_cast<double*>(X + pt_align16(nb * 4));
      ws.ld = (ws_ + 14) & ~3;
      ws.S = a.ws_search + slot * (size_t)a.search_cap;
      ws.keys = a.ws_keys + slot * (size_t)a.keys_cap;
      ws.Z = a.ws_sse + slot * (size_t)a.sse_cap;
      ws.Z1 = reinterpret_cast<double*>(X);  // over the histogram / LUT: dead once the search tile is written
      ws.ih = a.inv + spline_inverse_off(ho);
      ws.iw = a.inv + spline_inverse_off(wo);
      ws.cdf_q = hq_g;
      ws.cdf_v = hv_g;
      ws.fh = fh_g;
      ws.fw = fw_g;
      if (wide) {
        // [T | bucket table | key tile | raw keys + low bits, then the template CDF over them]: whatever of it fits;
        // the rest in the workspaces (the surface's, free until the SSD, takes the raw keys)
        const int npx = ws_ * hs;
        uint32_t* tab = reinterpret_cast<uint32_t*>(X);
        int used = offT + pt_align16(pt_wide_tab_bytes());
        const int kb = pt_align16(pt_keys_count(ws_, hs) * 2);
        if (used + kb <= a.r2_bytes) {
          ws.keys = reinterpret_cast<uint16_t*>(r2 + used);
          used += kb;
        }
        const int rb = pt_align16(npx * 4) + pt_align16(npx), jb = pt_align16((npx + 1) * 2);
        const bool raw_lds = used + rb <= a.r2_bytes;
        uint32_t* raw = raw_lds ? reinterpret_cast<uint32_t*>(r2 + used) : reinterpret_cast<uint32_t*>(ws.Z);
        uint8_t* low = reinterpret_cast<uint8_t*>(raw) + pt_align16(npx * 4);
        double* cdf_l = used + 2 * cdfb <= a.r2_bytes ? reinterpret_cast<double*>(r2 + used) : nullptr;
        // the interval table: behind the CDF copy, else over the raw keys in LDS (no CDF copy there), else in memory
        uint16_t* jt = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(ws.Z) + rb);
        if (cdf_l && used + 2 * cdfb + jb <= a.r2_bytes)
          jt = reinterpret_cast<uint16_t*>(r2 + used + 2 * cdfb);
        else if (!cdf_l && raw_lds)
          jt = reinterpret_cast<uint16_t*>(raw);
        pt_tile_prep_wide<TB>(ob, box, hist_n, ws, tab, raw, low, cdf_l, jt, scan_tmp, a.hp_rx, a.hp_ry,
                              (unsigned long long*)a.stamps);
      } else if (flt) {
        // Float frames (round 4; rounds 2-3: staged kernels only).  What extract_tile needs of np.unique is, per pixel,
        // the number of pixels at or below its NORMALISED value -- and from there on the stage is the 16-bit one: the
        // count is the key (it fits 16 bits: workspaces up to 255 pixels), the packed median network runs on the
        // counts, their matched values come from the interval table.  The tile is normalised in the frame's dtype with
        // NumPy's summation order and ranked by linear buckets exactly as the staged kernels do it (glh_kernels.h:
        // normalize_box_float, rank_values); values and their bucket order in the observer's float workspace.
        // (First form of this round: the whole staged stage run by this workgroup, its median by bisection over the
        // counts in memory -- 0.72 of 0.88 ms per frame at 1 024 points was that median.)
        if constexpr (SURF) {
          const int npx = ws_ * hs, wp = pt_keys_stride(ws_);
          uint32_t* tab = reinterpret_cast<uint32_t*>(X);
          int used = offT + pt_align16(pt_wide_tab_bytes());
          const int kb = pt_align16(pt_keys_count(ws_, hs) * 2);
          if (used + kb <= a.r2_bytes) {
            ws.keys = reinterpret_cast<uint16_t*>(r2 + used);
            used += kb;
          }
          double* fw = ob.fwork + (size_t)pt * ob.fwork_cap;  // 2 npx doubles of memory
          uint16_t* kt = ws.keys + 2 * wp + 2;
          const UDiv by_w = udiv_make(ws_);
          auto emit = [&](int idx, uint32_t cnt) {
            const int r = udiv(by_w, idx), c = idx - r * ws_;
            kt[r * wp + c] = (uint16_t)cnt;  // np.cumsum(counts)[inverse]
          };
          if (ob.bits == 32 && used + 8 * npx <= a.r2_bytes) {
            // a float32 tile that fits: [values | values in bucket order] as floats in LDS, the sums by the whole block
            // (round 4b: 1.29 -> 0.88 ms per frame at C3; the form below -- values as doubles in memory, the sums' trees
            // walked by one thread -- spent 175 k of the stage's 217 k cycles on the sums and the ranking)
            float* g = reinterpret_cast<float*>(r2 + used);
            normalize_box_f32_block<TB>(ob.frame, ob.width, ob.channels, box, g, tab, (unsigned long long*)a.stamps);
            rank_values<TB, PT_WIDE_BUCKETS>(g, g + npx, npx, tab, scan_tmp, emit);
          } else {
            // float64 frames and tiles beyond LDS: [values | values in bucket order] in the workspace; the float32
            // scratch of the normalisation (2 npx floats) over the second half
            float* scratch = reinterpret_cast<float*>(fw + npx);
            normalize_box_float<TB>(ob.frame, ob.width, ob.channels, ob.bits, box, fw, scratch, scratch + npx, wave_tot, nullptr,
                                    tab, PT_WIDE_BUCKETS);  // (the bucket table, not yet in use, lists the sums' leaves)
            rank_values<TB, PT_WIDE_BUCKETS>(fw, fw + npx, npx, tab, scan_tmp, emit);
          }
          PT_STAMP(13);
          double* cdf_l = used + 2 * cdfb <= a.r2_bytes ? reinterpret_cast<double*>(r2 + used) : nullptr;
          const int jb = pt_align16((npx + 1) * 2);
          uint16_t* jt = reinterpret_cast<uint16_t*>(ws.Z);  // (the surface's workspace: free until the SSD)
          if (cdf_l && used + 2 * cdfb + jb <= a.r2_bytes)
            jt = reinterpret_cast<uint16_t*>(r2 + used + 2 * cdfb);
          else if (!cdf_l && used + jb <= a.r2_bytes)
            jt = reinterpret_cast<uint16_t*>(r2 + used);
          pt_counts_finish<TB>(box, hist_n, ws, tab, cdf_l, jt, a.hp_rx, a.hp_ry, (unsigned long long*)a.stamps);
        }
      } else if (offT + hcl + pt_align16(pt_keys_count(ws_, hs) * 2) <= a.r2_bytes) {
        // only the float32 search tile is too large: the key tile stays in LDS (its own call, so that the
        // median's window loads keep their address space)
        ws.keys = reinterpret_cast<uint16_t*>(X + hcl);
        pt_tile_prep<TB, SURF>(ob, box, nb, hist_n, ws, scan_tmp, (unsigned long long*)a.stamps, a.hp_rx, a.hp_ry);
      } else {
        pt_tile_prep<TB, SURF>(ob, box, nb, hist_n, ws, scan_tmp, (unsigned long long*)a.stamps, a.hp_rx, a.hp_ry);
      }
      PT_STAMP(2);
      if (offT + pt_align16(hs * ws.ld * 4) <= a.r2_bytes) {
        // the histogram tables are dead and the search tile alone does fit behind the template: bring it in from
        // the workspace, so that the SSD reads LDS
        float* Sl = reinterpret_cast<float*>(r2 + offT);
        pt_stage<TB>(reinterpret_cast<float4*>(Sl), reinterpret_cast<const float4*>(ws.S),
                     hs * ws.ld / 4);  // ld is a multiple of 4
        __syncthreads();
        ws.S = Sl;
        pt_ssd<TB>(ws, tw, th, wo, ho);
      } else {
        pt_ssd<TB>(ws, tw, th, wo, ho);
      }
      PT_STAMP(3);
      const int lub = pt_align16(5 * (ho + wo) * 8);
      if (zb + (dense ? zb : lub) <= a.r2_bytes) {
        // The template tile and the histogram tables are dead: the SSD surface moves from the HBM workspace into
        // region 2, with the LU factors of a larger surface behind it, so that
        // the fit and the sampling of every particle read LDS.
        double* Zl = reinterpret_cast<double*>(r2);
        double* fl = Zl + zb / 8;
        pt_stage<TB>(Zl, static_cast<const double*>(ws.Z), ho * wo);
        if (need_lu) {
          for (int k = tid; k < 5 * ho; k += TB) fl[k] = fh_g[k];
          for (int k = tid; k < 5 * wo; k += TB) fl[5 * ho + k] = fw_g[k];
          ws.fh = fl;
          ws.fw = fl + 5 * ho;
        }
        __syncthreads();
        ws.Z = Zl;
        ws.Z1 = fl;
        if (!linear) pt_spline_fit<TB>(ws, wo, ho);
        PT_STAMP(4);
        if (!to_cells(Zl)) sample_all(Zl, std::false_type{});
      } else {
        // (the surface stays in its HBM workspace; the scratch of a dense fit -- at most 40 x 40 doubles: fused_plan keeps
        // that much of region 2 -- at the start of region 2, where everything is dead after the SSD)
        if (dense) ws.Z1 = reinterpret_cast<double*>(r2);
        if (!linear) pt_spline_fit<TB>(ws, wo, ho);
        PT_STAMP(4);
        sample_all(ws.Z, std::false_type{});  // (a surface this large is beyond the cell form as well)
      }
    }
    if (cells) sample_all(reinterpret_cast<const double*>(r2), std::true_type{});  // one instance for every branch
    __syncthreads();  // region 2 is free for the next observer
  }
  if (!c_ready) {  // every observer skipped (same-thread indices)
    if (SURF && !motion_term) {
      // ... and the motion model has no term either: update_weights leaves the weights as they are
      // (tracker.py:146-149); they are the input weights, one per input record
      for (int i = tid; i < N; i += TB) c[i] = W[uin ? (int)uin[i] : i];
      w_done = true;
      __syncthreads();
    } else {
      for (int i = tid; i < N; i += TB) c[i] = 0.0;
    }
  }
  PT_STAMP(5);
  if (outside) flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_SAMPLE_OUTSIDE, a.frame);
  if (!w_done) {
    const bool mterm = a.has_dem && motion_term;  // uniform
    if (RECOMP && mterm) {
      // (the last observer was skipped on this frame, so its pass did not append the motion model's term: the evolved
      // height once more, from the record)
      const double zs = m[17];
#pragma unroll 1
      for (int i = tid; i < N; i += TB) {
        const double2* src = Pin2 + (size_t)(COMMON || uin ? (int)uin[i] : i) * rec_stride;
        const double2 v0 = src[0], v1 = src[chunk_stride], v2 = src[2 * chunk_stride];
        double x[6] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y};
        double n[3];
        evolve_noise(rng_mode, a.normals, a.seed, a.step, pt, a.pt_base, i, N, n, third);
        evolve_cartesian_m<FAST>(x, m, n, tau, tau2);
        double ll = c[i], t = 0.0;
        if (zs != 0.0) {
          const double d = m[16] - x[2];
          t = (1.0 / (2.0 * (zs * zs))) * (d * d);
        }
        ll += t;
        c[i] = weight_of<FAST>(ll, tab32);
      }
    } else {
      double wn = mterm ? W[tid < N ? tid : 0] : 0.0;
      for (int i = tid; i < N; i += TB) {
        double ll = c[i];
        const double wi = wn;
        if (mterm) wn = W[i + TB < N ? i + TB : 0];
        if (mterm) ll += wi;  // the motion model's term is appended last (tracker.py:143)
        c[i] = weight_of<FAST>(ll, tab32);  // the weights stay in LDS until the gather of phase E
      }
    }
    __syncthreads();
  }

  PT_STAMP(6);
  // ---------------- D: w.sum() as NumPy's pairwise tree, cumsum(w / total), searchsorted ------
  double* node = reinterpret_cast<double*>(r2);
  double total = 1.0;
  if constexpr (!FAST) {
    {
      const int sub = tid & 7;
      for (int L = tid >> 3; L < a.nleaves; L += TB / 8) {
        const int off = p_leaf_off[L], len = p_leaf_len[L];
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
      for (int k = p_level_off[l] + tid; k < p_level_off[l + 1]; k += TB) {
        const int32_t* op = p_ops + 3 * k;
        node[op[0]] = node[op[1]] + node[op[2]];
      }
      __syncthreads();
    }
    total = node[p_roots[0]];
    for (int r = 1; r < a.nroots; ++r) total += node[p_roots[r]];
  }
  PT_STAMP(10);
  // cumsum(w / total) over contiguous segments, one per thread, WITHOUT storing it: c[] keeps the weights.
  // PPT > 0 (seg <= PPT): the quotients wait in registers for the second pass; otherwise they are recomputed.
  const int seg = (N + TB - 1) / TB;
  const int k0 = min(tid * seg, N), k1 = min(k0 + seg, N);
  double run = 0.0;
  double qn[NREG];
  if constexpr (PPT > 0 && FAST && (NREG & 1) == 0) {
    // whole segments of the full (even) length: aligned 16-byte reads, no guards (adding the 0.0 of an idle thread's
    // registers changes nothing: the sums are of non-negative weights, never -0.0)
    if (seg == NREG && N % seg == 0) {  // uniform
      const double2* c2 = reinterpret_cast<const double2*>(c + (k0 < N ? k0 : 0));
#pragma unroll
      for (int j = 0; j < NREG; j += 2) {
        const double2 q = c2[j >> 1];
        qn[j] = k0 < N ? q.x : 0.0;
        qn[j + 1] = k0 < N ? q.y : 0.0;
        run += qn[j];
        run += qn[j + 1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NREG; ++j) {
        qn[j] = k0 + j < k1 ? c[k0 + j] : 0.0;
        if (k0 + j < k1) run += qn[j];
      }
    }
  } else if constexpr (PPT > 0) {
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
      // (FAST: the raw weights are scanned; the positions are scaled by N / total instead)
      qn[j] = k0 + j < k1 ? (FAST ? c[k0 + j] : c[k0 + j] / total) : 0.0;
      if (k0 + j < k1) run += qn[j];
    }
  } else {
    for (int k = k0; k < k1; ++k) run += FAST ? c[k] : c[k] / total;
  }
  double incl = run, prev;
  if constexpr (FAST) {
    // (the DPP ladder: VALU moves instead of twelve ds_bpermute round trips; its association differs from the
    // shuffle scan's, so the exact arithmetic -- pinned bit for bit by the host-RNG goldens -- keeps that one)
    incl = wave_scan_add_f64(run);
    prev = wave_shr1_f64(incl);
  } else {
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      double t = __shfl_up(incl, off, WAVE);
      if (lane >= off) incl += t;
    }
    prev = __shfl_up(incl, 1, WAVE);
    if (lane == 0) prev = 0.0;
  }
  if (lane == WAVE - 1) wave_tot[wave] = incl;
  __syncthreads();
  double base = 0.0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  const double excl = base + prev;
  PT_STAMP(11);
  // element k's cumulative weight: the running sum of its segment, shifted by the segments before it
  // (thread 0's running sums are used as they are)
  // Region 2 once the tree nodes are dead: N uint16 (rank per output) and N uint32 (source and copies per rank).
  const int n16 = pt_align16(N * 2) / 2;
  uint16_t* ufill = reinterpret_cast<uint16_t*>(r2);  // rank + 1 of the source that serves output j
  uint32_t* usc = reinterpret_cast<uint32_t*>(ufill + n16);  // rank h: its source | copies of it << 16 (one word)
  // clast [TB]: the last cumulative weight of every segment.  Behind the rank tables when region 2 has the room (all but
  // the largest particle counts): the rank table can then be cleared here, a barrier earlier than where clast is dead
  // (round 4: one barrier less); otherwise over the tree nodes, as before.
  const int tables_bytes = pt_align16(n16 * 2 + N * 4);
  const bool clast_behind = tables_bytes + TB * (int)sizeof(double) <= a.r2_bytes;  // uniform
  double* clast = clast_behind ? reinterpret_cast<double*>(r2 + tables_bytes) : node + a.nnodes;
  clast[tid] = tid > 0 ? excl + run : run;
  if (FAST && tid == TB - 1) s_scale = (double)N / (excl + run);  // N / sum of the weights
  if (clast_behind)  // (every thread has read the tree's total before the barrier above)
    for (int q = tid; q < n16 / 8; q += TB) reinterpret_cast<uint4*>(ufill)[q] = make_uint4(0u, 0u, 0u, 0u);  // (n16: whole 16-byte words)
  PT_STAMP(12);
  const double u = s_u;  // the point's resample offset (drawn once, in the prologue)
  const double inv_n = 1.0 / (double)N;
  __shared__ int s_U;
  {
    // f(ck) = #{j : pos_j <= ck}, pos_j = (j + u) * (1 / n) exactly as tracker.py:173 rounds it.  The guess
    // floor(ck * n - u) + 1 is kept in float64 (integer valued, so (double)f == g bit for bit) and
    // VERIFIED with the two exact comparisons around it; only if one fails (an ulp-level boundary) does
    // the general walk run.
    const double dN = (double)N;
    pt_lds_barrier();  // clast (and, FAST, the position scale) are visible
    const double pos_scale = FAST ? s_scale : 0.0;
    auto count_le = [&](double ck) -> int {
      if constexpr (FAST) return count_le_fast(ck, pos_scale, u, N);
      double g = floor(ck * dN - u) + 1.0;
      g = g < 0.0 ? 0.0 : (g > dN ? dN : g);
      const bool below_ok = !(g > 0.0) || ((g - 1.0) + u) * inv_n <= ck;  // position g-1 is counted
      const bool above_ok = !(g < dN) || !((g + u) * inv_n <= ck);        // position g is not
      int f = (int)g;
      if (!(below_ok && above_ok)) {
        while (f < N && ((double)f + u) * inv_n <= ck) ++f;
        while (f > 0 && ((double)(f - 1) + u) * inv_n > ck) --f;
      }
      return f;
    };
    // Source k serves the positions [f(k-1), f(k)): np.searchsorted by its inverse.  The sources that serve at
    // least one position (the survivors) are RANKED in order; output j then only needs the rank of its source
    // (ufill, written at the head of every run and spread by an inclusive max-scan: ranks grow with the position),
    // and the gather below runs over the ranks: one re-evolved record per survivor, however many copies it has.
    const int f_first = k0 > 0 && k0 < N ? count_le(clast[tid - 1]) : 0;
    // pass 1: where every own source's run ends (f), how many own sources survive
    int fk[NREG];
    int nsurv = 0;
    {
      int f_prev = f_first;
      double run2 = 0.0;
      if constexpr (PPT > 0) {
#pragma unroll
        for (int j = 0; j < NREG; ++j) {
          const int k = k0 + j;
          fk[j] = f_prev;
          if (k < k1) {
            run2 += qn[j];
            int f = count_le(tid > 0 ? excl + run2 : run2);
            if (k == N - 1 && f < N) {  // positions beyond c[N-1] (searchsorted == N: IndexError in the reference)
              flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_RESAMPLE_CLAMP, a.frame);
              f = N;                    // clamp: the last source serves them
            }
            nsurv += f > f_prev;
            f_prev = f;
            fk[j] = f;
          }
        }
      } else {
        for (int k = k0; k < k1; ++k) {
          run2 += FAST ? c[k] : c[k] / total;
          int f = count_le(tid > 0 ? excl + run2 : run2);
          if (k == N - 1 && f < N) {
            flag_point(a.pt_status, a.pt_err_frame, pt, GLH_PT_RESAMPLE_CLAMP, a.frame);
            f = N;
          }
          nsurv += f > f_prev;
          f_prev = f;
        }
      }
    }
    // exclusive scan of the survivor counts over the block
    const int incl_s = (int)wave_scan_add_u32((uint32_t)nsurv);
    if (lane == WAVE - 1) scan_tmp[wave] = (uint32_t)incl_s;
    pt_lds_barrier();  // every thread has read clast by now: the tables may overwrite it and the tree nodes
    int rank = incl_s - nsurv;
    for (int w = 0; w < wave; ++w) rank += (int)scan_tmp[w];
    if (tid == TB - 1) s_U = rank + nsurv;
    if (!clast_behind) {  // (the rank table lies over clast: cleared only now)
      for (int q = tid; q < n16 / 8; q += TB) reinterpret_cast<uint4*>(ufill)[q] = make_uint4(0u, 0u, 0u, 0u);
      pt_lds_barrier();
    }
    // pass 2: survivors take their rank
    {
      int f_prev = f_first;
      double run2 = 0.0;
      if constexpr (PPT > 0) {
#pragma unroll
        for (int j = 0; j < NREG; ++j) {
          const int k = k0 + j;
          if (k < k1) {
            const int f = fk[j];
            if (f > f_prev) {
              usc[rank] = (uint32_t)k | ((uint32_t)(f - f_prev) << 16);
              ufill[f_prev] = (uint16_t)(rank + 1);
              ++rank;
            }
            f_prev = f;
          }
        }
      } else {
        for (int k = k0; k < k1; ++k) {
          run2 += FAST ? c[k] : c[k] / total;
          int f = count_le(tid > 0 ? excl + run2 : run2);
          if (k == N - 1 && f < N) f = N;
          if (f > f_prev) {
            usc[rank] = (uint32_t)k | ((uint32_t)(f - f_prev) << 16);
            ufill[f_prev] = (uint16_t)(rank + 1);
            ++rank;
          }
          f_prev = f;
        }
      }
    }
    pt_lds_barrier();
    if (s_U == 0 && tid == 0) {  // no comparison succeeded (NaN weights): every output is a copy of source 0
      usc[0] = (uint32_t)N << 16;  // (thread 0 owns the first segment: it reads this entry of ufill itself)
      ufill[0] = 1;
    }
    // Whole segments of an even length up to 10 (the benched shapes): the segment as 32-bit words, its prefix
    // maxima kept in registers until the maxima of the segments before it are known -- one read and one write per
    // word instead of two reads and two writes per entry.
    constexpr int MS_W = 5;
    const bool words = (seg & 1) == 0 && seg <= 2 * MS_W && N % seg == 0;  // uniform
    uint32_t* segw = reinterpret_cast<uint32_t*>(ufill + k0);            // (k0 is even: 4-byte aligned)
    uint32_t wv[MS_W];
    uint32_t runmax = 0;
    if (words) {
#pragma unroll
      for (int q = 0; q < MS_W; ++q) {
        wv[q] = 0;
        if (2 * q < seg && k0 < N) {
          const uint32_t w = segw[q];
          const uint32_t lo = max(w & 0xffffu, runmax), hi = max(w >> 16, lo);
          runmax = hi;
          wv[q] = lo | (hi << 16);
        }
      }
    } else {
      for (int j = k0; j < k1; ++j) {
        runmax = max(runmax, (uint32_t)ufill[j]);
        ufill[j] = (uint16_t)runmax;
      }
    }
    const uint32_t incl_m = wave_scan_max_u32(runmax);
    // (scan_tmp is reused: every thread consumed the survivor scan before the barrier behind pass 2)
    if (lane == WAVE - 1) scan_tmp[wave] = incl_m;
    uint32_t before = wave_shr1_u32(incl_m);
    pt_lds_barrier();
    for (int w = 0; w < wave; ++w) before = max(before, scan_tmp[w]);
    if (words) {
      const glh_us2 bb = {(unsigned short)before, (unsigned short)before};
#pragma unroll
      for (int q = 0; q < MS_W; ++q)
        if (2 * q < seg && k0 < N)
          segw[q] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(glh_us2, wv[q]), bb));
    } else if (before > 0) {
      for (int j = k0; j < k1; ++j) ufill[j] = (uint16_t)max((uint32_t)ufill[j], before);
    }
  }
  pt_lds_barrier();

  PT_STAMP(7);
  // ---------------- E + F: one re-evolved record per survivor, moments --------------------------
  double* Pout = a.particles_out + (size_t)pt * N * 6;
  double* Wout = a.weights_out + (size_t)pt * N;
  const int U = max(s_U, 1);
  double K[6];  // pivot of the shifted moments: the point's first evolved particle (parked by phase A)
#pragma unroll
  for (int k = 0; k < 6; ++k) K[k] = s_K[k];
  double s0 = 0.0, s1[6] = {0, 0, 0, 0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
#ifndef GLH_PT_GU
#define GLH_PT_GU 2
#endif
  // records in flight per thread.  The general code keeps ONE: with two, its 128 registers spill two record chunks inside this
  // loop (40 B of scratch); one record in flight, none -- TangentCartesianMotion -3.4 %, uint16 frames -3.3 %, over rasters
  // +-0 (profiles/ab_r04/r4j59_ab_gu1.txt).  The plain code has the registers for two (round 3: -1 % against one).
  // The 1 024-thread plain code (C4: one workgroup per CU, nothing else hides its latencies) takes three: -1.5 %, no scratch
  // (r4j71_ab_gu_plain.txt; C3 / C5 at 512 threads: one, two and three within noise).
#ifndef GLH_PT_GU_SURF
#define GLH_PT_GU_SURF 1
#endif
  constexpr int GU = SURF ? GLH_PT_GU_SURF : (TB >= 1024 ? GLH_PT_GU + 1 : GLH_PT_GU);
  // The record index of a survivor, uin[source], is a memory load the record loads depend on: the words of the NEXT
  // iteration (source | copies from the rank table, record index from memory) are fetched while this iteration's
  // records are evolved (C4, whose 16-wave workgroup has its CU to itself: -0.8 %; C3 / C5: -0.2 .. -0.4 %).
  uint32_t sc_n[GU];
  int rec_n[GU];
  double zp_n[GU];  // (tangent models over rasters: the parked height of the NEXT iteration's source -- a load from memory
                    // whose address is known as early as the record index's; fetched with the records it was a second
                    // exposed latency of every iteration of this one-record-in-flight loop)
  auto fetch_next = [&](int h0) {
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      const int h = h0 + g * TB;
      sc_n[g] = h < U ? usc[h] : 0u;
      rec_n[g] = COMMON || uin ? (int)uin[sc_n[g] & 0xffffu] : (int)(sc_n[g] & 0xffffu);
#ifndef GLH_ABLATE_ZP
      zp_n[g] = GLH_PT_ZPARK && GRID && tangent_pt ? ZP[sc_n[g] & 0xffffu] : 0.0;
#else
      zp_n[g] = 0.0;  // (diagnostic build: what the parked heights cost -- wrong results)
#endif
    }
  };
  fetch_next(tid);
  for (int h0 = tid; h0 < U; h0 += GU * TB) {
    int lo[GU], cnt[GU];
    double2 v[GU][3];
    double zp[GU];  // (tangent models: the evolved height phase A parked)
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      lo[g] = (int)(sc_n[g] & 0xffffu);
      cnt[g] = (int)(sc_n[g] >> 16);
      const double2* src = Pin2 + (size_t)rec_n[g] * rec_stride;
      v[g][0] = src[0]; v[g][1] = src[chunk_stride]; v[g][2] = src[2 * chunk_stride];
      zp[g] = zp_n[g];
    }
    fetch_next(h0 + GU * TB);  // (beyond U: slot 0 of the tables, a valid address; never used)
    // the records are evolved, stored and summed ONE AFTER ANOTHER (compiler barrier): only the loads overlap, the
    // Philox / Box-Muller temporaries of one record at a time are live
#pragma unroll
    for (int g = 0; g < GU; ++g) {
      if (g) asm volatile("" ::: "memory");
      double x[6] = {v[g][0].x, v[g][0].y, v[g][1].x, v[g][1].y, v[g][2].x, v[g][2].y};
      evolve_loaded(lo[g], x, zp[g]);
      const double w = c[lo[g]];
      const int h = h0 + g * TB;
      if (h < U) {
        typedef double pt_d2 __attribute__((ext_vector_type(2)));
        pt_d2* dst = reinterpret_cast<pt_d2*>(Pout) + h;  // planar: chunk c of record h at c N + h
        // streaming stores: the new state is not read again before the next launch, and should not displace
        // the pre-evolve records that the other workgroups' gathers are about to re-read from L2 / Infinity Cache
        __builtin_nontemporal_store(pt_d2{x[0], x[1]}, dst);
        __builtin_nontemporal_store(pt_d2{x[2], x[3]}, dst + N);
        __builtin_nontemporal_store(pt_d2{x[4], x[5]}, dst + 2 * N);
        __builtin_nontemporal_store(w, Wout + h);
        const double cw = (double)cnt[g] * w;  // the record stands for cnt identical particles
        s0 += cw;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double d = x[k] - K[k];
          double wd = cw * d;
          if constexpr (FAST) {
            s1[k] = glh_fma(cw, d, s1[k]);
            s2[k] = glh_fma(wd, d, s2[k]);
          } else {
            s1[k] += wd;
            s2[k] += wd * d;
          }
        }
      }
    }
  }
  {
    // which record every output is, and (debug) which source it came from
    uint16_t* uout = a.uidx_out + (size_t)pt * N;
    if (!a.idx_out && (N & 7) == 0) {
      // eight outputs per thread: one 16-byte LDS read, packed 16-bit max / subtract, one 16-byte store
      const glh_us2 one = {1, 1};
      for (int q = tid; q < N / 8; q += TB) {
        uint4 v = reinterpret_cast<const uint4*>(ufill)[q];
        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          w[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(glh_us2, w[k]), one) - one);
        reinterpret_cast<uint4*>(uout)[q] = v;
      }
    } else {
      for (int j = tid; j < N; j += TB) {
        const int h = max((int)ufill[j], 1) - 1;
        uout[j] = (uint16_t)h;
        if (a.idx_out) a.idx_out[(size_t)pt * N + j] = (int32_t)(usc[h] & 0xffffu);
      }
    }
  }
  PT_STAMP(8);
  {
    // The 13 sums over the workgroup.  c[] and region 2 are free (weights gathered, rank tables dead): the partial sums
    // are parked there, [sum][512 slots] (1 024 threads: lane pairs are added first); 16 lanes per sum then add 32 of
    // them each (two chains) and finish on a DPP row -- a quarter of the instructions of 13 wave reductions (which were
    // 6 % of the kernel's vector instructions).  A fixed order: the result depends on nothing but the partial sums.
    // (The launch allocates at least pt_park_bytes() of dynamic LDS: fused_step.)
    pt_lds_barrier();  // LDS only: the gather's stores drain in the background
    constexpr int NP = 512;
    double* M = c;
    if constexpr (TB > NP) {
      static_assert(TB == 2 * NP, "lane pairs");
      s0 = group_sum_dpp(s0, 2);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        s1[k] = group_sum_dpp(s1[k], 2);
        s2[k] = group_sum_dpp(s2[k], 2);
      }
    }
    if (TB == NP || !(tid & 1)) {
      const int slot = TB == NP ? tid : tid >> 1;
      M[slot] = s0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        M[(1 + k) * NP + slot] = s1[k];
        M[(7 + k) * NP + slot] = s2[k];
      }
    }
    pt_lds_barrier();
    double* tot = M + 13 * NP;
    if (tid < 13 * 16) {
      const double* src = M + (tid >> 4) * NP + (tid & 15);
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
      for (int i = 0; i < NP / 16; i += 2) {
        acc0 += src[16 * i];
        acc1 += src[16 * i + 16];
      }
      const double t = group_sum_dpp(acc0 + acc1, 16);
      if ((tid & 15) == 0) tot[tid >> 4] = t;
    }
    pt_lds_barrier();
    double S0 = 0.0, S1 = 0.0, S2 = 0.0;
    if (tid < 6) {
      S0 = tot[0];
      S1 = tot[1 + tid];
      S2 = tot[7 + tid];
    }
    if (tid < 6) {
      const double m1 = S1 / S0, m2 = S2 / S0;
      const double var = m2 - m1 * m1;
      double* out = a.moments + (size_t)pt * 12;
      out[tid] = K[tid] + m1;
      out[6 + tid] = sqrt(var > 0.0 ? var : 0.0);
    }
  }
  PT_STAMP(9);
}

}  // namespace glh
