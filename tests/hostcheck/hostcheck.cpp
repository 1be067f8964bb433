// hostcheck.cpp -- TEST-ONLY harness: compiles the `__host__ __device__` stage math of
// glimpse_amd/csrc (glh_math.h, glh_median.h, glh_host.h) with g++ so that its logic can be
// checked against the oracle on a machine without a GPU.  Never loaded by glimpse_amd; the
// product path has no CPU fallback.
#include <cstring>
#include <vector>

#include "../../glimpse_amd/csrc/glh_host.h"
#include "../../glimpse_amd/csrc/glh_math.h"
#include "../../glimpse_amd/csrc/glh_median.h"

using namespace glh;

extern "C" {

void hc_project(const double* cam, const double* xyz, int n, double* uv) {
  CamDev c;
  expand_camera(cam, &c);
  for (int i = 0; i < n; ++i) project(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], uv[2 * i], uv[2 * i + 1]);
}

int hc_search_box(const double* uv, int n, int tw, int th, double imgw, double imgh, int* box) {
  double mnu = INFINITY, mnv = INFINITY, mxu = -INFINITY, mxv = -INFINITY;
  int nan = 0;
  for (int i = 0; i < n; ++i) {
    double u = uv[2 * i], v = uv[2 * i + 1];
    if (std::isnan(u) || std::isnan(v)) { nan = 1; continue; }
    mnu = fmin(mnu, u); mnv = fmin(mnv, v); mxu = fmax(mxu, u); mxv = fmax(mxv, v);
  }
  return search_box(mnu, mnv, mxu, mxv, nan, tw, th, imgw, imgh, box);
}

int hc_template_box(double u, double v, int tw, int th, double imgw, double imgh, int* box, double* duv) {
  return template_box(u, v, tw, th, imgw, imgh, box, duv);
}

void hc_interp(const double* x, int nx, const double* xp, const double* fp, int n, double* out) {
  for (int i = 0; i < nx; ++i) out[i] = np_interp(x[i], xp, fp, n);
}

void hc_median25(const int* v, int n, int* out) {
  for (int i = 0; i < n; ++i) {
    int w[25];
    memcpy(w, v + 25 * i, sizeof w);
    out[i] = median25(w);
  }
}

int hc_reflect(int i, int n) { return reflect_index(i, n); }

// fit (explicit inverses for small surfaces, two passes of banded solves with spline_lu factors beyond) + evaluate
void hc_spline_sample(const double* z, int ho, int wo, const double* box, const double* uv, int n, double* out) {
  std::vector<double> c(z, z + (size_t)ho * wo), fh(5 * (size_t)ho), fw(5 * (size_t)wo);
  spline_lu(ho, fh.data());
  spline_lu(wo, fw.data());
  auto solve = [](double* x, int stride, int m, const double* f) {
    const double *l1 = f, *l2 = f + m, *u0i = f + 2 * m, *u1 = f + 3 * m, *u2 = f + 4 * m;
    for (int i = 1; i < m; ++i) {
      double y = x[(size_t)i * stride] - l1[i] * x[(size_t)(i - 1) * stride];
      if (i >= 2) y -= l2[i] * x[(size_t)(i - 2) * stride];
      x[(size_t)i * stride] = y;
    }
    for (int i = m - 1; i >= 0; --i) {
      double acc = x[(size_t)i * stride];
      if (i + 1 < m) acc -= u1[i] * x[(size_t)(i + 1) * stride];
      if (i + 2 < m) acc -= u2[i] * x[(size_t)(i + 2) * stride];
      x[(size_t)i * stride] = acc * u0i[i];
    }
  };
  if (spline_dense(ho, wo)) {
    // small surfaces: C = Ih . Z . Iw^T with the explicit inverses (spline_fit_dense of glh_kernels.h)
    std::vector<double> ih((size_t)ho * ho), iw((size_t)wo * wo), z1((size_t)ho * wo);
    spline_inverse(ho, ih.data());
    spline_inverse(wo, iw.data());
    for (int r = 0; r < ho; ++r)
      for (int cc = 0; cc < wo; ++cc) {
        double acc = 0.0;
        for (int k = 0; k < ho; ++k) acc += ih[(size_t)r * ho + k] * c[(size_t)k * wo + cc];
        z1[(size_t)r * wo + cc] = acc;
      }
    for (int r = 0; r < ho; ++r)
      for (int cc = 0; cc < wo; ++cc) {
        double acc = 0.0;
        for (int k = 0; k < wo; ++k) acc += z1[(size_t)r * wo + k] * iw[(size_t)cc * wo + k];
        c[(size_t)r * wo + cc] = acc;
      }
  } else {
    for (int cidx = 0; cidx < wo; ++cidx) solve(c.data() + cidx, wo, ho, fh.data());
    for (int r = 0; r < ho; ++r) solve(c.data() + (size_t)r * wo, 1, wo, fw.data());
  }
  double cu0 = cell_origin(box[0], box[2], wo), cv0 = cell_origin(box[1], box[3], ho);
  for (int i = 0; i < n; ++i) out[i] = spline_eval(c.data(), wo, ho, wo, cv0, cu0, uv[2 * i], uv[2 * i + 1]);
}

// NumPy pairwise sum through the same leaf plan + level-sorted tree the resample kernel runs
double hc_pairwise_sum(const double* w, int n) {
  PairwisePlan pl;
  pairwise_plan(n, pl);
  std::vector<double> node(pl.nnodes);
  for (size_t L = 0; L < pl.leaf_off.size(); ++L) {
    const double* x = w + pl.leaf_off[L];
    int m = pl.leaf_len[L];
    double res;
    if (m < 8) {
      res = 0.0;
      for (int i = 0; i < m; ++i) res += x[i];
    } else {
      double r[8];
      for (int j = 0; j < 8; ++j) r[j] = x[j];
      int body = m - (m & 7);
      for (int i = 8; i < body; i += 8)
        for (int j = 0; j < 8; ++j) r[j] += x[i + j];
      // butterfly order used on the GPU: xor 1, xor 2, xor 4
      double a01 = r[0] + r[1], a23 = r[2] + r[3], a45 = r[4] + r[5], a67 = r[6] + r[7];
      res = (a01 + a23) + (a45 + a67);
      for (int i = body; i < m; ++i) res += x[i];
    }
    node[L] = res;
  }
  int nlevels = (int)pl.level_off.size() - 1;
  for (int l = 0; l < nlevels; ++l)
    for (int k = pl.level_off[l]; k < pl.level_off[l + 1]; ++k)
      node[pl.ops[3 * k]] = node[pl.ops[3 * k + 1]] + node[pl.ops[3 * k + 2]];
  double total = node[pl.roots[0]];
  for (size_t r = 1; r < pl.roots.size(); ++r) total += node[pl.roots[r]];
  return total;
}

// max |poly-table basis - de Boor basis| over all intervals of n-site splines, nsamp points each
double hc_poly_basis_error(int nmax, int nsamp) {
  std::vector<double> tab(16 * GLH_NPOLY);
  basis_poly_table(tab.data());
  double worst = 0.0;
  for (int n = 4; n <= nmax; ++n)
    for (int q = 0; q <= n - 4; ++q) {
      double a = spline_interval_start(q), b = knot_local(q + 4, n);
      for (int k = 0; k <= nsamp; ++k) {
        double xl = a + (b - a) * k / nsamp;
        if (spline_interval(xl, n) != q) continue;  // right end belongs to the next interval
        double h1[4], h2[4];
        spline_basis_local(xl, q, n, h1);
        spline_basis_poly(tab.data(), xl, q, n, h2);
        for (int m = 0; m < 4; ++m) worst = fmax(worst, fabs(h1[m] - h2[m]));
      }
    }
  return worst;
}

void hc_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
  philox4x32_r<10>(c0, c1, c2, c3, k0, k1, out);  // Random123's philox4x32_10 (known-answer vectors)
}
// the round count the device streams use (GLH_PHILOX_ROUNDS), and any other
void hc_philox_rounds(int rounds, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                      unsigned* out) {
  if (rounds == 7) philox4x32_r<7>(c0, c1, c2, c3, k0, k1, out);
  else if (rounds == 10) philox4x32_r<10>(c0, c1, c2, c3, k0, k1, out);
  else philox4x32(c0, c1, c2, c3, k0, k1, out);
}
int hc_philox_device_rounds() { return GLH_PHILOX_ROUNDS; }
// fast-arithmetic restatements against the exact ones (host build: rcp_nr is 1 / x here)
double hc_exp_fast(double x) {
  static double tab[GLH_EXP_TAB];
  static bool init = false;
  if (!init) {
    for (int j = 0; j < GLH_EXP_TAB; ++j) tab[j] = exp2((double)j / GLH_EXP_TAB);
    init = true;
  }
  return weight_of<true>(-x, tab) - 1e-300;
}
// the fast arithmetic's sampling: per-cell power form, as a table (the fused kernel) and converted on the spot (the
// staged kernels); out[0] = coefficient form (exact arithmetic), out[1] = table, out[2] = on the spot
void hc_spline_cell_forms(const double* coef, int ho, int wo, double cv0, double cu0, const double* uv, int n,
                          double* out) {
  std::vector<double> tab(16 * GLH_NPOLY);
  basis_poly_table(tab.data());
  std::vector<double> cells((size_t)spline_cells(ho) * spline_cells(wo) * GLH_CELL_LD, 0.0);
  for (int qv = 0; qv < spline_cells(ho); ++qv)
    for (int qu = 0; qu < spline_cells(wo); ++qu)
      for (int r = 0; r < 4; ++r)
        spline_cell_row(tab.data(), coef, wo, ho, wo, qv, qu, r,
                        cells.data() + (size_t)(qv * spline_cells(wo) + qu) * GLH_CELL_LD + 4 * r);
  for (int i = 0; i < n; ++i) {
    const double u = uv[2 * i], v = uv[2 * i + 1];
    out[3 * i] = spline_eval_poly(tab.data(), coef, wo, ho, wo, cv0, cu0, u, v);
    out[3 * i + 1] = spline_eval_cell(cells.data(), ho, wo, cv0, cu0, u, v);
    out[3 * i + 2] = spline_eval_cell_direct(tab.data(), coef, wo, ho, wo, cv0, cu0, u, v);
  }
}
void hc_project_fast(const double* cam24, const double* xyz, int n, double* uv) {
  CamDev c;
  expand_camera(cam24, &c);
  const unsigned f = cam_flags(c);
  for (int i = 0; i < n; ++i) project_fast(c, f, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], uv[2 * i], uv[2 * i + 1]);
}
// raster_interval (glh_math.h): the interval of every x in the coordinate array g[n]
void hc_raster_interval(const double* g, int n, double lo, double hi, const double* x, int m, int* out) {
  double a, b;  // (lo, hi: the outer limits the guess is made from -- Raster.xlim for cell centres g)
  for (int i = 0; i < m; ++i) {
    out[i] = raster_interval(g, n, x[i], raster_guess(n, x[i], lo, (double)n / (hi - lo)), a, b);
    if (a != g[out[i]] || b != g[out[i] + 1]) out[i] = -1000;
  }
}
// raster_sample with and without a window of the raster around (cx, cy): values [m][2], hits = samples served by the window
int hc_raster_patch(const double* z, int nx, int ny, const double* gx, const double* gy, int sx, int sy, double xmin, double xmax,
                    double ymin, double ymax, double cx, double cy, const double* xy, int m, double* values) {
  const RasterDev r = raster_dev(z, gx, gy, nx, ny, sx, sy, xmin, xmax, ymin, ymax);
  RasterPatch p;
  raster_patch_origin(r, cx, cy, p.i0, p.j0, p.w, p.h);
  for (int j = 0; j < p.h; ++j)
    for (int i = 0; i < p.w; ++i) p.z[j * GLH_PATCH_W + i] = raster_node(r, p.i0 + i, p.j0 + j);
  for (int i = 0; i < p.w; ++i) p.ax[2 * i] = gx[p.i0 + i], p.ax[2 * i + 1] = 0.0;
  for (int j = 0; j < p.h; ++j) p.ay[2 * j] = gy[p.j0 + j], p.ay[2 * j + 1] = 0.0;
  for (int i = 0; i + 1 < p.w; ++i) p.ax[2 * i + 1] = rcp_nr(p.ax[2 * i + 2] - p.ax[2 * i]);
  for (int j = 0; j + 1 < p.h; ++j) p.ay[2 * j + 1] = rcp_nr(p.ay[2 * j + 2] - p.ay[2 * j]);
  p.fkx = r.kx;
  p.fky = r.ky;
  p.full = p.w == GLH_PATCH_W && p.h == GLH_PATCH_W;
  p.pair = p.full;
  // (count the hits by poisoning the raster itself: a sample that still reads it differs)
  int hits = 0;
  std::vector<double> poison((size_t)nx * ny, 1e300);
  RasterDev rp = r;
  rp.z = poison.data();
  for (int i = 0; i < m; ++i) {
    bool o1 = false, o2 = false, o3 = false;
    values[2 * i] = raster_sample(r, xy[2 * i], xy[2 * i + 1], 1, &o1);
    values[2 * i + 1] = raster_sample(r, xy[2 * i], xy[2 * i + 1], 1, &o2, &p);
    const double vp = raster_sample(rp, xy[2 * i], xy[2 * i + 1], 1, &o3, &p);
    if (!o3 && vp == values[2 * i + 1]) ++hits;
    // the pair form (one cell, one set of weights for two rasters on one grid): the same raster twice
    double a = 0.0, b = 0.0;
    if (raster_sample_pair(r, r, true, &p, &p, xy[2 * i], xy[2 * i + 1], a, b) && (a != values[2 * i] || b != values[2 * i]))
      values[2 * i + 1] = -1e300;
  }
  return hits;
}
// glh_math.h: count_fraction(k, n, 1 / n) against the IEEE quotient k / n for every 0 <= k <= n, n_lo <= n <= n_hi: the
// number of pairs that differ
long long hc_count_fraction_exhaustive(int n_lo, int n_hi) {
  long long bad = 0;
  for (int n = n_lo; n <= n_hi; ++n) {
    const double dn = (double)n, rn = 1.0 / dn;
    for (int k = 0; k <= n; ++k)
      if (count_fraction(k, dn, rn) != (double)k / dn) ++bad;
  }
  return bad;
}
// the fast-arithmetic form of the same samples (raster_bilinear_fast): without and with the window, [m][2]; and the pair form
void hc_raster_patch_fast(const double* z, int nx, int ny, const double* gx, const double* gy, int sx, int sy, double xmin,
                          double xmax, double ymin, double ymax, double cx, double cy, const double* xy, int m, double* values) {
  const RasterDev r = raster_dev(z, gx, gy, nx, ny, sx, sy, xmin, xmax, ymin, ymax);
  RasterPatch p;
  raster_patch_origin(r, cx, cy, p.i0, p.j0, p.w, p.h);
  for (int j = 0; j < p.h; ++j)
    for (int i = 0; i < p.w; ++i) p.z[j * GLH_PATCH_W + i] = raster_node(r, p.i0 + i, p.j0 + j);
  for (int i = 0; i < p.w; ++i) p.ax[2 * i] = gx[p.i0 + i], p.ax[2 * i + 1] = 0.0;
  for (int j = 0; j < p.h; ++j) p.ay[2 * j] = gy[p.j0 + j], p.ay[2 * j + 1] = 0.0;
  for (int i = 0; i + 1 < p.w; ++i) p.ax[2 * i + 1] = rcp_nr(p.ax[2 * i + 2] - p.ax[2 * i]);
  for (int j = 0; j + 1 < p.h; ++j) p.ay[2 * j + 1] = rcp_nr(p.ay[2 * j + 2] - p.ay[2 * j]);
  p.fkx = r.kx;
  p.fky = r.ky;
  p.full = p.w == GLH_PATCH_W && p.h == GLH_PATCH_W;
  p.pair = p.full;
  for (int i = 0; i < m; ++i) {
    bool o1 = false, o2 = false;
    values[2 * i] = raster_sample<true>(r, xy[2 * i], xy[2 * i + 1], 1, &o1);
    values[2 * i + 1] = raster_sample<true>(r, xy[2 * i], xy[2 * i + 1], 1, &o2, &p);
    double a = 0.0, b = 0.0;
    if (raster_sample_pair<true>(r, r, true, &p, &p, xy[2 * i], xy[2 * i + 1], a, b) &&
        (a != values[2 * i + 1] || b != values[2 * i + 1]))
      values[2 * i + 1] = -1e300;
  }
}
int hc_raster_uniform(const double* g, int n, double lo, double hi) { return raster_coordinates_uniform(g, n, lo, hi) ? 1 : 0; }
}
