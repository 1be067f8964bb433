// sanitize_main.cpp -- TEST-ONLY: the host-side tables and stage math of glimpse_amd/csrc (glh_host.h, glh_math.h,
// glh_median.h: what the C ABI builds on the host before every launch, and the inline functions shared with the kernels)
// exercised under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the
// pool).  Built and run by tests/test_hostcheck.py; never part of the product.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/glimpse_hip.h"
#include "../../glimpse_amd/csrc/glh_host.h"
#include "../../glimpse_amd/csrc/glh_math.h"
#include "../../glimpse_amd/csrc/glh_median.h"

using namespace glh;

#define REQUIRE(cond)                                                   \
  do {                                                                  \
    if (!(cond)) {                                                      \
      std::fprintf(stderr, "%s:%d: %s failed\n", __FILE__, __LINE__, #cond); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  // pairwise-sum plans for every particle count up to the LDS-resident limit, spot-checked against a direct sum
  for (int n : {1, 7, 8, 100, 127, 128, 129, 2000, 5000, 10000, 14900}) {
    PairwisePlan pl;
    pairwise_plan(n, pl);
    REQUIRE(!pl.leaf_off.empty() && pl.leaf_off.size() == pl.leaf_len.size());
    long total = 0;
    for (size_t i = 0; i < pl.leaf_len.size(); ++i) {
      REQUIRE(pl.leaf_off[i] >= 0 && pl.leaf_len[i] > 0 && pl.leaf_off[i] + pl.leaf_len[i] <= n);
      total += pl.leaf_len[i];
    }
    REQUIRE(total == n);
    REQUIRE(pl.ops.size() % 3 == 0);
    for (int v : pl.ops) REQUIRE(v >= 0 && v < pl.nnodes);
    for (int v : pl.roots) REQUIRE(v >= 0 && v < pl.nnodes);
  }
  // spline tables: LU factors and explicit inverses for every size, inverse * matrix = identity
  for (int n = 4; n <= 64; ++n) {
    std::vector<double> lu(5 * n);
    spline_lu(n, lu.data());
    for (double v : lu) REQUIRE(std::isfinite(v));
  }
  for (int n = 4; n <= GLH_SPL_DENSE_MAX; ++n) {
    std::vector<double> inv((size_t)n * n);
    spline_inverse(n, inv.data());
    for (double v : inv) REQUIRE(std::isfinite(v));
    REQUIRE(spline_inverse_off(n + 1) - spline_inverse_off(n) == (int64_t)n * n);
  }
  std::vector<double> tab(16 * GLH_NPOLY);
  basis_poly_table(tab.data());
  // basis: partition of unity on every interval, both forms
  for (int n = 4; n <= 40; ++n) {
    for (int t = 0; t < 50; ++t) {
      const double xl = U(rng) * (n - 1);
      const int q = spline_interval(xl, n);
      double h1[4], h2[4];
      spline_basis_local(xl, q, n, h1);
      spline_basis_poly(tab.data(), xl, q, n, h2);
      REQUIRE(std::fabs(h1[0] + h1[1] + h1[2] + h1[3] - 1.0) < 1e-12);
      for (int m = 0; m < 4; ++m) REQUIRE(std::fabs(h1[m] - h2[m]) < 1e-12);
    }
  }
  // spline evaluation on random coefficient grids, clamped arguments included
  for (int trial = 0; trial < 200; ++trial) {
    const int ho = 4 + (int)(U(rng) * 30), wo = 4 + (int)(U(rng) * 30);
    std::vector<double> coef((size_t)ho * wo);
    for (double& v : coef) v = U(rng);
    const double u = -3.0 + U(rng) * (wo + 6), v = -3.0 + U(rng) * (ho + 6);
    const double a = spline_eval_poly(tab.data(), coef.data(), wo, ho, wo, 0.0, 0.0, u, v);
    const double b = spline_eval_poly_fast(tab.data(), coef.data(), wo, ho, wo, 0.0, 0.0, u, v);
    const double c = spline_eval(coef.data(), wo, ho, wo, 0.0, 0.0, u, v);
    REQUIRE(std::isfinite(a) && std::fabs(a - b) < 1e-12 && std::fabs(a - c) < 1e-12);
    // the per-cell power form: the table and the on-the-spot conversion are the same number, within rounding of the
    // coefficient form
    std::vector<double> cells((size_t)spline_cells(ho) * spline_cells(wo) * GLH_CELL_LD, 0.0);
    for (int qv = 0; qv < spline_cells(ho); ++qv)
      for (int qu = 0; qu < spline_cells(wo); ++qu)
        for (int r = 0; r < 4; ++r)
          spline_cell_row(tab.data(), coef.data(), wo, ho, wo, qv, qu, r,
                          cells.data() + (size_t)(qv * spline_cells(wo) + qu) * GLH_CELL_LD + 4 * r);
    const double d = spline_eval_cell(cells.data(), ho, wo, 0.0, 0.0, u, v);
    const double e = spline_eval_cell_direct(tab.data(), coef.data(), wo, ho, wo, 0.0, 0.0, u, v);
    REQUIRE(d == e && std::fabs(a - d) < 1e-12);
  }
  // cameras, projection (both arithmetics), search / template boxes, np_interp, reflect, median network
  for (int trial = 0; trial < 500; ++trial) {
    double cam[GLH_CAM_LEN] = {0};
    cam[0] = U(rng) * 10; cam[1] = U(rng) * 10; cam[2] = 90 + U(rng) * 20;
    cam[3] = U(rng) * 40 - 20; cam[4] = -90 + U(rng) * 5; cam[5] = U(rng) * 4 - 2;
    cam[6] = 640; cam[7] = 480; cam[8] = 900; cam[9] = 900;
    for (int k = 0; k < 6; ++k) cam[12 + k] = trial % 3 ? 0.01 * (U(rng) - 0.5) : 0.0;
    cam[18] = trial % 4 ? 0.0 : 0.001; cam[20] = trial % 5 == 0;
    cam[21] = 6.3781e6; cam[22] = 0.13;
    CamDev c;
    expand_camera(cam, &c);
    double u, v, uf, vf;
    project(c, U(rng) * 20 - 10, U(rng) * 20 - 10, U(rng) * 2 - 1, u, v);
    project_fast(c, cam_flags(c), 0.5, -0.5, 0.1, uf, vf);
    project(c, 0.5, -0.5, 0.1, u, v);
    REQUIRE((std::isnan(u) && std::isnan(uf)) || std::fabs(u - uf) < 1e-8);
    double xyz[3];
    unproject(c, cam_flags(c), 320.0, 240.0, 1.0, 1, xyz);
    REQUIRE(std::isfinite(xyz[0]) && std::isfinite(xyz[1]) && std::isfinite(xyz[2]));
    int box[4];
    double duv[2];
    (void)search_box(10 + U(rng) * 600, 10 + U(rng) * 400, 20 + U(rng) * 600, 20 + U(rng) * 440, 0, 31, 31, 640, 480, box);
    (void)template_box(U(rng) * 640, U(rng) * 480, 31, 31, 640, 480, box, duv);
  }
  {
    std::vector<double> xp(50), fp(50);
    for (int i = 0; i < 50; ++i) { xp[i] = i * 0.02; fp[i] = std::sin(i * 0.1); }
    for (int i = 0; i < 200; ++i) REQUIRE(std::isfinite(np_interp(-0.2 + U(rng) * 1.4, xp.data(), fp.data(), 50)));
    for (int n : {1, 2, 5, 9})
      for (int i = -20; i < 20 + n; ++i) { const int r = n > 1 || (i >= 0 && i < 1) ? reflect_index(i, n) : 0; REQUIRE(r >= 0 && r < n); }
    for (int t = 0; t < 2000; ++t) {
      int v[25], w[25];
      for (int i = 0; i < 25; ++i) v[i] = w[i] = (int)(U(rng) * 766);
      const int m = median25(v);
      std::sort(w, w + 25);
      REQUIRE(m == w[12]);
    }
  }
  // Philox + uniform conversions, exp table form
  uint32_t out[4];
  philox4x32(1, 2, 3, 4, 5, 6, out);
  REQUIRE(u01_open(out[0], out[1]) > 0.0 && u01_halfopen(out[2], out[3]) < 1.0);
  double t32[GLH_EXP_TAB];
  for (int j = 0; j < GLH_EXP_TAB; ++j) t32[j] = std::exp2((double)j / GLH_EXP_TAB);
  for (double x : {0.0, 1e-300, 0.3, 17.0, 700.0, 744.0, 745.5, 1e4, 1e9})
    REQUIRE(weight_of<true>(x, t32) >= 1e-300 && weight_of<true>(x, t32) <= 1.0 + 1e-12);
  std::puts("SANITIZE_OK");
  return 0;
}
