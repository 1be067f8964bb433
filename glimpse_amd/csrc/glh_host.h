// glh_host.h -- host-side tables used by the C ABI: camera expansion, not-a-knot spline LU
// factors, NumPy pairwise-sum plan.  Plain C++ (also compiled by tests/hostcheck with g++).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "glh_math.h"

namespace glh {

inline void expand_camera(const double* v, CamDev* c) {
  *c = CamDev{};
  if (v[23] != 0.0) {
    // georeferenced raster image (Grid, raster.py:25-98): [0:2] (xlim[0], ylim[0]), [6:8] size, [8:10] d
    c->is_grid = 1;
    c->xyz[0] = v[0];
    c->xyz[1] = v[1];
    c->imgsz[0] = v[6];
    c->imgsz[1] = v[7];
    c->f[0] = v[8];
    c->f[1] = v[9];
    return;
  }
  const double d2r = M_PI / 180.0;
  // np.deg2rad(viewdir); C = cos, S = sin (camera.py:261-263)
  double C[3], S[3];
  for (int i = 0; i < 3; ++i) {
    double r = v[3 + i] * d2r;
    C[i] = std::cos(r);
    S[i] = std::sin(r);
  }
  for (int i = 0; i < 3; ++i) c->xyz[i] = v[i];
  // camera.py:264-280
  c->R[0] = C[0] * C[2] + S[0] * S[1] * S[2];
  c->R[1] = C[0] * S[1] * S[2] - C[2] * S[0];
  c->R[2] = -C[1] * S[2];
  c->R[3] = C[2] * S[0] * S[1] - C[0] * S[2];
  c->R[4] = S[0] * S[2] + C[0] * C[2] * S[1];
  c->R[5] = -C[1] * C[2];
  c->R[6] = C[1] * S[0];
  c->R[7] = C[0] * C[1];
  c->R[8] = S[1];
  for (int i = 0; i < 2; ++i) {
    c->imgsz[i] = v[6 + i];
    c->f[i] = v[8 + i];
    c->off[i] = v[6 + i] / 2 + v[10 + i];  // imgsz / 2 + c (camera.py:1507)
    c->p[i] = v[18 + i];
  }
  c->any_k = c->any_kden = c->any_p = 0;
  for (int i = 0; i < 6; ++i) {
    c->k[i] = v[12 + i];
    if (v[12 + i] != 0.0) {
      c->any_k = 1;
      if (i >= 3) c->any_kden = 1;
    }
  }
  if (c->p[0] != 0.0 || c->p[1] != 0.0) c->any_p = 1;
  c->has_corr = v[20] != 0.0;
  c->radius = v[21];
  c->refraction = v[22];
}

// Cubic coefficients [m][d] of the 4 non-zero B-splines on interval q of an n-site
// not-a-knot spline, in s = xl - start(q): sampled from the de Boor recursion at 4 points of
// the interval and interpolated (long double Vandermonde solve; the cubics are exact).
inline void basis_poly(int n, int q, double* out16) {
  const double a = spline_interval_start(q);
  const double b = knot_local(q + 4, n);  // interval end
  const long double wdt = (long double)(b - a);
  long double sp[4], V[4][4], rhs[4][4];
  for (int k = 0; k < 4; ++k) {
    sp[k] = wdt * k / 3.0L;
    double h[4];
    spline_basis_local((double)(a + (double)sp[k]), q, n, h);
    // evaluate in long double for the fit: redo the recursion with long double arithmetic
    long double hl[4] = {1, 0, 0, 0}, hh[3];
    const int l = q + 3;
    const long double x = (long double)a + sp[k];
    for (int j = 1; j <= 3; ++j) {
      for (int i = 0; i < j; ++i) hh[i] = hl[i];
      hl[0] = 0;
      for (int i = 0; i < j; ++i) {
        int li = l + i + 1, lj = li - j;
        long double tli = knot_local(li, n), tlj = knot_local(lj, n);
        long double f = hh[i] / (tli - tlj);
        hl[i] = hl[i] + f * (tli - x);
        hl[i + 1] = f * (x - tlj);
      }
    }
    for (int mm = 0; mm < 4; ++mm) rhs[k][mm] = hl[mm];
    long double p = 1;
    for (int d = 0; d < 4; ++d) {
      V[k][d] = p;
      p *= sp[k];
    }
  }
  // Gaussian elimination with partial pivoting on V c = rhs (4 right-hand sides)
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r)
      if (fabsl(V[r][col]) > fabsl(V[piv][col])) piv = r;
    for (int d = 0; d < 4; ++d) std::swap(V[col][d], V[piv][d]);
    for (int mm = 0; mm < 4; ++mm) std::swap(rhs[col][mm], rhs[piv][mm]);
    for (int r = col + 1; r < 4; ++r) {
      long double f = V[r][col] / V[col][col];
      for (int d = col; d < 4; ++d) V[r][d] -= f * V[col][d];
      for (int mm = 0; mm < 4; ++mm) rhs[r][mm] -= f * rhs[col][mm];
    }
  }
  long double cf[4][4];
  for (int mm = 0; mm < 4; ++mm) {
    for (int r = 3; r >= 0; --r) {
      long double acc = rhs[r][mm];
      for (int d = r + 1; d < 4; ++d) acc -= V[r][d] * cf[d][mm];
      cf[r][mm] = acc / V[r][r];
    }
  }
  for (int mm = 0; mm < 4; ++mm)
    for (int d = 0; d < 4; ++d) out16[4 * mm + d] = (double)cf[d][mm];
}

// GLH_NPOLY matrices indexed by spline_poly_index(n, q)
inline void basis_poly_table(double* out) {
  const int ng = 16;  // any n >= 9 gives the generic end/interior matrices
  const int qs[7] = {0, 1, 2, 5, ng - 6, ng - 5, ng - 4};
  for (int k = 0; k < 7; ++k) basis_poly(ng, qs[k], out + 16 * k);
  for (int n = 4; n <= 8; ++n)
    for (int q = 0; q <= n - 4; ++q) basis_poly(n, q, out + 16 * spline_poly_index(n, q));
}

// LU factors (no pivoting) of the not-a-knot collocation matrix of size n, packed as
// l1[n] l2[n] u0inv[n] u1[n] u2[n].  The matrix is diagonally dominant by rows.
inline void spline_lu(int n, double* out) {
  std::vector<double> a((size_t)n * 5, 0.0);  // a[i][d] = A[i][i + d - 2]
  for (int i = 0; i < n; ++i) {
    int q = spline_interval((double)i, n);
    double h[4];
    spline_basis((double)i, q, n, 0.0, h);
    for (int m = 0; m < 4; ++m) {
      int d = q + m - i + 2;
      if (h[m] != 0.0 && d >= 0 && d <= 4) a[(size_t)i * 5 + d] = h[m];
    }
  }
  double *l1 = out, *l2 = out + n, *u0i = out + 2 * n, *u1 = out + 3 * n, *u2 = out + 4 * n;
  for (int i = 0; i < n; ++i) l1[i] = l2[i] = u1[i] = u2[i] = 0.0;
  auto A = [&](int i, int j) -> double& { return a[(size_t)i * 5 + (j - i + 2)]; };
  for (int k = 0; k < n; ++k) {
    for (int i = k + 1; i < std::min(k + 3, n); ++i) {
      double m = A(i, k) / A(k, k);
      (i == k + 1 ? l1[i] : l2[i]) = m;
      for (int j = k; j < std::min(k + 3, n); ++j)
        if (j - i + 2 >= 0 && j - i + 2 <= 4) A(i, j) -= m * A(k, j);
      A(i, k) = 0.0;
    }
  }
  for (int i = 0; i < n; ++i) {
    u0i[i] = 1.0 / A(i, i);
    if (i + 1 < n) u1[i] = A(i, i + 1);
    if (i + 2 < n) u2[i] = A(i, i + 2);
  }
}

// Degree-k collocation matrix (glh_math.h: gspl_*) factored without pivoting (it is totally positive), bandwidth k on
// either side: out = [L: k arrays of n, L_d[i] at (i, i - d)] [u0inv: n] [U: k arrays of n, U_d[i] at (i, i + d)],
// (2 k + 1) n doubles.  Same elimination as oracle/spline.py: lu_general.
inline bool spline_lu_general(int n, int k, double* out) {
  // banded storage, (2 k + 1) columns per row: a(i, j) at band[i * W + (j - i + k)] -- the elimination below only ever
  // touches |i - j| <= k (round 4: this was a dense n x n matrix, O(dim^2) memory per size and O(dim^3) over the sizes
  // of a context, seconds at the 1280 .. 2000 pixel workspaces the Tracker's growth loop can reach)
  const int W = 2 * k + 1;
  std::vector<double> band((size_t)n * W, 0.0);
  auto A = [&](int i, int j) -> double& { return band[(size_t)i * W + (j - i + k)]; };
  for (int i = 0; i < n; ++i) {
    const int l = gspl_interval((double)i, n, k);
    double h[GLH_SPL_KMAX + 1];
    gspl_basis((double)i, l, n, k, h);
    for (int m = 0; m <= k; ++m) {
      const int j = l - k + m;
      // the support of row i lies within k of the diagonal: a basis value outside the band would be dropped from the
      // factorisation without notice -- refused instead (a knot layout that put one there is a bug to be seen)
      if (j - i >= -k && j - i <= k) A(i, j) = h[m];
      else if (h[m] != 0.0) return false;
    }
  }
  double* L = out;
  double* u0inv = out + (size_t)k * n;
  double* U = u0inv + n;
  for (size_t q = 0; q < (size_t)(2 * k + 1) * n; ++q) out[q] = 0.0;
  for (int c = 0; c < n; ++c) {
    for (int i = c + 1; i < std::min(c + k + 1, n); ++i) {
      const double m = A(i, c) / A(c, c);
      L[(size_t)(i - c - 1) * n + i] = m;
      for (int j = c; j < std::min(c + k + 1, n); ++j) A(i, j) -= m * A(c, j);
      A(i, c) = 0.0;
    }
  }
  for (int i = 0; i < n; ++i) {
    u0inv[i] = 1.0 / A(i, i);
    for (int d = 1; d <= k; ++d)
      if (i + d < n) U[(size_t)(d - 1) * n + i] = A(i, i + d);
  }
  return true;
}

// The same collocation matrix inverted explicitly (row-major n x n): for small surfaces the fit is two
// dense products, Ih . Z . Iw^T, in which every coefficient is an independent dot product (the banded solves
// above are two serial chains of n steps per line).  Gauss-Jordan with partial pivoting in long double, rounded
// once to double; the matrix is diagonally dominant by rows (condition number < 10).
inline void spline_inverse(int n, double* out) {
  std::vector<long double> a((size_t)n * n, 0.0L), b((size_t)n * n, 0.0L);
  for (int i = 0; i < n; ++i) {
    int q = spline_interval((double)i, n);
    double h[4];
    spline_basis((double)i, q, n, 0.0, h);
    for (int m = 0; m < 4; ++m)
      if (q + m >= 0 && q + m < n) a[(size_t)i * n + q + m] = (long double)h[m];
    b[(size_t)i * n + i] = 1.0L;
  }
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i)
      if (fabsl(a[(size_t)i * n + k]) > fabsl(a[(size_t)piv * n + k])) piv = i;
    if (piv != k)
      for (int j = 0; j < n; ++j) {
        std::swap(a[(size_t)k * n + j], a[(size_t)piv * n + j]);
        std::swap(b[(size_t)k * n + j], b[(size_t)piv * n + j]);
      }
    const long double d = 1.0L / a[(size_t)k * n + k];
    for (int j = 0; j < n; ++j) {
      a[(size_t)k * n + j] *= d;
      b[(size_t)k * n + j] *= d;
    }
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const long double m = a[(size_t)i * n + k];
      if (m == 0.0L) continue;
      for (int j = 0; j < n; ++j) {
        a[(size_t)i * n + j] -= m * a[(size_t)k * n + j];
        b[(size_t)i * n + j] -= m * b[(size_t)k * n + j];
      }
    }
  }
  for (size_t i = 0; i < (size_t)n * n; ++i) out[i] = (double)b[i];
}
// NumPy's pairwise float sum over n contiguous items (np.add.reduce: 8192-item chunks;
// <= 128-item leaves with 8 interleaved accumulators; split at n/2 rounded down to 8) as a
// tree the resample kernel can evaluate level by level:
//   nodes 0 .. nleaves-1 are the leaves (leaf_off/leaf_len);
//   ops[k] = (dst, a, b): node dst = node a + node b, sorted by level (level_off[l] .. [l+1]);
//   roots = the root node of every 8192-chunk, added left to right.
struct PairwisePlan {
  std::vector<int32_t> leaf_off, leaf_len;
  std::vector<int32_t> ops;        // 3 ints per op
  std::vector<int32_t> level_off;  // nlevels + 1
  std::vector<int32_t> roots;
  int nnodes = 0;
};

inline void pairwise_plan(int n, PairwisePlan& pl) {
  struct Op {
    int dst, a, b, level;
  };
  std::vector<Op> ops;
  std::vector<int> level_of;
  struct Rec {
    // leaves first: number them in order of appearance
    static int leaves(int o, int m, PairwisePlan& pl) {
      if (m <= 128) {
        pl.leaf_off.push_back(o);
        pl.leaf_len.push_back(m);
        return 1;
      }
      int n2 = m / 2;
      n2 -= n2 % 8;
      return leaves(o, n2, pl) + leaves(o + n2, m - n2, pl);
    }
  };
  for (int s = 0; s < n; s += 8192) Rec::leaves(s, std::min(8192, n - s), pl);
  const int nleaves = (int)pl.leaf_off.size();
  level_of.assign(nleaves, 0);
  int next_leaf = 0, next_node = nleaves;
  // second walk in the same order builds the internal nodes
  struct Build {
    static int run(int m, int& next_leaf, int& next_node, std::vector<Op>& ops, std::vector<int>& level_of) {
      if (m <= 128) return next_leaf++;
      int n2 = m / 2;
      n2 -= n2 % 8;
      int a = run(n2, next_leaf, next_node, ops, level_of);
      int b = run(m - n2, next_leaf, next_node, ops, level_of);
      int dst = next_node++;
      int lvl = 1 + std::max(level_of[a], level_of[b]);
      level_of.push_back(lvl);
      ops.push_back({dst, a, b, lvl});
      return dst;
    }
  };
  for (int s = 0; s < n; s += 8192)
    pl.roots.push_back(Build::run(std::min(8192, n - s), next_leaf, next_node, ops, level_of));
  pl.nnodes = next_node;
  int maxl = 0;
  for (auto& o : ops) maxl = std::max(maxl, o.level);
  pl.level_off.assign(maxl + 1, 0);
  std::stable_sort(ops.begin(), ops.end(), [](const Op& x, const Op& y) { return x.level < y.level; });
  int cur = 1;
  for (size_t k = 0; k < ops.size(); ++k) {
    while (cur < ops[k].level) pl.level_off[cur++] = (int)k;
    pl.ops.push_back(ops[k].dst);
    pl.ops.push_back(ops[k].a);
    pl.ops.push_back(ops[k].b);
  }
  while (cur <= maxl) pl.level_off[cur++] = (int)ops.size();
  // level_off[l-1] .. level_off[l] are the ops of level l (l = 1 .. maxl); level_off[0] = 0
  pl.level_off[0] = 0;
}

}  // namespace glh
