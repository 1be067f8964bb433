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

// NumPy's pairwise float sum over n contiguous items as a leaf list + postfix program
// (np.add.reduce: 8192-item chunks; <=128-item leaves; split at n/2 rounded down to 8).
inline void pairwise_plan(int n, std::vector<int32_t>& off, std::vector<int32_t>& len,
                          std::vector<int16_t>& prog) {
  struct Rec {
    static void run(int o, int m, std::vector<int32_t>& off, std::vector<int32_t>& len,
                    std::vector<int16_t>& prog) {
      if (m <= 128) {
        prog.push_back((int16_t)off.size());
        off.push_back(o);
        len.push_back(m);
        return;
      }
      int n2 = m / 2;
      n2 -= n2 % 8;
      run(o, n2, off, len, prog);
      run(o + n2, m - n2, off, len, prog);
      prog.push_back(-1);
    }
  };
  for (int s = 0; s < n; s += 8192) {
    Rec::run(s, std::min(8192, n - s), off, len, prog);
    prog.push_back(-2);
  }
}


}  // namespace glh
