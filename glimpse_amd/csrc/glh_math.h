// glh_math.h -- per-element stage math shared by the HIP kernels.
//
// Everything here is `__host__ __device__` so that tests/hostcheck can compile the very
// same arithmetic with g++ and compare it against the oracle on a machine without a GPU.
// (That harness is test-only; the product never runs this code on the CPU.)
//
// Operation order follows the reference expression by expression (citations relative to
// /root/reference/src/glimpse/), and the library is built with -ffp-contract=off, so the
// float64 results are bit-identical to NumPy wherever NumPy itself is deterministic.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GLH_HD __host__ __device__ __forceinline__
#else
#define GLH_HD inline
#endif

namespace glh {

// Camera expanded on the host at upload (camera.py:101, :239-280).
struct CamDev {
  double xyz[3];
  double R[9];      // Camera.R, row-major
  double f[2];
  double off[2];    // imgsz / 2 + c          (camera.py:1507)
  double imgsz[2];
  double k[6];
  double p[2];
  double radius;    // correction["radius"]     (camera.py:118-121)
  double refraction;
  int32_t has_corr;
  int32_t any_k;    // any(self.k)              (camera.py:1188)
  int32_t any_kden; // any(self.k[3:6])         (camera.py:1152)
  int32_t any_p;    // any(self.p)
  int32_t is_grid;  // georeferenced raster image instead of a camera (Grid.xyz_to_uv, raster.py:423-445):
                    // xyz[0:2] = (xlim[0], ylim[0]), f = d (cell size, signed), imgsz = size
  int32_t pad_;
};

// Which optional terms of the projection are active, as one word: the tests of the reference
// (`if self.correction`, `any(self.k)`, `if self.k[i]`, ... camera.py:1148-1196, :1446) depend on
// the camera only, so a kernel can take them from a scalar register and branch uniformly.
enum : uint32_t {
  CAM_F_CORR = 1u, CAM_F_ANYK = 2u, CAM_F_ANYKDEN = 4u, CAM_F_ANYP = 8u, CAM_F_K0 = 16u,  // K0 << i: k[i] != 0
  CAM_F_GRID = 1024u,
  CAM_F_DIRECTIONS = 2048u  // xyz are ray directions: no camera offset, no elevation correction (camera.py:1448-1449)
};
GLH_HD uint32_t cam_flags(const CamDev& c) {
  uint32_t f = 0;
  if (c.is_grid) return CAM_F_GRID;
  if (c.has_corr) f |= CAM_F_CORR;
  if (c.any_k) f |= CAM_F_ANYK;
  if (c.any_kden) f |= CAM_F_ANYKDEN;
  if (c.any_p) f |= CAM_F_ANYP;
  for (int i = 0; i < 6; ++i)
    if (c.k[i] != 0.0) f |= CAM_F_K0 << i;
  return f;
}

// Camera.xyz_to_uv (camera.py:591-628): _xyz_to_xy (:1435-1470), _distort (:1180-1196 with
// :1138-1163, :1165-1178), _xy_to_uv (:1499-1508).  `f` = cam_flags(c).
GLH_HD void project_f(const CamDev& c, uint32_t f, double x, double y, double z, double& u, double& v,
                      double* depth = nullptr) {  // depth: distance along the optical axis (camera.py:1468-1469)
  if (f & CAM_F_GRID) {  // Grid.xyz_to_uv: (xy - (xlim[0], ylim[0])) / d
    double gx = x - c.xyz[0], gy = y - c.xyz[1];
#if defined(__HIP_DEVICE_COMPILE__)
    // opaque to the optimiser: otherwise both division sequences are speculated above this (uniform) branch
    // and every perspective projection pays for them
    asm volatile("" : "+v"(gx), "+v"(gy));
#endif
    u = gx / c.f[0];
    v = gy / c.f[1];
    return;
  }
  double dx = x, dy = y, dz = z;
  if (!(f & CAM_F_DIRECTIONS)) {
    dx = x - c.xyz[0];
    dy = y - c.xyz[1];
    dz = z - c.xyz[2];
  }
  if ((f & CAM_F_CORR) && !(f & CAM_F_DIRECTIONS)) {
    // helpers.elevation_corrections (helpers.py:1790)
    double sq = dx * dx + dy * dy;
    dz += (c.refraction - 1.0) * sq / (2.0 * c.radius);
  }
  double cx = c.R[0] * dx + c.R[1] * dy + c.R[2] * dz;
  double cy = c.R[3] * dx + c.R[4] * dy + c.R[5] * dz;
  double cz = c.R[6] * dx + c.R[7] * dy + c.R[8] * dz;
  if (depth) *depth = cz;
  if (!(cz > 0.0)) {  // behind the camera (camera.py:1465-1466); NaN depth stays NaN too
    u = v = NAN;
    return;
  }
  double px = cx / cz;
  double py = cy / cz;
  double qx = px, qy = py;
  if (f & (CAM_F_ANYK | CAM_F_ANYP)) {
    double r2 = px * px + py * py;
    if (f & CAM_F_ANYK) {
      // (camera.py:1148-1153 skips a term whose k is zero; adding k * r2^n = +-0.0 instead leaves dr bit for bit
      // the same for every finite r2, and costs less than the three uniform branches)
      double dr = 1.0;
      dr += c.k[0] * r2;
      dr += c.k[1] * r2 * r2;
      dr += c.k[2] * r2 * r2 * r2;
      if (f & CAM_F_ANYKDEN) {
        double t = 1.0;
        if (f & (CAM_F_K0 << 3)) t += c.k[3] * r2;
        if (f & (CAM_F_K0 << 4)) t += c.k[4] * r2 * r2;
        if (f & (CAM_F_K0 << 5)) t += c.k[5] * r2 * r2 * r2;
        dr /= t;
      }
      qx = px * dr;
      qy = py * dr;
    }
    if (f & CAM_F_ANYP) {
      double xty = px * py;
      double dtx = 2.0 * xty * c.p[0] + c.p[1] * (r2 + 2.0 * (px * px));
      double dty = c.p[0] * (r2 + 2.0 * (py * py)) + 2.0 * xty * c.p[1];
      qx += dtx;
      qy += dty;
    }
  }
  u = qx * c.f[0] + c.off[0];
  v = qy * c.f[1] + c.off[1];
}
// ---- "fast" arithmetic (GLH_MATH_FAST; device-RNG runs): the same formulas with fused multiply-adds and
// Newton-refined reciprocals instead of IEEE divisions.  Every operation stays within ~1 ulp of the exact path
// (relative differences of the posteriors ~1e-13), but results are no longer NumPy's bit for bit -- which only the
// host-fed RNG mode (parity with np.random) needs.  The fused and the staged kernels share these functions, so
// they remain bit-identical to each other.
GLH_HD double glh_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fma(a, b, c);
#else
  return fma(a, b, c);
#endif
}
// k / n, correctly rounded -- the quotient an IEEE division gives, bit for bit -- for integers 0 <= k <= n <= 65 536 with
// rn = 1.0 / n (one division per n): q = RN(k rn) is within an ulp, the residual k - q n is exact in one fused multiply-add,
// and the correction rounds to the quotient (Markstein).  All 2.1e9 pairs compared with the division on the host
// (tests/hostcheck: hc_count_fraction_exhaustive); three instructions where the division takes a dozen and a v_rcp_f64.
GLH_HD double count_fraction(int k, double n, double rn) {
  const double dk = (double)k;
  const double q = dk * rn;
  return glh_fma(glh_fma(-q, n, dk), rn, q);
}
GLH_HD double rcp_nr(double x) {  // 1 / x: v_rcp_f64 + two Newton steps (<= 1 ulp for normal x)
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
#else
  return 1.0 / x;
#endif
}
GLH_HD double sqrt_nr(double x) {  // sqrt(x), x >= 0: v_rsq_f64 + Goldschmidt / Newton steps (<= 1 ulp for normal x; 0 -> 0)
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return x > 0.0 ? g : x;  // (0 and NaN come back as they are; rsq(0) = inf would make NaN of them)
#else
  return sqrt(x);
#endif
}
// min / max of two non-NaN doubles as ONE instruction (HIP's fmin / fmax canonicalise both operands first: three)
GLH_HD double min_nn(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a < b ? a : b;
#endif
}
GLH_HD double max_nn(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a > b ? a : b;
#endif
}

// project_f in fast arithmetic (perspective cameras and raster grids alike).
GLH_HD void project_fast(const CamDev& c, uint32_t f, double x, double y, double z, double& u, double& v) {
  if (f & CAM_F_GRID) {
    double gx = x - c.xyz[0], gy = y - c.xyz[1];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(gx), "+v"(gy));
#endif
    u = gx * rcp_nr(c.f[0]);
    v = gy * rcp_nr(c.f[1]);
    return;
  }
  double dx = x, dy = y, dz = z;
  if (!(f & CAM_F_DIRECTIONS)) {
    dx = x - c.xyz[0];
    dy = y - c.xyz[1];
    dz = z - c.xyz[2];
  }
  if ((f & CAM_F_CORR) && !(f & CAM_F_DIRECTIONS))
    dz = glh_fma((c.refraction - 1.0) * rcp_nr(2.0 * c.radius), glh_fma(dx, dx, dy * dy), dz);
  const double cx = glh_fma(c.R[0], dx, glh_fma(c.R[1], dy, c.R[2] * dz));
  const double cy = glh_fma(c.R[3], dx, glh_fma(c.R[4], dy, c.R[5] * dz));
  const double cz = glh_fma(c.R[6], dx, glh_fma(c.R[7], dy, c.R[8] * dz));
  if (!(cz > 0.0)) {
    u = v = NAN;
    return;
  }
  const double inv = rcp_nr(cz);
  const double px = cx * inv, py = cy * inv;
  double qx = px, qy = py;
  if (f & (CAM_F_ANYK | CAM_F_ANYP)) {
    const double r2 = glh_fma(px, px, py * py);
    if (f & CAM_F_ANYK) {
      double dr = glh_fma(r2, glh_fma(r2, glh_fma(r2, c.k[2], c.k[1]), c.k[0]), 1.0);
      if (f & CAM_F_ANYKDEN) dr *= rcp_nr(glh_fma(r2, glh_fma(r2, glh_fma(r2, c.k[5], c.k[4]), c.k[3]), 1.0));
      qx = px * dr;
      qy = py * dr;
    }
    if (f & CAM_F_ANYP) {
      const double xty2 = 2.0 * (px * py);
      qx += glh_fma(xty2, c.p[0], c.p[1] * glh_fma(2.0 * px, px, r2));
      qy += glh_fma(c.p[0], glh_fma(2.0 * py, py, r2), xty2 * c.p[1]);
    }
  }
  u = glh_fma(qx, c.f[0], c.off[0]);
  v = glh_fma(qy, c.f[1], c.off[1]);
}
// project_fast for the cameras most sequences have -- perspective, radial distortion without a denominator (k4..k6 = 0),
// no tangential terms, no elevation correction, world coordinates: the same operations in the same order without the
// uniform branches (a pinhole's k = 0 gives dr = 1 exactly and qx = px * 1).  The fused kernel's common instantiation
// in fast arithmetic compiles this form only; other cameras run on the general instantiation.
constexpr uint32_t CAM_F_NOT_SIMPLE = CAM_F_CORR | CAM_F_ANYKDEN | CAM_F_ANYP | CAM_F_GRID | CAM_F_DIRECTIONS;
GLH_HD void project_simple_fast(const CamDev& c, double x, double y, double z, double& u, double& v) {
  const double dx = x - c.xyz[0], dy = y - c.xyz[1], dz = z - c.xyz[2];
  const double cx = glh_fma(c.R[0], dx, glh_fma(c.R[1], dy, c.R[2] * dz));
  const double cy = glh_fma(c.R[3], dx, glh_fma(c.R[4], dy, c.R[5] * dz));
  const double cz = glh_fma(c.R[6], dx, glh_fma(c.R[7], dy, c.R[8] * dz));
  const double inv = rcp_nr(cz);
  const double px = cx * inv, py = cy * inv;
  const double r2 = glh_fma(px, px, py * py);
  const double dr = glh_fma(r2, glh_fma(r2, glh_fma(r2, c.k[2], c.k[1]), c.k[0]), 1.0);
  const double uu = glh_fma(px * dr, c.f[0], c.off[0]), vv = glh_fma(py * dr, c.f[1], c.off[1]);
  const bool front = cz > 0.0;
  u = front ? uu : NAN;
  v = front ? vv : NAN;
}
template <bool FAST>
GLH_HD void project_m(const CamDev& c, uint32_t f, double x, double y, double z, double& u, double& v) {
  if (FAST)
    project_fast(c, f, x, y, z, u, v);
  else
    project_f(c, f, x, y, z, u, v);
}

GLH_HD void project(const CamDev& c, double x, double y, double z, double& u, double& v) {
  project_f(c, cam_flags(c), x, y, z, u, v);
}

// ---- inverse projection: Camera.uv_to_xyz (camera.py:630-663) --------------------------------
// _radial_distortion / _tangential_distortion (camera.py:1138-1178) of camera coordinates (x, y)
GLH_HD double radial_factor(const CamDev& c, uint32_t f, double r2) {
  double dr = 1.0;
  if (f & (CAM_F_K0 << 0)) dr += c.k[0] * r2;
  if (f & (CAM_F_K0 << 1)) dr += c.k[1] * r2 * r2;
  if (f & (CAM_F_K0 << 2)) dr += c.k[2] * r2 * r2 * r2;
  if (f & CAM_F_ANYKDEN) {
    double t = 1.0;
    if (f & (CAM_F_K0 << 3)) t += c.k[3] * r2;
    if (f & (CAM_F_K0 << 4)) t += c.k[4] * r2 * r2;
    if (f & (CAM_F_K0 << 5)) t += c.k[5] * r2 * r2 * r2;
    dr /= t;
  }
  return dr;
}
GLH_HD void tangential_terms(const CamDev& c, double x, double y, double r2, double& dtx, double& dty) {
  const double xty = x * y;
  dtx = 2.0 * xty * c.p[0] + c.p[1] * (r2 + 2.0 * (x * x));
  dty = c.p[0] * (r2 + 2.0 * (y * y)) + 2.0 * xty * c.p[1];
}

// Camera._undistort (camera.py:1198-1230): identity, the closed-form cubic when only k1 is set
// (_undistort_k1, :1232-1264, Numerical Recipes' cubic roots) or the Oulu fixed point, 20 iterations
// (_undistort_oulu, :1305-1337).
GLH_HD void undistort(const CamDev& c, uint32_t f, double& x, double& y) {
  if (!(f & (CAM_F_ANYK | CAM_F_ANYP))) return;
  const uint32_t kbits = (f / CAM_F_K0) & 63u;
  if (kbits == 1u && !(f & CAM_F_ANYP)) {
    const double k1 = c.k[0];
    const double phi = atan2(y, x);
    const double Q = -1.0 / (3.0 * k1);
    const double R = -x / (2.0 * k1 * cos(phi));
    double r;
    if (R * R < Q * Q * Q) {
      const double th = acos(R * pow(Q, -1.5));
      r = -2.0 * sqrt(Q) * cos((th - 2.0 * M_PI) / 3.0);
    } else {
      const double sgn = R > 0.0 ? 1.0 : (R < 0.0 ? -1.0 : 0.0);
      const double A = -sgn * pow(fabs(R) + sqrt(R * R - Q * Q * Q), 1.0 / 3.0);
      const double B = A != 0.0 ? Q / A : 0.0;
      r = A + B;
    }
    x = cos(phi) * r;
    y = sin(phi) * r;
    return;
  }
  double ux = x, uy = y;
  for (int it = 0; it < 20; ++it) {
    const double r2 = ux * ux + uy * uy;
    if ((f & CAM_F_ANYP) && !(f & CAM_F_ANYK)) {
      double dtx, dty;
      tangential_terms(c, ux, uy, r2, dtx, dty);
      ux = x - dtx;
      uy = y - dty;
    } else {
      // (the reference's `any(k) and not any(k)` branch can never be taken, camera.py:1326)
      double dtx, dty;
      tangential_terms(c, ux, uy, r2, dtx, dty);
      const double inv = 1.0 / radial_factor(c, f, r2);
      ux = (x - dtx) * inv;
      uy = (y - dty) * inv;
    }
  }
  x = ux;
  y = uy;
}

// uv -> ray direction (or world point) at `depth` along the optical axis:
// _uv_to_xy (camera.py:1510-1519) then _xy_to_xyz (:1472-1497).
GLH_HD void unproject(const CamDev& c, uint32_t f, double u, double v, double depth, int directions, double* xyz) {
  double x = (u - c.off[0]) * (1.0 / c.f[0]);
  double y = (v - c.off[1]) * (1.0 / c.f[1]);
  undistort(c, f, x, y);
  for (int k = 0; k < 3; ++k) {
    double w = c.R[0 * 3 + k] * x + c.R[1 * 3 + k] * y;  // R.T[:, 0:2] @ xy
    w += c.R[2 * 3 + k];                                  // + R.T[:, 2]
    if (depth != 1.0) w *= depth;
    if (!directions) w += c.xyz[k];
    xyz[k] = w;
  }
}

// ---- gridded surfaces: Raster.sample at points (raster.py:913-1027) --------------------------
// scipy.interpolate.RegularGridInterpolator over the cell CENTRES (raster.py:891-900), made ascending
// (`x[::sign]`, `array.T[::sign[0], ::sign[1]]`), with `bounds_error=False, fill_value=None`: linear
// extrapolation in the half-cell border, after the reference's own bounds test against the OUTER
// limits (Grid.inbounds_xy, raster.py:313-337).
struct RasterDev {
  const double* z;   // [ny][nx] as Raster.array (row 0 = first y), null = no raster
  const double* gx;  // [nx] ascending cell-centre x (np.linspace, computed by the host like Grid.x)
  const double* gy;  // [ny] ascending cell-centre y
  int32_t nx, ny, sx, sy;  // sx, sy = sign of (dx, dy): which way array columns / rows run
  double xmin, xmax, ymin, ymax;
  double kx, ky;     // cells per unit length, nx / (xmax - xmin): the guess of raster_interval (raster_dev() makes them)
};
GLH_HD RasterDev raster_dev(const double* z, const double* gx, const double* gy, int nx, int ny, int sx, int sy, double xmin,
                            double xmax, double ymin, double ymax) {
  RasterDev r{z, gx, gy, nx, ny, sx, sy, xmin, xmax, ymin, ymax, 0.0, 0.0};
  r.kx = (double)nx / (xmax - xmin);
  r.ky = (double)ny / (ymax - ymin);
  return r;
}

// find_indices (scipy/interpolate/_rgi_cython.pyx): i = clip(searchsorted(g, x) - 1, 0, n - 2), i.e. the i with
// g[i] < x <= g[i + 1], clipped.  The coordinates are cell centres (Grid.x / Grid.y: np.linspace over the outer limits,
// uniform up to rounding -- glh_set_raster refuses coordinates further than a quarter cell from that), so the interval is
// guessed from the cell size (k = n / (max - min), made by the host) and moved by at most one: two loads of g, which the
// sample needs anyway, instead of log2(n) dependent ones (22 memory latencies per sample of a 2 000 x 2 000 DEM).
// With |g[j] - ideal_j| < d / 4 the guess floor((x - min) / d - 1 / 2) is within one of the answer, so one step ends
// where the search would.  ga, gb: g[i], g[i + 1].
GLH_HD int raster_guess(int n, double x, double lo_limit, double k) {
  const double t = (x - lo_limit) * k - 0.5;
  return t > 0.0 ? (t < (double)(n - 2) ? (int)t : n - 2) : 0;  // (NaN: 0; the caller has tested x against the limits)
}
// (g may be a window of the coordinates: g[(j - origin) * STRIDE] is coordinate j, for j = i - 1 .. i + 2 -- all this reads)
template <int STRIDE = 1>
GLH_HD int raster_interval(const double* g, int n, double x, int i, double& ga, double& gb, int origin = 0) {
  ga = g[(i - origin) * STRIDE];
  gb = g[(i + 1 - origin) * STRIDE];
  if (i > 0 && ga >= x) {
    --i;
    gb = ga;
    ga = g[(i - origin) * STRIDE];
  } else if (i < n - 2 && gb < x) {
    ++i;
    ga = gb;
    gb = g[(i + 1 - origin) * STRIDE];
  }
  return i;
}
// (host) the coordinates are those of a uniform grid over [lo_limit, hi_limit] to within a quarter cell
inline bool raster_coordinates_uniform(const double* g, int n, double lo_limit, double hi_limit) {
  const double d = (hi_limit - lo_limit) / (double)n;
  if (!(d > 0.0)) return false;
  for (int j = 0; j < n; ++j)
    if (!(fabs(g[j] - (lo_limit + ((double)j + 0.5) * d)) < 0.25 * d)) return false;
  return true;
}

// A window of a raster held near the samples (the fused kernel: LDS): nodes [i0, i0 + w) x [j0, j0 + h) in INTERVAL order
// (ascending coordinates, whichever way the array runs) and their coordinates.  A tracked point's particles fall into a few
// cells of a DEM; from the window a sample costs LDS reads where it cost two rounds of memory latency.
constexpr int GLH_PATCH_W = 12;
struct RasterPatch {
  double z[GLH_PATCH_W * GLH_PATCH_W];
  // per axis, node k of the window: its coordinate g[k] at [2 k] and rcp_nr(g[k + 1] - g[k]), the reciprocal width of the
  // interval it starts (fast arithmetic), at [2 k + 1] -- one 16-byte LDS read brings both
  double ax[2 * GLH_PATCH_W], ay[2 * GLH_PATCH_W];
  double fkx, fky;       // cells per unit length (the raster's kx, ky): the guess of raster_window_axis
  int32_t i0, j0, w, h;  // w = 0: nothing held
  int32_t full;          // w == h == GLH_PATCH_W: raster_sample_window serves this window
  int32_t pair;          // (window 0 only) ... and the dem_sigma window beside it lies on the same grid at the same origin
};
// What raster_window_axis needs of a window before it touches a node: its first coordinates and the cells per unit length.
// Uniform over the workgroup: the fused kernel reads them once, into scalar registers, instead of from LDS for every
// sample (an LDS read returns 16 bytes to each of 64 lanes whether or not they asked for the same address: a quarter of a
// sample's LDS traffic, and a dependent round trip).
struct RasterWin {
  double x0, kx, y0, ky;
};
// where the window around (x, y) starts, and how many nodes it holds
GLH_HD void raster_patch_origin(const RasterDev& r, double x, double y, int& i0, int& j0, int& w, int& h) {
  w = r.nx < GLH_PATCH_W ? r.nx : GLH_PATCH_W;
  h = r.ny < GLH_PATCH_W ? r.ny : GLH_PATCH_W;
  const bool in = x >= r.xmin && x <= r.xmax && y >= r.ymin && y <= r.ymax;  // (false for NaN: the window starts at 0)
  const int ic = in ? raster_guess(r.nx, x, r.xmin, r.kx) : 0, jc = in ? raster_guess(r.ny, y, r.ymin, r.ky) : 0;
  i0 = ic - (w / 2 - 1);
  j0 = jc - (h / 2 - 1);
  i0 = i0 < 0 ? 0 : (i0 > r.nx - w ? r.nx - w : i0);
  j0 = j0 < 0 ? 0 : (j0 > r.ny - h ? r.ny - h : j0);
}
// the raster's value at node (ix, iy) of the interval order
GLH_HD double raster_node(const RasterDev& r, int ix, int iy) {
  const int col = r.sx > 0 ? ix : r.nx - 1 - ix, row = r.sy > 0 ? iy : r.ny - 1 - iy;
  return r.z[(size_t)row * r.nx + col];
}

// The bilinear interpolant of one cell in fast arithmetic (GLH_MATH_FAST): the weights are (x - xa) x the interval's
// reciprocal width (a Newton reciprocal, rcp_nr: made once per interval when the window is loaded, per sample without a
// window -- the same bits) instead of two IEEE divisions, and the four products are three fused multiply-adds.  Within
// rounding of the exact form below; the fused and the staged kernels share it.
GLH_HD double raster_bilinear_fast(double z00, double z10, double z01, double z11, double tx, double ty) {
  const double a = glh_fma(tx, z10 - z00, z00), b = glh_fma(tx, z11 - z01, z01);
  return glh_fma(ty, b - a, a);
}
// is the interval guess g (and what one step around it reads) inside a window of `w` nodes starting at `o` of an axis of `n`?
GLH_HD bool raster_in_window(int g, int o, int w, int n) {
  const int l = g - o;
  return (l >= 1 || g == 0) && l >= 0 && (l + 2 < w || g == n - 2) && l + 1 < w;
}

// The common sample of the fused kernel in fast arithmetic, from the window ALONE (round 5): nothing of the raster's
// descriptor is read -- the loop that called raster_sample for every particle re-loaded the kernel arguments 23 times per
// particle and waited for each (scalar registers are short there).  One axis: cells from the window's first node,
// f = (x - g[0]) k; for f in [1, W - 2) the interval is floor(f) moved by at most one (the coordinates are within a quarter
// cell of a uniform grid's, so is g[0]: floor(f) is within one of the answer), x lies strictly inside the window's span --
// hence inside the raster's limits, no bounds test -- and the interval is not clipped.  The same interval (g[i] < x <=
// g[i + 1]) and the same weight (x - g[i]) * rcp_nr(g[i + 1] - g[i]) as raster_sample<true> finds: bit for bit its value.
// Three dependent LDS round trips per sample (the window's origin and cell size, the guessed nodes, the cell's corners) and
// one rarely taken branch: the first form of this -- the interval step as nested branches, every load behind its own
// wait -- took six and fifteen exec-mask instructions per axis.
GLH_HD bool raster_window_axis(const double* a, double g0, double k, double x, int& li, double& t) {
  const double f = (x - g0) * k;
  const bool ok = f >= 1.0 && f < (double)(GLH_PATCH_W - 2);  // (NaN: false)
  int i = (int)fmin(fmax(f, 1.0), (double)(GLH_PATCH_W - 3));  // in [1, W - 3] whatever f is: the reads below stay inside
  double ga = a[2 * i], r = a[2 * i + 1];
  const double gb = a[2 * i + 2];
  // (both comparisons always: as a short-circuit the second node's read sat behind a branch on the first, a round trip more)
  if ((int)!(ga < x) | (int)!(x <= gb)) {
    // the guess is one off: x within rounding of a node, or coordinates that are not exactly a uniform grid's (rare)
    i += ga >= x ? -1 : 1;  // in [0, W - 2]
    ga = a[2 * i];
    r = a[2 * i + 1];
  }
  li = i;
  t = (x - ga) * r;
  return ok;
}
// p1: the window of a second raster on the same grid at the same origin (PAIR), sampled with the same cell and weights
template <bool PAIR>
GLH_HD bool raster_sample_window(const RasterPatch* p0, const RasterPatch* p1, double x, double y, double& v0, double& v1,
                                 const RasterWin* win = nullptr) {
#ifdef GLH_ABLATE_SAMPLE  // (diagnostic build: what the samples cost where they stand -- the value is not the raster's)
  v0 = x * 1e-9 + y * 1e-9;
  if constexpr (PAIR) v1 = 0.3 + x * 1e-12;
  return true;
#endif
  int li, lj;
  double tx, ty;
  const bool okx = raster_window_axis(p0->ax, win ? win->x0 : p0->ax[0], win ? win->kx : p0->fkx, x, li, tx);
  const bool oky = raster_window_axis(p0->ay, win ? win->y0 : p0->ay[0], win ? win->ky : p0->fky, y, lj, ty);
  const double* z = p0->z + lj * GLH_PATCH_W + li;
  v0 = raster_bilinear_fast(z[0], z[1], z[GLH_PATCH_W], z[GLH_PATCH_W + 1], tx, ty);
  if constexpr (PAIR) {
    const double* zz = p1->z + lj * GLH_PATCH_W + li;
    v1 = raster_bilinear_fast(zz[0], zz[1], zz[GLH_PATCH_W], zz[GLH_PATCH_W + 1], tx, ty);
  }
  return okx & oky;  // (false: the values are of a clamped cell -- the caller samples the raster itself)
}

// ONE raster at TWO points through its window (the tangent models' step samples the surface under the particle before and
// after the move, a fraction of a cell apart): the second point usually lies in the first one's cell -- g[i] < x <= g[i + 1]
// on both axes, the very test that defines the interval -- and then takes its corners, coordinates and reciprocal widths:
// no guess, no node reads.  Otherwise it is sampled on its own.  The same values as two calls of raster_sample_window.
GLH_HD bool raster_sample_window2(const RasterPatch* p, double xa, double ya, double xb, double yb, double& va, double& vb,
                                  const RasterWin* win = nullptr) {
  int li, lj;
  double tx, ty;
  const bool okx = raster_window_axis(p->ax, win ? win->x0 : p->ax[0], win ? win->kx : p->fkx, xa, li, tx);
  const bool oky = raster_window_axis(p->ay, win ? win->y0 : p->ay[0], win ? win->ky : p->fky, ya, lj, ty);
  const double* z = p->z + lj * GLH_PATCH_W + li;
  const double z00 = z[0], z10 = z[1], z01 = z[GLH_PATCH_W], z11 = z[GLH_PATCH_W + 1];
  va = raster_bilinear_fast(z00, z10, z01, z11, tx, ty);
  if (!(okx & oky)) return false;
  // (the cell's own coordinates again: (x - g) * r and t agree only to rounding, so the test is made on g itself)
  const double gxa = p->ax[2 * li], rxa = p->ax[2 * li + 1], gxb = p->ax[2 * li + 2];
  const double gya = p->ay[2 * lj], rya = p->ay[2 * lj + 1], gyb = p->ay[2 * lj + 2];
  if ((int)(gxa < xb) & (int)(xb <= gxb) & (int)(gya < yb) & (int)(yb <= gyb)) {
    vb = raster_bilinear_fast(z00, z10, z01, z11, (xb - gxa) * rxa, (yb - gya) * rya);
    return true;
  }
  double unused;
  return raster_sample_window<false>(p, p, xb, yb, vb, unused, win);
}

// order 1: bilinear (RegularGridInterpolator method 'linear'); order 0: 'nearest'.  Sets *oob when the
// point is outside the raster's outer limits (the reference raises ValueError there).  `patch`: a window of THIS raster
// (or null); a sample whose interval and its neighbours lie inside is read from it -- same values, same arithmetic.
// FAST: raster_bilinear_fast for order 1 (order 0 has one form).
template <bool FAST = false>
GLH_HD double raster_sample(const RasterDev& r, double x, double y, int order, bool* oob, const RasterPatch* patch = nullptr,
                            const RasterWin* win = nullptr) {  // win: the window's origin and cell size in registers, or null
  if constexpr (FAST) {
    if (patch && order == 1 && patch->full) {  // (uniform)
      double v, unused;
      if (raster_sample_window<false>(patch, patch, x, y, v, unused, win)) return v;
    }
#ifdef GLH_ABLATE_COLD  // (diagnostic build: no general code behind the window -- what its inlined copies cost the hot path)
    if (patch) {
      *oob = true;
      return NAN;
    }
#endif
  }
  // (Tried in round 5: this general code as a real function called from the particle loops -- 20 .. 30 inlined copies of it
  // make the raster instantiations 370 KB -- does not compile: a call inside divergent control flow ends in the backend's
  // "illegal VGPR to SGPR copy", also when the whole wave makes the call.)
  if (!(x >= r.xmin && x <= r.xmax && y >= r.ymin && y <= r.ymax)) {
    *oob = true;
    return NAN;
  }
  double xa, xb, ya, yb;
  const int gi = raster_guess(r.nx, x, r.xmin, r.kx), gj = raster_guess(r.ny, y, r.ymin, r.ky);
  if (patch && order == 1) {
    // the step of raster_interval reads nodes guess - 1 .. guess + 2 (clipped to the raster): all inside the window?
    if (raster_in_window(gi, patch->i0, patch->w, r.nx) && raster_in_window(gj, patch->j0, patch->h, r.ny)) {
      const int i0 = raster_interval<2>(patch->ax, r.nx, x, gi, xa, xb, patch->i0) - patch->i0;
      const int i1 = raster_interval<2>(patch->ay, r.ny, y, gj, ya, yb, patch->j0) - patch->j0;
      const double* z = patch->z + i1 * GLH_PATCH_W + i0;
      if constexpr (FAST)
        return raster_bilinear_fast(z[0], z[1], z[GLH_PATCH_W], z[GLH_PATCH_W + 1], (x - xa) * patch->ax[2 * i0 + 1],
                                    (y - ya) * patch->ay[2 * i1 + 1]);
      const double y0 = (x - xa) / (xb - xa);
      const double y1 = (y - ya) / (yb - ya);
      return z[0] * (1.0 - y0) * (1.0 - y1) + z[GLH_PATCH_W] * (1.0 - y0) * y1 + z[1] * y0 * (1.0 - y1) +
             z[GLH_PATCH_W + 1] * y0 * y1;
    }
  }
  const int i0 = raster_interval(r.gx, r.nx, x, gi, xa, xb), i1 = raster_interval(r.gy, r.ny, y, gj, ya, yb);
  if constexpr (FAST)
    if (order == 1)
      return raster_bilinear_fast(raster_node(r, i0, i1), raster_node(r, i0 + 1, i1), raster_node(r, i0, i1 + 1),
                                  raster_node(r, i0 + 1, i1 + 1), (x - xa) * rcp_nr(xb - xa), (y - ya) * rcp_nr(yb - ya));
  const double y0 = (x - xa) / (xb - xa);
  const double y1 = (y - ya) / (yb - ya);
  if (order == 0) return raster_node(r, y0 <= 0.5 ? i0 : i0 + 1, y1 <= 0.5 ? i1 : i1 + 1);
  return raster_node(r, i0, i1) * (1.0 - y0) * (1.0 - y1) + raster_node(r, i0, i1 + 1) * (1.0 - y0) * y1 +
         raster_node(r, i0 + 1, i1) * y0 * (1.0 - y1) + raster_node(r, i0 + 1, i1 + 1) * y0 * y1;
}

// Two rasters on ONE grid (a DEM and its uncertainty usually are) sampled at one point through their windows: the cell and
// the two weights are found once.  `same_grid`: the host found the two rasters' coordinate arrays bit-identical
// (glh_set_raster; equal limits alone would not say that).  Returns false (nothing done) unless both samples can be served
// from the windows -- the caller then samples each raster on its own.  Same values, same arithmetic as raster_sample.
template <bool FAST = false>
GLH_HD bool raster_sample_pair(const RasterDev& r0, const RasterDev& r1, bool same_grid, const RasterPatch* p0,
                               const RasterPatch* p1, double x, double y, double& v0, double& v1,
                               const RasterWin* win = nullptr) {
  if constexpr (FAST) {
    // (pair: uniform.  What the windows do not serve goes to the two single samples -- the same values: this function's
    // general form below is the exact arithmetic's)
    return p0->pair && raster_sample_window<true>(p0, p1, x, y, v0, v1, win);
  }
  const bool same = same_grid && r0.nx == r1.nx && r0.ny == r1.ny && r0.xmin == r1.xmin && r0.xmax == r1.xmax &&
                    r0.ymin == r1.ymin && r0.ymax == r1.ymax && p0->i0 == p1->i0 && p0->j0 == p1->j0 && p0->w == p1->w &&
                    p0->h == p1->h;  // (uniform)
  if (!same) return false;
  if (!(x >= r0.xmin && x <= r0.xmax && y >= r0.ymin && y <= r0.ymax)) return false;
  const int gi = raster_guess(r0.nx, x, r0.xmin, r0.kx), gj = raster_guess(r0.ny, y, r0.ymin, r0.ky);
  if (!(raster_in_window(gi, p0->i0, p0->w, r0.nx) && raster_in_window(gj, p0->j0, p0->h, r0.ny))) return false;
  double xa, xb, ya, yb;
  const int i0 = raster_interval<2>(p0->ax, r0.nx, x, gi, xa, xb, p0->i0) - p0->i0;
  const int i1 = raster_interval<2>(p0->ay, r0.ny, y, gj, ya, yb, p0->j0) - p0->j0;
  const double* z = p0->z + i1 * GLH_PATCH_W + i0;
  const double* zz = p1->z + i1 * GLH_PATCH_W + i0;
  if constexpr (FAST) {
    const double tx = (x - xa) * p0->ax[2 * i0 + 1], ty = (y - ya) * p0->ay[2 * i1 + 1];
    v0 = raster_bilinear_fast(z[0], z[1], z[GLH_PATCH_W], z[GLH_PATCH_W + 1], tx, ty);
    v1 = raster_bilinear_fast(zz[0], zz[1], zz[GLH_PATCH_W], zz[GLH_PATCH_W + 1], tx, ty);
    return true;
  }
  const double y0 = (x - xa) / (xb - xa);
  const double y1 = (y - ya) / (yb - ya);
  v0 = z[0] * (1.0 - y0) * (1.0 - y1) + z[GLH_PATCH_W] * (1.0 - y0) * y1 + z[1] * y0 * (1.0 - y1) + z[GLH_PATCH_W + 1] * y0 * y1;
  v1 = zz[0] * (1.0 - y0) * (1.0 - y1) + zz[GLH_PATCH_W] * (1.0 - y0) * y1 + zz[1] * y0 * (1.0 - y1) + zz[GLH_PATCH_W + 1] * y0 * y1;
  return true;
}

// The surface height / its sigma under (x, y) for one point: the context's raster when the point's
// flag says so (m[20] dem, m[21] dem_sigma), else the constant m[16] / m[17] (an infinite 1 x 1
// raster in the reference, motion.py:136-141, raster.py:1021-1026).
struct Surfaces {
  RasterDev dem, dem_sigma, viewshed;
  int32_t same_grid;  // the dem and dem_sigma coordinate arrays are bit-identical (the host compared them: glh_set_raster)
  int32_t pad_;
};
// (patches: windows of the dem [0] and the dem_sigma [1] raster, or null)
// (wins: the two windows' RasterWin, or null)
template <bool FAST = false>
GLH_HD double dem_at(const double* m, const Surfaces& s, double x, double y, bool* oob, const RasterPatch* patches = nullptr,
                     const RasterWin* wins = nullptr) {
  return m[20] != 0.0 ? raster_sample<FAST>(s.dem, x, y, 1, oob, patches, wins) : m[16];
}
template <bool FAST = false>
GLH_HD double dem_sigma_at(const double* m, const Surfaces& s, double x, double y, bool* oob,
                           const RasterPatch* patches = nullptr, const RasterWin* wins = nullptr) {
  return m[21] != 0.0 ? raster_sample<FAST>(s.dem_sigma, x, y, 1, oob, patches ? patches + 1 : nullptr,
                                            wins ? wins + 1 : nullptr)
                      : m[17];
}

// Search box (tracker.py:580-603).  Returns 0 and fills box (l,t,r,b) when the box is
// inside the image (Camera.inframe, camera.py:700-718), 1 otherwise.  `kcols` / `krows`: the interpolation orders
// that set the least surface size (tracker.py:585-590: "ky" widens the columns, "kx" the rows); 3 unless
// Tracker(interpolation=...) says otherwise.
GLH_HD int search_box(double minu, double minv, double maxu, double maxv, int has_nan, int tw,
                      int th, double imgw, double imgh, int* box, int kcols = 3, int krows = 3) {
  if (has_nan) return 1;  // NaN min/max -> garbage ints -> out of bounds (tracker.py:597)
  double lo_u = minu - tw * 0.5, hi_u = maxu + tw * 0.5;
  double lo_v = minv - th * 0.5, hi_v = maxv + th * 0.5;
  double ncols = (double)kcols - ((hi_u - lo_u) - tw);
  if (ncols > 0.0) {
    lo_u += -ncols * 0.5;
    hi_u += ncols * 0.5;
  }
  double nrows = (double)krows - ((hi_v - lo_v) - th);
  if (nrows > 0.0) {
    lo_v += -nrows * 0.5;
    hi_v += nrows * 0.5;
  }
  double l = floor(lo_u), t = floor(lo_v), r = ceil(hi_u), b = ceil(hi_v);
  if (!(l >= 0.0 && t >= 0.0 && r <= imgw && b <= imgh && l <= imgw && t <= imgh && r >= 0.0 &&
        b >= 0.0))
    return 1;
  box[0] = (int)l;
  box[1] = (int)t;
  box[2] = (int)r;
  box[3] = (int)b;
  return 0;
}

// Grid.snap_box on an image grid (raster.py:414-421, :372-388) + duv (tracker.py:556).
// Returns 0 when the box is inside the image, 1 otherwise (IndexError in the reference).
GLH_HD int template_box(double u, double v, int tw, int th, double imgw, double imgh, int* box,
                        double* duv) {
  double x0 = u - tw * 0.5, x1 = u + tw * 0.5;
  double y0 = v - th * 0.5, y1 = v + th * 0.5;
  if (!(x0 >= 0.0 && x0 <= imgw && x1 >= 0.0 && x1 <= imgw && y0 >= 0.0 && y0 <= imgh &&
        y1 >= 0.0 && y1 <= imgh))
    return 1;
  box[0] = (int)floor(x0 + 0.5);
  box[1] = (int)floor(y0 + 0.5);
  box[2] = (int)floor(x1 + 0.5);
  box[3] = (int)floor(y1 + 0.5);
  duv[0] = u - (double)(box[0] + box[2]) / 2.0;
  duv[1] = v - (double)(box[1] + box[3]) / 2.0;
  return 0;
}

// np.interp for one x (numpy/_core/src/multiarray/compiled_base.c: arr_interp), with the
// default left = fp[0], right = fp[n-1].  xp is non-decreasing, n >= 1.
// In two halves: the interval search (its result -- or the two out-of-range cases -- as one number) and the
// evaluation, so that a caller with many x of few distinct values searches once per value.
constexpr int NP_INTERP_LEFT = -1;  // x < xp[0]
GLH_HD int np_interp_find(double x, const double* xp, int n, int lo = 0, int hi = -1) {
  if (n == 1) return 0;
  if (x > xp[n - 1]) return n - 1;  // (evaluates to fp[n - 1])
  if (x < xp[0]) return NP_INTERP_LEFT;
  if (hi < 0) hi = n - 1;  // invariant: xp[lo] <= x, and (hi == n-1 or x < xp[hi])
  if (x >= xp[n - 1]) {
    lo = n - 1;
  } else {
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (x >= xp[mid])
        lo = mid;
      else
        hi = mid;
    }
  }
  return lo;
}
GLH_HD double np_interp_at(int j, double x, const double* xp, const double* fp, int n) {
  if (j == NP_INTERP_LEFT) return fp[0];
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double res = slope * (x - xp[j]) + fp[j];
  if (isnan(res)) {
    res = slope * (x - xp[j + 1]) + fp[j + 1];
    if (isnan(res) && fp[j] == fp[j + 1]) res = fp[j];
  }
  return res;
}
// (np_interp itself keeps its one-piece form: the 8-bit LUT code of the fused kernel is compiled from it)
GLH_HD double np_interp(double x, const double* xp, const double* fp, int n) {
  if (n == 1) return fp[0];
  if (x > xp[n - 1]) return fp[n - 1];
  if (x < xp[0]) return fp[0];
  int lo = 0, hi = n - 1;  // invariant: xp[lo] <= x, and (hi == n-1 or x < xp[hi])
  if (x >= xp[n - 1]) {
    lo = n - 1;
  } else {
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (x >= xp[mid])
        lo = mid;
      else
        hi = mid;
    }
  }
  int j = lo;
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double res = slope * (x - xp[j]) + fp[j];
  if (isnan(res)) {
    res = slope * (x - xp[j + 1]) + fp[j + 1];
    if (isnan(res) && fp[j] == fp[j + 1]) res = fp[j];
  }
  return res;
}

// Index with edge-repeating reflection (scipy.ndimage mode='reflect': d c b a | a b c d | d c b a).
GLH_HD int reflect_index(int i, int n) {
  while (i < 0 || i >= n) {
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
  }
  return i;
}

// The other boundary modes of scipy.ndimage.median_filter (Tracker(highpass={"size": ..., "mode": ...}), tracker.py:59,
// :530): 'nearest' (a a a | a b c d | d d d), 'mirror' (d c b | a b c d | c b a), 'wrap' (a b c d | a b c d | a b c d).
// The library carries the mode IN the window's half height: ry | mode << 4 (GLH_HP_RY / GLH_HP_MODE), so that every test
// "is this the 5 x 5 default" (ry == 2) also asks for the default boundary, and only the window functions decode it.
#define GLH_HP_REFLECT 0
#define GLH_HP_NEAREST 1
#define GLH_HP_MIRROR 2
#define GLH_HP_WRAP 3
#define GLH_HP_RY(packed) ((packed) & 15)
#define GLH_HP_MODE(packed) ((packed) >> 4)
GLH_HD int border_index(int i, int n, int mode) {
  if (mode == GLH_HP_REFLECT) return reflect_index(i, n);
  if (mode == GLH_HP_NEAREST) return i < 0 ? 0 : (i >= n ? n - 1 : i);
  if (mode == GLH_HP_MIRROR) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
      if (i < 0) i = -i;
      if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
  }
  i %= n;  // GLH_HP_WRAP
  return i < 0 ? i + n : i;
}

// ---- not-a-knot bicubic spline (observer.py:210 == FITPACK regrid/bispev with s=0) ------
// knot i (0 <= i < n+4) in unit-spaced local coordinates: [0]*4, 2..n-3, [n-1]*4
GLH_HD double knot_local(int i, int n) {
  return i <= 3 ? 0.0 : (i >= n ? (double)(n - 1) : (double)(i - 2));
}

// interval q (0 <= q <= n-4) that holds local coordinate xl in [0, n-1]
// The spline fit of a surface of up to GLH_SPL_DENSE_MAX coefficients a side goes through the explicit inverses of the two
// collocation matrices (glh_host.h: spline_inverse): two dense products, every coefficient an independent dot product.
// Larger surfaces use the banded LU solves -- two serial chains of n dependent steps per line.  Round 5 raised the bound
// from "both inverses fit in 512 doubles" (16 x 16) to 40 x 40: phase stamps showed the banded fit at 15.7 k cycles for a
// 19 x 20 surface against 3.2 k for the dense fit of an 11 x 11 one -- the points with the wider clouds, the slowest of
// every launch, paid it (tools/experiments/slow_points.py).  Inverses of up to GLH_SPL_DENSE_NINV doubles (one entry per
// thread of the fused kernel) are staged in LDS, larger ones are read from the table in memory (every workgroup reads the
// same few: cache hits).
constexpr int GLH_SPL_DENSE_MAX = 40;    // largest side fitted by explicit inverses
constexpr int GLH_SPL_DENSE_NINV = 512;  // ho^2 + wo^2 <= this: the inverses are staged in LDS
GLH_HD bool spline_dense(int ho, int wo) { return ho <= GLH_SPL_DENSE_MAX && wo <= GLH_SPL_DENSE_MAX; }
GLH_HD int64_t spline_inverse_off(int n) {  // offset of the n x n inverse in the packed table (sizes 4 .. MAX)
  // sum_{m=4}^{n-1} m^2
  const int64_t k = n - 1;
  return k * (k + 1) * (2 * k + 1) / 6 - 14;
}

GLH_HD int spline_interval(double xl, int n) {
  int m = (int)floor(xl) - 1;
  if (m < 0) m = 0;
  if (m > n - 4) m = n - 4;
  return m;
}

// The 4 non-zero cubic B-splines on interval q at x (de Boor, as FITPACK's fpbspl);
// knots are x0 + knot_local(i, n).
GLH_HD void spline_basis(double x, int q, int n, double x0, double* h) {
  int l = q + 3;
  double hh[3];
  h[0] = 1.0;
  h[1] = h[2] = h[3] = 0.0;
  for (int j = 1; j <= 3; ++j) {
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.0;
    for (int i = 0; i < j; ++i) {
      int li = l + i + 1;
      int lj = li - j;
      double tli = x0 + knot_local(li, n);
      double tlj = x0 + knot_local(lj, n);
      double f = hh[i] / (tli - tlj);
      h[i] = h[i] + f * (tli - x);
      h[i + 1] = f * (x - tlj);
    }
  }
}

// Cell-centre origin along one axis (observer.py:203-208):
//   d = (b1 - b0) / n ;  c0 = b0 + d * 0.5   (np.arange start)
GLH_HD double cell_origin(double b0, double b1, int n) { return b0 + ((b1 - b0) / n) * 0.5; }

// Same basis in local coordinates xl = x - x0 (sites at 0..n-1).  The knot differences are
// small integers (1..4, and 5 only when n == 6: the first and last not-a-knot intervals are
// 2 wide), so the de Boor divisions become multiplications by exact (1, 1/2, 1/4) or
// correctly rounded (1/3, 1/5) reciprocals: <= 1 ulp from the division form.
GLH_HD void spline_basis_local(double xl, int q, int n, double* h) {
  const int l = q + 3;
  const double rcp[6] = {0.0, 1.0, 0.5, 1.0 / 3.0, 0.25, 0.2};
  double hh[3];
  h[0] = 1.0;
  h[1] = h[2] = h[3] = 0.0;
  for (int j = 1; j <= 3; ++j) {
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.0;
    for (int i = 0; i < j; ++i) {
      const int li = l + i + 1, lj = li - j;
      const int ki = li <= 3 ? 0 : (li >= n ? n - 1 : li - 2);
      const int kj = lj <= 3 ? 0 : (lj >= n ? n - 1 : lj - 2);
      const double f = hh[i] * rcp[ki - kj];
      h[i] = h[i] + f * ((double)ki - xl);
      h[i + 1] = f * (xl - (double)kj);
    }
  }
}

// ---- piecewise-polynomial form of the same basis -------------------------------------------
// On knot interval q the 4 non-zero B-splines are cubics in s = xl - start(q).  Their
// coefficient matrices depend on n only near the ends: for n >= 9 there are 7 distinct
// matrices (3 at the left end, 1 uniform interior, 3 at the right end); n = 4..8 have their
// own (1 + 2 + 3 + 4 + 5).  GLH_NPOLY matrices of 16 doubles, built on the host
// (glh_host.h: basis_poly_table) from the de Boor recursion above.
#define GLH_NPOLY 22
GLH_HD int spline_poly_index(int n, int q) {
  if (n >= 9) return q <= 2 ? q : (q >= n - 6 ? 4 + q - (n - 6) : 3);
  // n = 4: 7 | 5: 8..9 | 6: 10..12 | 7: 13..16 | 8: 17..21
  return 7 + ((n - 4) * (n - 3)) / 2 + q;
}
GLH_HD double spline_interval_start(int q) { return q == 0 ? 0.0 : (double)(q + 1); }

// h[m] = ((c3 s + c2) s + c1) s + c0 with the matrix laid out [m][d]
GLH_HD void spline_basis_poly(const double* tab, double xl, int q, int n, double* h) {
  const double* c = tab + 16 * spline_poly_index(n, q);
  const double s = xl - spline_interval_start(q);
  for (int m2 = 0; m2 < 4; ++m2) h[m2] = ((c[4 * m2 + 3] * s + c[4 * m2 + 2]) * s + c[4 * m2 + 1]) * s + c[4 * m2];
}

GLH_HD double spline_eval_poly(const double* tab, const double* coef, int ld, int ho, int wo, double cv0,
                               double cu0, double u, double v) {
  double vl = v - cv0, ul = u - cu0;
  const double vmax = (double)(ho - 1), umax = (double)(wo - 1);
  vl = vl < 0.0 ? 0.0 : (vl > vmax ? vmax : vl);
  ul = ul < 0.0 ? 0.0 : (ul > umax ? umax : ul);
  const int qv = spline_interval(vl, ho);
  const int qu = spline_interval(ul, wo);
  double hv[4], hu[4];
  spline_basis_poly(tab, vl, qv, ho, hv);
  spline_basis_poly(tab, ul, qu, wo, hu);
  double sp = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double* row = coef + (size_t)(qv + i) * ld + qu;
    for (int j = 0; j < 4; ++j) sp += row[j] * hv[i] * hu[j];
  }
  return sp;
}

// spline_eval_poly in fast arithmetic: Horner steps and the tensor sum as fused multiply-adds, the tensor summed
// row by row (20 operations instead of 48), clamps as min / max (the arguments are never NaN here: a NaN
// projection skips the observer before anything is sampled).
GLH_HD double spline_eval_poly_fast(const double* tab, const double* coef, int ld, int ho, int wo, double cv0,
                                    double cu0, double u, double v) {
  const double vmax = (double)(ho - 1), umax = (double)(wo - 1);
  const double vl = min_nn(max_nn(v - cv0, 0.0), vmax), ul = min_nn(max_nn(u - cu0, 0.0), umax);
  const int qv = spline_interval(vl, ho);
  const int qu = spline_interval(ul, wo);
  double hv[4], hu[4];
  {
    const double* c = tab + 16 * spline_poly_index(ho, qv);
    const double s = vl - spline_interval_start(qv);
    for (int m2 = 0; m2 < 4; ++m2)
      hv[m2] = glh_fma(glh_fma(glh_fma(c[4 * m2 + 3], s, c[4 * m2 + 2]), s, c[4 * m2 + 1]), s, c[4 * m2]);
  }
  {
    const double* c = tab + 16 * spline_poly_index(wo, qu);
    const double s = ul - spline_interval_start(qu);
    for (int m2 = 0; m2 < 4; ++m2)
      hu[m2] = glh_fma(glh_fma(glh_fma(c[4 * m2 + 3], s, c[4 * m2 + 2]), s, c[4 * m2 + 1]), s, c[4 * m2]);
  }
  double sp = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double* row = coef + (size_t)(qv + i) * ld + qu;
    const double si = glh_fma(row[3], hu[3], glh_fma(row[2], hu[2], glh_fma(row[1], hu[1], row[0] * hu[0])));
    sp = glh_fma(hv[i], si, sp);
  }
  return sp;
}
// ---- per-cell power form of the fitted surface (fast arithmetic only) ----------------------------------------
// On cell (qv, qu) the spline is a bicubic in the local offsets (sv, su): sum_ab P[a][b] sv^a su^b with
// P = Mv^T * Z[qv..qv+3][qu..qu+3] * Mu (M = the cell's basis matrices above, [m][d]).  Built once per surface, a
// particle then reads ONE contiguous 128-byte block and runs 15 fused multiply-adds, instead of two basis matrices
// (256 B), 16 scattered coefficients and 44 operations.  Cells are GLH_CELL_LD doubles apart (144 B: rows stay
// 16-byte aligned and consecutive cells start 4 banks apart).
constexpr int GLH_CELL_LD = 18;
GLH_HD int spline_cells(int n) { return n - 3; }
// row a (the power of sv) of cell (qv, qu): out4[b], b = the power of su
GLH_HD void spline_cell_row(const double* tab, const double* coef, int ld, int ho, int wo, int qv, int qu, int a,
                            double* out4) {
  const double* mv = tab + 16 * spline_poly_index(ho, qv);
  const double* mu = tab + 16 * spline_poly_index(wo, qu);
  const double* z = coef + (size_t)qv * ld + qu;
  double t[4];
  for (int j = 0; j < 4; ++j)
    t[j] = glh_fma(mv[12 + a], z[3 * ld + j], glh_fma(mv[8 + a], z[2 * ld + j], glh_fma(mv[4 + a], z[ld + j], mv[a] * z[j])));
  for (int b = 0; b < 4; ++b)
    out4[b] = glh_fma(t[3], mu[12 + b], glh_fma(t[2], mu[8 + b], glh_fma(t[1], mu[4 + b], t[0] * mu[b])));
}
// interval q = spline_interval(xl, n) and the offset xl - spline_interval_start(q) of a clamped coordinate, without
// leaving float64: q = clamp(floor(xl) - 1, 0, n - 4) and start = q + min(q, 1) (0 for the double-width first
// interval, q + 1 after it) are small integers, exact in either type
GLH_HD double spline_cell_offset(double xl, int n, int& q) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double fl = __builtin_floor(xl);
#else
  const double fl = floor(xl);
#endif
  const double qd = min_nn(max_nn(fl - 1.0, 0.0), (double)(n - 4));
  q = (int)qd;
  return xl - (qd + min_nn(qd, 1.0));
}
// Horner in su along each row, then in sv across the rows; p = 16 doubles [a][b]
GLH_HD double spline_cell_horner(const double* p, double sv, double su) {
  double r[4];
  for (int a = 0; a < 4; ++a)
    r[a] = glh_fma(glh_fma(glh_fma(p[4 * a + 3], su, p[4 * a + 2]), su, p[4 * a + 1]), su, p[4 * a]);
  return glh_fma(glh_fma(glh_fma(r[3], sv, r[2]), sv, r[1]), sv, r[0]);
}
// sample a surface held in per-cell form (cells = GLH_CELL_LD doubles per cell, row-major over (qv, qu))
GLH_HD double spline_eval_cell(const double* cells, int ho, int wo, double cv0, double cu0, double u, double v) {
  const double vmax = (double)(ho - 1), umax = (double)(wo - 1);
  const double vl = min_nn(max_nn(v - cv0, 0.0), vmax), ul = min_nn(max_nn(u - cu0, 0.0), umax);
  int qv, qu;
  const double sv = spline_cell_offset(vl, ho, qv), su = spline_cell_offset(ul, wo, qu);
  return spline_cell_horner(cells + (size_t)(qv * spline_cells(wo) + qu) * GLH_CELL_LD, sv, su);
}
// the same value without a cell table: the particle's cell is converted on the spot (the staged kernels; every
// operation is the one spline_cell_row / spline_eval_cell perform, so the result is bit-identical)
GLH_HD double spline_eval_cell_direct(const double* tab, const double* coef, int ld, int ho, int wo, double cv0,
                                      double cu0, double u, double v) {
  const double vmax = (double)(ho - 1), umax = (double)(wo - 1);
  const double vl = min_nn(max_nn(v - cv0, 0.0), vmax), ul = min_nn(max_nn(u - cu0, 0.0), umax);
  int qv, qu;
  const double sv = spline_cell_offset(vl, ho, qv), su = spline_cell_offset(ul, wo, qu);
  double p[16];
  for (int a = 0; a < 4; ++a) spline_cell_row(tab, coef, ld, ho, wo, qv, qu, a, p + 4 * a);
  return spline_cell_horner(p, sv, su);
}
template <bool FAST>
GLH_HD double spline_eval_poly_m(const double* tab, const double* coef, int ld, int ho, int wo, double cv0,
                                 double cu0, double u, double v) {
  return FAST ? spline_eval_poly_fast(tab, coef, ld, ho, wo, cv0, cu0, u, v)
              : spline_eval_poly(tab, coef, ld, ho, wo, cv0, cu0, u, v);
}

// w = exp(-ll) + 1e-300 (tracker.py:149).  Exact: the library exp.  Fast: exp(x) = 2^m * T[j] * P(r) with
// x = (32 m + j) ln2 / 32 + r, |r| <= ln2 / 64, T[j] = 2^(j/32) (a 32-entry table the caller provides: LDS) and
// P the degree-5 Taylor polynomial (|r|^6 / 720 < 2.3e-15): ~1e-15 relative, half the instructions.
constexpr int GLH_EXP_TAB = 32;
GLH_HD double exp_fast(double x, const double* tab32) {
  const double t = x * 0x1.71547652b82fep+5;  // 32 / ln 2
#if defined(__HIP_DEVICE_COMPILE__)
  const double kf = __builtin_rint(t);
#else
  const double kf = rint(t);
#endif
  double r = glh_fma(kf, -0x1.62e42fefa39efp-6, x);  // ln 2 / 32 = hi + lo
  r = glh_fma(kf, -0x1.abc9e3b39803fp-61, r);
  const int ki = (int)kf;
  const double T = tab32[ki & (GLH_EXP_TAB - 1)];
  double p = glh_fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = glh_fma(p, r, 1.0 / 6.0);
  p = glh_fma(p, r, 0.5);
  p = glh_fma(p, r, 1.0);
  p = glh_fma(p, r, 1.0);
  return ldexp(T * p, ki >> 5);
}
template <bool FAST>
GLH_HD double weight_of(double ll, const double* tab32) {
  if (FAST) {
    // exp underflows to 0 below -745.2, like the library's.  Arguments below -746 are replaced by -800 (whose
    // 2^-1154 scaling rounds to 0 as well) rather than branched around: ki stays a small int, a NaN stays a NaN
    const double x = -ll;
    return exp_fast(x < -746.0 ? -800.0 : x, tab32) + 1e-300;
  }
  return exp(-ll) + 1e-300;
}

// RectBivariateSpline(kx = ky = 1, s = 0) (Tracker(interpolation={"kx": 1, "ky": 1}), tracker.py:60, :623): the
// degree-1 B-spline through the surface values, i.e. their bilinear interpolant; arguments clamped like fpbisp's;
// the two non-zero basis functions on the unit knot interval [i, i + 1] are (i + 1 - x) and (x - i) (fpbspl).
GLH_HD double spline_eval_linear(const double* z, int ld, int ho, int wo, double cv0, double cu0, double u, double v) {
  double vl = v - cv0, ul = u - cu0;
  const double vmax = (double)(ho - 1), umax = (double)(wo - 1);
  vl = vl < 0.0 ? 0.0 : (vl > vmax ? vmax : vl);
  ul = ul < 0.0 ? 0.0 : (ul > umax ? umax : ul);
  int iv = (int)floor(vl), iu = (int)floor(ul);
  if (iv > ho - 2) iv = ho - 2;
  if (iu > wo - 2) iu = wo - 2;
  const double hv[2] = {(double)(iv + 1) - vl, vl - (double)iv}, hu[2] = {(double)(iu + 1) - ul, ul - (double)iu};
  double sp = 0.0;
  for (int i = 0; i < 2; ++i) {
    const double* row = z + (size_t)(iv + i) * ld + iu;
    for (int j = 0; j < 2; ++j) sp += row[j] * hv[i] * hu[j];
  }
  return sp;
}

// ---- any order 1 .. 5 of RectBivariateSpline(kx, ky), s = 0 (Tracker(interpolation=...), tracker.py:60, :623) ----
// FITPACK (fpgrre, interpolation case) puts the interior knots at the data sites for odd degrees and midway between
// them for even degrees; on unit-spaced sites 0 .. n-1 both read t[i] = i - (k + 1) / 2 for i = k+1 .. n-1, between
// k + 1 knots at 0 and k + 1 at n - 1.  Restated in oracle/spline.py (knot_general ... eval_general) and pinned to the
// reference by tests/golden/g23_orders.npz.
constexpr int GLH_SPL_KMAX = 5;
GLH_HD double gspl_knot(int i, int n, int k) {
  return i <= k ? 0.0 : (i >= n ? (double)(n - 1) : (double)i - 0.5 * (double)(k + 1));
}
// l with t[l] <= xl < t[l+1], clamped to [k, n-1] (fpbisp)
GLH_HD int gspl_interval(double xl, int n, int k) {
  int l = (int)floor(xl + 0.5 * (double)(k + 1));
  if (l < k) l = k;
  if (l > n - 1) l = n - 1;
  return l;
}
// the k + 1 non-zero B-splines B_{l-k} .. B_l at x (FITPACK fpbspl); h [GLH_SPL_KMAX + 1]
GLH_HD void gspl_basis(double x, int l, int n, int k, double* h) {
  double hh[GLH_SPL_KMAX + 1];
  h[0] = 1.0;
  for (int j = 1; j <= k; ++j) {
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.0;
    for (int i = 0; i < j; ++i) {
      const int li = l + i + 1, lj = li - j;
      const double tli = gspl_knot(li, n, k), tlj = gspl_knot(lj, n, k);
      const double f = hh[i] / (tli - tlj);
      h[i] = h[i] + f * (tli - x);
      h[i + 1] = f * (x - tlj);
    }
  }
}
// the tensor spline of degree kv along the rows axis and ku along the columns axis, arguments clamped to the outermost
// sites (fpbisp); coef [ho][wo], row stride ld
GLH_HD double spline_eval_general(const double* coef, int ld, int ho, int wo, int kv, int ku, double cv0, double cu0,
                                  double u, double v) {
  double vl = v - cv0, ul = u - cu0;
  const double vmax = (double)(ho - 1), umax = (double)(wo - 1);
  vl = vl < 0.0 ? 0.0 : (vl > vmax ? vmax : vl);
  ul = ul < 0.0 ? 0.0 : (ul > umax ? umax : ul);
  const int lv = gspl_interval(vl, ho, kv), lu = gspl_interval(ul, wo, ku);
  double hv[GLH_SPL_KMAX + 1], hu[GLH_SPL_KMAX + 1];
  gspl_basis(vl, lv, ho, kv, hv);
  gspl_basis(ul, lu, wo, ku, hu);
  double sp = 0.0;
  for (int i = 0; i <= kv; ++i) {
    const double* row = coef + (size_t)(lv - kv + i) * ld + (lu - ku);
    for (int j = 0; j <= ku; ++j) sp += row[j] * hv[i] * hu[j];
  }
  return sp;
}

// Evaluate the tensor spline with coefficients coef[ho][wo] (row stride ld) at (u, v);
// arguments are clamped to the outermost cell centres (FITPACK fpbisp).
GLH_HD double spline_eval(const double* coef, int ld, int ho, int wo, double cv0, double cu0,
                          double u, double v) {
  double vl = v - cv0, ul = u - cu0;
  const double vmax = (double)(ho - 1), umax = (double)(wo - 1);
  vl = vl < 0.0 ? 0.0 : (vl > vmax ? vmax : vl);
  ul = ul < 0.0 ? 0.0 : (ul > umax ? umax : ul);
  const int qv = spline_interval(vl, ho);
  const int qu = spline_interval(ul, wo);
  double hv[4], hu[4];
  spline_basis_local(vl, qv, ho, hv);
  spline_basis_local(ul, qu, wo, hu);
  double sp = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double* row = coef + (size_t)(qv + i) * ld + qu;
    for (int j = 0; j < 4; ++j) sp += row[j] * hv[i] * hu[j];
  }
  return sp;
}

// ---- Philox4x32 (Salmon et al. 2011), counter-based: no state in HBM ----------------------
// The device streams use 7 rounds: the smallest round count of Philox4x32 that passes BigCrush ("Crush-resistant",
// Salmon et al. 2011, table 2; the paper's own safety margin is 10).  philox4x32_r<10> is Random123's
// philox4x32_10 (known-answer vectors in tests/test_hostcheck.py); the statistical tests of the 7-round streams
// are tests/test_gpu_rng_statistics.py.
#ifndef GLH_PHILOX_ROUNDS
#define GLH_PHILOX_ROUNDS 7
#endif
GLH_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

template <int ROUNDS>
GLH_HD void philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                         uint32_t k1, uint32_t* out) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
  // The key is uniform (it comes from the seed).  Opaque to the optimiser at every call: otherwise the 2 x ROUNDS round
  // keys are hoisted out of the particle loops as kernel-wide constants, spilled for lack of scalar registers and
  // restored by ~9 v_readlane per call; recomputed here they are scalar adds beside the vector work.
  asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
#endif
  for (int r = 0; r < ROUNDS; ++r) {
    // one 32 x 32 -> 64 product per multiplier (a single v_mad_u64_u32 on gfx950) gives hi and lo
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// the generator of every device stream (evolve / init noise, resampling offsets)
GLH_HD void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                       uint32_t* out) {
  philox4x32_r<GLH_PHILOX_ROUNDS>(c0, c1, c2, c3, k0, k1, out);
}

// two uint32 -> uniform double in (0, 1): 53 random bits, never 0 (safe for log)
GLH_HD double u01_open(uint32_t a, uint32_t b) {
  uint64_t x = (((uint64_t)a << 32) | b) >> 11;  // 53 bits
  return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}

// [0, 1) like np.random.random()
GLH_HD double u01_halfopen(uint32_t a, uint32_t b) {
  uint64_t x = (((uint64_t)a << 32) | b) >> 11;
  return (double)x * (1.0 / 9007199254740992.0);
}

}  // namespace glh
