"""Oracle: bicubic-spline sampling of the SSE surface (test infrastructure only).

`sample_tile` restates `Observer.sample_tile` (/root/reference/src/glimpse/
track/observer.py:178-214, with `helpers.in_box` helpers.py:815-832) with the
same SciPy call the reference makes (`RectBivariateSpline(cv, cu, tile, kx=3,
ky=3)`, s=0).

`fit_notaknot` / `eval_notaknot` are the closed-form restatement the HIP
kernels implement: the interpolating tensor-product
cubic B-spline on the not-a-knot knot vector
    t = [x0]*4 + [x2 .. x_{n-3}] + [x_{n-1}]*4,
coefficients from two passes of banded (|i-j| <= 2) collocation solves,
evaluation by de Boor's recursion with the argument clamped to [x0, x_{n-1}]
(FITPACK fpbisp behaviour).  tests/test_oracle_spline.py checks the two agree.
"""
import numpy as np
import scipy.interpolate


def in_box(points, box):
    """helpers.py:831-832 (box = [xmin, ymin, xmax, ymax])."""
    box = np.asarray(box, dtype=float).reshape(2, -1)
    return np.all((points >= box[0, :]) & (points <= box[1, :]), axis=1)


def cell_centres(box, shape):
    """observer.py:203-208."""
    du = (box[2] - box[0]) / shape[1]
    dv = (box[3] - box[1]) / shape[0]
    cu = np.arange(box[0] + du * 0.5, box[2])
    cv = np.arange(box[1] + dv * 0.5, box[3])
    return cu, cv


def sample_tile(uv, tile, box, kx=3, ky=3):
    """observer.py:201-214 (grid=False)."""
    if not np.all(in_box(uv, box)):
        raise ValueError("Some sampling points are outside box")
    cu, cv = cell_centres(box, tile.shape)
    f = scipy.interpolate.RectBivariateSpline(cv, cu, tile, kx=kx, ky=ky)
    return f(uv[:, 1], uv[:, 0], grid=False)


# ---- closed-form restatement (what the kernels do) ----


def knot_local(i, n):
    """i-th knot (0 <= i < n+4) in unit-spaced local coordinates (site j at j)."""
    i = np.asarray(i)
    return np.where(i <= 3, 0, np.where(i >= n, n - 1, i - 2)).astype(float)


def basis4(x, q, n, x0=0.0):
    """The 4 non-zero cubic B-spline basis values on knot interval q (de Boor).

    Interval q (0 <= q <= n-4) spans knots t[q+3] .. t[q+4]; the basis functions
    are B_q .. B_{q+3}.  Mirrors FITPACK's fpbspl recursion.
    """
    l = q + 3
    t = lambda i: x0 + knot_local(i, n)  # noqa: E731
    h = np.zeros(4)
    hh = np.zeros(3)
    h[0] = 1.0
    for j in range(1, 4):
        hh[:j] = h[:j]
        h[0] = 0.0
        for i in range(j):
            li = l + i + 1
            lj = li - j
            f = hh[i] / (t(li) - t(lj))
            h[i] = h[i] + f * (t(li) - x)
            h[i + 1] = f * (x - t(lj))
    return h


def interval_of(xl, n):
    """Knot interval of local coordinate xl in [0, n-1]:  clamp(floor(xl)-1, 0, n-4)."""
    m = int(np.floor(xl))
    return min(max(m - 1, 0), n - 4)


def collocation_bands(n):
    """Banded collocation matrix A[i, j] = B_j(site i), |i - j| <= 2.

    Returned as `ab` (n, 5): ab[i, d] = A[i, i + d - 2] (zero outside the matrix).
    """
    ab = np.zeros((n, 5))
    for i in range(n):
        q = interval_of(float(i), n)
        h = basis4(float(i), q, n)
        for m in range(4):
            j = q + m
            d = j - i + 2
            if abs(h[m]) > 0:
                assert 0 <= d <= 4, (n, i, j, h)
                ab[i, d] = h[m]
    return ab


def lu_bands(ab):
    """LU without pivoting of a banded matrix with lower/upper bandwidth 2.

    Returns (l1, l2, u0inv, u1, u2): row i of L has l1[i] at (i,i-1), l2[i] at
    (i,i-2); row i of U has 1/u0inv[i] at (i,i), u1[i] at (i,i+1), u2[i] at (i,i+2).
    The matrix is diagonally dominant by rows, so no pivoting is needed.
    """
    n = ab.shape[0]
    a = np.zeros((n, n))
    for i in range(n):
        for d in range(5):
            j = i + d - 2
            if 0 <= j < n:
                a[i, j] = ab[i, d]
    l1 = np.zeros(n)
    l2 = np.zeros(n)
    for k in range(n):
        for i in range(k + 1, min(k + 3, n)):
            m = a[i, k] / a[k, k]
            if i == k + 1:
                l1[i] = m
            else:
                l2[i] = m
            a[i, k : min(k + 3, n)] -= m * a[k, k : min(k + 3, n)]
            a[i, k] = 0.0
    u0inv = 1.0 / np.diag(a)
    u1 = np.zeros(n)
    u2 = np.zeros(n)
    u1[: n - 1] = np.diag(a, 1)
    u2[: n - 2] = np.diag(a, 2)
    return l1, l2, u0inv, u1, u2


def solve_lines(lu, b):
    """Solve A x = b along axis 0 of b (n, m) with the factors from `lu_bands`."""
    l1, l2, u0inv, u1, u2 = lu
    n = b.shape[0]
    y = np.array(b, dtype=float)
    for i in range(1, n):
        y[i] = y[i] - l1[i] * y[i - 1]
        if i >= 2:
            y[i] = y[i] - l2[i] * y[i - 2]
    x = y
    for i in range(n - 1, -1, -1):
        acc = x[i]
        if i + 1 < n:
            acc = acc - u1[i] * x[i + 1]
        if i + 2 < n:
            acc = acc - u2[i] * x[i + 2]
        x[i] = acc * u0inv[i]
    return x


def fit_notaknot(z):
    """Coefficients C with A_rows . C . A_cols^T = z (columns pass, then rows pass)."""
    ho, wo = z.shape
    c = solve_lines(lu_bands(collocation_bands(ho)), np.asarray(z, dtype=float))
    c = solve_lines(lu_bands(collocation_bands(wo)), c.T).T
    return c


def eval_notaknot(coef, cv0, cu0, v, u):
    """Evaluate at points (v, u); site (r, c) sits at (cv0 + r, cu0 + c)."""
    ho, wo = coef.shape
    out = np.empty(len(u))
    for k in range(len(u)):
        vv = min(max(v[k], cv0), cv0 + (ho - 1))
        uu = min(max(u[k], cu0), cu0 + (wo - 1))
        qv = interval_of(vv - cv0, ho)
        qu = interval_of(uu - cu0, wo)
        hv = basis4(vv, qv, ho, cv0)
        hu = basis4(uu, qu, wo, cu0)
        sp = 0.0
        for i in range(4):
            for j in range(4):
                sp += coef[qv + i, qu + j] * hv[i] * hu[j]
        out[k] = sp
    return out


# ---- any order 1 .. 5 (RectBivariateSpline(kx, ky), s = 0): closed form of what the general kernels do ----
# FITPACK (fpgrre, interpolation case) puts the interior knots at the data sites for odd degrees and midway between them
# for even degrees; with unit-spaced sites j = 0 .. n-1 both read  t[i] = i - (k + 1) / 2  for i = k+1 .. n-1, between
# k + 1 knots at 0 and k + 1 knots at n - 1.


def knot_general(i, n, k):
    """i-th knot (0 <= i < n + k + 1) of the degree-k interpolating spline on sites 0 .. n-1."""
    if i <= k:
        return 0.0
    if i >= n:
        return float(n - 1)
    return i - 0.5 * (k + 1)


def interval_general(xl, n, k):
    """l with t[l] <= xl < t[l+1], clamped to [k, n-1] (fpbisp)."""
    return int(min(max(np.floor(xl + 0.5 * (k + 1)), k), n - 1))


def basis_general(x, l, n, k):
    """The k + 1 non-zero B-splines B_{l-k} .. B_l at x (FITPACK fpbspl)."""
    h = np.zeros(k + 1)
    hh = np.zeros(k + 1)
    h[0] = 1.0
    for j in range(1, k + 1):
        hh[:j] = h[:j]
        h[0] = 0.0
        for i in range(j):
            li, lj = l + i + 1, l + i + 1 - j
            f = hh[i] / (knot_general(li, n, k) - knot_general(lj, n, k))
            h[i] = h[i] + f * (knot_general(li, n, k) - x)
            h[i + 1] = f * (x - knot_general(lj, n, k))
    return h


def collocation_general(n, k):
    """Dense n x n collocation matrix A[i, j] = B_j(site i); banded with |i - j| <= k."""
    a = np.zeros((n, n))
    for i in range(n):
        l = interval_general(float(i), n, k)
        a[i, l - k : l + 1] = basis_general(float(i), l, n, k)
    return a


def lu_general(n, k):
    """LU without pivoting of the collocation matrix (totally positive), bandwidth k on either side:
    (L [n][k]: L[i][d-1] at (i, i-d); u0inv [n]; U [n][k]: U[i][d-1] at (i, i+d))."""
    a = collocation_general(n, k)
    L = np.zeros((n, k))
    for c in range(n):
        for i in range(c + 1, min(c + k + 1, n)):
            m = a[i, c] / a[c, c]
            L[i, i - c - 1] = m
            a[i, c : min(c + k + 1, n)] -= m * a[c, c : min(c + k + 1, n)]
            a[i, c] = 0.0
    U = np.zeros((n, k))
    for i in range(n):
        for d in range(1, k + 1):
            if i + d < n:
                U[i, d - 1] = a[i, i + d]
    return L, 1.0 / np.diag(a), U


def solve_general(lu, b):
    """Solve A x = b along axis 0 of b (n, m) with the factors from `lu_general`."""
    L, u0inv, U = lu
    n, k = L.shape
    y = np.array(b, dtype=float)
    for i in range(1, n):
        for d in range(1, min(k, i) + 1):
            y[i] = y[i] - L[i, d - 1] * y[i - d]
    x = y
    for i in range(n - 1, -1, -1):
        acc = x[i]
        for d in range(1, k + 1):
            if i + d < n:
                acc = acc - U[i, d - 1] * x[i + d]
        x[i] = acc * u0inv[i]
    return x


def fit_general(z, kx, ky):
    """Coefficients of the interpolating spline of degree kx along the rows axis and ky along the columns axis."""
    ho, wo = z.shape
    c = solve_general(lu_general(ho, kx), np.asarray(z, dtype=float))
    return solve_general(lu_general(wo, ky), c.T).T


def eval_general(coef, kx, ky, cv0, cu0, v, u):
    """Evaluate at points (v, u), arguments clamped to the outermost sites (fpbisp)."""
    ho, wo = coef.shape
    out = np.empty(len(u))
    for p in range(len(u)):
        vl = min(max(v[p] - cv0, 0.0), ho - 1.0)
        ul = min(max(u[p] - cu0, 0.0), wo - 1.0)
        lv, lu_ = interval_general(vl, ho, kx), interval_general(ul, wo, ky)
        hv, hu = basis_general(vl, lv, ho, kx), basis_general(ul, lu_, wo, ky)
        sp = 0.0
        for i in range(kx + 1):
            for j in range(ky + 1):
                sp += coef[lv - kx + i, lu_ - ky + j] * hv[i] * hu[j]
        out[p] = sp
    return out
