"""Stage math of the HIP kernels, compiled for the CPU (tests/hostcheck) and checked against
the oracle / the reference goldens.  Test-only: the product never runs this code on the CPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import resample as oresample
from oracle import tiles as otiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp")
OUT = os.path.join(ROOT, "tests", "hostcheck", "_build", "libhostcheck.so")


@pytest.fixture(scope="module")
def hc():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [SRC] + [os.path.join(ROOT, "glimpse_amd", "csrc", f) for f in ("glh_math.h", "glh_median.h", "glh_host.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", OUT, SRC], check=True)
    lib = C.CDLL(OUT)
    lib.hc_pairwise_sum.restype = C.c_double
    lib.hc_poly_basis_error.restype = C.c_double
    return lib


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_projection_bit_exact_on_reference_goldens(hc, golden):
    g = golden("g1_projection.npz")
    for cam, xyz, uv in zip(g["cams"], g["xyz"], g["uv"]):
        got = np.empty_like(uv)
        hc.hc_project(p(np.ascontiguousarray(cam)), p(np.ascontiguousarray(xyz)), len(xyz), p(got))
        assert np.array_equal(np.isnan(got), np.isnan(uv))
        ok = ~np.isnan(uv[:, 0])
        np.testing.assert_allclose(got[ok], uv[ok], rtol=1e-13, atol=1e-9)


def test_search_and_template_boxes(hc, golden):
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.integers(1, 50))
        size = (int(rng.choice([15, 31, 9])), int(rng.choice([15, 31, 11])))
        uv = rng.uniform(-5, 140, 2) + rng.standard_normal((n, 2)) * rng.choice([0.01, 1.0, 8.0])
        want = otiles.search_box(uv, size)
        inb = (want >= 0).all() and (want <= np.array([128.0, 128.0])).all()
        box = np.zeros(4, dtype=np.int32)
        rc = hc.hc_search_box(p(np.ascontiguousarray(uv)), n, size[0], size[1], C.c_double(128.0), C.c_double(128.0), p(box))
        assert (rc == 0) == bool(inb)
        if inb:
            np.testing.assert_array_equal(box, want.ravel())
    uvn = np.array([[10.0, 10.0], [np.nan, 3.0]])
    assert hc.hc_search_box(p(uvn), 2, 15, 15, C.c_double(128.0), C.c_double(128.0), p(box)) == 1
    for trial in range(300):
        uv = rng.uniform(-3, 131, 2)
        size = (int(rng.choice([15, 31, 8])), int(rng.choice([15, 31, 10])))
        duv = np.zeros(2)
        rc = hc.hc_template_box(C.c_double(uv[0]), C.c_double(uv[1]), size[0], size[1], C.c_double(128.0),
                                C.c_double(128.0), p(box), p(duv))
        try:
            want = otiles.snap_box(uv, size, (128, 128))
        except IndexError:
            assert rc == 1
            continue
        assert rc == 0
        np.testing.assert_array_equal(box, want)
        np.testing.assert_array_equal(duv, uv - want.reshape(2, -1).mean(axis=0))


def test_np_interp_bit_exact(hc):
    rng = np.random.default_rng(1)
    for n in [1, 2, 3, 17, 225, 961]:
        xp = np.sort(rng.random(n))
        xp[-1] = 1.0
        fp = np.sort(rng.standard_normal(n))
        x = np.concatenate((rng.random(500), xp, [0.0, 1.0, xp[0] - 1e-3, 1.5]))
        out = np.empty_like(x)
        hc.hc_interp(p(x), len(x), p(xp), p(fp), n, p(out))
        np.testing.assert_array_equal(out, np.interp(x, xp, fp))


def test_median_and_reflect(hc):
    import scipy.ndimage

    rng = np.random.default_rng(2)
    v = rng.integers(0, 766, (5000, 25)).astype(np.int32)
    v[:100] = rng.integers(0, 3, (100, 25))
    out = np.empty(len(v), dtype=np.int32)
    hc.hc_median25(p(v), len(v), p(out))
    np.testing.assert_array_equal(out, np.sort(v, axis=1)[:, 12])
    for n in [1, 2, 3, 5, 16]:
        a = np.arange(n)
        pad = np.pad(a, 2 * n + 2, mode="symmetric")
        for i in range(-2 * n - 2, 3 * n + 2):
            assert hc.hc_reflect(i, n) == pad[i + 2 * n + 2]
    # scipy's 'reflect' == numpy 'symmetric' (edge-repeating) for the 5x5 window
    img = rng.integers(0, 255, (9, 7))
    want = scipy.ndimage.median_filter(img, size=(5, 5), mode="reflect")
    got = np.empty_like(img)
    for r in range(9):
        for c in range(7):
            w = np.array([[img[hc.hc_reflect(r + dr, 9), hc.hc_reflect(c + dc, 7)] for dc in range(-2, 3)]
                          for dr in range(-2, 3)], dtype=np.int32).ravel()
            o = np.empty(1, dtype=np.int32)
            hc.hc_median25(p(w), 1, p(o))
            got[r, c] = o[0]
    np.testing.assert_array_equal(got, want)


def test_spline_fit_and_eval_match_reference(hc, golden):
    g = golden("g4_spline.npz")
    for i in range(8):
        sse, box, uv, val = g[f"s{i}_sse"], g[f"s{i}_box"], g[f"s{i}_uv"], g[f"s{i}_val"]
        z = np.ascontiguousarray(sse, dtype=np.float64)
        out = np.empty(len(uv))
        hc.hc_spline_sample(p(z), z.shape[0], z.shape[1], p(np.ascontiguousarray(box)), p(np.ascontiguousarray(uv)),
                            len(uv), p(out))
        np.testing.assert_allclose(out, val, rtol=0, atol=5e-13)


def test_polynomial_basis_table_equals_de_boor(hc):
    # the 22 per-interval cubic matrices reproduce the de Boor basis for every n and interval
    assert hc.hc_poly_basis_error(64, 37) < 4e-16


def test_pairwise_sum_plan_is_numpy_sum(hc):
    rng = np.random.default_rng(3)
    for n in [1, 5, 8, 9, 100, 128, 129, 1000, 2000, 5000, 8191, 8192, 8193, 10000, 16000, 19000]:
        for t in range(5):
            w = np.exp(-rng.random(n) * 40) + 1e-300
            assert hc.hc_pairwise_sum(p(w), n) == w.sum() == oresample.numpy_pairwise_sum(w)


def test_philox_known_answers(hc):
    # Random123 known-answer vectors for philox4x32-10
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        out = (C.c_uint * 4)()
        hc.hc_philox(*ctr, *key, out)
        assert tuple(out) == want


def _philox_python(rounds, ctr, key):
    """Independent restatement of Philox4x32-R (Salmon et al. 2011, fig. 2) in Python integers."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(rounds):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, \
            p0 & 0xFFFFFFFF
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [c0, c1, c2, c3]


def test_philox_device_round_count_matches_python_restatement(hc):
    """The device streams use Philox4x32-7 (the Crush-resistant minimum); the 10-round form is pinned by Random123's
    known-answer vectors above, every other round count by the same round function restated in Python."""
    hc.hc_philox_device_rounds.restype = C.c_int
    assert hc.hc_philox_device_rounds() == 7
    rng = np.random.default_rng(3)
    for rounds in (7, 10):
        for _ in range(50):
            ctr = [int(v) for v in rng.integers(0, 2 ** 32, 4)]
            key = [int(v) for v in rng.integers(0, 2 ** 32, 2)]
            out = (C.c_uint * 4)()
            hc.hc_philox_rounds(rounds, *[C.c_uint(v) for v in ctr + key], out)
            assert list(out) == _philox_python(rounds, ctr, key)
    # and the Python restatement reproduces a Random123 vector at 10 rounds
    assert _philox_python(10, [0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def test_fast_arithmetic_restatements_stay_within_a_few_ulp(hc, golden):
    """GLH_MATH_FAST helpers against the exact path (host build of the same code): the table exp over the whole
    range of log likelihoods, and the FMA projection on the reference's projection goldens."""
    hc.hc_exp_fast.restype = C.c_double
    x = -np.concatenate((np.linspace(0, 50, 20001), np.geomspace(1e-12, 745.0, 5000), [0.0, 1e-300, 744.9, 746.5, 1e6]))
    got = np.array([hc.hc_exp_fast(C.c_double(v)) for v in x])
    want = np.exp(x)
    ok = want > 1e-290
    assert np.abs(got[ok] / want[ok] - 1).max() < 4e-15
    assert (got[~ok] >= 0).all() and (got[~ok] < 2e-290).all()
    pos = np.array([hc.hc_exp_fast(C.c_double(v)) for v in (1e-9, 0.5, 3.0)])
    np.testing.assert_allclose(pos, np.exp([1e-9, 0.5, 3.0]), rtol=4e-15)
    g = golden("g1_projection.npz")
    for cam, xyz, uv in zip(g["cams"], g["xyz"], g["uv"]):
        got = np.empty_like(uv)
        hc.hc_project_fast(p(np.ascontiguousarray(cam)), p(np.ascontiguousarray(xyz)), len(xyz), p(got))
        assert np.array_equal(np.isnan(got), np.isnan(uv))
        ok = ~np.isnan(uv[:, 0])
        np.testing.assert_allclose(got[ok], uv[ok], rtol=1e-12, atol=1e-8)


def test_host_tables_and_stage_math_under_sanitizers():
    """AddressSanitizer + UndefinedBehaviorSanitizer over the host-side tables of the C ABI (pairwise-sum plans, spline
    LU factors / inverses / basis polynomials, camera expansion) and the inline stage math shared with the kernels
    (tests/hostcheck/sanitize_main.cpp).  GPU sanitizers are not available on the pool; this is the CPU build."""
    src = os.path.join(ROOT, "tests", "hostcheck", "sanitize_main.cpp")
    exe = os.path.join(ROOT, "tests", "hostcheck", "_build", "sanitize_main")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-ffp-contract=off", "-o", exe, src], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0 and "SANITIZE_OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_cell_form_of_the_fitted_surface_equals_the_coefficient_form(hc):
    """GLH_MATH_FAST samples a fitted surface in per-cell power form (glh_math.h: spline_cell_row / spline_eval_cell):
    for every surface side 4 .. 34 (the short sides have their own basis matrices) and arguments inside, on the knots,
    on the border and outside (clamped), (a) the table the fused kernel builds and the on-the-spot conversion of the
    staged kernels are the same float64, (b) both are the exact arithmetic's coefficient form to rounding."""
    rng = np.random.default_rng(23)
    worst = 0.0
    for ho, wo in [(n, m) for n in range(4, 20) for m in (4, 5, 8, 9, 13, 20)] + [(34, 34), (12, 31), (31, 12)]:
        coef = np.ascontiguousarray(rng.standard_normal((ho, wo)) * 3.0)
        n = 400
        uv = np.empty((n, 2))
        uv[:, 0] = rng.uniform(-2.0, wo + 1.0, n)
        uv[:, 1] = rng.uniform(-2.0, ho + 1.0, n)
        uv[:40, 0] = rng.integers(0, wo, 40)           # on the knots
        uv[40:80, 1] = rng.integers(0, ho, 40)
        uv[80:90] = [[0.0, 0.0], [wo - 1.0, ho - 1.0], [1.0, 1.0], [2.0, 2.0], [wo - 2.0, ho - 2.0], [wo - 3.0, 1.5],
                     [1.999999999, 2.000000001], [wo - 1.0 - 1e-12, 0.5], [0.5, ho - 1.0 + 1e-9], [-0.0, 3.0]]
        cu0, cv0 = 100.25, -7.5
        uv_abs = np.ascontiguousarray(uv + [cu0, cv0])
        out = np.empty((n, 3))
        hc.hc_spline_cell_forms(p(coef), ho, wo, C.c_double(cv0), C.c_double(cu0), p(uv_abs), n, p(out))
        assert np.isfinite(out).all()
        np.testing.assert_array_equal(out[:, 1], out[:, 2])
        scale = np.abs(coef).max()
        worst = max(worst, np.abs(out[:, 1] - out[:, 0]).max() / scale)
    assert worst < 5e-14, worst


def test_raster_interval_is_the_clipped_searchsorted(hc):
    """find_indices of scipy's RegularGridInterpolator: clip(searchsorted(g, x) - 1, 0, n - 2).  The kernels guess the
    interval from the cell size and move by at most one (glh_math.h: raster_interval): same answer on the cell centres
    of a uniform grid (np.linspace, what Raster.x is) -- at and around every centre, between them, outside the ends, with
    centres perturbed by up to a fifth of a cell; coordinates further than a quarter cell from uniform are refused by the
    library (raster_coordinates_uniform)."""
    rng = np.random.default_rng(5)
    grids = [np.linspace(-13.7, 250.3, n) for n in (2, 3, 17, 400, 2001)] + [np.linspace(4.1e5, 4.3e5, 1999) + 0.25]
    jitter = np.linspace(0.0, 300.0, 601)
    jitter[1:-1] += rng.uniform(-0.1, 0.1, 599)  # (cells of 0.5: up to a fifth of a cell)
    grids.append(jitter)
    for g in grids:
        g = np.ascontiguousarray(g, dtype=np.float64)
        assert (np.diff(g) > 0).all()
        x = np.concatenate([g, np.nextafter(g, -np.inf), np.nextafter(g, np.inf), (g[:-1] + g[1:]) / 2,
                            rng.uniform(g[0] - 3.0, g[-1] + 3.0, 4000), [g[0] - 1e9, g[-1] + 1e9]])
        want = np.clip(np.searchsorted(g, x, side="left") - 1, 0, len(g) - 2).astype(np.int32)
        got = np.empty(len(x), dtype=np.int32)
        # (outer limits as Raster._limits makes them from cell centres: half a cell beyond the ends)
        d = (g[-1] - g[0]) / (len(g) - 1)
        lo, hi = C.c_double(g[0] - d / 2), C.c_double(g[-1] + d / 2)
        assert hc.hc_raster_uniform(p(g), len(g), lo, hi) == 1
        hc.hc_raster_interval(p(g), len(g), lo, hi, p(np.ascontiguousarray(x)), len(x), p(got))
        np.testing.assert_array_equal(got, want)
    for g in (np.sort(rng.uniform(0, 100, 300)), np.cumsum(rng.uniform(0.01, 5.0, 64) ** 3)):
        g = np.ascontiguousarray(g)
        d = (g[-1] - g[0]) / (len(g) - 1)
        assert hc.hc_raster_uniform(p(g), len(g), C.c_double(g[0] - d / 2), C.c_double(g[-1] + d / 2)) == 0


def test_count_fraction_is_the_ieee_quotient_for_every_count(hc):
    """glh_math.h: count_fraction -- k / n in three instructions (a multiplication by 1 / n and a fused residual correction)
    in the tile stage of 16-bit and float frames, where k is a pixel's cumulative count and n the tile's pixel count: bit
    for bit the IEEE division for EVERY pair 0 <= k <= n <= 65 536 (2.1e9 of them: tiles of the fused step have fewer
    than 65 536 pixels)."""
    hc.hc_count_fraction_exhaustive.restype = C.c_longlong
    assert hc.hc_count_fraction_exhaustive(1, 65536) == 0


def test_raster_window_serves_the_same_samples(hc):
    """A window of a raster around a point (glh_math.h: RasterPatch -- the fused kernel keeps one per surface in LDS): the
    samples it serves are bit for bit the raster's own, for either orientation of the array, at the raster's edges and
    corners, for rasters smaller than the window; samples outside it fall through to the raster."""
    rng = np.random.default_rng(9)
    for nx, ny, sx, sy in ((300, 200, 1, -1), (40, 50, -1, 1), (7, 9, 1, 1), (2, 2, 1, -1), (12, 13, -1, -1)):
        z = np.ascontiguousarray(rng.standard_normal((ny, nx)))
        xlim, ylim = (100.0, 100.0 + 2.0 * nx), (-50.0, -50.0 + 3.0 * ny)
        gx = np.ascontiguousarray(np.linspace(xlim[0] + 1.0, xlim[1] - 1.0, nx))
        gy = np.ascontiguousarray(np.linspace(ylim[0] + 1.5, ylim[1] - 1.5, ny))
        for cx, cy in ((xlim[0] + 0.3, ylim[0] + 0.1), (xlim[1] - 0.2, ylim[1] - 0.4), (np.mean(xlim), np.mean(ylim)),
                       (xlim[0] + 9.0, ylim[1] - 7.0)):
            near = np.column_stack((cx + rng.uniform(-6, 6, 400), cy + rng.uniform(-9, 9, 400)))
            on_nodes = np.column_stack((rng.choice(gx, 60), rng.choice(gy, 60)))
            far = np.column_stack((rng.uniform(*xlim, 100), rng.uniform(*ylim, 100)))
            xy = np.ascontiguousarray(np.vstack((near, on_nodes, far, [[cx, cy]])))
            xy[:, 0] = np.clip(xy[:, 0], *xlim)
            xy[:, 1] = np.clip(xy[:, 1], *ylim)
            vals = np.empty((len(xy), 2))
            hits = hc.hc_raster_patch(p(z), nx, ny, p(gx), p(gy), sx, sy, C.c_double(xlim[0]), C.c_double(xlim[1]),
                                      C.c_double(ylim[0]), C.c_double(ylim[1]), C.c_double(cx), C.c_double(cy), p(xy),
                                      len(xy), p(vals))
            np.testing.assert_array_equal(vals[:, 0], vals[:, 1])
            assert np.isfinite(vals).all()
            assert hits >= (300 if min(nx, ny) >= 12 else len(xy) if max(nx, ny) <= 12 else 1), hits
            # fast arithmetic (raster_bilinear_fast: reciprocal interval widths, three fused multiply-adds): the window
            # serves the raster's own fast samples bit for bit, and both stay within rounding of the exact form
            fast = np.empty((len(xy), 2))
            hc.hc_raster_patch_fast(p(z), nx, ny, p(gx), p(gy), sx, sy, C.c_double(xlim[0]), C.c_double(xlim[1]),
                                    C.c_double(ylim[0]), C.c_double(ylim[1]), C.c_double(cx), C.c_double(cy), p(xy),
                                    len(xy), p(fast))
            np.testing.assert_array_equal(fast[:, 0], fast[:, 1])
            np.testing.assert_allclose(fast[:, 0], vals[:, 0], rtol=0, atol=1e-14 * np.abs(z).max())
