"""The device random streams (GLH_RNG_PHILOX: Philox4x32-7 + float32 Box-Muller) every bench number runs on.

(1) The evolve noise itself, read back exactly: with zero state, zero mean acceleration and unit acceleration sigma one
    evolve step leaves v = n, so the velocities ARE the normals.  10^6 particles x 3 components: moments, a
    Kolmogorov-Smirnov test against N(0, 1), tail counts, and correlations along every index of the counter
    (particle, component, point, step).
(2) The initialisation noise the same way (six normals per particle).
(3) Distributional parity of a tracking run: device draws against the oracle on np.random -- no bias between the
    posteriors over 256 tracks, and their disagreement is the disagreement of two oracle runs on different seeds."""
import numpy as np
import pytest
import scipy.stats

pytestmark = pytest.mark.gpu


def _evolve_noise(lib, P, N, steps, seed, math):
    """(steps, P, N, 3) evolve normals of a context, through the public ABI."""
    params = np.zeros((P, lib.MOTION_LEN))
    params[:, 13:16] = 1.0  # axyz_sigma: acceleration = 0 + 1 * n
    out = []
    with lib.Context(P, N, 1, max_frames=steps + 1) as ctx:
        ctx.observer_init(0, 2, 64, 64, 1, 0.3)
        ctx.begin_sequence(P, N, (15, 15))
        ctx.set_motion_cartesian(params)
        ctx.set_math(math)
        for s in range(1, steps + 1):
            ctx.set_particles(np.zeros((P, N, 6)))
            ctx.evolve(1.0, seed=seed, step=s)
            p = ctx.get_particles()
            np.testing.assert_array_equal(p[..., 0:3], 0.5 * p[..., 3:6])  # x = 0 + 1 * 0 + 0.5 * n * 1
            out.append(p[..., 3:6].copy())
    return np.stack(out)


def _check_standard_normal(z, label):
    n = z.size
    se = 1 / np.sqrt(n)
    assert abs(z.mean()) < 5 * se, (label, z.mean())
    assert abs(z.var() - 1) < 5 * np.sqrt(2) * se, (label, z.var())
    assert abs(scipy.stats.skew(z)) < 5 * np.sqrt(6) * se, (label, scipy.stats.skew(z))
    assert abs(scipy.stats.kurtosis(z)) < 5 * np.sqrt(24) * se, (label, scipy.stats.kurtosis(z))
    ks = scipy.stats.kstest(z, "norm")
    assert ks.pvalue > 1e-4, (label, ks)
    # tails: counts beyond 3 and 4 sigma against the binomial expectation (Box-Muller on float32 reaches |z| <= 6.6)
    for k in (3.0, 4.0):
        p = 2 * scipy.stats.norm.sf(k)
        count = int((np.abs(z) > k).sum())
        assert abs(count - n * p) < 6 * np.sqrt(n * p) + 3, (label, k, count, n * p)
    assert np.abs(z).max() < 6.7


@pytest.mark.parametrize("math", ["fast", "exact"])
def test_evolve_noise_is_standard_normal_and_uncorrelated(math):
    from glimpse_amd import _lib as lib

    P, N, S = 100, 5000, 2  # 10^6 particles
    z = _evolve_noise(lib, P, N, S, seed=20240607, math=math)  # (S, P, N, 3)
    _check_standard_normal(z.ravel(), "all")
    for k in range(3):
        _check_standard_normal(z[..., k].ravel(), f"component {k}")
    n = z[..., 0].size
    lim = 5 / np.sqrt(n)

    def corr(a, b):
        return float(np.corrcoef(a.ravel(), b.ravel())[0, 1])

    assert abs(corr(z[:, :, :-1, :], z[:, :, 1:, :])) < lim      # neighbouring particles (counter word 0)
    assert abs(corr(z[:, :-1], z[:, 1:])) < lim                    # neighbouring points (counter word 1)
    assert abs(corr(z[0], z[1])) < lim                             # consecutive steps (counter word 2)
    for a, b in ((0, 1), (0, 2), (1, 2)):                          # components of one Philox block / Box-Muller pair
        assert abs(corr(z[..., a], z[..., b])) < 3 * lim
        assert abs(corr(z[..., a] ** 2, z[..., b] ** 2)) < 3 * lim  # (uncorrelated AND no shared radius)
    # the streams are keyed on the global point index: a shard sees the same numbers
    with lib.Context(10, N, 1, max_frames=3) as ctx:
        ctx.observer_init(0, 2, 64, 64, 1, 0.3)
        ctx.begin_sequence(10, N, (15, 15))
        params = np.zeros((10, lib.MOTION_LEN))
        params[:, 13:16] = 1.0
        ctx.set_motion_cartesian(params)
        ctx.set_math(math)
        ctx.set_point_offset(40)
        ctx.set_particles(np.zeros((10, N, 6)))
        ctx.evolve(1.0, seed=20240607, step=1)
        np.testing.assert_array_equal(ctx.get_particles()[..., 3:6], z[0, 40:50])


def test_initialisation_noise_is_standard_normal():
    from glimpse_amd import _lib as lib

    P, N = 40, 5000
    params = np.zeros((P, lib.MOTION_LEN))
    params[:, 2:4] = 1.0    # xy_sigma
    params[:, 7:10] = 1.0   # vxyz_sigma
    params[:, 17] = 1.0     # dem_sigma: z = 0 + 1 * n
    with lib.Context(P, N, 1, max_frames=2) as ctx:
        ctx.observer_init(0, 2, 64, 64, 1, 0.3)
        ctx.begin_sequence(P, N, (15, 15))
        ctx.set_motion_cartesian(params)
        ctx.init_particles(seed=99)
        z = ctx.get_particles()  # xy = n0 n1, z = n2, vxyz = n3 n4 n5
    _check_standard_normal(z.ravel(), "init")
    c = np.corrcoef(z.reshape(-1, 6).T)
    assert np.abs(c - np.eye(6)).max() < 5 / np.sqrt(P * N)


def test_device_draws_give_the_oracle_posterior_distribution(golden):
    """256 tracks of the miniature configuration 2 (the g8_c2mini scene: 256^2 frames, k1-k3, N = 200): the device
    stream against the oracle on np.random.  Neither is `right' track by track (different draws); as estimators of
    the same posterior they must not differ on average, and must scatter like two oracle runs scatter."""
    from glimpse_amd import _lib as lib
    from oracle import motion as omotion
    from oracle import tracker as otracker

    g = golden("g8_c2mini.npz")
    frames, cams = g["obs0_frames"], g["obs0_cams"]
    T = len(frames)
    N = int(g["n_particles"][0])
    tile = tuple(int(v) for v in g["tile_size"])
    base = g["params"][0]
    rng = np.random.default_rng(5)
    P = 256
    params = np.tile(base, (P, 1))
    params[:, 0:2] = rng.uniform(-6.0, 6.0, (P, 2))  # 256 seeds in the middle of the golden's scene (same texture, motion)
    with lib.Context(P, N, 1, max_tile=31, max_search_dim=128, max_frames=T) as ctx:
        ctx.observer_init(0, T, frames.shape[2], frames.shape[1], 1, float(g["sigmas"][0]))
        ctx.observer_set_cameras(0, cams)
        for i, f in enumerate(frames):
            ctx.observer_upload_frame(0, i, f)
        ctx.begin_sequence(P, N, tile)
        ctx.set_motion_cartesian(params)
        ctx.set_math("fast")
        ctx.set_frame(0)
        ctx.init_particles(seed=1)
        ctx.init_templates(0, 0)
        ctx.record_moments(0)
        ctx.track(list(range(1, T)), list(np.diff(g["datetimes_days"])), [[i] for i in range(1, T)], seed=1)
        dev = ctx.get_moments(0, T)[-1]  # (P, 12) at the last frame
        ok = (ctx.point_status() == 0) & (ctx.observer_status()[0] == lib.OBS_OK)
    observers = [otracker.Observer(list(frames), cams, float(g["sigmas"][0]))]
    matching = np.arange(T)[:, None]
    taus = np.diff(g["datetimes_days"])

    def oracle_run(seed):
        np.random.seed(seed)
        out = np.full((P, 12), np.nan)
        models = [omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13],
                                          axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=N) for q in params]
        res = otracker.track(models, observers, matching, taus, tile_size=tile)
        for p in range(P):
            if res["errors"][p] is None:
                out[p, 0:6], out[p, 6:12] = res["means"][p][-1], res["sigmas"][p][-1]
        return out

    a, b = oracle_run(11), oracle_run(12)
    good = ok & np.isfinite(a).all(axis=1) & np.isfinite(b).all(axis=1)
    assert good.sum() >= 0.9 * P
    n = int(good.sum())
    for k in (0, 1, 3, 4):  # x, y, vx, vy (z and vz carry no noise in this configuration)
        d_dev, d_ora = (dev[good, k] - a[good, k]), (b[good, k] - a[good, k])
        # robust scatter: with 200 particles a track now and then locks onto a neighbouring correlation peak (in the
        # oracle as on the device), and one such track dominates a root mean square
        mad_dev, mad_ora = np.median(np.abs(d_dev)), np.median(np.abs(d_ora))
        assert 0.75 < mad_dev / mad_ora < 1.3, (k, mad_dev, mad_ora)
        p95_dev, p95_ora = np.percentile(np.abs(d_dev), 95), np.percentile(np.abs(d_ora), 95)
        assert 0.7 < p95_dev / p95_ora < 1.4, (k, p95_dev, p95_ora)
        # no bias: the median difference is zero within 4 standard errors of a median
        assert abs(np.median(d_dev)) < 4 * 1.2533 * 1.4826 * mad_dev / np.sqrt(n), (k, np.median(d_dev), mad_dev)
        assert (np.abs(d_dev) > 10 * p95_ora).mean() < 0.02  # outliers stay rare
    # the posterior widths agree too
    for k in (6, 7, 9, 10):
        ratio = np.median(dev[good, k]) / np.median(a[good, k])
        assert 0.9 < ratio < 1.1, (k, ratio)
