"""Oracle: particle resampling and moments (test infrastructure only).

Follows /root/reference/src/glimpse/track/tracker.py:
  * systematic / stratified / residual / choice   tracker.py:168-209
  * gather of particles AND weights               tracker.py:222-223
  * particle_mean / compute_particle_sigma / particle_covariance  tracker.py:72-104
"""
import numpy as np


def systematic(weights, u):
    """tracker.py:168-176 with the single `np.random.random()` draw passed as u."""
    n = len(weights)
    weights = weights / weights.sum()
    positions = (np.arange(n) + u) * (1 / n)
    cumulative_weight = np.cumsum(weights)
    return np.searchsorted(cumulative_weight, positions)


def stratified(weights, u):
    """tracker.py:178-186 with `np.random.random(n)` passed as u (n,)."""
    n = len(weights)
    weights = weights / weights.sum()
    positions = (np.arange(n) + u) * (1 / n)
    cumulative_weight = np.cumsum(weights)
    return np.searchsorted(cumulative_weight, positions)


def residual(weights, u):
    """tracker.py:188-203; `u` = the `np.random.random(n - sum(reps))` draw (or callable)."""
    n = len(weights)
    weights = weights / weights.sum()
    repetitions = (n * weights).astype(int)
    initial_indexes = np.repeat(np.arange(n), repetitions)
    residuals = weights - repetitions
    residuals *= 1 / residuals.sum()
    cumulative_sum = np.cumsum(residuals)
    cumulative_sum[-1] = 1.0
    m = n - len(initial_indexes)
    draws = u(m) if callable(u) else np.asarray(u)[:m]
    additional_indexes = np.searchsorted(cumulative_sum, draws)
    return np.hstack((initial_indexes, additional_indexes))


def choice(weights, u):
    """tracker.py:205-209: `np.random.choice(np.arange(n), size=(n,), replace=True, p=weights)` with the n
    uniforms it draws passed as u.  The legacy `RandomState.choice` (numpy/random/mtrand.pyx, the
    replace=True / p given branch) computes cdf = p.cumsum(); cdf /= cdf[-1];
    idx = cdf.searchsorted(random_sample(n), side='right')."""
    weights = weights / weights.sum()
    cdf = np.cumsum(weights)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, np.asarray(u, dtype=float), side="right")


METHODS = {"systematic": systematic, "stratified": stratified, "residual": residual, "choice": choice}


def particle_mean(particles, weights):
    """tracker.py:76."""
    return np.average(particles, weights=weights, axis=0)


def particle_sigma(particles, weights, mean):
    """tracker.py:101-104."""
    variance = np.average((particles - mean) ** 2, weights=weights, axis=0)
    return np.sqrt(variance)


def particle_covariance(particles, weights):
    """tracker.py:82."""
    return np.cov(particles.T, aweights=weights, ddof=0)


NUMPY_BUFSIZE = 8192  # np.getbufsize(): the reduction's inner loop sees chunks of this many items


def numpy_pairwise_sum_plan(n):
    """Leaf blocks (offset, length, chunk) of `np.sum` over n contiguous float64 items.

    `np.add.reduce` hands the inner loop (`DOUBLE_add` -> `DOUBLE_pairwise_sum`)
    chunks of NUMPY_BUFSIZE items and adds the chunk results to the running
    total in order.  Inside a chunk, blocks of <= 128 items are summed with 8
    interleaved accumulators; longer ranges are split at n/2 rounded down to a
    multiple of 8 and the halves added.  The HIP resample kernel reproduces
    this tree so that `weights.sum()` (tracker.py:172) is bit-exact.
    """
    leaves = []

    def rec(off, m, chunk):
        if m <= 128:
            leaves.append((off, m, chunk))
            return
        n2 = m // 2
        n2 -= n2 % 8
        rec(off, n2, chunk)
        rec(off + n2, m - n2, chunk)

    for ci, s in enumerate(range(0, n, NUMPY_BUFSIZE)):
        rec(s, min(NUMPY_BUFSIZE, n - s), ci)
    return leaves


def _pairwise_leaf(x):
    m = len(x)
    if m < 8:
        res = 0.0
        for v in x:
            res += v
        return res
    r = [x[j] for j in range(8)]
    i = 8
    while i < m - (m % 8):
        for j in range(8):
            r[j] += x[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < m:
        res += x[i]
        i += 1
    return res


def _pairwise_rec(x):
    m = len(x)
    if m <= 128:
        return _pairwise_leaf(x)
    n2 = m // 2
    n2 -= n2 % 8
    return _pairwise_rec(x[:n2]) + _pairwise_rec(x[n2:])


def numpy_pairwise_sum(a):
    """Bit-exact emulation of `a.sum()` for a contiguous float64 vector (see the plan)."""
    a = np.asarray(a, dtype=float)
    acc = None
    for s in range(0, len(a), NUMPY_BUFSIZE):
        part = _pairwise_rec(a[s : s + NUMPY_BUFSIZE])
        acc = part if acc is None else acc + part
    return 0.0 if acc is None else acc
