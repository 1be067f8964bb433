"""`Tracks`: result container (/root/reference/src/glimpse/track/tracks.py:52-213).
Same constructor, array shapes and properties, plus the steps that follow a tracking run:
`reverse` (tracks.py:131-149), `from_multiple` (:151-191, merging e.g. a forward and a backward
run) and `average` (:193-213).  These are host-side NumPy on (tracks, times, 6) result arrays, as in
the reference; plotting / animation are out of scope."""
import warnings as _warnings

import numpy as np


def combine_normals(means, sigmas, weights=None, normalize=False, correlation=0, axis=None, keepdims=False,
                    ignore_nan=False):
    """Mean and standard deviation of a weighted sum of normal variables along `axis`
    (helpers.sum_normals, helpers.py:523-610; linear propagation of uncertainty with one common
    correlation coefficient):

        m = sum_i w_i m_i          var = sum_i w_i^2 s_i^2 + 2 rho sum_{i<j} w_i w_j s_i s_j

    NaNs must coincide in `means` and `sigmas`; they are skipped in the sums and make the result NaN
    where any (ignore_nan=False) or all (ignore_nan=True) inputs are missing.  `normalize` rescales
    the weights to sum to one over the non-missing inputs."""
    means = np.asarray(means, dtype=float)
    sigmas = np.asarray(sigmas, dtype=float)
    missing = np.isnan(means)
    if (missing != np.isnan(sigmas)).any():
        raise ValueError("Means and sigmas have missing values at different indices")
    if (sigmas == 0).any():
        raise ValueError("Sigmas cannot be zero")
    w = np.ones(means.shape) if weights is None else np.asarray(weights, dtype=float)
    if normalize:
        with _warnings.catch_warnings():
            _warnings.simplefilter("ignore", RuntimeWarning)
            w = w * (1 / np.nansum(w * ~missing, axis=axis, keepdims=True))
    total = np.nansum(w * means, axis=axis, keepdims=True)
    var = np.nansum(w ** 2 * sigmas ** 2, axis=axis, keepdims=True)
    gone = missing.all(axis=axis, keepdims=True) if ignore_nan else missing.any(axis=axis, keepdims=True)
    total[gone] = np.nan
    var[gone] = np.nan
    if correlation:
        n = means.size if axis is None else means.shape[axis]
        i, j = np.triu_indices(n=n, k=1)
        ws = w * sigmas if axis is not None else (w * sigmas).ravel()
        ax = 0 if axis is None else axis
        cross = np.take(ws, i, axis=ax) * np.take(ws, j, axis=ax)
        if axis is None:
            extra = np.nansum(correlation * cross)
        else:
            extra = np.nansum(correlation * cross, axis=axis, keepdims=True)
        var = var + 2 * extra
    sd = np.sqrt(var)
    if not keepdims:
        if axis is None:
            return total.reshape(-1)[0], sd.reshape(-1)[0]
        total, sd = np.squeeze(total, axis=axis), np.squeeze(sd, axis=axis)
    return total, sd


class Tracks:
    def __init__(self, datetimes, time_unit, means, sigmas=None, covariances=None, particles=None, weights=None,
                 tracker=None, images=None, params=None, errors=None, warnings=None):
        def stack(x):
            if np.iterable(x) and not isinstance(x, np.ndarray):
                return np.stack(x, axis=0)
            return x

        self.datetimes = np.asarray(datetimes)
        self.time_unit = time_unit
        self.means = stack(means)
        self.sigmas = stack(sigmas)
        self.covariances = stack(covariances)
        self.particles = stack(particles)
        self.weights = stack(weights)
        self.tracker = tracker
        self.images = images if images is None else np.asarray(images)
        self.params = params
        self.errors = errors if errors is None else np.asarray(errors, dtype=object)
        self.warnings = warnings if warnings is None else np.asarray(warnings, dtype=object)

    def reverse(self):
        """Reverse the temporal order in place (tracks.py:131-149)."""
        self.datetimes = self.datetimes[::-1]
        for key in ("means", "sigmas", "covariances", "particles", "weights", "images"):
            value = getattr(self, key)
            if value is not None:
                setattr(self, key, value[::-1] if value.ndim == 1 or key == "images" else value[:, ::-1, ...])

    @classmethod
    def from_multiple(cls, runs, ignore_nan=False):
        """Merge runs over identical timesteps (tracks.py:151-191): per time step, the inverse-variance
        weighted average of the runs' distributions, assumed uncorrelated."""
        runs = list(runs)
        datetimes = {tuple(run.datetimes) for run in runs}
        if len(datetimes) != 1:
            raise ValueError("Datetimes are not equal for all runs")
        time_unit = {run.time_unit for run in runs}
        if len(time_unit) != 1:
            raise ValueError(f"Time units are not equal for all runs: {time_unit}")
        means = np.stack([run.means for run in runs], axis=3)
        sigmas = np.stack([run.sigmas for run in runs], axis=3)
        means, sigmas = combine_normals(means, sigmas, weights=sigmas ** -2, normalize=True, correlation=0, axis=3,
                                        ignore_nan=ignore_nan)
        return cls(datetimes=datetimes.pop(), time_unit=time_unit.pop(), means=means, sigmas=sigmas)

    def average(self, ignore_nan=False):
        """Time-averaged mean and sigma of each track (tracks.py:193-213): inverse-variance weights,
        time steps assumed fully correlated."""
        return combine_normals(self.means, self.sigmas, weights=self.sigmas ** -2, normalize=True, correlation=1,
                               axis=1, ignore_nan=ignore_nan)

    @property
    def xyz(self):
        return self.means[:, :, 0:3]

    @property
    def vxyz(self):
        return self.means[:, :, 3:6]

    @property
    def xyz_sigma(self):
        if self.sigmas is not None:
            return self.sigmas[:, :, 0:3]
        if self.covariances is not None:
            return np.sqrt(self.covariances[:, :, (0, 1, 2), (0, 1, 2)])

    @property
    def vxyz_sigma(self):
        if self.sigmas is not None:
            return self.sigmas[:, :, 3:6]
        if self.covariances is not None:
            return np.sqrt(self.covariances[:, :, (3, 4, 5), (3, 4, 5)])

    @property
    def endpoints(self):
        valid = ~np.isnan(self.means[:, :, 0])
        first = np.argmax(valid, axis=1)
        last = valid.shape[1] - 1 - np.argmax(valid[:, ::-1], axis=1)
        first_valid = valid[np.arange(len(first)), first]
        return first_valid, first[first_valid], last[first_valid]

    @property
    def success(self):
        if self.errors is not None:
            return np.array([error is None for error in self.errors])
