"""`Tracks`: result container (/root/reference/src/glimpse/track/tracks.py:52-213).
Same constructor, array shapes and properties, plus the steps that follow a tracking run:
`reverse` (tracks.py:131-149), `from_multiple` (:151-191, merging e.g. a forward and a backward
run) and `average` (:193-213).  These are host-side NumPy on (tracks, times, 6) result arrays, as in
the reference; plotting / animation are out of scope."""
import numpy as np


def fuse_normals(means, sigmas, axis, comonotone=False, ignore_nan=False):
    """Precision-weighted combination of normal estimates along `axis` (what the reference does with
    helpers.sum_normals(weights=sigma**-2, normalize=True), tracks.py:185-189 and :207-213).

    With weights w_i = s_i^-2 / sum_j s_j^-2 over the entries that are present,

        mean = sum_i w_i m_i        sigma = sqrt(sum_i (w_i s_i)^2)      independent estimates (several runs)
                                    sigma = sum_i w_i s_i                fully correlated ones (`comonotone`: the
                                                                          time steps of one track; the variance of a
                                                                          sum with correlation 1 is the square of the
                                                                          summed sigmas)

    NaN marks a missing estimate and must coincide in `means` and `sigmas`; the result is missing where any
    (ignore_nan=False) or every (ignore_nan=True) entry along the axis is."""
    m = np.asarray(means, dtype=float)
    s = np.asarray(sigmas, dtype=float)
    present = ~np.isnan(m)
    if (present != ~np.isnan(s)).any():
        raise ValueError("Means and sigmas have missing values at different indices")
    if (s == 0).any():
        raise ValueError("Sigmas cannot be zero")
    with np.errstate(invalid="ignore", divide="ignore"):
        precision = np.where(present, 1.0 / (s * s), 0.0)
        w = precision * (1.0 / precision.sum(axis=axis, keepdims=True))
        mean = np.where(present, w * m, 0.0).sum(axis=axis)
        spread = np.where(present, w * s, 0.0)
        sigma = spread.sum(axis=axis) if comonotone else np.sqrt((spread * spread).sum(axis=axis))
    gone = (~present).all(axis=axis) if ignore_nan else (~present).any(axis=axis)
    mean[gone] = np.nan
    sigma[gone] = np.nan
    return mean, sigma


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
        """Fuse several runs over the same timesteps (e.g. forward + backward; tracks.py:151-191): the product of
        the runs' independent normal distributions per track, time and state component, i.e. precisions add,

            mean = sum_r m_r / s_r^2 / sum_r 1 / s_r^2,    sigma = sqrt(sum_r (w_r s_r)^2),  w_r = s_r^-2 / sum s^-2

        over the runs that have a value; the result is missing where any run (ignore_nan=False) or every run
        (ignore_nan=True) is missing."""
        runs = list(runs)
        head = runs[0]
        for run in runs[1:]:
            if len(run.datetimes) != len(head.datetimes) or any(a != b for a, b in zip(run.datetimes, head.datetimes)):
                raise ValueError("Datetimes are not equal for all runs")
            if run.time_unit != head.time_unit:
                raise ValueError(f"Time units are not equal for all runs: {set(r.time_unit for r in runs)}")
        m = np.stack([np.asarray(run.means, dtype=float) for run in runs], axis=-1)
        sd = np.stack([np.asarray(run.sigmas, dtype=float) for run in runs], axis=-1)
        means, sigmas = fuse_normals(m, sd, axis=m.ndim - 1, ignore_nan=ignore_nan)
        return cls(datetimes=head.datetimes, time_unit=head.time_unit, means=means, sigmas=sigmas)

    def average(self, ignore_nan=False):
        """Time-averaged mean and sigma of each track (tracks.py:193-213): precision weights, the time steps of a
        track taken as fully correlated."""
        return fuse_normals(self.means, self.sigmas, axis=1, comonotone=True, ignore_nan=ignore_nan)

    # ---- views of the state vector (x, y, z, vx, vy, vz) ----------------------------------------
    def _spread(self, lo, hi):
        """Standard deviations of components [lo, hi): `sigmas`, or the diagonal of `covariances`."""
        if self.sigmas is not None:
            return self.sigmas[..., lo:hi]
        if self.covariances is not None:
            return np.sqrt(np.diagonal(self.covariances, axis1=-2, axis2=-1)[..., lo:hi])
        return None

    xyz = property(lambda self: self.means[..., 0:3], doc="positions (tracks, times, 3)")
    vxyz = property(lambda self: self.means[..., 3:6], doc="velocities (tracks, times, 3)")
    xyz_sigma = property(lambda self: self._spread(0, 3))
    vxyz_sigma = property(lambda self: self._spread(3, 6))

    @property
    def endpoints(self):
        """tracks.py:116-123: (mask of the tracks with any valid time step, index of their first valid step, of
        their last)."""
        ok = ~np.isnan(self.means[..., 0])
        has = ok.any(axis=1)
        steps = np.arange(ok.shape[1])
        first = np.where(ok, steps, ok.shape[1]).min(axis=1)
        last = np.where(ok, steps, -1).max(axis=1)
        return has, first[has], last[has]

    @property
    def success(self):
        """Per track: True where no error was recorded (None when the run kept no error list)."""
        if self.errors is None:
            return None
        return np.fromiter((e is None for e in self.errors), dtype=bool, count=len(self.errors))
