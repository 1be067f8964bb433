"""`Tracks`: result container (/root/reference/src/glimpse/track/tracks.py:52-129).
Same constructor, array shapes and properties; merging / plotting are out of scope."""
import numpy as np


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
