"""User-defined motion models for the tests: plain objects with the interface `glimpse.Tracker` asks of a motion
model (track/motion.py:13-89 -- initialize_particles, evolve_particles, optionally compute_log_likelihoods), written for
these tests.  tools/make_golden.py runs the REFERENCE Tracker on them (g16), tests/test_gpu_api.py runs
glimpse_amd.Tracker on the same classes: same np.random stream in, same tracks out."""
import numpy as np


class DriftMotion:
    """Constant velocity drawn once, no acceleration, no likelihood term (compute_log_likelihoods absent)."""

    def __init__(self, xy, time_unit, n=150, v=(0.15, 0.0), v_sigma=0.1, xy_sigma=0.15):
        self.xy, self.time_unit, self.n, self.v, self.v_sigma, self.xy_sigma = xy, time_unit, n, v, v_sigma, xy_sigma

    def initialize_particles(self):
        p = np.zeros((self.n, 6))
        p[:, 0:2] = np.asarray(self.xy) + self.xy_sigma * np.random.randn(self.n, 2)
        p[:, 3:5] = np.asarray(self.v) + self.v_sigma * np.random.randn(self.n, 2)
        return p

    def evolve_particles(self, particles, dt):
        tau = dt.total_seconds() / self.time_unit.total_seconds()
        particles[:, 0:3] += tau * particles[:, 3:6]


class SpeedPriorMotion(DriftMotion):
    """Random-walk velocity with a prior on the speed: compute_log_likelihoods returns an array."""

    def __init__(self, *args, speed=0.15, speed_sigma=0.03, walk=0.03, **kw):
        super().__init__(*args, **kw)
        self.speed, self.speed_sigma, self.walk = speed, speed_sigma, walk

    def evolve_particles(self, particles, dt):
        tau = dt.total_seconds() / self.time_unit.total_seconds()
        particles[:, 3:5] += self.walk * abs(tau) ** 0.5 * np.random.randn(len(particles), 2)
        particles[:, 0:3] += tau * particles[:, 3:6]

    def compute_log_likelihoods(self, particles):
        speed = np.hypot(particles[:, 3], particles[:, 4])
        return 0.5 * ((speed - self.speed) / self.speed_sigma) ** 2


class NoTermMotion(DriftMotion):
    """compute_log_likelihoods present but returning None (motion.py:74-89)."""

    def compute_log_likelihoods(self, particles):
        return None


def shape_and_weight(particles, weights):
    """A `reduce_particles` function the worker processes of `track(parallel=N)` can import (not a lambda)."""
    import numpy as np

    return particles.shape, float(np.nansum(weights))
