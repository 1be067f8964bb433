"""Motion models (/root/reference/src/glimpse/track/motion.py).

`CartesianMotion` (motion.py:92-204) is the model the GPU path implements natively: the
Tracker reads its parameters and evolves the particles on the device.  Its methods are kept
as host NumPy conveniences with the reference's semantics (legacy `np.random` draws in the
same order) for users who call them directly; the Tracker does not use them.  The other
models of the reference (Cylindrical, Tangent*) need gridded DEMs and are listed as "next"
(SURVEY.md 8(f) rank 1).
"""
import numpy as np


class Motion:
    """Interface illustration (motion.py:13-89)."""

    def __init__(self, xy, time_unit, n=1000, vxyz_sigma=(0, 0, 0)):
        self.xy = xy
        self.time_unit = time_unit
        self.n = n
        self.vxyz_sigma = vxyz_sigma


class CartesianMotion(Motion):
    def __init__(self, xy, time_unit, dem, dem_sigma=None, n=1000, xy_sigma=(0, 0), vxyz=(0, 0, 0),
                 vxyz_sigma=(0, 0, 0), axyz=(0, 0, 0), axyz_sigma=(0, 0, 0)):
        """motion.py:121-147.  `dem` / `dem_sigma` must be numbers (constant surfaces): gridded
        rasters are "next"; `dem_sigma=None` crashes in the reference (KeyError 'buf_xsize',
        SURVEY.md 7.4 item 8), so a number is required here."""
        if not np.isscalar(dem) or dem_sigma is None or not np.isscalar(dem_sigma):
            raise NotImplementedError("CartesianMotion needs scalar dem and dem_sigma on the GPU path")
        self.xy = xy
        self.time_unit = time_unit
        self.dem = float(dem)
        self.dem_sigma = float(dem_sigma)
        self.n = int(n)
        self.xy_sigma = xy_sigma
        self.vxyz = vxyz
        self.vxyz_sigma = vxyz_sigma
        self.axyz = axyz
        self.axyz_sigma = axyz_sigma

    def params(self):
        """GLH_MOTION_LEN doubles (include/glimpse_hip.h)."""
        def v(x, n):
            return np.broadcast_to(np.asarray(x, dtype=float), (n,))
        return np.concatenate((v(self.xy, 2), v(self.xy_sigma, 2), v(self.vxyz, 3), v(self.vxyz_sigma, 3),
                               v(self.axyz, 3), v(self.axyz_sigma, 3), [self.dem, self.dem_sigma]))

    def initialize_particles(self):
        """motion.py:149-163."""
        particles = np.zeros((self.n, 6), dtype=float)
        particles[:, 0:2] = self.xy + self.xy_sigma * np.random.randn(self.n, 2)
        particles[:, 2] = self.dem
        particles[:, 2] += self.dem_sigma * np.random.randn(self.n)
        particles[:, 3:6] = self.vxyz + self.vxyz_sigma * np.random.randn(self.n, 3)
        return particles

    def evolve_particles(self, particles, dt):
        """motion.py:165-179 (in place)."""
        n = len(particles)
        time_units = dt.total_seconds() / self.time_unit.total_seconds()
        axyz = self.axyz + self.axyz_sigma * np.random.randn(n, 3)
        particles[:, 0:3] += time_units * particles[:, 3:6] + 0.5 * axyz * time_units ** 2
        particles[:, 3:6] += time_units * axyz

    def compute_log_likelihoods(self, particles):
        """motion.py:181-204."""
        ll = np.zeros(len(particles), dtype=float)
        if self.dem_sigma != 0:
            ll[:] = 1 / (2 * self.dem_sigma ** 2) * (self.dem - particles[:, 2]) ** 2
        return ll
