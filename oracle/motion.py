"""Oracle: CartesianMotion (test infrastructure only).

Follows /root/reference/src/glimpse/track/motion.py:92-204 with scalar (constant)
`dem` / `dem_sigma`, which the reference wraps in infinite 1x1 rasters
(motion.py:136-141) sampled as constants (raster.py:1021-1026).

Random numbers are drawn from the legacy global `np.random` stream in the
reference's order: init `randn(n,2)`, `randn(n)`,
`randn(n,3)`; each evolve `randn(n,3)`.  A `draws` recorder can be passed so
the very same numbers can be fed to the HIP path.
"""
import numpy as np


class CartesianMotion:
    def __init__(
        self,
        xy,
        time_unit=1.0,
        dem=0.0,
        dem_sigma=0.0,
        n=1000,
        xy_sigma=(0, 0),
        vxyz=(0, 0, 0),
        vxyz_sigma=(0, 0, 0),
        axyz=(0, 0, 0),
        axyz_sigma=(0, 0, 0),
    ):
        self.xy = np.asarray(xy, dtype=float)
        self.time_unit = time_unit
        self.dem = float(dem)
        self.dem_sigma = float(dem_sigma)
        self.n = int(n)
        self.xy_sigma = np.asarray(xy_sigma, dtype=float)
        self.vxyz = np.asarray(vxyz, dtype=float)
        self.vxyz_sigma = np.asarray(vxyz_sigma, dtype=float)
        self.axyz = np.asarray(axyz, dtype=float)
        self.axyz_sigma = np.asarray(axyz_sigma, dtype=float)

    def params17(self):
        """[xy2, xy_sigma2, vxyz3, vxyz_sigma3, axyz3, axyz_sigma3, dem, dem_sigma]  (18)."""
        return np.concatenate(
            (self.xy, self.xy_sigma, self.vxyz, self.vxyz_sigma, self.axyz,
             self.axyz_sigma, [self.dem, self.dem_sigma])
        )

    def initialize_particles(self, normals=None):
        """motion.py:149-163.  `normals` (n, 6) = [randn(n,2) | randn(n) | randn(n,3)]."""
        n = self.n
        if normals is None:
            normals = np.column_stack(
                (np.random.randn(n, 2), np.random.randn(n), np.random.randn(n, 3))
            )
        particles = np.zeros((n, 6), dtype=float)
        particles[:, 0:2] = self.xy + self.xy_sigma * normals[:, 0:2]
        particles[:, 2] = np.full(n, self.dem)
        z_sigma = np.full(n, self.dem_sigma)
        particles[:, 2] += z_sigma * normals[:, 2]
        particles[:, 3:6] = self.vxyz + self.vxyz_sigma * normals[:, 3:6]
        return particles, normals

    def evolve_particles(self, particles, time_units, normals=None):
        """motion.py:165-179 (in place).  `normals` (n, 3) = randn(n, 3)."""
        n = len(particles)
        if normals is None:
            normals = np.random.randn(n, 3)
        axyz = self.axyz + self.axyz_sigma * normals
        particles[:, 0:3] += time_units * particles[:, 3:6] + 0.5 * axyz * time_units ** 2
        particles[:, 3:6] += time_units * axyz
        return normals

    def compute_log_likelihoods(self, particles):
        """motion.py:181-204."""
        z = np.full(len(particles), self.dem)
        z_sigma = np.full(len(particles), self.dem_sigma)
        nonzero = np.nonzero(z_sigma)[0]
        log_likelihoods = np.zeros(len(particles), dtype=float)
        log_likelihoods[nonzero] = (
            1 / (2 * z_sigma[nonzero] ** 2) * (z[nonzero] - particles[nonzero, 2]) ** 2
        )
        return log_likelihoods
