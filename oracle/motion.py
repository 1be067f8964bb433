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

from .raster import Raster, sample as _surf


def _surface(v):
    return v if isinstance(v, Raster) else float(v)


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
        self.dem = _surface(dem)
        self.dem_sigma = _surface(dem_sigma)
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
        particles[:, 2] = _surf(self.dem, particles[:, 0:2])
        z_sigma = _surf(self.dem_sigma, particles[:, 0:2])
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
        z = _surf(self.dem, particles[:, 0:2])
        z_sigma = _surf(self.dem_sigma, particles[:, 0:2])
        nonzero = np.nonzero(z_sigma)[0]
        log_likelihoods = np.zeros(len(particles), dtype=float)
        log_likelihoods[nonzero] = (
            1 / (2 * z_sigma[nonzero] ** 2) * (z[nonzero] - particles[nonzero, 2]) ** 2
        )
        return log_likelihoods


def _cyl_velocity(v):
    """(radius rate, theta[, z rate]) -> Cartesian components (motion.py:276-285, :479-487)."""
    cols = [v[:, 0] * np.cos(v[:, 1]), v[:, 0] * np.sin(v[:, 1])]
    if v.shape[1] == 3:
        cols.append(v[:, 2])
    return np.column_stack(cols)


def _cyl_acceleration(particles, a):
    """(radius accel, theta rate[, z accel]) -> Cartesian components (motion.py:297-307, :500-510)."""
    vx = particles[:, 3]
    vy = particles[:, 4]
    vr = np.sqrt(vx ** 2 + vy ** 2)
    cols = [a[:, 0] * (vx / vr) - vy * a[:, 1], a[:, 0] * (vy / vr) + vx * a[:, 1]]
    if a.shape[1] == 3:
        cols.append(a[:, 2])
    return np.column_stack(cols)


class CylindricalMotion(CartesianMotion):
    """motion.py:207-311.  Constructor arguments are named like CartesianMotion's: `vxyz` etc. hold
    (radius rate, theta, dz/dt) = the reference's vrthz / vrthz_sigma / arthz / arthz_sigma."""

    KIND = 1

    def initialize_particles(self, normals=None):
        n = self.n
        if normals is None:
            normals = np.column_stack((np.random.randn(n, 2), np.random.randn(n), np.random.randn(n, 3)))
        particles = np.zeros((n, 6), dtype=float)
        particles[:, 0:2] = self.xy + self.xy_sigma * normals[:, 0:2]
        particles[:, 2] = _surf(self.dem, particles[:, 0:2])
        particles[:, 2] += _surf(self.dem_sigma, particles[:, 0:2]) * normals[:, 2]
        particles[:, 3:6] = _cyl_velocity(self.vxyz + self.vxyz_sigma * normals[:, 3:6])
        return particles, normals

    def evolve_particles(self, particles, time_units, normals=None):
        n = len(particles)
        if normals is None:
            normals = np.random.randn(n, 3)
        axyz = _cyl_acceleration(particles, self.axyz + self.axyz_sigma * normals)
        particles[:, 0:3] += time_units * particles[:, 3:6] + 0.5 * axyz * time_units ** 2
        particles[:, 3:6] += time_units * axyz
        return normals


class TangentCartesianMotion:
    """motion.py:314-412 with constant surfaces.  `vxy`, `vxy_sigma`, `axy`, `axy_sigma` are 2-vectors."""

    KIND = 2
    CYL = False

    def __init__(self, xy, time_unit=1.0, dem=0.0, dem_sigma=0.0, n=1000, xy_sigma=(0, 0), vxy=(0, 0),
                 vxy_sigma=(0, 0), axy=(0, 0), axy_sigma=(0, 0), slope_sigma=0.0):
        self.xy = np.asarray(xy, dtype=float)
        self.time_unit = time_unit
        self.dem, self.dem_sigma, self.n = _surface(dem), _surface(dem_sigma), int(n)
        self.xy_sigma = np.asarray(xy_sigma, dtype=float)
        self.vxy, self.vxy_sigma = np.asarray(vxy, dtype=float), np.asarray(vxy_sigma, dtype=float)
        self.axy, self.axy_sigma = np.asarray(axy, dtype=float), np.asarray(axy_sigma, dtype=float)
        self.slope_sigma = float(slope_sigma)

    def initialize_particles(self, normals=None):
        """`normals` (n, 6) = [randn(n,2) | randn(n) | randn(n,2) | unused]."""
        n = self.n
        if normals is None:
            normals = np.column_stack((np.random.randn(n, 2), np.random.randn(n), np.random.randn(n, 2), np.zeros(n)))
        particles = np.zeros((n, 6), dtype=float)
        particles[:, 0:2] = self.xy + self.xy_sigma * normals[:, 0:2]
        z_offsets = _surf(self.dem_sigma, particles[:, 0:2]) * normals[:, 2]
        particles[:, 2] = _surf(self.dem, particles[:, 0:2]) + z_offsets
        v = self.vxy + self.vxy_sigma * normals[:, 3:5]
        particles[:, 3:5] = _cyl_velocity(v) if self.CYL else v
        return particles, normals

    def evolve_particles(self, particles, time_units, normals=None):
        """`normals` (n, 3) = [randn(n,2) | randn(n)]."""
        n = len(particles)
        if normals is None:
            normals = np.column_stack((np.random.randn(n, 2), np.random.randn(n)))
        a = self.axy + self.axy_sigma * normals[:, 0:2]
        axy = _cyl_acceleration(particles, a) if self.CYL else a
        dxy = time_units * particles[:, 3:5] + 0.5 * axy * time_units ** 2
        z_offsets = particles[:, 2] - _surf(self.dem, particles[:, 0:2])
        z_offsets += self.slope_sigma * normals[:, 2] * (dxy ** 2).sum(axis=1) ** 0.5
        particles[:, 0:2] += dxy
        particles[:, 2] = _surf(self.dem, particles[:, 0:2]) + z_offsets
        particles[:, 3:5] += time_units * axy
        return normals

    def compute_log_likelihoods(self, particles):
        """Base Motion.compute_log_likelihoods (motion.py:76-89): no term."""
        return None


class TangentCylindricalMotion(TangentCartesianMotion):
    """motion.py:415-522 (vxy etc. hold the reference's vrth / vrth_sigma / arth / arth_sigma)."""

    KIND = 3
    CYL = True
