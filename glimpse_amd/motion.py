"""Motion models (/root/reference/src/glimpse/track/motion.py).

`CartesianMotion` (motion.py:92-204), `CylindricalMotion` (:207-311), `TangentCartesianMotion`
(:314-412) and `TangentCylindricalMotion` (:415-522) with constant or gridded (`glimpse_amd.Raster`)
`dem` / `dem_sigma` surfaces: the Tracker reads their parameters (`params_table`) and initialises /
evolves the particles on the device.  Their methods are host NumPy code with the reference's semantics
(legacy `np.random` draws in the same order) for users who call them directly and for the per-track
loop (`Tracker._track_custom`: residual resampling on the np.random stream); the batched Tracker does
not use them.  `Motion` is the reference's minimal model / interface statement (motion.py:13-89).
"""
import itertools
import operator

import numpy as np

from .raster import Raster

_CARTESIAN_VECTORS = operator.attrgetter("xy", "xy_sigma", "vxyz", "vxyz_sigma", "axyz", "axyz_sigma")
_CARTESIAN_SURFACES = operator.attrgetter("dem", "dem_sigma")


def _surface(value, name, cls):
    """dem / dem_sigma: a number (constant surface) or a glimpse_amd.Raster."""
    if isinstance(value, Raster):
        return value
    if value is None or not np.isscalar(value):
        raise NotImplementedError(f"{cls}: {name} must be a number or a glimpse_amd.Raster")
    return float(value)


def _sample(surface, xy):
    """Raster.sample at points, or the constant (an infinite 1 x 1 raster in the reference)."""
    if isinstance(surface, Raster):
        return surface.sample(xy)
    return np.full(len(xy), surface)


def _columns(values, width):
    """(len(values), width) float64 from per-model parameters (sequences of `width` numbers, or scalars that broadcast)."""
    try:
        a = np.concatenate(values)
        if a.size == len(values) * width:
            return a.reshape(len(values), width).astype(np.float64, copy=False)
    except (ValueError, TypeError):
        pass
    return np.array([np.broadcast_to(np.asarray(v, dtype=np.float64), (width,)) for v in values])


def params_table(models):
    """[P][GLH_MOTION_FULL_LEN] table of glh_set_motion for a list of motion models: `fill_params` of every model,
    column by column when all of them are Cartesian / Cylindrical models (thousands of rows per run: one NumPy
    concatenation per parameter instead of a dozen slice assignments per model)."""
    if isinstance(models, ModelBlock):
        return models.table
    P = len(models)
    table = np.zeros((P, 24))
    kinds = set(map(type, models))
    if P and kinds == {CartesianMotion}:
        # the usual batch, thousands of CartesianMotion models over constant surfaces: ONE pass over the models and ONE
        # concatenation of their 6 P parameter vectors (what NumPy charges per small array is the cost here)
        try:
            vectors = list(itertools.chain.from_iterable(map(_CARTESIAN_VECTORS, models)))
            flat = np.concatenate(vectors)
            surfaces = np.array(list(map(_CARTESIAN_SURFACES, models)), dtype=np.float64)  # (TypeError: a Raster)
            if flat.size == 16 * P and surfaces.shape == (P, 2):
                table[:, 0:16] = flat.reshape(P, 16)
                table[:, 16:18] = surfaces
                return table  # (kind 0, no slope sigma, no rasters)
        except (ValueError, TypeError):
            pass  # scalars that broadcast, gridded surfaces: the general forms below
        table[:] = 0.0
    if P and kinds <= {CartesianMotion, CylindricalMotion}:
        rates = [m._rates() for m in models]
        table[:, 0:2] = _columns([m.xy for m in models], 2)
        table[:, 2:4] = _columns([m.xy_sigma for m in models], 2)
        for k, (a, b) in enumerate(((4, 7), (7, 10), (10, 13), (13, 16))):
            table[:, a:b] = _columns([r[k] for r in rates], 3)
        dem, dem_sigma = [m.dem for m in models], [m.dem_sigma for m in models]
        rd = np.fromiter((isinstance(d, Raster) for d in dem), dtype=bool, count=P)
        rs = np.fromiter((isinstance(d, Raster) for d in dem_sigma), dtype=bool, count=P)
        table[:, 16] = [0.0 if r else d for d, r in zip(dem, rd)]
        table[:, 17] = [0.0 if r else d for d, r in zip(dem_sigma, rs)]
        table[:, 18] = [m.KIND for m in models]
        table[:, 20], table[:, 21] = rd, rs
        return table
    for row, model in zip(table, models):
        model.fill_params(row)
    return table


class _RowView:
    """What the batched Tracker reads of ONE model of a `ModelBlock` (it never calls a device model's methods)."""

    __slots__ = ("xy", "n", "time_unit", "dem", "dem_sigma", "KIND", "TANGENT")

    def __init__(self, block, i):
        row = block.table[i]
        self.xy = row[0:2]
        self.n, self.time_unit = block.n, block.time_unit
        self.dem = block.dem if row[20] else float(row[16])
        self.dem_sigma = block.dem_sigma if row[21] else float(row[17])
        self.KIND = int(row[18])
        self.TANGENT = self.KIND in (2, 3)


class ModelBlock:
    """A block of DEVICE motion models that share one batch (one particle count, one time unit, at most one gridded dem
    and one gridded dem_sigma: `tracker._batches`) as ONE parameter table: everything the batched Tracker uses of them
    (`params_table`, the rasters, n, time_unit).  `Tracker.track(parallel=N)` hands its workers their tracks in this
    form -- thousands of model objects cost more to pickle and unpickle than their tracks take to run (4 096
    CartesianMotion objects: 25 ms; their table: 0.8 MB in one piece).  A read-only sequence: an index gives a view of
    that model's parameters (not a model: no methods), a slice another block."""

    def __init__(self, table, n, time_unit, dem=None, dem_sigma=None):
        self.table = np.ascontiguousarray(table, dtype=np.float64)
        self.n, self.time_unit = int(n), time_unit
        self.dem, self.dem_sigma = dem, dem_sigma  # the Raster of the rows flagged in columns 20 / 21, or None

    @classmethod
    def from_models(cls, models):
        """Of a list of device models that `tracker._batches` holds in one batch."""
        table = params_table(models)
        rasters = []
        for col, attr in ((20, "dem"), (21, "dem_sigma")):
            rows = np.nonzero(table[:, col])[0]
            rasters.append(getattr(models[int(rows[0])], attr) if len(rows) else None)
        return cls(table, models[0].n, models[0].time_unit, *rasters)

    def __len__(self):
        return len(self.table)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return ModelBlock(self.table[i], self.n, self.time_unit, self.dem, self.dem_sigma)
        return _RowView(self, int(i))

    def __iter__(self):
        return (_RowView(self, i) for i in range(len(self.table)))

    def raster(self, attr):
        """The Raster some row uses as its `attr` ("dem" / "dem_sigma"), or None."""
        return getattr(self, attr) if self.table[:, 20 if attr == "dem" else 21].any() else None


class Motion:
    """The minimal motion model of the reference (motion.py:13-89), which doubles as the statement of the interface a
    `Tracker` asks of any model: every particle starts AT `xy` with z = 0 and a velocity drawn around zero, moves with
    that velocity, and contributes no likelihood term.  Used as it is, or subclassed, it runs like any user-defined model:
    its methods on the host, everything else on the device (`Tracker._track_custom`)."""

    def __init__(self, xy, time_unit, n=1000, vxyz_sigma=(0, 0, 0)):
        self.xy = xy
        self.time_unit = time_unit
        self.n = n
        self.vxyz_sigma = vxyz_sigma

    def initialize_particles(self):
        """(n, 6) particles (x, y, z, vx, vy, vz): motion.py:54-64 -- one randn(n, 3) draw, for the velocities."""
        particles = np.zeros((self.n, 6), dtype=float)
        particles[:, 0:2] = self.xy
        particles[:, 3:6] = self.vxyz_sigma * np.random.randn(self.n, 3)
        return particles

    def evolve_particles(self, particles, dt):
        """In place, motion.py:66-76: positions advance by the velocities, no draw."""
        steps = dt.total_seconds() / self.time_unit.total_seconds()
        particles[:, 0:3] += steps * particles[:, 3:6]

    def compute_log_likelihoods(self, particles):
        """motion.py:78-89: this model has no likelihood term."""
        return None


class CartesianMotion(Motion):
    def __init__(self, xy, time_unit, dem, dem_sigma=None, n=1000, xy_sigma=(0, 0), vxyz=(0, 0, 0),
                 vxyz_sigma=(0, 0, 0), axyz=(0, 0, 0), axyz_sigma=(0, 0, 0)):
        """motion.py:121-147.  `dem` / `dem_sigma`: numbers (constant surfaces) or `glimpse_amd.Raster`s;
        `dem_sigma=None` crashes in the reference (KeyError 'buf_xsize'), so a value is
        required here."""
        self.xy = xy
        self.time_unit = time_unit
        self.dem = _surface(dem, "dem", "CartesianMotion")
        self.dem_sigma = _surface(dem_sigma, "dem_sigma", "CartesianMotion")
        self.n = int(n)
        self.xy_sigma = xy_sigma
        self.vxyz = vxyz
        self.vxyz_sigma = vxyz_sigma
        self.axyz = axyz
        self.axyz_sigma = axyz_sigma

    KIND = 0          # GLH_MOTION_CARTESIAN
    N_INIT_V = 3      # velocity normals drawn by initialize_particles: randn(n, 3)
    TANGENT = False   # evolve draws randn(n, 3) (False) or randn(n, 2) then randn(n) (True)

    def _rates(self):
        """(velocity, its sigma, acceleration, its sigma) in the model's own coordinates, 3 components each."""
        return self.vxyz, self.vxyz_sigma, self.axyz, self.axyz_sigma

    def fill_params(self, row):
        """Write this model into one row of the [P][GLH_MOTION_FULL_LEN] table of glh_set_motion (slice assignment
        broadcasts scalars: the Tracker fills thousands of rows per run)."""
        v, vs, a, as_ = self._rates()
        row[0:2], row[2:4] = self.xy, self.xy_sigma
        row[4:7], row[7:10], row[10:13], row[13:16] = v, vs, a, as_
        row[16] = 0.0 if isinstance(self.dem, Raster) else self.dem
        row[17] = 0.0 if isinstance(self.dem_sigma, Raster) else self.dem_sigma
        row[18], row[19] = self.KIND, getattr(self, "slope_sigma", 0.0)
        row[20], row[21] = isinstance(self.dem, Raster), isinstance(self.dem_sigma, Raster)

    def params_full(self):
        """GLH_MOTION_FULL_LEN doubles (include/glimpse_hip.h): params() | kind | slope_sigma | raster flags | 0 0."""
        row = np.zeros(24)
        self.fill_params(row)
        return row

    def params(self):
        """GLH_MOTION_LEN doubles (include/glimpse_hip.h)."""
        return self.params_full()[:18]

    def initialize_particles(self):
        """motion.py:149-163."""
        particles = np.zeros((self.n, 6), dtype=float)
        particles[:, 0:2] = self.xy + self.xy_sigma * np.random.randn(self.n, 2)
        particles[:, 2] = _sample(self.dem, particles[:, 0:2])
        particles[:, 2] += _sample(self.dem_sigma, particles[:, 0:2]) * np.random.randn(self.n)
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
        z = _sample(self.dem, particles[:, 0:2])
        z_sigma = _sample(self.dem_sigma, particles[:, 0:2])
        nonzero = np.nonzero(z_sigma)[0]
        ll = np.zeros(len(particles), dtype=float)
        ll[nonzero] = 1 / (2 * z_sigma[nonzero] ** 2) * (z[nonzero] - particles[nonzero, 2]) ** 2
        return ll


class CylindricalMotion(CartesianMotion):
    """motion.py:207-311: like CartesianMotion with velocity / acceleration given as (radius rate,
    direction theta in radians, dz/dt)."""

    KIND = 1

    def __init__(self, xy, time_unit, dem, dem_sigma=None, n=1000, xy_sigma=(0, 0), vrthz=(0, 0, 0),
                 vrthz_sigma=(0, 0, 0), arthz=(0, 0, 0), arthz_sigma=(0, 0, 0)):
        self.dem = _surface(dem, "dem", "CylindricalMotion")
        self.dem_sigma = _surface(dem_sigma, "dem_sigma", "CylindricalMotion")
        self.xy, self.time_unit, self.n, self.xy_sigma = xy, time_unit, int(n), xy_sigma
        self.vrthz, self.vrthz_sigma, self.arthz, self.arthz_sigma = vrthz, vrthz_sigma, arthz, arthz_sigma

    def _rates(self):
        return self.vrthz, self.vrthz_sigma, self.arthz, self.arthz_sigma

    def initialize_particles(self):
        """motion.py:262-286."""
        particles = np.zeros((self.n, 6), dtype=float)
        particles[:, 0:2] = self.xy + self.xy_sigma * np.random.randn(self.n, 2)
        particles[:, 2] = _sample(self.dem, particles[:, 0:2])
        particles[:, 2] += _sample(self.dem_sigma, particles[:, 0:2]) * np.random.randn(self.n)
        v = self.vrthz + self.vrthz_sigma * np.random.randn(self.n, 3)
        particles[:, 3:6] = np.column_stack((v[:, 0] * np.cos(v[:, 1]), v[:, 0] * np.sin(v[:, 1]), v[:, 2]))
        return particles

    def evolve_particles(self, particles, dt):
        """motion.py:288-311 (in place)."""
        n = len(particles)
        time_units = dt.total_seconds() / self.time_unit.total_seconds()
        vx, vy = particles[:, 3], particles[:, 4]
        vr = np.sqrt(vx ** 2 + vy ** 2)
        arthz = self.arthz + self.arthz_sigma * np.random.randn(n, 3)
        axyz = np.column_stack((arthz[:, 0] * (vx / vr) - vy * arthz[:, 1],
                                arthz[:, 0] * (vy / vr) + vx * arthz[:, 1], arthz[:, 2]))
        particles[:, 0:3] += time_units * particles[:, 3:6] + 0.5 * axyz * time_units ** 2
        particles[:, 3:6] += time_units * axyz


class TangentCartesianMotion(Motion):
    """motion.py:314-412: particles move tangent to the mean surface; their height keeps its offset
    from the surface plus a random walk ~ slope_sigma * horizontal distance.  No log likelihood of
    its own (Motion.compute_log_likelihoods returns None, motion.py:76-89)."""

    KIND = 2
    N_INIT_V = 2
    TANGENT = True

    def __init__(self, xy, time_unit, dem, dem_sigma=0, n=1000, xy_sigma=(0, 0), vxy=(0, 0), vxy_sigma=(0, 0),
                 axy=(0, 0), axy_sigma=(0, 0), slope_sigma=0):
        self.dem = _surface(dem, "dem", type(self).__name__)
        self.dem_sigma = _surface(dem_sigma, "dem_sigma", type(self).__name__)
        self.xy, self.time_unit, self.n, self.xy_sigma = xy, time_unit, int(n), xy_sigma
        self.vxy, self.vxy_sigma, self.axy, self.axy_sigma = vxy, vxy_sigma, axy, axy_sigma
        self.slope_sigma = float(slope_sigma)

    def _v4(self):
        return self.vxy, self.vxy_sigma, self.axy, self.axy_sigma

    def fill_params(self, row):
        """One row of the [P][GLH_MOTION_FULL_LEN] table (two horizontal components; the third slots stay 0)."""
        v, vs, a, as_ = self._v4()
        row[0:2], row[2:4] = self.xy, self.xy_sigma
        row[4:6], row[7:9], row[10:12], row[13:15] = v, vs, a, as_
        row[16] = 0.0 if isinstance(self.dem, Raster) else self.dem
        row[17] = 0.0 if isinstance(self.dem_sigma, Raster) else self.dem_sigma
        row[18], row[19] = self.KIND, self.slope_sigma
        row[20], row[21] = isinstance(self.dem, Raster), isinstance(self.dem_sigma, Raster)

    def params_full(self):
        row = np.zeros(24)
        self.fill_params(row)
        return row

    def params(self):
        return self.params_full()[:18]

    def _initial_velocity(self, normals):
        return self.vxy + self.vxy_sigma * normals

    def _acceleration(self, particles, normals):
        return self.axy + self.axy_sigma * normals

    def initialize_particles(self):
        """motion.py:382-394 / :470-488."""
        particles = np.zeros((self.n, 6), dtype=float)
        particles[:, 0:2] = self.xy + self.xy_sigma * np.random.randn(self.n, 2)
        z_offsets = _sample(self.dem_sigma, particles[:, 0:2]) * np.random.randn(self.n)
        particles[:, 2] = _sample(self.dem, particles[:, 0:2]) + z_offsets
        particles[:, 3:5] = self._initial_velocity(np.random.randn(self.n, 2))
        return particles

    def evolve_particles(self, particles, dt):
        """motion.py:396-412 / :490-522 (in place)."""
        n = len(particles)
        time_units = dt.total_seconds() / self.time_unit.total_seconds()
        axy = self._acceleration(particles, np.random.randn(n, 2))
        dxy = time_units * particles[:, 3:5] + 0.5 * axy * time_units ** 2
        z_offsets = particles[:, 2] - _sample(self.dem, particles[:, 0:2])
        z_offsets += self.slope_sigma * np.random.randn(n) * (dxy ** 2).sum(axis=1) ** 0.5
        particles[:, 0:2] += dxy
        particles[:, 2] = _sample(self.dem, particles[:, 0:2]) + z_offsets
        particles[:, 3:5] += time_units * axy

    def compute_log_likelihoods(self, particles):
        return None


class TangentCylindricalMotion(TangentCartesianMotion):
    """motion.py:415-522: TangentCartesianMotion with velocity / acceleration as (radius rate, theta)."""

    KIND = 3

    def __init__(self, xy, time_unit, dem, dem_sigma=0, n=1000, xy_sigma=(0, 0), vrth=(0, 0), vrth_sigma=(0, 0),
                 arth=(0, 0), arth_sigma=(0, 0), slope_sigma=0):
        super().__init__(xy, time_unit, dem, dem_sigma, n, xy_sigma, slope_sigma=slope_sigma)
        self.vrth, self.vrth_sigma, self.arth, self.arth_sigma = vrth, vrth_sigma, arth, arth_sigma

    def _v4(self):
        return self.vrth, self.vrth_sigma, self.arth, self.arth_sigma

    def _initial_velocity(self, normals):
        vrth = self.vrth + self.vrth_sigma * normals
        return np.column_stack((vrth[:, 0] * np.cos(vrth[:, 1]), vrth[:, 0] * np.sin(vrth[:, 1])))

    def _acceleration(self, particles, normals):
        vx, vy = particles[:, 3], particles[:, 4]
        vr = np.sqrt(vx ** 2 + vy ** 2)
        arth = self.arth + self.arth_sigma * normals
        return np.column_stack((arth[:, 0] * (vx / vr) - vy * arth[:, 1], arth[:, 0] * (vy / vr) + vx * arth[:, 1]))
