"""Synthetic workloads named after BASELINE.json's configs (host-side input builders).

C1  1 point x 100 particles, 5 frames 512^2, pinhole, 15x15
C2  256 points x 2 000 particles, 50 frames 2048^2, radial k1-k3, 15x15
C3  4 096 points x 5 000 particles, 100 frames 2048^2, 31x31      (single-GPU roofline run)
C4  10 000 points x 10 000 particles, 100 frames, sharded 8 ways  (1 250 points per GPU)
C5  two-camera observer, 2 048 points x 5 000 particles, dem_sigma > 0 (512 points per GPU of 4)

Scene and motion parameters follow SURVEY.md section 8(d): textured plane z = 0 moving at
0.15 units/frame seen from 100 units above at f = 1000 px (10 px per unit), particle cloud
sigma ~ 2 px (sigma = 0.2 units), time unit = one frame.
"""
import numpy as np

from . import synth

CONFIGS = {
    "C1": dict(points=1, particles=100, frames=5, imgsz=(512, 512), k=(0, 0, 0), tile=(15, 15), observers=1),
    "C2": dict(points=256, particles=2000, frames=50, imgsz=(2048, 2048), k=(0.05, -0.01, 0.002), tile=(15, 15),
               observers=1),
    "C3": dict(points=4096, particles=5000, frames=100, imgsz=(2048, 2048), k=(0.05, -0.01, 0.002), tile=(31, 31),
               observers=1),
    "C4": dict(points=10000, particles=10000, frames=100, imgsz=(2048, 2048), k=(0.05, -0.01, 0.002),
               tile=(31, 31), observers=1, shards=8),
    "C5": dict(points=2048, particles=5000, frames=100, imgsz=(2048, 2048), k=(0.05, -0.01, 0.002), tile=(31, 31),
               observers=2, shards=4, dem_sigma=0.5),
}

SIGMA = 0.2
VELOCITY = (0.15, 0.0)


def motion_params(xy, dem_sigma=0.0, sigma=SIGMA):
    """[P][18] CartesianMotion parameters (include/glimpse_hip.h GLH_MOTION_LEN layout)."""
    xy = np.atleast_2d(np.asarray(xy, dtype=float))
    p = np.zeros((len(xy), 18))
    p[:, 0:2] = xy
    p[:, 2:4] = sigma
    p[:, 4:7] = (VELOCITY[0], VELOCITY[1], 0.0)
    p[:, 7:10] = (sigma, sigma, 0.05 if dem_sigma else 0.0)
    p[:, 13:16] = (sigma / 4, sigma / 4, 0.01 if dem_sigma else 0.0)
    p[:, 16] = 0.0
    p[:, 17] = dem_sigma
    return p


class Workload:
    """Cameras, frames and per-point motion parameters of one (shard of a) config."""

    def __init__(self, name, n_frames=None, n_points=None, n_particles=None, shard=0, seed=0, imgsz=None):
        cfg = dict(CONFIGS[name])
        self.name = name
        self.cfg = cfg
        shards = cfg.get("shards", 1)
        self.P = n_points if n_points is not None else cfg["points"] // shards
        self.N = n_particles if n_particles is not None else cfg["particles"]
        self.T = n_frames if n_frames is not None else cfg["frames"]
        self.tile = cfg["tile"]
        self.O = cfg["observers"]
        self.imgsz = tuple(imgsz) if imgsz is not None else cfg["imgsz"]
        self.dem_sigma = cfg.get("dem_sigma", 0.0)
        self.seed = seed
        k = tuple(cfg["k"]) + (0, 0, 0)
        cam0 = synth.nadir_camera(self.imgsz, f=1000.0, height=100.0, k=k)
        self.cams = [cam0]
        self.sigmas = [0.3]
        if self.O == 2:
            # oblique second station looking at the scene centre
            self.cams.append(synth.pack_camera(imgsz=self.imgsz, f=1200.0, k=(0.03, 0, 0), xyz=(40, -30, 90),
                                               viewdir=(-53.13, -60.9, 0)))
            self.sigmas.append(0.3)
        self._scene = None
        border = 0.5 * max(self.tile) + 110.0 if min(self.imgsz) >= 1024 else 0.5 * max(self.tile) + 60.0
        # the scene drifts 1.5 px per frame: long sequences keep their points further from the image border, so that
        # the last frame's search boxes are still inside the image
        border += 10.0 * max(abs(VELOCITY[0]), abs(VELOCITY[1])) * max(0, self.T - 32)
        if self.O == 2:
            border += 200.0  # keep the points inside the oblique view too
        # different shards track different points of the same scene
        if self.O == 1:
            self.xy = synth.grid_points(cam0, self.P, border_px=border, seed=1000 + shard)
        else:
            # keep only seeds that every camera sees with a full template + search margin
            margin = 0.5 * max(self.tile) + (110.0 if min(self.imgsz) >= 1024 else 40.0)
            margin += 20.0 * max(abs(VELOCITY[0]), abs(VELOCITY[1])) * max(0, self.T - 32)  # (f = 1200, oblique: <= 2 px/frame)
            want, factor = self.P, 2
            while True:
                cand = synth.grid_points(cam0, want * factor, border_px=border - 200.0 + margin, seed=1000 + shard)
                ok = np.ones(len(cand), dtype=bool)
                xyz = np.column_stack((cand, np.zeros(len(cand))))
                for cam in self.cams:
                    uv = synth.project(cam, xyz)
                    ok &= (uv[:, 0] > margin) & (uv[:, 0] < cam[6] - margin)
                    ok &= (uv[:, 1] > margin) & (uv[:, 1] < cam[7] - margin)
                if ok.sum() >= want or factor > 64:
                    break
                factor *= 2
            if ok.sum() < want:
                raise ValueError(f"only {ok.sum()} of {want} points are visible in both cameras")
            self.xy = cand[ok][:want]
        self.params = motion_params(self.xy, dem_sigma=self.dem_sigma)

    def slice(self, lo, hi):
        """The same workload restricted to points [lo, hi) (a shard of a strong split)."""
        sub = Workload.__new__(Workload)
        sub.__dict__.update(self.__dict__)
        sub.P = hi - lo
        sub.xy = self.xy[lo:hi]
        sub.params = self.params[lo:hi]
        return sub

    channels = 1  # 3: RGB frames (what a time-lapse JPEG decodes to)
    bits = 8      # 16: the same scene on a 16-bit sensor (uint16 frames); 32 / 64: the caller brings float frames

    @property
    def scene(self):
        """The textured ground, built when the first frame is asked for (callers that bring their frames never pay
        for the texture).  It must cover every camera's footprint."""
        if self._scene is None:
            self._scene = synth.default_scene(self.cams[-1], seed=self.seed, velocity=VELOCITY, n_frames=self.T,
                                              margin=30.0)
        return self._scene

    def frame(self, obs, t):
        return self.scene.render(self.cams[obs], float(t), channels=self.channels, bits=self.bits)

    def frames(self, obs):
        return [self.frame(obs, t) for t in range(self.T)]

    def describe(self, motion="cartesian"):
        name = {"cartesian": "CartesianMotion", "cylindrical": "CylindricalMotion",
                "tangent_cartesian": "TangentCartesianMotion", "tangent_cylindrical": "TangentCylindricalMotion"}[motion]
        return {
            "workload": f"{self.name}: {self.P} points x {self.N} particles x {self.T} frames "
                        f"{self.imgsz[0]}x{self.imgsz[1]} {'uint' if self.bits <= 16 else 'float'}{self.bits}{' RGB' if self.channels == 3 else ''}, tile {self.tile[0]}x{self.tile[1]}, "
                        f"{self.O} observer(s), {name}, radial k={tuple(self.cfg['k'])}",
            "points_per_gpu": self.P,
            "particles": self.N,
            "frames": self.T,
            "tile": list(self.tile),
            "observers": self.O,
        }


def setup_context(ctx, wl, frames=None, channels=None):
    """Upload cameras + frames of a workload into a glimpse_amd._lib.Context and start a sequence."""
    channels = wl.channels if channels is None else channels
    for o in range(wl.O):
        ctx.observer_init(o, wl.T, wl.imgsz[0], wl.imgsz[1], channels, wl.sigmas[o])
        if wl.bits != 8:  # (uint16 frames, or the scene as float32 / float64 reflectances)
            ctx.observer_set_depth(o, {16: np.uint16, 32: np.float32, 64: np.float64}[wl.bits])
        ctx.observer_set_cameras(o, np.tile(wl.cams[o], (wl.T, 1)))
        for t in range(wl.T):
            f = frames[o][t] if frames is not None else wl.frame(o, t)
            ctx.observer_upload_frame(o, t, f)
    ctx.begin_sequence(wl.P, wl.N, wl.tile)
    ctx.set_motion_cartesian(wl.params)
