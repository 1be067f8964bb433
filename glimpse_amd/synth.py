"""Synthetic image sequences for tests and benchmarks (host-side data generator).

Not part of the hot path: this only manufactures inputs (SURVEY.md 8(d)
"Synthetic inputs").  Scene: a planar textured ground z = 0 that moves with a
constant world velocity; frames are rendered by inverse-mapping pixel centres
through the camera model onto the plane (the inverse projection follows the
semantics of the reference's `Camera.uv_to_xyz`, camera.py:630-663, with the
Oulu fixed-point undistortion, camera.py:1305-1337) and a bilinear texture
lookup.  Everything is seeded with PCG64 (`np.random.default_rng`), which is
version-stable.
"""
import numpy as np
import scipy.ndimage

CAM_LEN = 24


def pack_camera(imgsz, f, c=(0, 0), k=(0,) * 6, p=(0, 0), xyz=(0, 0, 0), viewdir=(0, 0, 0),
                correction=False):
    """Camera arguments -> the 24-double vector used across the C ABI (include/glimpse_hip.h)."""
    v = np.zeros(CAM_LEN, dtype=np.float64)
    v[0:3] = xyz
    v[3:6] = viewdir
    v[6:8] = np.broadcast_to(np.asarray(imgsz, dtype=float), (2,))
    v[8:10] = np.broadcast_to(np.asarray(f, dtype=float), (2,))
    v[10:12] = np.broadcast_to(np.asarray(c, dtype=float), (2,))
    v[12:12 + len(k)] = k
    v[18:18 + len(p)] = p
    if correction is True:
        correction = {}
    if isinstance(correction, dict):
        v[20] = 1.0
        v[21] = correction.get("radius", 6.3781e6)
        v[22] = correction.get("refraction", 0.13)
    return v


def rotation_matrix(viewdir):
    rad = np.deg2rad(np.asarray(viewdir, dtype=float))
    C, S = np.cos(rad), np.sin(rad)
    return np.array(
        [
            [C[0] * C[2] + S[0] * S[1] * S[2], C[0] * S[1] * S[2] - C[2] * S[0], -C[1] * S[2]],
            [C[2] * S[0] * S[1] - C[0] * S[2], S[0] * S[2] + C[0] * C[2] * S[1], -C[1] * C[2]],
            [C[1] * S[0], C[0] * C[1], S[1]],
        ]
    )


def _distort(cam, xy):
    k, p = cam[12:18], cam[18:20]
    r2 = np.sum(xy ** 2, axis=1)
    dr = 1 + k[0] * r2 + k[1] * r2 ** 2 + k[2] * r2 ** 3
    if np.any(k[3:6]):
        dr = dr / (1 + k[3] * r2 + k[4] * r2 ** 2 + k[5] * r2 ** 3)
    xty = xy[:, 0] * xy[:, 1]
    dt = np.column_stack(
        (2 * xty * p[0] + p[1] * (r2 + 2 * xy[:, 0] ** 2), p[0] * (r2 + 2 * xy[:, 1] ** 2) + 2 * xty * p[1])
    )
    return dr, dt


def uv_to_ground(cam, uv, z=0.0, iterations=20):
    """Pixel coordinates -> world xy on the plane z (rays from the camera centre)."""
    xy = (uv - (cam[6:8] * 0.5 + cam[10:12])) * (1 / cam[8:10])
    if np.any(cam[12:20]):
        u = xy
        for _ in range(iterations):
            dr, dt = _distort(cam, u)
            u = (xy - dt) / dr[:, None]
        xy = u
    R = rotation_matrix(cam[3:6])
    d = xy @ R[0:2, :] + R[2, :]
    s = (z - cam[2]) / d[:, 2]
    return cam[0:2] + s[:, None] * d[:, 0:2]


def project(cam, xyz):
    """World xyz (n, 3) -> pixel uv (n, 2) for the packed camera vector (host NumPy; workload set-up only)."""
    xyz = np.atleast_2d(np.asarray(xyz, dtype=float))
    R = rotation_matrix(cam[3:6])
    c = (xyz - cam[0:3]) @ R.T
    xy = c[:, 0:2] / c[:, 2:3]
    xy[c[:, 2] <= 0] = np.nan
    dr, dt = _distort(cam, xy)
    return (xy * dr[:, None] + dt) * cam[8:10] + (cam[6:8] * 0.5 + cam[10:12])


def make_texture(size, seed=0, blur=2.0):
    """Seeded white noise, Gaussian-blurred, rescaled to [0, 255] float32 (size x size)."""
    rng = np.random.default_rng(seed)
    t = rng.standard_normal((size, size)).astype(np.float32)
    t = scipy.ndimage.gaussian_filter(t, blur, mode="wrap")
    t -= t.min()
    t *= 255.0 / t.max()
    return t


_GROUND_MAPS = {}  # packed camera -> world xy on z = 0 of every pixel centre (a property of the camera alone)


def ground_rows(cam, r0, r1):
    """World xy on z = 0 of the pixel centres of image rows [r0, r1) (row-major)."""
    nx = int(cam[6])
    cu, cv = np.meshgrid(np.arange(nx) + 0.5, np.arange(r0, r1) + 0.5)
    return uv_to_ground(cam, np.column_stack((cu.ravel(), cv.ravel()))).astype(np.float64)


_RGB_LUT = None


def gray_to_rgb(img):
    """The 8-bit RGB frame of a gray one: a deterministic, channel-dependent remap, so that the channels differ."""
    global _RGB_LUT
    if _RGB_LUT is None:
        g = np.arange(256, dtype=np.int32)
        _RGB_LUT = (np.clip(g + ((g * 7) % 5) - 2, 0, 255).astype(np.uint8), np.clip(255 - g // 2, 0, 255).astype(np.uint8))
    return np.stack((img, _RGB_LUT[0][img], _RGB_LUT[1][img]), axis=-1)


class Scene:
    """Planar textured ground moving at a constant world velocity."""

    def __init__(self, texture, texel, origin, velocity=(0.15, 0.0)):
        self.texture = texture
        self.texel = float(texel)  # world units per texel
        self.origin = np.asarray(origin, dtype=float)  # world xy of texel (0, 0) centre
        self.velocity = np.asarray(velocity, dtype=float)

    def ground_map(self, cam):
        key = cam.tobytes()
        if key not in _GROUND_MAPS:
            # in bands of rows: the undistortion iterates over its whole operand, which should stay in cache
            ny = int(cam[7])
            _GROUND_MAPS[key] = np.concatenate([ground_rows(cam, r, min(ny, r + 64)) for r in range(0, ny, 64)])
        return _GROUND_MAPS[key]

    def render(self, cam, t, channels=1, bits=8):
        """uint8 (bits=8) or uint16 (bits=16) frame (ny, nx) or (ny, nx, 3) at time t (in time units)."""
        nx, ny = int(cam[6]), int(cam[7])
        xy = self.ground_map(cam) - self.velocity * t
        tx = (xy[:, 0] - self.origin[0]) / self.texel
        ty = (xy[:, 1] - self.origin[1]) / self.texel
        # texture rows follow -y so that the image is not mirrored for a nadir camera
        vals = scipy.ndimage.map_coordinates(self.texture, [ty, tx], order=1, mode="wrap")
        if bits == 16:
            # the same scene on a 16-bit sensor: 257 levels per 8-bit level, so tiles hold thousands of distinct values
            img = np.clip(np.rint(vals * 257.0), 0, 65535).astype(np.uint16).reshape(ny, nx)
            if channels == 3:
                g = img.astype(np.int64)
                img = np.stack((img, np.clip(g + ((g * 7) % 1291) - 600, 0, 65535).astype(np.uint16),
                                np.clip(65535 - g // 2, 0, 65535).astype(np.uint16)), axis=2)
            return img
        img = np.clip(np.rint(vals), 0, 255).astype(np.uint8).reshape(ny, nx)
        if channels == 3:
            img = gray_to_rgb(img)
        return img


def nadir_camera(imgsz, f=1000.0, height=100.0, k=(0,) * 6, p=(0, 0), xyz_offset=(0, 0)):
    return pack_camera(imgsz=imgsz, f=f, k=k, p=p, xyz=(xyz_offset[0], xyz_offset[1], height),
                       viewdir=(0, -90, 0))


def make_sequence(cam, n_frames, seed=0, velocity=(0.15, 0.0), texel=0.05, channels=1, scene=None):
    """Frames 0..n_frames-1 (time unit = 1 frame) seen by a static camera."""
    if scene is None:
        scene = default_scene(cam, seed=seed, velocity=velocity, texel=texel, n_frames=n_frames)
    return [scene.render(cam, float(t), channels=channels) for t in range(n_frames)], scene


def default_scene(cam, seed=0, velocity=(0.15, 0.0), texel=0.05, n_frames=10, margin=8.0):
    """Texture covering the camera footprint on z = 0 (wraps periodically beyond it)."""
    nx, ny = cam[6], cam[7]
    corners = np.array([[0.5, 0.5], [nx - 0.5, 0.5], [nx - 0.5, ny - 0.5], [0.5, ny - 0.5]])
    g = uv_to_ground(cam, corners)
    span = max(np.ptp(g[:, 0]), np.ptp(g[:, 1])) + 2 * margin + np.abs(velocity).max() * n_frames
    size = int(np.ceil(span / texel))
    origin = g.min(axis=0) - margin - np.abs(np.asarray(velocity)) * n_frames
    return Scene(make_texture(size, seed=seed), texel, origin, velocity)


def grid_points(cam, n_points, border_px=40.0, seed=0, jitter=0.25):
    """World xy of `n_points` seeds on a jittered grid inside the image, z = 0."""
    nx, ny = cam[6], cam[7]
    cols = int(np.ceil(np.sqrt(n_points * nx / ny)))
    rows = int(np.ceil(n_points / cols))
    us = np.linspace(border_px, nx - border_px, cols)
    vs = np.linspace(border_px, ny - border_px, rows)
    uu, vv = np.meshgrid(us, vs)
    uv = np.column_stack((uu.ravel(), vv.ravel()))[:n_points]
    rng = np.random.default_rng(seed)
    du = (us[1] - us[0]) if cols > 1 else 0.0
    dv = (vs[1] - vs[0]) if rows > 1 else 0.0
    lim = np.minimum(jitter * np.array([du, dv]), border_px * 0.25)
    uv = uv + (rng.random(uv.shape) - 0.5) * 2 * lim
    return uv_to_ground(cam, uv)
