"""Oracle: world -> image projection (test infrastructure only).

Follows /root/reference/src/glimpse/camera.py:
  * state vector `_vector[20]`            camera.py:101, 128-198
  * `Camera.R`                            camera.py:239-280
  * `Camera._xyz_to_xy`                   camera.py:1435-1470
  * `Camera._radial_distortion`           camera.py:1138-1163
  * `Camera._tangential_distortion`       camera.py:1165-1178
  * `Camera._distort`                     camera.py:1180-1196
  * `Camera._xy_to_uv`                    camera.py:1499-1508
  * `Camera.inframe`                      camera.py:700-718
  * `helpers.elevation_corrections`       helpers.py:1771-1790

A camera is a float64 vector of CAM_LEN = 24 values:
  [0:3]  xyz        [3:6]  viewdir (deg: yaw, pitch, roll)
  [6:8]  imgsz      [8:10] f        [10:12] c
  [12:18] k1..k6    [18:20] p1, p2
  [20] correction flag (0/1)   [21] radius   [22] refraction
  [23] 0 camera; 1 georeferenced raster image (Grid.xyz_to_uv, raster.py:423-445) with
       [0:2] = (xlim[0], ylim[0]), [6:8] = size, [8:10] = d
The first 20 entries are exactly the reference's `_vector`.
"""
import numpy as np

CAM_LEN = 24
DEFAULT_RADIUS = 6.3781e6  # camera.py:120
DEFAULT_REFRACTION = 0.13  # camera.py:120


def make_camera(
    imgsz,
    f,
    c=(0, 0),
    k=(0, 0, 0, 0, 0, 0),
    p=(0, 0),
    xyz=(0, 0, 0),
    viewdir=(0, 0, 0),
    correction=False,
):
    """Pack camera arguments (camera.py:77-123) into a CAM_LEN vector."""
    v = np.zeros(CAM_LEN, dtype=float)
    v[0:3] = xyz
    v[3:6] = viewdir
    v[6:8] = np.broadcast_to(np.asarray(imgsz, dtype=float), (2,))
    v[8:10] = np.broadcast_to(np.asarray(f, dtype=float), (2,))
    v[10:12] = np.broadcast_to(np.asarray(c, dtype=float), (2,))
    kk = np.zeros(6)
    kk[: len(k)] = k
    v[12:18] = kk
    pp = np.zeros(2)
    pp[: len(p)] = p
    v[18:20] = pp
    if correction is True:
        correction = {}
    if isinstance(correction, dict):
        corr = {"radius": DEFAULT_RADIUS, "refraction": DEFAULT_REFRACTION, **correction}
        v[20] = 1.0
        v[21] = corr["radius"]
        v[22] = corr["refraction"]
    return v


def rotation_matrix(viewdir):
    """camera.py:263-280."""
    radians = np.deg2rad(np.asarray(viewdir, dtype=float))
    C = np.cos(radians)
    S = np.sin(radians)
    return np.array(
        [
            [
                C[0] * C[2] + S[0] * S[1] * S[2],
                C[0] * S[1] * S[2] - C[2] * S[0],
                -C[1] * S[2],
            ],
            [
                C[2] * S[0] * S[1] - C[0] * S[2],
                S[0] * S[2] + C[0] * C[2] * S[1],
                -C[1] * C[2],
            ],
            [C[1] * S[0], C[0] * C[1], S[1]],
        ]
    )


def xyz_to_xy(cam, xyz):
    """camera.py:1435-1470 (directions=False)."""
    xyz = np.asarray(xyz, dtype=float)
    dxyz = xyz - cam[0:3]
    if cam[20] != 0:
        # helpers.py:1790
        sq = np.sum(dxyz[:, 0:2] ** 2, axis=1)
        dxyz[:, 2] += (cam[22] - 1) * sq / (2 * cam[21])
    R = rotation_matrix(cam[3:6])
    xyz_c = np.matmul(R, dxyz.T).T
    with np.errstate(invalid="ignore", divide="ignore"):
        xy = xyz_c[:, 0:2] / xyz_c[:, 2:3]
    behind = xyz_c[:, 2] <= 0
    xy[behind] = np.nan
    return xy


def distort(cam, xy):
    """camera.py:1180-1196 with :1138-1163 and :1165-1178."""
    k = cam[12:18]
    p = cam[18:20]
    if not any(k) and not any(p):
        return xy
    dxy = xy.copy()
    r2 = np.sum(xy ** 2, axis=1)
    if any(k):
        dr = 1
        if k[0]:
            dr = dr + k[0] * r2
        if k[1]:
            dr = dr + k[1] * r2 * r2
        if k[2]:
            dr = dr + k[2] * r2 * r2 * r2
        if any(k[3:6]):
            temp = 1
            if k[3]:
                temp = temp + k[3] * r2
            if k[4]:
                temp = temp + k[4] * r2 * r2
            if k[5]:
                temp = temp + k[5] * r2 * r2 * r2
            dr = dr / temp
        dxy *= dr[:, None]
    if any(p):
        xty = xy[:, 0] * xy[:, 1]
        dtx = 2 * xty * p[0] + p[1] * (r2 + 2 * xy[:, 0] ** 2)
        dty = p[0] * (r2 + 2 * xy[:, 1] ** 2) + 2 * xty * p[1]
        dxy += np.column_stack((dtx, dty))
    return dxy


def grid_vector(size, xlim, ylim):
    """CAM_LEN vector of a raster image (an orthophoto observer image)."""
    v = np.zeros(CAM_LEN, dtype=float)
    v[0], v[1] = xlim[0], ylim[0]
    v[6:8] = size
    v[8:10] = (np.diff(xlim)[0] / size[0], np.diff(ylim)[0] / size[1])
    v[23] = 1.0
    return v


def xyz_to_uv(cam, xyz):
    """camera.py:591-628 / :1499-1508: uv = distort(xy) * f + (imgsz / 2 + c)."""
    if cam[23] != 0:  # Grid.xyz_to_uv (raster.py:445)
        xyz = np.atleast_2d(np.asarray(xyz, dtype=float))
        return (xyz[:, 0:2] - (cam[0], cam[1])) / cam[8:10]
    xy = xyz_to_xy(cam, np.atleast_2d(xyz))
    xy = distort(cam, xy)
    return xy * cam[8:10] + (cam[6:8] / 2 + cam[10:12])


def inframe(cam, uv):
    """camera.py:700-718."""
    with np.errstate(invalid="ignore"):
        return np.all((uv >= 0) & (uv <= cam[6:8]), axis=1)


def _radial(cam, r2):
    """camera.py:1138-1163."""
    k = cam[12:18]
    dr = np.ones_like(r2)
    if k[0]:
        dr = dr + k[0] * r2
    if k[1]:
        dr = dr + k[1] * r2 * r2
    if k[2]:
        dr = dr + k[2] * r2 * r2 * r2
    if np.any(k[3:6]):
        temp = np.ones_like(r2)
        if k[3]:
            temp = temp + k[3] * r2
        if k[4]:
            temp = temp + k[4] * r2 * r2
        if k[5]:
            temp = temp + k[5] * r2 * r2 * r2
        dr = dr / temp
    return dr[:, None]


def _tangential(cam, xy, r2):
    """camera.py:1165-1178."""
    p = cam[18:20]
    xty = xy[:, 0] * xy[:, 1]
    return np.column_stack((2 * xty * p[0] + p[1] * (r2 + 2 * xy[:, 0] ** 2),
                            p[0] * (r2 + 2 * xy[:, 1] ** 2) + 2 * xty * p[1]))


def undistort(cam, xy):
    """Camera._undistort (camera.py:1198-1230): identity, the closed-form cubic for k1 alone (:1232-1264),
    else the Oulu fixed point with 20 iterations (:1305-1337)."""
    k, p = cam[12:18], cam[18:20]
    if not np.any(k) and not np.any(p):
        return xy
    if k[0] and not np.any(k[1:]) and not np.any(p):
        phi = np.arctan2(xy[:, 1], xy[:, 0])
        Q = -1 / (3 * k[0])
        R = -xy[:, 0] / (2 * k[0] * np.cos(phi))
        three = R ** 2 < Q ** 3
        r = np.full(len(xy), np.nan)
        if np.any(three):
            th = np.arccos(R[three] * Q ** -1.5)
            r[three] = -2 * np.sqrt(Q) * np.cos((th - 2 * np.pi) / 3)
        one = ~three
        if np.any(one):
            A = -np.sign(R[one]) * (np.abs(R[one]) + np.sqrt(R[one] ** 2 - Q ** 3)) ** (1.0 / 3)
            B = np.zeros(A.shape)
            nz = A != 0
            B[nz] = Q / A[nz]
            r[one] = A + B
        return np.column_stack((np.cos(phi), np.sin(phi))) * r[:, None]
    uxy = xy
    for _ in range(20):
        r2 = np.sum(uxy ** 2, axis=1)
        if np.any(p) and not np.any(k):
            uxy = xy - _tangential(cam, uxy, r2)
        else:
            uxy = (xy - _tangential(cam, uxy, r2)) * (1 / _radial(cam, r2))
    return uxy


def uv_to_xyz(cam, uv, directions=True, depth=1):
    """Camera.uv_to_xyz (camera.py:630-663): _uv_to_xy (:1510-1519) then _xy_to_xyz (:1472-1497)."""
    uv = np.atleast_2d(np.asarray(uv, dtype=float))
    xy = (uv - (cam[6:8] * 0.5 + cam[10:12])) * (1 / cam[8:10])
    xy = undistort(cam, xy)
    R = rotation_matrix(cam[3:6])
    xyz = np.matmul(R.T[:, 0:2], xy.T).T
    xyz += R.T[:, 2]
    if not isinstance(depth, (int, float)) or depth != 1:
        xyz *= np.atleast_1d(depth).reshape(-1, 1)
    if not directions:
        xyz += cam[0:3]
    return xyz
