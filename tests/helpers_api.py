"""Build glimpse_amd API objects (Image / Observer / Tracker / CartesianMotion) from a g8 fixture."""
import datetime

import numpy as np

import glimpse_amd

T0 = datetime.datetime(2020, 1, 1)
DAY = datetime.timedelta(days=1)


def camera_from(vec):
    corr = False
    if vec[20]:
        corr = {"radius": vec[21], "refraction": vec[22]}
    return glimpse_amd.Camera(imgsz=vec[6:8], f=vec[8:10], c=vec[10:12], k=vec[12:18], p=vec[18:20], xyz=vec[0:3],
                              viewdir=vec[3:6], correction=corr)


def observers_from(g):
    observers = []
    for o in range(int(g["n_obs"])):
        frames, cams, days = g[f"obs{o}_frames"], g[f"obs{o}_cams"], g[f"obs{o}_days"]
        images = [glimpse_amd.Image("synthetic", cam=camera_from(cams[i]), datetime=T0 + float(days[i]) * DAY,
                                    array=frames[i]) for i in range(len(frames))]
        observers.append(glimpse_amd.Observer(images, sigma=float(g["sigmas"][o])))
    return observers


def models_from(g):
    out = []
    for p, n in zip(g["params"], g["n_particles"]):
        out.append(glimpse_amd.CartesianMotion(xy=p[0:2], time_unit=DAY, dem=float(p[16]), dem_sigma=float(p[17]),
                                               n=int(n), xy_sigma=p[2:4], vxyz=p[4:7], vxyz_sigma=p[7:10],
                                               axyz=p[10:13], axyz_sigma=p[13:16]))
    return out
