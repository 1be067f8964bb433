"""cProfile of the warm glimpse_amd.Tracker.track(rng="philox") call at C3 (where does the host time go?)."""
import cProfile
import datetime
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glimpse_amd as g  # noqa: E402
from glimpse_amd import workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = workloads.Workload(name)
T = wl.T if hasattr(wl, "T") else 100
frames = [wl.frames(o) for o in range(wl.O)]
t_start = datetime.datetime(2020, 1, 1)
unit = datetime.timedelta(days=1)
observers = []
for o in range(wl.O):
    v = wl.cams[o]
    images = [g.Image(cam=g.Camera(imgsz=v[6:8], f=v[8:10], c=v[10:12], k=v[12:18], p=v[18:20], xyz=v[0:3], viewdir=v[3:6]),
                      datetime=t_start + t * unit, array=np.asarray(frames[o][t])) for t in range(len(frames[o]))]
    observers.append(g.Observer(images, sigma=wl.sigmas[o]))
models = [g.CartesianMotion(xy=q[0:2], time_unit=unit, dem=q[16], dem_sigma=q[17], n=wl.N, xy_sigma=q[2:4], vxyz=q[4:7],
                            vxyz_sigma=q[7:10], axyz=q[10:13], axyz_sigma=q[13:16]) for q in wl.params]
tracker = g.Tracker(observers)
tracker.track(models, tile_size=wl.tile, rng="philox", seed=1)
t0 = time.perf_counter()
tracker.track(models, tile_size=wl.tile, rng="philox", seed=1)
print("warm call", time.perf_counter() - t0, "workspace side", tracker._ctx_key[-1], "grown", getattr(tracker, "_grown_dim", None))
pr = cProfile.Profile()
pr.enable()
tracker.track(models, tile_size=wl.tile, rng="philox", seed=1)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
