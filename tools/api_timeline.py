"""Wall time of every device-context call inside one warm glimpse_amd.Tracker.track(rng="philox") at C3 (no profiler:
the context's methods are wrapped with timers)."""
import collections
import datetime
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glimpse_amd as g  # noqa: E402
from glimpse_amd import _lib, workloads  # noqa: E402

wl = workloads.Workload(sys.argv[1] if len(sys.argv) > 1 else "C3")
frames = [wl.frames(o) for o in range(wl.O)]
t_start, unit = datetime.datetime(2020, 1, 1), datetime.timedelta(days=1)
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
acc = collections.OrderedDict()


def wrap(name, fn):
    def inner(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return inner


for name in dir(_lib.Context):
    if not name.startswith("_") and callable(getattr(_lib.Context, name)):
        setattr(_lib.Context, name, wrap(name, getattr(_lib.Context, name)))
import glimpse_amd.tracker as tr  # noqa: E402
tr.params_table = wrap("params_table (python)", tr.params_table)
tr._batches = wrap("_batches (python)", tr._batches)
import gc  # noqa: E402
for rep in range(int(os.environ.get("API_REPS", "1"))):
    acc.clear()
    if os.environ.get("API_GC_OFF"):
        gc.disable()
    t0 = time.perf_counter()
    tracks = tracker.track(models, tile_size=wl.tile, rng="philox", seed=1)
    total = time.perf_counter() - t0
    if os.environ.get("API_KEEP"):
        keep = tracks  # (the previous result stays alive: its 40 MB of arrays are not unmapped and mapped again)
    print(f"warm call {1e3 * total:.2f} ms = {1e3 * total / (wl.T - 1):.4f} ms/step")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]:
        print(f"  {k:32s} {1e3 * v:8.3f} ms")
    print(f"  {'(everything else)':32s} {1e3 * (total - sum(acc.values())):8.3f} ms")
