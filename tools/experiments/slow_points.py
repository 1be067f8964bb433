"""Experiment (round 5): which phases make a slow point slow?  Phase stamps of one frame: the medians over all points
beside the medians over the slowest 5 %.      python tools/experiments/slow_points.py C2 256 2000 30"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glimpse_amd import _lib, workloads  # noqa: E402

name, P, N, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N)
frames = [wl.frames(o) for o in range(wl.O)]
NAMES = ["A evolve+project", "B tile_prep", "B ssd", "B spline_fit", "C sample", "C exp", "D resample", "E gather", "F moments"]
with _lib.Context(wl.P, wl.N, wl.O, max_frames=T, max_search_dim=320) as ctx:
    workloads.setup_context(ctx, wl, frames)
    ctx.set_math("fast")
    ctx.set_track_streams(1)
    ctx.set_frame(0)
    ctx.init_particles(seed=3)
    for o in range(wl.O):
        ctx.init_templates(o, 0)
    ctx.record_moments(0)
    ctx.phase_stamps()
    for i in range(1, T):
        ctx.step(i, 1.0, [i] * wl.O, seed=3)
    ctx.sync()
    st = ctx.phase_stamps().astype(np.int64)
    bx = ctx.search_boxes()
d = np.diff(st[:, :10], axis=1)
tot = d.sum(axis=1)
slow = np.argsort(tot)[-max(3, P // 20):]
area = [(bx[o][:, 2] - bx[o][:, 0]) * (bx[o][:, 3] - bx[o][:, 1]) for o in range(wl.O)]
print(f"{name} {P} x {N}: lifetime median {np.median(tot):.0f}, slowest 5 % median {np.median(tot[slow]):.0f}")
print("  tile pixels (observer 0): median", np.median(area[0]), " slowest 5 %:", np.median(area[0][slow]))
for k, n in enumerate(NAMES):
    print(f"  {n:18s} all {np.median(d[:, k]):8.0f}   slowest 5 % {np.median(d[slow, k]):8.0f}   (+{np.median(d[slow, k]) - np.median(d[:, k]):.0f})")
a = st[:, 15:20]
for label, x in (("A loop", a[:, 2] - a[:, 1]), ("A wait+reduce", a[:, 3] - a[:, 2]), ("A box", a[:, 4] - a[:, 3])):
    print(f"  {label:18s} all {np.median(x):8.0f}   slowest 5 % {np.median(x[slow]):8.0f}")
print("  the three slowest points:")
for p_ in np.argsort(tot)[-3:][::-1]:
    tiles = ["%dx%d" % (bx[o][p_, 2] - bx[o][p_, 0], bx[o][p_, 3] - bx[o][p_, 1]) for o in range(wl.O)]
    print(f"    point {p_}: lifetime {tot[p_]}, tiles {tiles}, phases " + " ".join(f"{n.split()[-1]}={d[p_, k]}" for k, n in enumerate(NAMES)))
print("  the median point's tiles:", ["%dx%d" % (np.median(bx[o][:, 2] - bx[o][:, 0]), np.median(bx[o][:, 3] - bx[o][:, 1])) for o in range(wl.O)])
