"""Experiment (round 5): is a point's workgroup lifetime persistent from frame to frame?  A launch of one round of
workgroups ends with its slowest point; several frames in ONE launch (every workgroup loops over the frames of its point)
would replace the sum of the per-frame maxima by the maximum of the per-point sums -- a gain only if the slow points of
one frame are not the slow points of the next.

    python tools/experiments/lifetime_persistence.py C2 256 2000 40
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glimpse_amd import _lib, workloads  # noqa: E402

name, P, N, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N)
frames = [wl.frames(o) for o in range(wl.O)]
with _lib.Context(wl.P, wl.N, wl.O, max_frames=T, max_search_dim=320) as ctx:
    workloads.setup_context(ctx, wl, frames)
    ctx.set_math("fast")
    ctx.set_track_streams(1)
    ctx.set_frame(0)
    ctx.init_particles(seed=3)
    for o in range(wl.O):
        ctx.init_templates(o, 0)
    ctx.record_moments(0)
    ctx.phase_stamps()  # arm
    life = []
    for i in range(1, T):
        ctx.step(i, 1.0, [i] * wl.O, seed=3)
        ctx.sync()
        st = ctx.phase_stamps().astype(np.int64)
        life.append(st[:, 9] - st[:, 0])
life = np.array(life[8:], dtype=float)  # (steady state) [frames][points]
per_frame_max = life.max(axis=1)
print(f"{name} {P} x {N}: frames {life.shape[0]}, lifetime median {np.median(life):.0f}, mean of the per-frame maxima {per_frame_max.mean():.0f}")
print(f"  sum of per-frame maxima {per_frame_max.sum():.0f}  vs  maximum of per-point sums {life.sum(axis=0).max():.0f}  "
      f"(ratio {life.sum(axis=0).max() / per_frame_max.sum():.3f}; mean-based bound {life.mean() * life.shape[0] / per_frame_max.sum():.3f})")
c = np.corrcoef(life[:-1].ravel(), life[1:].ravel())[0, 1]
print(f"  correlation of a point's lifetime with its lifetime in the next frame: {c:.3f}")
slow = np.argsort(life.mean(axis=0))[-5:]
print("  slowest points' mean lifetimes:", life.mean(axis=0)[slow].round(0), "overall mean", life.mean().round(0))
