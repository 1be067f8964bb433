"""Diagnostic: mean cycles per phase of the fused per-point kernel on a synthetic workload.

    python tools/phase_probe.py [C3] [points] [particles] [frames]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimpse_amd import _lib, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
P = int(sys.argv[2]) if len(sys.argv) > 2 else None
N = int(sys.argv[3]) if len(sys.argv) > 3 else None
T = int(sys.argv[4]) if len(sys.argv) > 4 else 8
wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N)
wl.bits = int(os.environ.get("GLH_BITS", "8"))          # 16: uint16 frames (pt_tile_prep_wide)
wl.channels = int(os.environ.get("GLH_CHANNELS", "1"))
HP = os.environ.get("GLH_HP")                            # e.g. 3: a 3 x 3 median, i.e. the general instantiation
if wl.bits >= 32:  # float frames as bench.py's C3_f32 leg makes them: the 8-bit scene scaled to [0, 1]
    wl.bits = 8
    ft = np.float32 if os.environ["GLH_BITS"] == "32" else np.float64
    frames = [[np.asarray(f, dtype=ft) * ft(1.0 / 255.0) for f in wl.frames(o)] for o in range(wl.O)]
    wl.bits = int(os.environ["GLH_BITS"])
else:
    frames = [wl.frames(o) for o in range(wl.O)]
NAMES = ["", "A evolve+project", "B tile_prep", "B ssd", "B spline_fit", "C sample", "C exp", "D resample",
         "E gather", "F moments"]
with _lib.Context(wl.P, wl.N, wl.O, max_frames=T, max_search_dim=255 if wl.bits >= 16 else 320) as ctx:
    workloads.setup_context(ctx, wl, frames)
    if HP:
        ctx.set_highpass((int(HP), int(HP)))
    if os.environ.get("GLH_MOTION") or os.environ.get("GLH_DEM"):  # e.g. tangent_cartesian / gridded (bench.py: apply_motion)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        bench.apply_motion(ctx, wl, os.environ.get("GLH_MOTION", "cartesian"), os.environ.get("GLH_DEM", "constant"))
    ctx.set_math(os.environ.get("GLH_MATH", "fast"))
    ctx.set_frame(0)
    ctx.init_particles(seed=3)
    for o in range(wl.O):
        ctx.init_templates(o, 0)
    ctx.record_moments(0)
    ctx.phase_stamps()  # arm
    ctx.profile_enable(True)
    for i in range(1, T):
        if i == T - 1:
            ctx.profile_reset()
        ctx.step(i, 1.0, [i] * wl.O, seed=3)
    ctx.sync()
    ms = {k: v for k, v in ctx.profile_get().items() if v[0] > 0}
    st = ctx.phase_stamps().astype(np.int64)
    sub = st[:, 10:13]
    tp = st[:, 13:15]
    print('  tile_prep split (median ticks): fetch+hist', np.median(tp[:,0]-st[:,1]), 'scan+lut', np.median(tp[:,1]-tp[:,0]), 'median+write', np.median(st[:,2]-tp[:,1]))
    asub = st[:, 15:20]
    print('  A split (median ticks): prologue', np.median(asub[:,0]-st[:,0]), 'record staging', np.median(asub[:,1]-asub[:,0]), 'particle loop', np.median(asub[:,2]-asub[:,1]), 'wait for the other waves + reduce', np.median(asub[:,3]-asub[:,2]), 'box (thread 0)', np.median(asub[:,4]-asub[:,3]), 'barrier', np.median(st[:,1]-asub[:,4]))
    if wl.bits == 32:  # the float32 branch stamps 20..22 (normalize_box_f32_block) and 13 (ranked)
        fs = st[:, 20:23]
        print('  float split (median ticks): fetch', np.median(fs[:,0]-st[:,1]), 'sum', np.median(fs[:,1]-fs[:,0]), 'squares + sum + write', np.median(fs[:,2]-fs[:,1]),
              'rank', np.median(tp[:,0]-fs[:,2]), 'matched values', np.median(tp[:,1]-tp[:,0]), 'highpass', np.median(st[:,2]-tp[:,1]))
    st = st[:, :10]
    d = np.diff(st, axis=1)  # (P, 9)
    tot = d.sum(axis=1)
    span = st[:, -1].max() - st[:, 0].min()
    print(f"{wl.describe()['workload']}")
    print(f"  last step kernels (ms): {ms}")
    print(f"  kernel span {span} ticks; per-point total: min {tot.min()} median {np.median(tot):.0f} max {tot.max()}")
    print(f"  per-point total percentiles 50/90/99: {np.percentile(tot, [50, 90, 99])}")
    k_ms = ms["point_step"][0]
    print(f"  sum of workgroup lifetimes / kernel time = {tot.sum() / (k_ms * 1e-3) / 1e9:.1f} G tick-slots/s "
          f"(512 resident workgroups at 100 MHz ticks would be 51.2; a dispatch tail shows as less)")
    t0 = st[:, 0] - st[:, 0].min()
    t1 = st[:, -1] - st[:, 0].min()
    print(f"  block start ticks percentiles 25/50/75/100: {np.percentile(t0, [25, 50, 75, 100])}; last end {t1.max()}")
    for k in range(9):
        print(f"  {NAMES[k + 1]:18s} median {np.median(d[:, k]):10.0f} ticks  {100 * d[:, k].sum() / tot.sum():5.1f} %")
    if (sub > 0).all():
        print(f"  D split (median ticks): sum tree {np.median(sub[:, 0] - st[:, 6]):.0f}, scan {np.median(sub[:, 1] - sub[:, 0]):.0f}, "
              f"fix-up {np.median(sub[:, 2] - sub[:, 1]):.0f}, search+scatter {np.median(st[:, 7] - sub[:, 2]):.0f}")
    bx = ctx.search_boxes()[0]
    wsz, hsz = bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]
    print("  search tile w x h median:", np.median(wsz), np.median(hsz), "max:", wsz.max(), hsz.max())
    slow = np.argsort(tot)[-5:]
    print("  slowest points: total", tot[slow], "tile", list(zip(wsz[slow], hsz[slow])))
