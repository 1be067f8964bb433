"""Where the time of one warm Tracker.track() call goes beyond the frame loop (round 5): cProfile of the drop-in call on C3.

    python tools/experiments/api_profile.py [WORKLOAD] [FRAMES]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import bench
    import glimpse_amd as g
    from glimpse_amd import workloads

    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 101
    wl = workloads.Workload(name, n_frames=T)
    frames = bench.render_frames(wl, bench.usable_cores())
    observers, models = bench._api_objects(wl, frames, T)
    tracker = g.Tracker(observers, device=0)
    tracker.track(models, tile_size=wl.tile, rng="philox", seed=1)
    for _ in range(2):
        t0 = time.perf_counter()
        tracker.track(models, tile_size=wl.tile, rng="philox", seed=1)
        print(f"warm call {time.perf_counter() - t0:.4f} s", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    tracker.track(models, tile_size=wl.tile, rng="philox", seed=1)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)
    tracker.close()


if __name__ == "__main__":
    main()
