"""How one run from TIFF files is chunked into glh_track calls (round 5: the leg's call time varied 0.06 .. 0.10 s)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import bench
    from glimpse_amd import workloads

    T = 101
    wl = workloads.Workload("C3", n_frames=T)
    frames = bench.render_frames(wl, bench.usable_cores())
    keys = ("call_seconds", "call_seconds_min_max", "glh_track_calls", "frame_loop_waited_for_decoders_seconds",
            "decode_ms_per_frame_per_core")
    for rep in range(3):
        for fmt in sys.argv[1:] or ("tiff",):
            r = bench.from_files_leg(wl, frames, T, 1234, 0, fmt, 0.054)
            print(fmt, {k: (round(r[k], 4) if isinstance(r[k], float) else r[k]) for k in keys}, flush=True)


if __name__ == "__main__":
    main()
