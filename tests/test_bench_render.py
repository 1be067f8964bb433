"""bench.py's frame rendering: forked helpers write into shared memory, ground maps are computed in row bands and shared
between the scenes of a run, RGB frames are the channel remap of the gray ones -- all of it must give exactly the frames
a Workload renders by itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_render_all_equals_direct_rendering():
    import bench
    from glimpse_amd import synth, workloads

    T, size = 6, (256, 192)
    c1 = workloads.Workload("C1", n_frames=T, n_points=1, n_particles=50, imgsz=size)
    c5 = workloads.Workload("C5", n_frames=T, n_points=3, n_particles=50, imgsz=size)
    u16 = workloads.Workload("C1", n_frames=T, n_points=1, n_particles=50, imgsz=size)
    u16.bits = 16
    synth._GROUND_MAPS.clear()
    got = bench.render_all({"C1": c1, "C5": c5, "u16": u16}, 3, rgb_of=("rgb", "C1"))
    assert got["C1"][0].shape == (T, size[1], size[0]) and got["C1"][0].dtype == np.uint8
    assert got["rgb"][0].shape == (T, size[1], size[0], 3) and got["u16"][0].dtype == np.uint16
    assert len(got["C5"]) == 2
    # the same workloads rendered one frame at a time, from fresh ground maps
    synth._GROUND_MAPS.clear()
    d1 = workloads.Workload("C1", n_frames=T, n_points=1, n_particles=50, imgsz=size)
    d5 = workloads.Workload("C5", n_frames=T, n_points=3, n_particles=50, imgsz=size)
    for t in range(T):
        np.testing.assert_array_equal(got["C1"][0][t], d1.frame(0, t))
        for o in range(2):
            np.testing.assert_array_equal(got["C5"][o][t], d5.frame(o, t))
    d1.channels = 3
    np.testing.assert_array_equal(got["rgb"][0][T - 1], d1.frame(0, T - 1))
    d1.channels, d1.bits = 1, 16
    np.testing.assert_array_equal(got["u16"][0][2], d1.frame(0, 2))
    # one worker: no helpers, same frames
    serial = bench.render_all({"C1": workloads.Workload("C1", n_frames=T, n_points=1, n_particles=50, imgsz=size)}, 1)
    np.testing.assert_array_equal(serial["C1"][0], got["C1"][0])


def test_ground_map_bands_equal_the_whole_map():
    from glimpse_amd import synth

    cam = synth.nadir_camera((96, 80), f=300.0, height=50.0, k=(0.05, -0.01, 0.002, 0, 0, 0))
    whole = synth.ground_rows(cam, 0, 80)
    parts = np.concatenate([synth.ground_rows(cam, r, min(80, r + 32)) for r in range(0, 80, 32)])
    np.testing.assert_array_equal(whole, parts)
    g = np.arange(256, dtype=np.uint8).reshape(16, 16)
    rgb = synth.gray_to_rgb(g)
    i = g.astype(np.int32)
    np.testing.assert_array_equal(rgb[..., 0], g)
    np.testing.assert_array_equal(rgb[..., 1], np.clip(i + ((i * 7) % 5) - 2, 0, 255))
    np.testing.assert_array_equal(rgb[..., 2], np.clip(255 - i // 2, 0, 255))


def test_pmc_traffic_is_scaled_to_a_frame_by_the_points_of_the_profiled_launch():
    """roofline.traffic: the committed PMC figure is per LAUNCH of the profiled run (two streams: half of the points); the
    bench line reports the bytes of a frame update of all the workload's points, whatever the launches of this run hold."""
    import json

    import bench
    from glimpse_amd import workloads

    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    wl = workloads.Workload("C3", n_frames=3)
    entry = table["C3:4096x5000"]["k_point_step"]
    per_frame = bench.pmc_traffic_per_frame(wl, "k_point_step")
    assert per_frame == entry["hbm_bytes_per_launch"] * wl.P / entry["points_per_launch"]
    # measured traffic within a few percent of the algorithmic figure at C3 (96 B of state per particle-frame + tiles)
    assert 0.85 < per_frame / (100.2 * wl.P * wl.N) < 1.1
    assert "not measured in this run" in bench.pmc_traffic_source(wl, "k_point_step")
    assert bench.pmc_traffic_per_frame(workloads.Workload("C1", n_frames=3), "k_point_step") is None
