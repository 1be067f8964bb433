"""Contexts come and go without leaving device memory behind: every workspace a sequence grows on first use (float / 16-bit
workspaces, the second stream's events, phase stamps, debug buffers) is released by glh_destroy."""
import numpy as np
import pytest

from glimpse_amd import _lib, workloads

pytestmark = pytest.mark.gpu


def _sequence(bits, streams, T=4):
    wl = workloads.Workload("C3", n_frames=T, n_points=24, n_particles=900, imgsz=(384, 384))
    frames = [wl.frames(o) for o in range(wl.O)]
    if bits == 32:
        frames = [[np.asarray(f, dtype=np.float32) * np.float32(1.0 / 255.0) for f in fo] for fo in frames]
    elif bits == 16:
        frames = [[np.asarray(f, dtype=np.uint16) * np.uint16(257) for f in fo] for fo in frames]
    wl.bits = bits
    with _lib.Context(wl.P, wl.N, wl.O, device_id=0, max_tile=max(wl.tile), max_search_dim=160, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_math("fast")
        ctx.set_track_streams(streams)
        ctx.phase_stamps()  # (arms the diagnostic stamps: one more buffer to release)
        ctx.set_frame(0)
        ctx.init_particles(seed=5)
        ctx.init_templates(0, 0)
        ctx.record_moments(0)
        fr = list(range(1, T))
        ctx.track(fr, [1.0] * len(fr), [[j] for j in fr], seed=5)
        assert ctx.last_track_streams() == streams
        assert (ctx.point_status() == 0).all()
        return ctx.get_moments(0, T)


def test_contexts_release_their_device_memory():
    _sequence(8, 2)  # (the library's one-time allocations -- module, streams' pools -- happen here)
    _sequence(16, 1)
    _sequence(32, 2)
    free0, total = _lib.device_memory(0)
    first = None
    for rep in range(6):
        for bits, streams in ((8, 2), (16, 1), (32, 2)):
            m = _sequence(bits, streams)
            if rep == 0 and bits == 8:
                first = m
            elif bits == 8:
                np.testing.assert_array_equal(m, first)  # (and a fresh context repeats the run bit for bit)
    free1, _ = _lib.device_memory(0)
    assert free0 - free1 < 32 << 20, f"{(free0 - free1) >> 20} MiB of {total >> 20} gone after 18 contexts"
