"""glh_track on two streams (glh_set_track_streams): the two halves of the batch run their frame loops concurrently;
the results are those of one stream bit for bit (same kernel, same per-point arithmetic, same Philox keys)."""
import numpy as np
import pytest

from glimpse_amd import _lib, workloads

pytestmark = pytest.mark.gpu


def _run(wl, frames, T, streams, math="fast", seed=3):
    with _lib.Context(wl.P, wl.N, wl.O, device_id=0, max_tile=max(wl.tile), max_search_dim=192, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_math(math)
        ctx.set_track_streams(streams)
        ctx.set_frame(0)
        ctx.init_particles(seed=seed)
        for o in range(wl.O):
            ctx.init_templates(o, 0)
        ctx.record_moments(0)
        fr = list(range(1, T))
        # two calls: the second starts from a compact state and must order itself behind the first on both streams
        cut = T // 2
        ctx.track(fr[:cut], [1.0] * cut, [[j] * wl.O for j in fr[:cut]], seed=seed)
        used = [ctx.last_track_streams()]
        ctx.track(fr[cut:], [1.0] * (len(fr) - cut), [[j] * wl.O for j in fr[cut:]], seed=seed)
        used.append(ctx.last_track_streams())
        assert (ctx.point_status() == 0).all()
        return dict(moments=ctx.get_moments(0, T), particles=ctx.get_particles(), weights=ctx.get_weights(),
                    status=ctx.observer_status_frames(1, T - 1), used=used)


@pytest.mark.parametrize("name,P,N", [("C3", 37, 700), ("C5", 10, 600), ("C4", 5, 10000)])
def test_two_streams_equal_one_stream(name, P, N):
    T = 7
    wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=(512, 512) if name != "C5" else None)
    frames = [wl.frames(o) for o in range(wl.O)]
    one = _run(wl, frames, T, 1)
    two = _run(wl, frames, T, 2)
    assert one["used"] == [1, 1] and two["used"] == [2, 2]
    for key in ("moments", "particles", "weights", "status"):
        np.testing.assert_array_equal(one[key], two[key])
    assert np.isfinite(two["moments"]).all()


def test_automatic_choice_follows_the_batch_size():
    T = 3
    wl = workloads.Workload("C3", n_frames=T, n_points=8, n_particles=256, imgsz=(512, 512))
    frames = [wl.frames(o) for o in range(wl.O)]
    assert _run(wl, frames, T, 0)["used"] == [1, 1]  # a handful of points: one launch per frame
    # more points than the chip has compute units (256 on an MI355X): the two halves on two streams
    cus = _lib.device_compute_units(0)
    wl = workloads.Workload("C3", n_frames=T, n_points=cus + 8, n_particles=128, imgsz=(512, 512))
    assert _run(wl, [wl.frames(0)], T, 0)["used"] == [2, 2]
    wl = workloads.Workload("C3", n_frames=T, n_points=cus, n_particles=128, imgsz=(512, 512))
    assert _run(wl, [wl.frames(0)], T, 0)["used"] == [1, 1]
