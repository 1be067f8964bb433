"""Zero-copy hand-off of the library-owned moments history to torch (what bench.py gives RCCL)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_moments_device_pointer_is_a_valid_torch_view():
    import torch

    import bench
    from glimpse_amd import _lib, workloads

    T, P, N = 3, 4, 600
    wl = workloads.Workload("C2", n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
    frames = [wl.frames(0)]
    with _lib.Context(P, N, 1, max_search_dim=160, max_frames=T) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_frame(0)
        ctx.init_particles(seed=1)
        ctx.init_templates(0, 0)
        ctx.record_moments(0)
        for i in range(1, T):
            ctx.step(i, 1.0, [i], seed=1)
        ctx.sync()
        ptr, nbytes = ctx.moments_device()
        assert nbytes == T * P * 12 * 8
        view = torch.as_tensor(bench.DevArray(ptr, (T, P, 12)), device="cuda:0")
        assert view.dtype == torch.float64 and view.data_ptr() == ptr
        np.testing.assert_array_equal(view.cpu().numpy(), ctx.get_moments(0, T))
        # a collective-style consumer: contiguous, sendable as is
        assert view.is_contiguous()
        gathered = [torch.empty_like(view)]
        gathered[0].copy_(view)
        np.testing.assert_array_equal(gathered[0].cpu().numpy(), ctx.get_moments(0, T))
