"""Zero-copy hand-off of the library-owned moments history to torch (for callers that issue their own collective:
examples/torch_interop.py -- the product package itself never imports torch).

Runs in a fresh interpreter: torch bundles its own HIP runtime and must be imported BEFORE
libglimpse_hip.so is loaded (a torch job would have done exactly that); inside the shared pytest process
other tests have already loaded the library."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/examples")
import torch                      # first: see the module docstring
import numpy as np
import torch_interop
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
    view = torch.as_tensor(torch_interop.DeviceArray(ptr, (T, P, 12)), device="cuda:0")
    assert view.dtype == torch.float64 and view.data_ptr() == ptr and view.is_contiguous()
    want = ctx.get_moments(0, T)
    np.testing.assert_array_equal(view.cpu().numpy(), want)
    recv = torch.empty_like(view)   # what dist.gather fills on rank 0
    recv.copy_(view)
    np.testing.assert_array_equal(recv.cpu().numpy(), want)
print("INTEROP_OK")
"""


def test_moments_device_pointer_is_a_valid_torch_view(tmp_path):
    script = tmp_path / "interop.py"
    script.write_text(SCRIPT.format(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "INTEROP_OK" in r.stdout, r.stderr[-3000:]
