"""Multi-GPU path on CPU: point sharding + the one end-of-sequence gather, world_size 2, gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from glimpse_amd import sharding  # noqa: E402


@pytest.mark.parametrize("n,world", [(10, 1), (10, 3), (4096, 8), (10000, 8), (5, 8), (0, 2)])
def test_shard_range_partitions_points_in_order(n, world):
    blocks = [sharding.shard_range(n, world, r) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = sharding.shard_sizes(n, world)
    assert sum(sizes) == n and max(sizes) - min(sizes) <= 1


def test_shard_range_rejects_bad_rank():
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from glimpse_amd import sharding
rank, world = sharding.init(backend="gloo")
assert world == 2
P, T = 7, 5                                  # ragged: 4 + 3 points
lo, hi = sharding.shard_range(P, world, rank)
full_means = np.arange(P * T * 6, dtype=float).reshape(P, T, 6)
full_sig = -full_means
status = np.arange(P, dtype=np.int64) % 3
got = sharding.gather_points([full_means[lo:hi], full_sig[lo:hi], status[lo:hi]], P)
if rank == 0:
    assert np.array_equal(got[0], full_means) and np.array_equal(got[1], full_sig)
    assert np.array_equal(got[2], status)
    print("GATHER_OK")
else:
    assert got is None
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_gather_points_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for rc, out, err in outs:
        assert rc == 0, err[-2000:]
    assert "GATHER_OK" in outs[0][1]
