"""Multi-GPU path on CPU: point sharding, the torch-free group (FileStore + host transport, world_size 2), the launcher,
and the example of a caller-issued gather inside a torch.distributed job (examples/torch_interop.py, gloo)."""
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
sys.path.insert(0, {root!r} + "/examples")
import torch_interop
from glimpse_amd import sharding
rank, world = torch_interop.init(backend="gloo")
assert world == 2
P, T = 7, 5                                  # ragged: 4 + 3 points
lo, hi = sharding.shard_range(P, world, rank)
full_means = np.arange(P * T * 6, dtype=float).reshape(P, T, 6)
full_sig = -full_means
status = np.arange(P, dtype=np.int64) % 3
got = torch_interop.gather_points([full_means[lo:hi], full_sig[lo:hi], status[lo:hi]], P)
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


GROUP_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from glimpse_amd import sharding
group = sharding.Group.from_env()
assert group.world == 2 and group.store is not None
P, T = 7, 5                                  # ragged: 4 + 3 points
sizes = sharding.shard_sizes(P, group.world)
lo, hi = sharding.shard_range(P, group.world, group.rank)
full = np.arange(P * T * 12, dtype=float).reshape(P, T, 12)
status = (np.arange(P) % 3).astype(np.uint32)


class FakeCtx:                               # the host transport only needs these three calls
    def get_moments(self, f0, n):
        return np.ascontiguousarray(np.transpose(full[lo:hi, f0:f0 + n], (1, 0, 2)))
    def point_status(self):
        return status[lo:hi]
    def sync(self):
        pass


ctx = FakeCtx()
assert group.attach(ctx, "host") == "host"
group.barrier()
assert group.max(10.0 + group.rank) == 11.0
got = group.gather_moments(ctx, 0, T, sizes)
arrs = group.gather_arrays([full[lo:hi], status[lo:hi]])
if group.rank == 0:
    mom, st = got
    assert mom.shape == (T, P, 12) and np.array_equal(mom, np.transpose(full, (1, 0, 2)))
    assert np.array_equal(st, status)
    assert np.array_equal(arrs[0], full) and np.array_equal(arrs[1], status)
    print("GROUP_OK")
else:
    assert got is None and arrs is None
path = group.store.path
group.close()
if group.rank == 0:
    assert not os.path.exists(path)
"""


def test_group_host_transport_world2(tmp_path):
    """The torch-free multi-rank path (FileStore rendezvous + host transport): what `bench.py --gpus 2` and
    `Tracker.track(parallel=2)` run when RCCL cannot make a communicator."""
    script = tmp_path / "group_worker.py"
    script.write_text(GROUP_WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLH_RENDEZVOUS_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for rc, out, err in outs:
        assert rc == 0, err[-2000:]
    assert "GROUP_OK" in outs[0][1]


def test_bench_launcher_starts_n_ranks(tmp_path, monkeypatch):
    """`bench.py --gpus N` without a launcher starts N rank processes with the torchrun environment before anything
    touches a GPU, relays rank 0's line and fails if a rank fails (the ranks are stubbed: no GPU here)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys, json\n"
                    "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                    "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                    "assert os.environ['LOCAL_RANK'] == os.environ['RANK']\n"
                    "open(os.path.join(os.path.dirname(__file__), f'seen.{r}'), 'w').write(str(w))\n"
                    "if r == 0: print(json.dumps({'n_gpus': w}))\n"
                    "sys.exit(int(os.environ.get('STUB_FAIL_RANK', '-1')) == r)\n")
    monkeypatch.setattr(bench, "__file__", str(stub))
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: p)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    args = bench.parse_args(["--gpus", "3"])
    assert bench.launch(args) == 0
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("seen.")) == ["seen.0", "seen.1", "seen.2"]
    monkeypatch.setenv("STUB_FAIL_RANK", "2")
    assert bench.launch(args) == 1


def test_launcher_returns_when_a_rank_dies_before_the_rendezvous(tmp_path, monkeypatch):
    """Rank 1 dies before it ever reaches the store; rank 0 and rank 2 are waiting in FileStore.get for its barrier
    file.  The launcher marks the rendezvous directory as aborted, the waiting ranks raise at once (not after the
    store's 600 s timeout), and launch() returns non-zero within seconds."""
    import importlib.util
    import time

    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stub = tmp_path / "stub_wait.py"
    stub.write_text("import os, sys\n"
                    f"sys.path.insert(0, {ROOT!r})\n"
                    "from glimpse_amd import sharding\n"
                    "if os.environ['RANK'] == '1':\n"
                    "    sys.exit(7)\n"
                    "g = sharding.Group.from_env()\n"
                    "try:\n"
                    "    g.store.barrier()\n"
                    "except RuntimeError as e:\n"
                    "    assert 'aborted' in str(e), e\n"
                    "    open(os.path.join(os.path.dirname(__file__), 'aborted.' + os.environ['RANK']), 'w').write(str(e))\n"
                    "    sys.exit(5)\n"
                    "sys.exit(0)\n")
    monkeypatch.setattr(bench, "__file__", str(stub))
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: p)
    monkeypatch.setenv("GLH_RENDEZVOUS_DIR", str(tmp_path))
    monkeypatch.setenv("GLH_LAUNCH_GRACE", "30")  # the ranks must leave by themselves, not by the kill timer
    args = bench.parse_args(["--gpus", "3"])
    t0 = time.monotonic()
    assert bench.launch(args, argv=["--gpus", "3"]) == 1
    assert time.monotonic() - t0 < 20
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("aborted.")) == ["aborted.0", "aborted.2"]


REATTACH_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
from glimpse_amd import sharding

class FakeCtx:
    def sync(self): pass

g = sharding.Group.from_env()
# two attaches on ONE group (a new context for the next sequence): each has its own store keys, so the second never
# reads the first one's id / votes
assert g.attach(FakeCtx(), "host") == "host"
assert g.attach(FakeCtx(), "host") == "host"
assert g._attaches == 2
g.barrier()
g.close()
print("REATTACH_OK")
"""


def test_group_attach_twice_uses_fresh_keys(tmp_path):
    script = tmp_path / "reattach_worker.py"
    script.write_text(REATTACH_WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLH_RENDEZVOUS_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and "REATTACH_OK" in out, err[-2000:]
