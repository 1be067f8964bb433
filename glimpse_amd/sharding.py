"""Multi-GPU sharding of tracked points (the reference's `parallel=` of track/tracker.py:381-387).

Tracks are independent (each owns its particles, weights, templates and random draws), so
the points are split into contiguous blocks, one block per process = per GPU, and nothing is exchanged while
a sequence runs.  At the end the per-point posterior moments are collected on rank 0 with ONE exchange, in
`motion_models` order.

The product path is torch-free: `Group` rendezvouses through a directory (`FileStore`: the 128-byte RCCL id and
a few words travel as files; one node, like the reference's process pool) and the gather itself is RCCL over
xGMI inside libglimpse_hip.so (`glh_gather_moments`).  If RCCL cannot make the communicator (librccl missing,
several ranks told to share one GPU) the group says so (`transport == "host"`) and gathers host copies through
the same directory instead -- never silently.

    # launched as N processes with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT set (torchrun does; so does
    # `bench.py --gpus N` and `Tracker.track(parallel=N)`)
    group = sharding.Group.from_env()
    lo, hi = sharding.shard_range(len(models), group.world, group.rank)
    ctx = _lib.Context(hi - lo, n, O, device_id=group.local_rank); ...; ctx.set_point_offset(lo)
    group.attach(ctx)
    ... track ...
    moments, status = group.gather_moments(ctx, 0, T, sharding.shard_sizes(len(models), group.world))   # rank 0

Callers that already live inside a torch.distributed job and want to issue the collective themselves: `examples/torch_interop.py`
(a zero-copy view of the library's history buffer, the same gather with gloo / nccl); nothing in this package imports torch.
"""
import os
import tempfile
import time

import numpy as np


def shard_range(n_points, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; the remainder goes to the first ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("need 0 <= rank < world")
    base, extra = divmod(int(n_points), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_points, world):
    return [hi - lo for lo, hi in (shard_range(n_points, world, r) for r in range(world))]


class _stdout_to_stderr:
    """File descriptor 1 -> 2 for the duration (native libraries write to the descriptor, not to sys.stdout): a
    launcher that parses this process's stdout sees only what the caller prints."""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        try:  # what the library printed sits in the C runtime's buffer: push it out while 1 still points at stderr
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)


# ---- rendezvous -------------------------------------------------------------------------------
class FileStore:
    """Key -> bytes through a directory shared by the ranks of one node (tmpfs when there is one).

    Writes are atomic (temporary name + rename), reads poll.  Keys are used once; `barrier` counts its calls, so
    the k-th barrier of every rank meets the k-th of the others."""

    def __init__(self, path, rank, world, timeout=600.0):
        self.path, self.rank, self.world, self.timeout = path, int(rank), int(world), float(timeout)
        os.makedirs(path, exist_ok=True)
        self._barriers = 0

    @staticmethod
    def default_path(token):
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        return os.path.join(os.environ.get("GLH_RENDEZVOUS_DIR", base), f"glh_rdzv_{token}")

    def put(self, key, data):
        tmp = os.path.join(self.path, f".{key}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, os.path.join(self.path, key))

    ABORT = "abort"  # written by a rank (or the launcher) that is giving up: every waiting `get` then raises

    def abort(self, why=""):
        """Tell every rank that is (or will be) waiting on this store that the job is over."""
        try:
            self.put(self.ABORT, str(why).encode())
        except OSError:
            pass

    def get(self, key):
        target = os.path.join(self.path, key)
        stop = os.path.join(self.path, self.ABORT)
        deadline = time.monotonic() + self.timeout
        delay = 0.0002
        while not os.path.exists(target):
            if os.path.exists(stop):
                try:
                    with open(stop, "rb") as f:
                        why = f.read().decode(errors="replace")
                except OSError:
                    why = ""
                raise RuntimeError(f"rank {self.rank}: the job was aborted while waiting for {key!r}: {why}")
            if time.monotonic() > deadline:
                raise TimeoutError(f"rank {self.rank}: no {key!r} in {self.path} after {self.timeout:.0f} s")
            time.sleep(delay)
            delay = min(delay * 1.5, 0.005)
        with open(target, "rb") as f:
            return f.read()

    def put_array(self, key, a):
        tmp = os.path.join(self.path, f".{key}.{self.rank}.tmp.npy")
        np.save(tmp, np.ascontiguousarray(a))
        os.replace(tmp, os.path.join(self.path, key + ".npy"))

    def get_array(self, key, mmap=False):
        self.get(key + ".npy")
        return np.load(os.path.join(self.path, key + ".npy"), mmap_mode="r" if mmap else None)

    def barrier(self, tag="b"):
        k = self._barriers
        self._barriers += 1
        self.put(f"{tag}.{k}.{self.rank}", b"1")
        for r in range(self.world):
            self.get(f"{tag}.{k}.{r}")

    def leave(self):
        """Last call of every rank: the others say that they are done reading, rank 0 waits for them and removes
        the directory (nobody is left polling a file that has been deleted)."""
        if self.rank != 0:
            self.put(f"left.{self.rank}", b"1")
            return
        import shutil

        for r in range(1, self.world):
            self.get(f"left.{r}")
        shutil.rmtree(self.path, ignore_errors=True)


class Group:
    """The ranks of one job: rank / world / local_rank, the store, and (after `attach`) the transport."""

    def __init__(self, rank=0, world=1, local_rank=0, store=None):
        self.rank, self.world, self.local_rank = int(rank), int(world), int(local_rank)
        self.store = store
        self.transport = "none" if world == 1 else "host"
        self._ctx = None
        self._seq = 0
        self._attaches = 0  # attach() calls so far: store keys are write-once, every attach has its own

    @classmethod
    def from_env(cls, env=None):
        """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT as torchrun (and our own launchers) export them.  The store
        directory is keyed on the port and on the launcher's pid, so that stale files of an earlier job are never
        read."""
        env = os.environ if env is None else env
        world = int(env.get("WORLD_SIZE", "1"))
        rank = int(env.get("RANK", "0"))
        local = int(env.get("LOCAL_RANK", str(rank)))
        if world == 1:
            return cls(0, 1, local, None)
        token = env.get("GLH_RENDEZVOUS_TOKEN") or f"{env.get('MASTER_PORT', '0')}_{os.getppid()}"
        return cls(rank, world, local, FileStore(FileStore.default_path(token), rank, world))

    # -- transport
    def attach(self, ctx, transport=None):
        """Make the RCCL communicator on `ctx` (collective).  `transport`: None = RCCL if every rank can, "rccl" =
        RCCL or raise, "host" = skip RCCL.  Returns the transport in force."""
        self._ctx = ctx
        if self.world == 1 and transport != "rccl":
            self.transport = "none"
            return self.transport
        from . import _lib

        want = transport or os.environ.get("GLH_COMM") or None
        ok, why = False, ""
        # (a second attach on this group -- a new context for the next sequence, a retry after falling back to the
        # host transport -- must not read the previous communicator's id or votes)
        k = self._attaches
        self._attaches += 1
        if want != "host":
            # rank 0 makes the id; an EMPTY id tells the others that it could not (they then skip the collective
            # ncclCommInitRank instead of waiting for a root that never comes)
            cid = b""
            if self.rank == 0:
                try:
                    cid = _lib.comm_unique_id()
                except _lib.GlhError as e:  # librccl missing
                    why = str(e)
                if self.store is not None:
                    self.store.put(f"rccl_id.{k}", cid)
            else:
                cid = self.store.get(f"rccl_id.{k}")
                why = "" if cid else "rank 0 could not make an RCCL id"
            if cid:
                try:
                    with _stdout_to_stderr():  # librccl prints a version banner on stdout when it initialises
                        ctx.comm_init(cid, self.rank, self.world)
                    ok = True
                except _lib.GlhError as e:  # e.g. several ranks on one GPU
                    why = str(e)
        if self.store is not None:  # all or nothing
            self.store.put(f"rccl_ok.{k}.{self.rank}", b"1" if ok else b"0")
            every = all(self.store.get(f"rccl_ok.{k}.{r}") == b"1" for r in range(self.world))
        else:
            every = ok
        if ok and not every:
            ctx.comm_destroy()
        if want == "rccl" and not every:
            raise RuntimeError(f"RCCL transport requested but unavailable: {why or 'another rank failed'}")
        self.transport = "rccl" if every else "host"
        self.why_host = why
        return self.transport

    def barrier(self):
        if self.world == 1 and self.transport != "rccl":
            if self._ctx is not None:
                self._ctx.sync()
            return
        if self.transport == "rccl":
            self._ctx.comm_barrier()
        else:
            if self._ctx is not None:
                self._ctx.sync()
            self.store.barrier()

    def max(self, value):
        """Max over the ranks of a host float (every rank gets it)."""
        if self.transport == "rccl":
            return self._ctx.comm_max(value)
        if self.world == 1:
            return float(value)
        k = self._seq
        self._seq += 1
        self.store.put(f"max.{k}.{self.rank}", repr(float(value)).encode())
        return max(float(self.store.get(f"max.{k}.{r}").decode()) for r in range(self.world))

    def gather_moments(self, ctx, frame0, n_frames, points_per_rank, root=0, download=True):
        """Every rank's moments history [n_frames][P_rank][12] and status words to `root`: (moments
        (n_frames, sum P, 12), status (sum P,)) there, None elsewhere.  RCCL inside the library, or host copies
        through the store when the group runs on the "host" transport.  download=False (RCCL): the blocks stay on
        the root's device -- the exchange is all that happens -- and `ctx.gathered()` fetches them later."""
        if self.transport == "rccl":
            return ctx.gather_moments(frame0, n_frames, points_per_rank, root=root, download=download)
        mine = ctx.get_moments(frame0, n_frames)
        status = ctx.point_status()
        if self.world == 1:
            return mine, status
        k = self._seq
        self._seq += 1
        if self.rank != root:
            self.store.put_array(f"mom.{k}.{self.rank}", mine)
            self.store.put_array(f"st.{k}.{self.rank}", status)
            return None
        moms = [mine if r == root else self.store.get_array(f"mom.{k}.{r}") for r in range(self.world)]
        sts = [status if r == root else self.store.get_array(f"st.{k}.{r}") for r in range(self.world)]
        for r, m in enumerate(moms):
            if m.shape != (n_frames, points_per_rank[r], 12):
                raise ValueError(f"rank {r} sent moments {m.shape}, expected {(n_frames, points_per_rank[r], 12)}")
        return np.concatenate(moms, axis=1), np.concatenate(sts)

    def gather_arrays(self, arrays, root=0):
        """Per-point host arrays (leading axis = this rank's points) to `root` through the store, concatenated in
        rank order; None elsewhere."""
        if self.world == 1:
            return list(arrays)
        k = self._seq
        self._seq += 1
        if self.rank != root:
            for j, a in enumerate(arrays):
                self.store.put_array(f"arr.{k}.{j}.{self.rank}", a)
            return None
        out = []
        for j, a in enumerate(arrays):
            parts = [np.asarray(a) if r == root else self.store.get_array(f"arr.{k}.{j}.{r}") for r in range(self.world)]
            out.append(np.concatenate(parts, axis=0))
        return out

    def close(self):
        if self._ctx is not None and self.transport == "rccl":
            self._ctx.comm_destroy()
        if self.store is not None:
            self.store.leave()
