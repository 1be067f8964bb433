"""Worker processes of `Tracker.track(parallel=N)` (the reference's process pool, track/tracker.py:381-387,
helpers.py:2008-2017; its frames reach the pool through anonymous shared memory, image.py:209 `sharedmem.copy`).

The reference maps `process` over the tracks on `parallel` CPU processes.  Here the workers are GPU ranks:

* N PERSISTENT processes, started once (fresh interpreters: nothing of the parent's GPU state is inherited), one
  context each on GPU (worker mod device count), kept between `track` calls -- a second call finds its context, its
  uploaded frames and its workspaces in place;
* the observers' frames travel ONCE, through `multiprocessing.shared_memory` blocks the workers map (never pickled:
  what crosses the pipes is the tracks' parameter table -- `motion.ModelBlock`; the model objects themselves only for
  blocks that are not one device batch --, a few arguments, and per-track errors / warnings);
* every worker tracks a contiguous block of the tracks with `point_offset` = its first track (the device RNG is keyed
  on the global track index: the parallel run draws what the single-process run draws);
* the posterior history [T][P][12] and the status words are collected on worker 0 with ONE exchange -- the grouped
  ncclSend / ncclRecv of `glh_gather_moments` (sharding.Group, RCCL over xGMI) when every worker can join the
  communicator; otherwise (librccl missing, several workers told to share one GPU) every worker hands its host copy over
  through shared memory, and the run says so: `Tracks.transport`, a warning of the `glimpse_amd` logger.  Particles,
  weights and covariances (`return_particles`, `return_covariances`) are per-worker host downloads either way.
"""
import atexit
import copy
import logging
import multiprocessing as mp
import multiprocessing.connection as mpc
import os
import traceback
import weakref
import zlib
from multiprocessing import shared_memory

import numpy as np

from . import sharding

log = logging.getLogger("glimpse_amd")
_BIG = 1 << 16  # arrays of a reply beyond this many bytes travel through shared memory, not the pipe


class without_main:
    """Context manager: processes started inside it (spawn) do NOT import the parent's `__main__` module again.

    multiprocessing's spawn re-runs the main script in every child (as `__mp_main__`) so that objects defined there can be
    unpickled; a script without an `if __name__ == "__main__":` guard then runs whole in every child -- tracking, starting
    pools of its own, opening the GPU fifteen times.  The decoder processes of a run from image files are started by a plain
    `Tracker.track()` call: nobody expects to guard a script for that, and nothing they receive is defined in `__main__`.
    The workers of `track(parallel=N)` skip the import too unless a motion model or a `reduce_particles` function comes from
    `__main__` (then the script needs its guard, as every multiprocessing program does)."""

    def __enter__(self):
        import multiprocessing.spawn as spawn

        self._spawn, self._orig = spawn, spawn.get_preparation_data

        def prep(name):
            d = self._orig(name)
            d.pop("init_main_from_path", None)
            d.pop("init_main_from_name", None)
            return d

        spawn.get_preparation_data = prep

    def __exit__(self, *exc):
        self._spawn.get_preparation_data = self._orig


class _with_main:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        pass


def _in_child():
    """This process is a worker / decoder of glimpse_amd: it must not start pools of its own."""
    return os.environ.get("GLH_POOL_CHILD") == "1"


# ---- arrays through shared memory -----------------------------------------------------------------
def shm_room(nbytes):
    """/dev/shm has room for a block of `nbytes` (and a margin).  A shared-memory block is a sparse file on a tmpfs: making
    it never fails, WRITING beyond what the tmpfs holds is a SIGBUS that kills the writer (a container's default /dev/shm is
    64 MB) -- so the room is looked at before every block is made."""
    try:
        st = os.statvfs("/dev/shm")
    except OSError:
        return True  # (no /dev/shm to look at: the block's creation decides)
    return st.f_bavail * st.f_frsize >= int(nbytes) + (8 << 20)


def _export(a):
    """ndarray -> a picklable handle (the bytes in a shared-memory block the receiver unlinks); small arrays as they are
    (and large ones too, through the pipe, when shared memory has no room for them)."""
    if not isinstance(a, np.ndarray) or a.nbytes < _BIG or a.dtype == object or not shm_room(a.nbytes):
        return a
    a = np.ascontiguousarray(a)
    shm = shared_memory.SharedMemory(create=True, size=a.nbytes)
    np.ndarray(a.shape, a.dtype, buffer=shm.buf)[...] = a
    handle = ("__shm__", shm.name, a.shape, a.dtype.str)
    shm.close()  # (the receiver unlinks it; should it never get there, the resource tracker -- the parent's, shared by the
    return handle  # workers -- removes the block when the parent ends)


def _import(h):
    if not (isinstance(h, tuple) and len(h) == 4 and h[0] == "__shm__"):
        return h
    shm = shared_memory.SharedMemory(name=h[1])
    try:
        return np.array(np.ndarray(h[2], np.dtype(h[3]), buffer=shm.buf))  # (a copy: the block goes away)
    finally:
        shm.close()
        try:
            shm.unlink()
        except FileNotFoundError:
            pass


class SharedFrames:
    """The frames of a list of Observers in shared memory, one block per Observer: (images, height, width[, bands]).
    Images that still live in files are decoded once, here (a thread pool: Pillow releases the GIL)."""

    def __init__(self, observers):
        self.blocks, self.spec = [], []
        try:
            for obs in observers:
                self._share(obs)
        except Exception:
            self.close()
            raise
        self.key = self.key_of(observers)  # (after the reads: a cached read has put its array on the image)
        # the key is made of object identities: the objects stay alive as long as the key is compared with (an array freed
        # and another allocated at its address would otherwise pass for the one that was shared)
        self._alive = [(obs, list(obs.images), [getattr(img, "array", None) for img in obs.images]) for obs in observers]

    @staticmethod
    def key_of(observers):
        """What the shared copy is a copy OF: the image objects (and their pixel arrays), in order.  (Pixels changed IN
        PLACE are not seen: `Tracker.forget_frames()` makes the next parallel call share the frames again.)"""
        return tuple((id(obs), obs.sigma, tuple((id(img), id(getattr(img, "array", None))) for img in obs.images))
                     for obs in observers)

    def _share(self, obs):
        from concurrent.futures import ThreadPoolExecutor

        def pixels(img):
            return np.asarray(img.read(cache=obs.cache))

        first = pixels(obs.images[0])
        n = len(obs.images)
        try:
            if not shm_room(n * first.nbytes):
                raise OSError("the tmpfs is too small")
            shm = shared_memory.SharedMemory(create=True, size=max(1, n * first.nbytes))
        except OSError as e:
            raise MemoryError(f"Tracker.track(parallel=...): no room in shared memory (/dev/shm) for an observer's "
                              f"{n} frames of {first.nbytes} bytes: {e}") from e
        self.blocks.append(shm)
        block = np.ndarray((n,) + first.shape, first.dtype, buffer=shm.buf)
        block[0] = first
        rest = list(range(1, n))
        on_disk = [k for k in rest if getattr(obs.images[k], "array", None) is None]
        if len(on_disk) > 1:
            with ThreadPoolExecutor(max_workers=min(16, len(os.sched_getaffinity(0)), len(on_disk))) as pool:
                for k, a in zip(rest, pool.map(lambda k: pixels(obs.images[k]), rest)):
                    block[k] = self._same(a, first, k)
        else:
            for k in rest:
                block[k] = self._same(pixels(obs.images[k]), first, k)
        # the image objects without their pixels (cameras, datetimes, paths: small), for the workers to hang the views on
        bare = []
        for img in obs.images:
            c = copy.copy(img)
            c.array = None
            c.__dict__.pop("_resized", None)
            bare.append(c)
        self.spec.append(dict(shm=shm.name, shape=block.shape, dtype=first.dtype.str, images=bare, sigma=obs.sigma,
                              cache=obs.cache, cls=type(obs)))

    @staticmethod
    def _same(a, first, k):
        if a.shape != first.shape or a.dtype != first.dtype:
            raise ValueError(f"image {k}: {a.dtype} {a.shape}, the observer's first image is {first.dtype} {first.shape}")
        return a

    def nbytes(self):
        return sum(b.size for b in self.blocks)

    def close(self):
        for shm in self.blocks:
            try:
                shm.close()
                shm.unlink()
            except (FileNotFoundError, OSError):
                pass
        self.blocks = []
        self._alive = []


class _ArrayRef:
    """Stands in for a Raster's array while the Raster crosses a pipe: the pixels are in a shared-memory block."""

    def __init__(self, name, shape, dtype):
        self.name, self.shape, self.dtype = name, tuple(shape), dtype


class SharedRasters:
    """The arrays of the Rasters a parallel call sends along (a DEM, its uncertainty, the viewshed: tens of megabytes that
    every model of a block references) in shared memory, once per array; `lent()` swaps them for references while the
    jobs are pickled."""

    def __init__(self):
        self.blocks = {}  # id(array) -> (array kept alive, SharedMemory)

    def ref(self, array):
        key = id(array)
        held = self.blocks.get(key)
        if held is None or held[0] is not array:
            a = np.ascontiguousarray(array)
            if not shm_room(a.nbytes):
                return array  # (no room: the array is pickled with its Raster)
            shm = shared_memory.SharedMemory(create=True, size=max(1, a.nbytes))
            np.ndarray(a.shape, a.dtype, buffer=shm.buf)[...] = a
            held = self.blocks[key] = (array, shm, a.shape, a.dtype.str)
        return _ArrayRef(held[1].name, held[2], held[3])

    def lent(self, rasters):
        """Context manager: inside it every Raster of `rasters` (arrays beyond _BIG only) holds an _ArrayRef."""
        shared = self

        class _Lend:
            def __enter__(self):
                self.saved = []
                for r in rasters:
                    a = getattr(r, "array", None)
                    if isinstance(a, np.ndarray) and a.nbytes >= _BIG:
                        self.saved.append((r, a))
                        r.array = shared.ref(a)

            def __exit__(self, *exc):
                for r, a in self.saved:
                    r.array = a

        return _Lend()

    def close(self):
        for _, shm, _, _ in self.blocks.values():
            try:
                shm.close()
                shm.unlink()
            except (BufferError, FileNotFoundError, OSError):
                pass
        self.blocks = {}


def _resolve_rasters(state, rasters):
    """(worker) Rasters that arrived with an _ArrayRef get a read-only view of the shared block (attached once per block)."""
    held = state.setdefault("raster_shm", {})
    for r in rasters:
        ref = getattr(r, "array", None)
        if isinstance(ref, _ArrayRef):
            if ref.name not in held:
                held[ref.name] = shared_memory.SharedMemory(name=ref.name)
            a = np.ndarray(ref.shape, np.dtype(ref.dtype), buffer=held[ref.name].buf)
            a.flags.writeable = False
            r.array = a


def rasters_of(models, viewshed=None):
    """The distinct Raster objects a block of motion models (and the Tracker's viewshed) brings."""
    from .motion import ModelBlock
    from .raster import Raster

    seen = {}
    if isinstance(models, ModelBlock):
        brought = [models.dem, models.dem_sigma]
    else:
        brought = [getattr(m, attr, None) for m in models for attr in ("dem", "dem_sigma")]
    for r in [viewshed] + brought:
        if isinstance(r, Raster):
            seen[id(r)] = r
    return list(seen.values())


def attach_observers(spec):
    """(worker) Observers whose images read from the shared blocks.  Returns (observers, blocks to keep alive)."""
    observers, keep = [], []
    for s in spec:
        shm = shared_memory.SharedMemory(name=s["shm"])  # (the parent owns the block and unlinks it)
        keep.append(shm)
        block = np.ndarray(s["shape"], np.dtype(s["dtype"]), buffer=shm.buf)
        block.flags.writeable = False
        for k, img in enumerate(s["images"]):
            img.array = block[k]
        observers.append(s["cls"](s["images"], sigma=s["sigma"], cache=s["cache"]))
    return observers, keep


# ---- the worker -----------------------------------------------------------------------------------
def _frames_digest(state, _args):
    """Diagnostic: what this worker sees in the shared blocks -- per observer (shape, dtype, crc32 of every frame)."""
    return [(tuple(np.asarray(obs.images[0].read()).shape), str(np.asarray(obs.images[0].read()).dtype),
             [zlib.crc32(np.ascontiguousarray(img.read()).tobytes()) for img in obs.images]) for obs in state["observers"]]


def _track_block(state, args):
    """One block of tracks on this worker's Tracker (made on the first call for these frames, kept afterwards), then the
    collective gather of the posterior history when the parent asked for it."""
    import time

    from .tracker import Tracker

    t0 = time.perf_counter()
    _resolve_rasters(state, rasters_of(args["models"], args["tracker"].get("viewshed")))
    tracker = state.get("tracker")
    made = tracker is None
    if made:
        tracker = state["tracker"] = Tracker(state["observers"], device=state["device"], **args["tracker"])
        state["attached"] = None
    else:
        for k, v in args["tracker"].items():
            if k == "viewshed":
                tracker.viewshed = v
            elif getattr(tracker, k) != v:
                tracker.close()
                tracker = state["tracker"] = Tracker(state["observers"], device=state["device"], **args["tracker"])
                state["attached"], made = None, True
                break
    if args["np_seed"] is not None:
        np.random.seed(int(args["np_seed"]))
    t = tracker.track(args["models"], _catch_errors=args["catch"], **args["kw"])
    t_track = time.perf_counter() - t0
    out = {k: getattr(t, k) for k in ("datetimes", "time_unit", "covariances", "particles", "weights", "images")}
    out["errors"], out["warnings"] = list(t.errors), list(t.warnings)
    out["reduced"] = getattr(t, "reduced", None)
    # (rows of a failed track are NaN from the frame where it failed: the parent repeats that on the gathered history)
    out["nan_from"] = [(p, int(np.argmax(np.isnan(t.means[p][:, 0])))) for p, e in enumerate(t.errors)
                       if e is not None and np.isnan(t.means[p][:, 0]).any()]
    out["last_particles"], out["last_weights"] = (tracker.particles, tracker.weights) if args["want_last"] else (None, None)
    out["context_made"] = made or state.get("ctx_id") != id(tracker._ctx)
    out["track_seconds"] = t_track
    gathered = None
    transport = "host"
    if args["gather"]:
        group, ctx = state["group"], tracker._ctx
        call = args["call"]
        # a worker whose context is new (first call, another shape, grown workspaces) needs the communicator made again,
        # and making it is collective: every worker learns whether ANY context changed
        changed = state.get("ctx_id") != id(ctx) or state.get("attached") is None
        group.store.put(f"ctxchg.{call}.{group.rank}", b"1" if changed else b"0")
        if any(group.store.get(f"ctxchg.{call}.{r}") == b"1" for r in range(group.world)):
            if state.get("attached") == "rccl" and not changed:
                ctx.comm_destroy()
            # (workers that share a GPU cannot make a communicator -- ncclCommInitRank refuses a device twice --: the parent
            # says so and the attempt, a second or two of RCCL bootstrap at the first call, is not made)
            state["attached"] = group.attach(ctx, "host" if args.get("host_only") else None)
            state["why_host"] = args.get("host_only") or getattr(group, "why_host", "")
        else:
            group._ctx = ctx
        state["ctx_id"] = id(ctx)
        transport = state["attached"]
        if transport == "rccl":
            ntimes = len(t.datetimes)
            gathered = group.gather_moments(ctx, 0, ntimes, args["sizes"])  # (moments (T, sum P, 12), status) on rank 0
    else:
        state["ctx_id"] = id(tracker._ctx)
    out["transport"] = transport
    out["why_host"] = state.get("why_host", "")
    # The history goes into the parent's result block (one shared-memory block for all workers, kept between calls: its
    # pages are mapped once), rows [lo, hi) of (tracks, times, 12) = means | sigmas: worker 0 writes everybody's after the
    # RCCL gather, every worker its own otherwise.  Blocks of another shape (ragged runs) travel as arrays of the reply.
    block = _result_block(state, args.get("result"))
    lo, hi = args["rows"]
    if gathered is not None and block is not None:
        block[:, :, :] = np.transpose(gathered[0], (1, 0, 2))
        out["in_block"] = "all"
    elif gathered is not None:
        out["gathered"] = gathered[0]
    elif transport != "rccl" or not args["gather"]:
        m, sg = t.means, t.sigmas
        if (block is not None and isinstance(m, np.ndarray) and m.shape == (hi - lo, block.shape[1], 6)
                and (sg is None or (isinstance(sg, np.ndarray) and sg.shape == m.shape))):
            block[lo:hi, :, 0:6] = m
            if sg is not None:
                block[lo:hi, :, 6:12] = sg
            out["in_block"] = "rows"
        else:
            out["means"], out["sigmas"] = m, sg
    out["seconds"] = time.perf_counter() - t0
    return out


def _result_block(state, spec):
    """(worker) the parent's result block as an array (tracks, times, 12), attached once per block."""
    if spec is None:
        return None
    name, shape = spec
    held = state.get("result_shm")
    if held is None or held[0] != name:
        if held is not None:
            held[1].close()
        held = state["result_shm"] = (name, shared_memory.SharedMemory(name=name))
    return np.ndarray(shape, np.float64, buffer=held[1].buf)


def _rasters_digest(state, rasters):
    """Diagnostic: crc32 of the arrays of the Rasters that came with the message (through shared memory when large)."""
    _resolve_rasters(state, rasters)
    return [(tuple(r.array.shape), zlib.crc32(np.ascontiguousarray(r.array).tobytes())) for r in rasters]


_HANDLERS = {"track": _track_block, "digest": _frames_digest, "rasters": _rasters_digest}


def _worker_main(conn, rank, world, device, token):
    """The loop of one worker process."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), GLH_RENDEZVOUS_TOKEN=token, GLH_POOL_CHILD="1",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    store = sharding.FileStore(sharding.FileStore.default_path(token), rank, world) if world > 1 else None
    state = {"group": sharding.Group(rank, world, rank, store), "device": device, "observers": None, "blocks": []}
    orphan = False
    try:
        while True:
            try:
                msg = conn.recv()
            except (EOFError, OSError):
                orphan = True  # (the parent is gone without a "stop": nobody else removes the rendezvous directory)
                break
            kind, args = msg
            if kind == "stop":
                break
            try:
                if kind == "frames":
                    tracker = state.pop("tracker", None)
                    if tracker is not None:
                        tracker.close()
                    for shm in state["blocks"]:
                        shm.close()
                    state["observers"], state["blocks"] = attach_observers(args)
                    reply = True
                else:
                    reply = _HANDLERS[kind](state, args)
                    if isinstance(reply, dict):
                        reply = {k: [_export(x) for x in v] if isinstance(v, list) and v and isinstance(v[0], np.ndarray)
                                 else _export(v) for k, v in reply.items()}
                conn.send(("ok", reply))
            except BaseException as e:  # noqa: BLE001  (reported to the parent, which decides)
                conn.send(("error", (repr(e), traceback.format_exc())))
    finally:
        tracker = state.get("tracker")
        if tracker is not None:
            try:
                tracker.close()
            except Exception:  # noqa: BLE001
                pass
        for shm in state["blocks"]:
            shm.close()
        for shm in state.get("raster_shm", {}).values():
            try:
                shm.close()
            except BufferError:
                pass
        if orphan:  # (every rank: one that starts late makes the directory again)
            import shutil

            shutil.rmtree(sharding.FileStore.default_path(token), ignore_errors=True)


# ---- the pool (parent) ----------------------------------------------------------------------------
_POOLS = weakref.WeakSet()


def _close_all():
    for pool in list(_POOLS):
        pool.close()


atexit.register(_close_all)


class WorkerPool:
    """`n` persistent worker processes.  `call(kind, [args per worker])` sends one message to every worker and returns
    their replies in order; a worker that dies or raises takes the call down with a RuntimeError (and the pool with it)."""

    def __init__(self, n, devices, import_main=False):
        if _in_child():
            raise RuntimeError("a glimpse_amd worker process tried to start workers of its own: the main script runs again "
                               "in every worker -- put it behind `if __name__ == '__main__':`")
        ctx = mp.get_context("spawn")  # fresh interpreters: the parent may have initialised the GPU runtime
        self.import_main = bool(import_main)
        # ONE resource tracker for the parent and the workers (started here, before they are: they inherit it): a block made
        # on one side and unlinked on the other is then registered and unregistered in the same place
        from multiprocessing import resource_tracker

        resource_tracker.ensure_running()
        self.n = n
        self.token = f"pool_{os.getpid()}_{id(self):x}"
        self.calls = 0
        self.frames = None  # SharedFrames the workers hold
        self.results = None  # shared-memory block the workers write the posterior history into
        self.rasters = SharedRasters()  # arrays of the Rasters the calls send along
        self.procs, self.conns = [], []
        for rank in range(n):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(child, rank, n, devices[rank], self.token), daemon=True)
            with (_with_main() if self.import_main else without_main()):
                p.start()
            child.close()
            self.procs.append(p)
            self.conns.append(parent)
        _POOLS.add(self)

    def alive(self):
        return bool(self.procs) and all(p.is_alive() for p in self.procs)

    def call(self, kind, args, timeout=3600.0):
        import time

        self.calls += 1
        failure = None
        for r, (conn, a) in enumerate(zip(self.conns, args)):
            try:
                conn.send((kind, a))
            except (OSError, ValueError):  # (the worker is gone)
                failure = f"worker {r} closed its pipe (exit code {self.procs[r].exitcode})"
                break
        replies = [None] * self.n
        pending = set(range(self.n))
        deadline = time.monotonic() + timeout
        while pending and failure is None:
            ready = mpc.wait([self.conns[r] for r in pending], timeout=0.05)
            for r in list(pending):
                if self.conns[r] in ready:
                    try:
                        status, value = self.conns[r].recv()
                    except (EOFError, OSError):
                        failure = f"worker {r} closed its pipe"
                        break
                    if status == "error":
                        failure = f"worker {r} raised {value[0]}\n{value[1]}"
                        break
                    replies[r] = value
                    pending.discard(r)
                elif not ready and not self.procs[r].is_alive():
                    failure = f"worker {r} died (exit code {self.procs[r].exitcode})"
                    break
            if time.monotonic() > deadline:
                failure = f"no reply from workers {sorted(pending)} after {timeout:.0f} s"
        if failure is not None:
            # the others may be waiting for the lost worker inside a collective: tell them through the store, then stop all
            path = sharding.FileStore.default_path(self.token)
            if self.n > 1:
                sharding.FileStore(path, -1, self.n).abort(failure.splitlines()[0])
            self.close(kill=True)
            raise RuntimeError("Tracker.track(parallel=...): " + failure)
        return replies

    def result_block(self, shape):
        """(name, shape) of a float64 shared-memory block of at least this shape, kept between calls (grown on demand)."""
        nbytes = int(np.prod(shape)) * 8
        if (self.results is None or self.results.size < nbytes) and not shm_room(nbytes):
            return None  # (no room: the workers send their histories with their replies)
        if self.results is None or self.results.size < nbytes:
            if self.results is not None:
                try:
                    self.results.close()
                    self.results.unlink()
                except (BufferError, FileNotFoundError, OSError):
                    pass
            self.results = shared_memory.SharedMemory(create=True, size=max(nbytes, 1))
        return self.results.name, tuple(int(v) for v in shape)

    def result_array(self, shape):
        """A COPY of the block's contents as (tracks, times, 12)."""
        return np.array(self.result_view(shape))

    def result_view(self, shape):
        """The block's contents as (tracks, times, 12) WITHOUT a copy: to be read and dropped before the next call (the
        workers write into the same pages) and before the pool is closed."""
        return np.ndarray(shape, np.float64, buffer=self.results.buf)

    def share(self, observers):
        """The workers see these observers' frames (shared once; again only when the image objects changed)."""
        key = SharedFrames.key_of(observers)
        if self.frames is not None and self.frames.key == key:
            return False
        frames = SharedFrames(observers)
        self.call("frames", [frames.spec] * self.n)
        if self.frames is not None:
            self.frames.close()
        self.frames = frames
        return True

    def close(self, kill=False):
        for conn in self.conns:
            try:
                if not kill:
                    conn.send(("stop", None))
            except (OSError, ValueError):
                pass
        for p in self.procs:
            p.join(0.1 if kill else 10.0)
            if p.is_alive():
                p.terminate()
                p.join(5.0)
        for conn in self.conns:
            conn.close()
        self.procs, self.conns = [], []
        if self.frames is not None:
            self.frames.close()
            self.frames = None
        if self.results is not None:
            try:
                self.results.close()
                self.results.unlink()
            except (BufferError, FileNotFoundError, OSError):
                pass
            self.results = None
        self.rasters.close()
        import shutil

        shutil.rmtree(sharding.FileStore.default_path(self.token), ignore_errors=True)
