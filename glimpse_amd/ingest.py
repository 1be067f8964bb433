"""Decoding image files in worker PROCESSES (the frames of a sequence that still live in files: image.py:137-214).

Pillow releases the GIL while it decodes, but what surrounds the decode does not -- file reads in Python-level chunks,
the copy into an ndarray -- and sixteen decoding THREADS reached 4x one thread for JPEG and 1.2x for uncompressed TIFF
(profiles/r05: the run from files spent 73 % of its time waiting for pixels).  A `DecodePool` is `n` persistent processes
(fresh interpreters; they never touch the GPU) that decode into the slots of ONE shared-memory ring; the parent uploads
a slot to the device (pinned staging + copy stream) and hands it back.  What crosses the pipes: the image object without
pixels (path, camera, datetime) one way, a slot number the other.
"""
import atexit
import copy
import multiprocessing as mp
import os
import queue
import time
import traceback
import weakref
from multiprocessing import shared_memory

import numpy as np


def _decode_main(tasks, done, ring_name, slot_bytes, parent):
    """One decoder process: (job, image, slot) -> the image's pixels at its camera size in ring[slot].  `parent`: the pid
    of the process that owns the pool."""
    os.environ["GLH_POOL_CHILD"] = "1"
    ring = shared_memory.SharedMemory(name=ring_name)
    try:
        while True:
            try:
                item = tasks.get(timeout=5.0)
            except queue.Empty:
                # (a queue never reports its writer's death -- every process that holds it is a writer: a decoder whose
                # parent was killed would wait here for ever)
                if os.getppid() != parent:
                    break
                continue
            if item is None:
                break
            job, img, slot = item
            t0 = time.perf_counter()
            try:
                a = np.ascontiguousarray(img.read(cache=False))
                if a.nbytes > slot_bytes:
                    raise ValueError(f"{img.path}: {a.nbytes} bytes, the ring's slots hold {slot_bytes}")
                np.ndarray(a.shape, a.dtype, buffer=ring.buf, offset=slot * slot_bytes)[...] = a
                done.put((job, slot, a.shape, a.dtype.str, time.perf_counter() - t0, None))
            except BaseException as e:  # noqa: BLE001  (the parent raises it where the frame is needed)
                done.put((job, slot, None, None, time.perf_counter() - t0, (repr(e), traceback.format_exc())))
    finally:
        ring.close()


_POOLS = weakref.WeakSet()


def _close_all():
    for pool in list(_POOLS):
        pool.close()


atexit.register(_close_all)


class DecodePool:
    RING_BYTES = 512 << 20

    def __init__(self, n, slot_bytes, slots=None):
        from multiprocessing import resource_tracker

        from .parallel import _in_child, without_main

        if _in_child():
            raise RuntimeError("a glimpse_amd worker process tried to start image decoders")
        resource_tracker.ensure_running()  # (one tracker for the parent and the decoders: see parallel.WorkerPool)
        ctx = mp.get_context("spawn")
        self.n, self.slot_bytes = n, int(slot_bytes)
        # two slots per decoder (one being filled, one on its way to the device) while the ring stays within RING_BYTES
        self.slots = slots or max(4, min(2 * n + 2, self.RING_BYTES // max(1, self.slot_bytes)))
        from .parallel import shm_room

        while self.slots > 4 and not shm_room(self.slots * self.slot_bytes):
            self.slots -= 1
        if not shm_room(self.slots * self.slot_bytes):  # (the caller decodes in threads instead)
            raise OSError(f"/dev/shm has no room for a ring of {self.slots} frames of {self.slot_bytes} bytes")
        self.ring = shared_memory.SharedMemory(create=True, size=self.slots * self.slot_bytes)
        self.tasks, self.done = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_decode_main,
                                  args=(self.tasks, self.done, self.ring.name, self.slot_bytes, os.getpid()),
                                  daemon=True) for _ in range(n)]
        with without_main():  # (the decoders never run the caller's main script: parallel.without_main)
            for p in self.procs:
                p.start()
        self.free = list(range(self.slots))
        # The ring page-locked for the device (hipHostRegister): the parent then uploads a slot without the copy into a
        # staging buffer (glh_observer_upload_frame_pinned).  Without a device (or if the driver refuses) the slots are
        # uploaded through the library's own staging ring.
        self.pinned, self._addr = False, None
        try:
            from . import _lib

            probe = np.frombuffer(self.ring.buf, dtype=np.uint8)
            addr = probe.ctypes.data
            del probe
            _lib.host_register(addr, self.slots * self.slot_bytes)
            self.pinned, self._addr = True, addr
        except Exception:  # noqa: BLE001
            pass
        _POOLS.add(self)

    def alive(self):
        return bool(self.procs) and all(p.is_alive() for p in self.procs)

    def submit(self, job, img):
        """Queue one image (an object with .read(); sent without its pixels).  False when no slot is free."""
        if not self.free:
            return False
        bare = copy.copy(img)
        bare.array = None
        bare.__dict__.pop("_resized", None)
        self.tasks.put((job, bare, self.free.pop()))
        return True

    def result(self, block=True, timeout=600.0):
        """(job, view of the slot's pixels, slot, decoder seconds) of some finished image, or None (non-blocking, nothing
        finished).  The view is valid until `release(slot)`."""
        deadline = time.monotonic() + timeout
        while True:
            try:
                job, slot, shape, dtype, seconds, err = self.done.get(block, 0.05) if block else self.done.get_nowait()
                break
            except queue.Empty:
                if not block:
                    return None
                if not self.alive():
                    raise RuntimeError("an image decoder process died") from None
                if time.monotonic() > deadline:
                    raise TimeoutError("no decoded image after %.0f s" % timeout) from None
        if err is not None:
            self.free.append(slot)
            raise RuntimeError(f"decoding failed: {err[0]}\n{err[1]}")
        view = np.ndarray(shape, np.dtype(dtype), buffer=self.ring.buf, offset=slot * self.slot_bytes)
        return job, view, slot, seconds

    def release(self, slot):
        self.free.append(slot)

    def drain(self, timeout=60.0):
        """Forget everything in flight (a run that ended early): wait for the queued images, free their slots.  A slot that
        does not come back within `timeout` (its result was taken and never released) ends the pool: the next run makes
        a new one."""
        deadline = time.monotonic() + timeout
        while len(self.free) < self.slots and self.alive():
            try:
                item = self.done.get(True, 0.05)
            except queue.Empty:
                if time.monotonic() > deadline:
                    self.close()
                    return
                continue
            self.free.append(item[1])

    def close(self):
        for _ in self.procs:
            try:
                self.tasks.put(None)
            except (OSError, ValueError):
                pass
        for p in self.procs:
            p.join(5.0)
            if p.is_alive():
                p.terminate()
                p.join(2.0)
        self.procs = []
        for q in (self.tasks, self.done):
            try:
                q.close()
                q.join_thread()
            except (OSError, ValueError, AttributeError):
                pass
        if self.ring is not None:
            if self.pinned:
                try:
                    from . import _lib

                    _lib.host_unregister(self._addr)
                except Exception:  # noqa: BLE001
                    pass
                self.pinned = False
            try:
                self.ring.close()
            except BufferError:  # (a view of a slot is still alive somewhere: collect, then let the mapping go with the process)
                import gc

                gc.collect()
                try:
                    self.ring.close()
                except BufferError:
                    pass
            except OSError:
                pass
            try:
                self.ring.unlink()
            except (FileNotFoundError, OSError):
                pass
            self.ring = None
