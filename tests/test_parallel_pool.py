"""The worker pool of `Tracker.track(parallel=N)` (glimpse_amd/parallel.py) without a GPU: persistent processes, the
observers' frames shared once through shared memory -- never pickled --, large reply arrays through shared memory, a dying
worker takes the call down instead of hanging it.  (The tracking itself on the workers: tests/test_gpu_api.py.)"""
import datetime
import multiprocessing.connection as mpc
import os
import zlib

import numpy as np
import pytest

import glimpse_amd
from glimpse_amd import parallel

T0 = datetime.datetime(2021, 6, 1)
DAY = datetime.timedelta(days=1)


def _observers(rng, n=3, size=(320, 256)):
    cam = glimpse_amd.Camera(imgsz=size, f=(400, 400), xyz=(0, 0, 50), viewdir=(0, -90, 0))
    gray = [glimpse_amd.Image("mem", cam=cam, datetime=T0 + k * DAY, array=rng.integers(0, 255, size[::-1], dtype=np.uint8))
            for k in range(n)]
    rgb = [glimpse_amd.Image("mem", cam=cam, datetime=T0 + k * DAY,
                             array=rng.integers(0, 255, size[::-1] + (3,), dtype=np.uint8)) for k in range(n)]
    return [glimpse_amd.Observer(gray, sigma=0.3), glimpse_amd.Observer(rgb, sigma=0.5)]


@pytest.fixture
def pickled(monkeypatch):
    """Sizes of everything this process sends through multiprocessing pipes."""
    sizes = []
    original = mpc._ForkingPickler.dumps.__func__

    def dumps(cls, obj, protocol=None):
        data = original(cls, obj, protocol)
        sizes.append(len(data))
        return data

    monkeypatch.setattr(mpc._ForkingPickler, "dumps", classmethod(dumps))
    return sizes


def test_frames_reach_the_workers_through_shared_memory_only(pickled):
    rng = np.random.default_rng(0)
    observers = _observers(rng)
    frame_bytes = sum(img.array.nbytes for obs in observers for img in obs.images)
    pool = parallel.WorkerPool(2, [0, 0])
    try:
        assert pool.share(observers) is True
        assert pool.frames.nbytes() >= frame_bytes
        want = [[zlib.crc32(img.array.tobytes()) for img in obs.images] for obs in observers]
        for digest in pool.call("digest", [None, None]):
            assert [d[2] for d in digest] == want
            assert digest[0][0] == (256, 320) and digest[1][0] == (256, 320, 3) and digest[0][1] == "uint8"
        # what crossed the pipes: the image objects without their pixels, twice -- a small fraction of one frame
        assert sum(pickled) < 0.05 * frame_bytes, (sum(pickled), frame_bytes)
        assert max(pickled) < observers[0].images[0].array.nbytes
        # the same observers again: nothing is sent; the workers are the same processes
        pids = [p.pid for p in pool.procs]
        sent = len(pickled)
        assert pool.share(observers) is False and len(pickled) == sent
        assert [p.pid for p in pool.procs] == pids and pool.alive()
        # another pixel array on one image: shared again, the workers see the new pixels
        observers[0].images[1].array = rng.integers(0, 255, (256, 320), dtype=np.uint8)
        assert pool.share(observers) is True
        digest = pool.call("digest", [None, None])[1]
        assert digest[0][2][1] == zlib.crc32(observers[0].images[1].array.tobytes())
        # pixels changed IN PLACE are not noticed by themselves; after Tracker.forget_frames() (the key is dropped) they are
        observers[1].images[0].array[...] = 7
        assert pool.share(observers) is False
        pool.frames.key = None
        assert pool.share(observers) is True
        assert pool.call("digest", [None, None])[0][1][2][0] == zlib.crc32(observers[1].images[0].array.tobytes())
        # the caller's images keep their own arrays (the workers got copies of the objects without pixels)
        assert all(img.array is not None and img.array.flags.writeable for obs in observers for img in obs.images)
    finally:
        blocks = [b.name for b in pool.frames.blocks] if pool.frames else []
        pool.close()
    assert not pool.alive()
    for name in blocks:  # the blocks are gone with the pool
        assert not os.path.exists("/dev/shm/" + name.lstrip("/"))


def test_large_arrays_of_a_reply_travel_through_shared_memory():
    a = np.arange(100000, dtype=np.float64).reshape(100, 1000)
    handle = parallel._export(a)
    assert isinstance(handle, tuple) and handle[0] == "__shm__"
    back = parallel._import(handle)
    np.testing.assert_array_equal(back, a)
    assert not os.path.exists("/dev/shm/" + handle[1].lstrip("/"))  # unlinked by the receiver
    small = np.arange(10.0)
    assert parallel._export(small) is small and parallel._import(small) is small
    assert parallel._import(None) is None


def test_a_failing_worker_takes_the_call_down(monkeypatch):
    pool = parallel.WorkerPool(2, [0, 0])
    try:
        with pytest.raises(RuntimeError, match="worker [01] raised KeyError"):
            pool.call("no such handler", [None, None])
        assert not pool.alive()  # the pool is closed: the next parallel call starts a new one
    finally:
        pool.close()
    pool = parallel.WorkerPool(2, [0, 0])
    try:
        pool.procs[1].terminate()
        pool.procs[1].join(10)
        with pytest.raises(RuntimeError, match="worker 1 (died|closed its pipe)"):
            pool.call("digest", [None, None], timeout=30)
    finally:
        pool.close()


def test_decoder_processes_fill_the_shared_ring(tmp_path):
    """glimpse_amd/ingest.py: image files decoded by worker processes into the slots of one shared-memory ring -- the
    pixels of Image.read (here with a camera at half the file's size: the nearest-neighbour resampling of image.py:188-193
    happens in the decoder), any order of completion, a slot is reused once released, a missing file raises where its
    result is asked for."""
    PIL = pytest.importorskip("PIL.Image")
    from glimpse_amd import ingest

    rng = np.random.default_rng(4)
    cam = glimpse_amd.Camera(imgsz=(160, 120), f=(200, 200))
    half = glimpse_amd.Camera(imgsz=(80, 60), f=(100, 100))
    images, want = [], []
    for k in range(9):
        a = rng.integers(0, 255, (120, 160) if k % 2 else (120, 160, 3), dtype=np.uint8)
        path = tmp_path / f"f{k}.png"
        PIL.fromarray(a).save(path)
        img = glimpse_amd.Image(str(path), cam=half if k == 4 else cam, datetime=T0 + k * DAY)
        images.append(img)
        want.append(img.read(cache=False))
    pool = ingest.DecodePool(3, 120 * 160 * 3, slots=4)
    try:
        got, queued = {}, 0
        while len(got) < len(images):
            while queued < len(images) and pool.submit(queued, images[queued]):
                queued += 1
            job, view, slot, seconds = pool.result()
            got[job] = np.array(view)
            assert seconds >= 0.0
            pool.release(slot)
        for k, a in enumerate(want):
            np.testing.assert_array_equal(got[k], a)
        assert got[4].shape == (60, 80, 3)
        assert sorted(pool.free) == [0, 1, 2, 3] and all(img.array is None for img in images)
        pool.submit(99, glimpse_amd.Image(str(tmp_path / "missing.png"), cam=cam, datetime=T0))
        with pytest.raises(RuntimeError, match="decoding failed"):
            pool.result()
        assert pool.alive() and sorted(pool.free) == [0, 1, 2, 3]
    finally:
        name = pool.ring.name
        pool.close()
    assert not os.path.exists("/dev/shm/" + name.lstrip("/"))


def test_raster_arrays_travel_through_shared_memory_once(pickled):
    """The Rasters a parallel call sends along (a DEM and its uncertainty on every motion model, the viewshed): their arrays
    go into shared memory once, the pipes carry references; the caller's Rasters have their arrays back after the call."""
    rng = np.random.default_rng(2)
    dem = glimpse_amd.Raster(rng.standard_normal((300, 400)), x=(0, 800), y=(600, 0))
    small = glimpse_amd.Raster(np.ones((4, 5)), x=(0, 10), y=(8, 0))  # (below the threshold: pickled as it is)
    models = [glimpse_amd.CartesianMotion(xy=(10.0 + k, 20.0), time_unit=DAY, dem=dem, dem_sigma=small, n=100,
                                          xy_sigma=(0.2, 0.2), vxyz=(0.1, 0, 0), vxyz_sigma=(0.1, 0.1, 0)) for k in range(50)]
    rasters = parallel.rasters_of(models, viewshed=None)
    assert len(rasters) == 2 and any(r is dem for r in rasters)
    pool = parallel.WorkerPool(2, [0, 0])
    try:
        want = [(r.array.shape, zlib.crc32(np.ascontiguousarray(r.array).tobytes())) for r in rasters]
        for _ in range(2):
            sent = sum(pickled)
            with pool.rasters.lent(rasters):
                assert isinstance(dem.array, parallel._ArrayRef) and isinstance(small.array, np.ndarray)
                got = pool.call("rasters", [rasters, rasters])
            assert got[0] == want and got[1] == want
            assert isinstance(dem.array, np.ndarray) and dem.array.shape == (300, 400)
            assert sum(pickled) - sent < 0.02 * dem.array.nbytes
        assert len(pool.rasters.blocks) == 1  # (the second call found the block)
        names = [held[1].name for held in pool.rasters.blocks.values()]
    finally:
        pool.close()
    for name in names:
        assert not os.path.exists("/dev/shm/" + name.lstrip("/"))


def test_pools_do_not_run_an_unguarded_main_script_again(tmp_path):
    """A script WITHOUT `if __name__ == "__main__":` that starts decoders and workers at its top level: multiprocessing's
    spawn would run it again in every child (each starting pools of its own); glimpse_amd's children skip the import of
    the parent's main module (parallel.without_main), so the script runs exactly once."""
    import subprocess
    import sys

    script = tmp_path / "unguarded.py"
    marker = tmp_path / "ran.txt"
    script.write_text(f"""
import sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
with open({str(marker)!r}, "a") as f:
    f.write("run\\n")
from glimpse_amd import ingest, parallel
pool = ingest.DecodePool(2, 1 << 16, slots=4)
assert pool.alive()
pool.close()
workers = parallel.WorkerPool(2, [0, 0])
assert workers.call("rasters", [[], []]) == [[], []]
workers.close()
print("done")
""")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().endswith("done")
    assert marker.read_text() == "run\n"


def test_a_block_of_models_travels_as_one_parameter_table():
    """motion.ModelBlock: what `Tracker.track(parallel=N)` sends a worker instead of thousands of model objects -- the
    parameter table the batched Tracker reads, the rasters, n and the time unit; views of single models for the few
    places that look one up."""
    import pickle

    from glimpse_amd import tracker as tracker_module
    from glimpse_amd.motion import ModelBlock, params_table

    rng = np.random.default_rng(5)
    dem = glimpse_amd.Raster(rng.standard_normal((30, 40)), x=(0, 80), y=(60, 0))
    models = []
    for k in range(40):
        cls = (glimpse_amd.CartesianMotion, glimpse_amd.CylindricalMotion, glimpse_amd.TangentCartesianMotion,
               glimpse_amd.TangentCylindricalMotion)[k % 4]
        kw = dict(xy=(10.0 + k, 20.0), time_unit=DAY, dem=dem if k % 3 == 0 else 5.0, dem_sigma=0.25 * (k % 5), n=64,
                  xy_sigma=(0.2, 0.3))
        kw.update([dict(vxyz=(0.1 * k, 0, 0), vxyz_sigma=(0.1, 0.1, 0)), dict(vrthz=(0.1 * k, 0.2, 0), vrthz_sigma=(0.1, 0.1, 0)),
                   dict(vxy=(0.1 * k, 0), vxy_sigma=(0.1, 0.1), slope_sigma=0.1),
                   dict(vrth=(0.1 * k, 0.2), vrth_sigma=(0.1, 0.1), slope_sigma=0.2)][k % 4])
        models.append(cls(**kw))
    assert tracker_module._batches(models) == [0]
    block = ModelBlock.from_models(models)
    assert len(block) == 40 and tracker_module._batches(block) == [0]
    np.testing.assert_array_equal(params_table(block), params_table(models))
    assert block.raster("dem") is dem and block.raster("dem_sigma") is None
    assert parallel.rasters_of(block) == [dem] and parallel.rasters_of(models) == [dem]
    back = pickle.loads(pickle.dumps(block))
    np.testing.assert_array_equal(back.table, block.table)
    assert back.n == 64 and back.time_unit == DAY and isinstance(back.dem, glimpse_amd.Raster)
    for k in (0, 1, 7, 39):
        view, m = block[k], models[k]
        assert view.n == m.n and view.time_unit == m.time_unit and view.TANGENT == m.TANGENT and view.KIND == m.KIND
        assert (view.dem is dem) == (m.dem is dem) and (view.dem is dem or view.dem == m.dem)
        assert view.dem_sigma == m.dem_sigma and tuple(view.xy) == tuple(m.xy)
        assert tracker_module._on_device(view)
    part = block[10:25]
    np.testing.assert_array_equal(part.table, params_table(models[10:25]))
    assert [v.KIND for v in part] == [m.KIND for m in models[10:25]]
    constant = ModelBlock.from_models([m for m in models if m.dem is not dem])
    assert constant.raster("dem") is None and parallel.rasters_of(constant) == []


def test_no_block_is_made_that_dev_shm_cannot_hold(monkeypatch):
    """A shared-memory block is a sparse file: writing past what /dev/shm holds is a SIGBUS, not an exception.  Every block
    is therefore checked against the free space first; without room the frames refuse (MemoryError), the decoders refuse
    (OSError: the Tracker then decodes in threads), reply arrays and raster arrays go through the pipes."""
    from glimpse_amd import ingest

    assert parallel.shm_room(1 << 10) is True
    assert parallel.shm_room(1 << 60) is False
    monkeypatch.setattr(parallel, "shm_room", lambda nbytes: False)
    big = np.zeros((300, 400))
    assert parallel._export(big) is big
    with pytest.raises(MemoryError, match="no room in shared memory"):
        parallel.SharedFrames(_observers(np.random.default_rng(1), n=2))
    with pytest.raises(OSError, match="no room"):
        ingest.DecodePool(2, 1 << 16)
    shared = parallel.SharedRasters()
    assert shared.ref(big) is big and not shared.blocks
    pool = parallel.WorkerPool.__new__(parallel.WorkerPool)
    pool.results = None
    assert pool.result_block((10, 10, 12)) is None


def test_decoders_and_workers_end_when_their_parent_is_killed(tmp_path):
    """The parent of a pool dies without a word (SIGKILL): the workers see their pipe close, the decoders -- whose queue
    never reports a dead writer -- notice that they have been adopted, and all of them end."""
    import subprocess
    import sys
    import time

    script = tmp_path / "killed.py"
    script.write_text(f"""
import os, signal, sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from glimpse_amd import ingest, parallel
if __name__ == "__main__":
    pool = ingest.DecodePool(2, 1 << 16, slots=4)
    workers = parallel.WorkerPool(2, [0, 0])
    print(" ".join(str(v) for v in [p.pid for p in pool.procs + workers.procs] + [os.getpid()]), flush=True)
    os.kill(os.getpid(), signal.SIGKILL)
""")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    pids = [int(v) for v in out.stdout.split()]
    assert len(pids) == 5 and out.returncode == -9
    pids, parent_pid = pids[:4], pids[4]

    def alive(pid):
        try:
            with open(f"/proc/{pid}/stat") as f:
                return f.read().split(") ")[-1][0] != "Z"
        except OSError:
            return False

    deadline = time.monotonic() + 30.0
    while any(alive(p) for p in pids) and time.monotonic() < deadline:
        time.sleep(0.25)
    assert not any(alive(p) for p in pids), [p for p in pids if alive(p)]
    assert not [name for name in os.listdir("/dev/shm") if name.startswith("glh_rdzv_pool_%d_" % parent_pid)]
