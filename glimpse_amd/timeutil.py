"""Exact integer-microsecond datetime helpers shared by Tracker.match_datetimes and Observer.subset
(the reference works on float timestamps and full distance matrices: helpers.py:161-204, 1831-1854, 1883-1951)."""
import datetime

import numpy as np

_US = datetime.timedelta(microseconds=1)


def _offsets_us(dts, ref):
    """Datetimes as exact integer microseconds from `ref` (timedelta arithmetic, no float timestamps)."""
    return np.fromiter(((d - ref) // _US for d in dts), dtype=np.int64, count=len(dts))


def nearest_in_sorted(times_us, queries_us):
    """For every query the index of the nearest entry of the ascending `times_us` and its distance (microseconds).
    One binary search per query (np.searchsorted) instead of the reference's len(queries) x len(times) distance
    matrix (helpers.py:1831-1854); a tie goes to the earlier entry, like np.argmin over that matrix."""
    times_us = np.asarray(times_us, dtype=np.int64)
    queries_us = np.asarray(queries_us, dtype=np.int64)
    right = np.clip(np.searchsorted(times_us, queries_us, side="left"), 0, len(times_us) - 1)
    left = np.maximum(right - 1, 0)
    d_left, d_right = np.abs(queries_us - times_us[left]), np.abs(times_us[right] - queries_us)
    take_left = d_left <= d_right
    return np.where(take_left, left, right), np.where(take_left, d_left, d_right)


def select_datetimes(datetimes, start=None, end=None, snap=None, maxdt=None,
                     origin=datetime.datetime(1970, 1, 1, 0, 0, 0)):
    """helpers.py:1883-1951: boolean mask of the ascending `datetimes` inside [start, end] (inclusive) and, with
    `snap`, nearest to a multiple of `snap` from `origin` by no more than `maxdt` (default snap / 2)."""
    datetimes = list(datetimes)
    t = _offsets_us(datetimes, origin)
    keep = np.ones(len(t), dtype=bool)
    step = None if not snap else snap // _US
    if start:
        lo = (start - origin) // _US
        keep &= t >= lo
    else:
        lo = int(t[0]) - (step or 0)
    if end:
        hi = (end - origin) // _US
        keep &= t <= hi
    else:
        hi = int(t[-1]) + (step or 0)
    if lo > hi:
        raise ValueError("Start datetime is after end datetime")
    if step:
        first = -((-lo) // step) * step  # first multiple of snap at or after the lower bound
        targets = np.arange(first, hi + 1, step, dtype=np.int64)
        reach = (snap * 0.5 if maxdt is None else maxdt) // _US
        idx, dist = nearest_in_sorted(t, targets)
        on_grid = np.zeros(len(t), dtype=bool)
        on_grid[idx[dist <= reach]] = True
        keep &= on_grid
    return keep
