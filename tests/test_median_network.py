"""Prove the 25-input median selection network (glimpse_amd/csrc/glh_median.h) correct.

0/1 principle: a comparator network selects the median of every input iff it does so
for every binary input.  All 2^25 binary inputs are run bit-parallel (one bit per input
vector), so the proof takes well under a second.
"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def network_pairs():
    src = open(os.path.join(ROOT, "glimpse_amd", "csrc", "glh_median.h")).read()
    body = src[src.index("#define GLH_MED25_NETWORK(X)"):src.index("// clang-format on")]
    return [(int(a), int(b)) for a, b in re.findall(r"X\((\d+),(\d+)\)", body)]


def test_median25_network_zero_one_principle():
    pairs = network_pairs()
    assert len(pairs) == 99
    nbits = 25
    total = 1 << nbits
    words = total // 64
    idx = np.arange(total, dtype=np.uint32)
    wires = []
    for w in range(nbits):
        bits = ((idx >> w) & 1).astype(np.uint8)
        wires.append(np.packbits(bits, bitorder="little").view(np.uint64))
    assert wires[0].shape == (words,)
    for a, b in pairs:
        assert a < b
        lo = wires[a] & wires[b]
        hi = wires[a] | wires[b]
        wires[a], wires[b] = lo, hi
    # popcount >= 13 <=> median of 25 binary values is 1
    pop = np.zeros(total, dtype=np.uint8)
    for w in range(nbits):
        pop += ((idx >> w) & 1).astype(np.uint8)
    expect = np.packbits((pop >= 13).astype(np.uint8), bitorder="little").view(np.uint64)
    assert np.array_equal(wires[12], expect)


# ---- round 4: the shared-row form (GLH_SORT5 / GLH_MERGE55 / GLH_MID6 + the rank-5 formula of med_fin_pk) ---------------
def _pairs(name):
    src = open(os.path.join(ROOT, "glimpse_amd", "csrc", "glh_median.h")).read()
    body = src[src.index("#define " + name + "(X)"):]
    end = re.search(r"[^\\]\n", body).end()  # the macro ends at the first line without a continuation
    return [(int(a), int(b)) for a, b in re.findall(r"X\((\d+),(\d+)\)", body[:end])]


def _apply(pairs, wires):
    """comparator X(a, b): minimum to wire a, maximum to wire b (a > b occurs)."""
    w = list(wires)
    for a, b in pairs:
        lo, hi = np.minimum(w[a], w[b]), np.maximum(w[a], w[b])
        w[a], w[b] = lo, hi
    return w


def _sorted_patterns(n):
    return np.array([[0] * (n - k) + [1] * k for k in range(n + 1)], dtype=np.int8)


def _sorted_group_inputs(groups):
    """every 0/1 input whose groups are sorted ascending: (tests, sum(groups))."""
    grids = np.meshgrid(*[np.arange(g + 1) for g in groups], indexing="ij")
    cols = []
    for g, k in zip(groups, grids):
        pat = _sorted_patterns(g)[k.ravel()]  # (tests, g)
        cols.append(pat)
    return np.concatenate(cols, axis=1)


def _fin(x, y):
    m = x[0]
    for k in range(1, 6):
        m = np.maximum(m, np.minimum(x[k], y[5 - k]))
    return m


def test_sort5_network_sorts():
    pairs = _pairs("GLH_SORT5_NETWORK")
    assert len(pairs) == 9
    x = np.array([[(i >> b) & 1 for b in range(5)] for i in range(32)], dtype=np.int8)
    w = _apply(pairs, [x[:, k] for k in range(5)])
    np.testing.assert_array_equal(np.stack(w, 1), np.sort(x, axis=1))


def test_merge55_network_merges():
    pairs = _pairs("GLH_MERGE55_NETWORK")
    assert len(pairs) == 13
    x = _sorted_group_inputs([5, 5])
    w = _apply(pairs, [x[:, k] for k in range(10)])
    np.testing.assert_array_equal(np.stack(w, 1), np.sort(x, axis=1))


def test_mid6_network_selects_ranks_7_to_12():
    pairs = _pairs("GLH_MID6_NETWORK")
    x = _sorted_group_inputs([10, 10])
    w = _apply(pairs, [x[:, k] for k in range(20)])
    np.testing.assert_array_equal(np.stack(w[7:13], 1), np.sort(x, axis=1)[:, 7:13])


def test_rank5_formula_of_two_sorted_lists():
    x = _sorted_group_inputs([6, 5])
    got = _fin([x[:, k] for k in range(6)], [x[:, 6 + k] for k in range(5)])
    np.testing.assert_array_equal(got, np.sort(x, axis=1)[:, 5])


def _shared_row_medians(rows):
    """rows: 8 lists of 5 arrays (unsorted) -> the 4 medians of the windows rows j .. j + 4, exactly as
    glh_point.h: pt_highpass_write composes the pieces."""
    s5, m55, mid = _pairs("GLH_SORT5_NETWORK"), _pairs("GLH_MERGE55_NETWORK"), _pairs("GLH_MID6_NETWORK")
    srt = [_apply(s5, r) for r in rows]
    merge = lambda a, b: _apply(m55, a + b)
    mid6 = lambda p, q: _apply(mid, p + q)[7:13]
    m12, m34, m56 = merge(srt[1], srt[2]), merge(srt[3], srt[4]), merge(srt[5], srt[6])
    qa, qb = mid6(m12, m34), mid6(m34, m56)
    return [_fin(qa, srt[0]), _fin(qa, srt[5]), _fin(qb, srt[2]), _fin(qb, srt[7])]


def test_shared_row_composition_zero_one_principle():
    # all 6^8 = 1 679 616 binary inputs with sorted rows (the row sort is proven above, so these are all the inputs
    # the later stages can see); min / max commute with thresholds, hence the composition is exact for every input
    x = _sorted_group_inputs([5] * 8)
    rows = [[x[:, 5 * r + k] for k in range(5)] for r in range(8)]
    got = _shared_row_medians(rows)
    for j in range(4):
        ones = x[:, 5 * j:5 * j + 25].sum(axis=1)
        np.testing.assert_array_equal(got[j], (ones >= 13).astype(np.int8))


def test_shared_row_composition_on_random_keys():
    rng = np.random.default_rng(5)
    x = rng.integers(0, 766, size=(20000, 8, 5)).astype(np.int32)
    x[:5000] = rng.integers(0, 4, size=(5000, 8, 5))  # many ties
    rows = [[x[:, r, k] for k in range(5)] for r in range(8)]
    got = _shared_row_medians(rows)
    for j in range(4):
        np.testing.assert_array_equal(got[j], np.sort(x[:, j:j + 5].reshape(len(x), 25), axis=1)[:, 12])
