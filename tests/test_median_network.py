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
