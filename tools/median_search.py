"""Search for cheap min / max programs that select order statistics from PRE-SORTED groups (round 4: the 5 x 5 median
of the fused tile stage, glh_median.h: GLH_MED25_SHARED).

The 0/1 principle holds for any network of min / max operations (they commute with every monotone map), so a program
is correct for all inputs iff it is correct for all 0/1 inputs; with groups that arrive sorted only the 0/1 inputs that
are sorted inside every group occur: 6 per group of 5, 11 per group of 10.  A wire's value over all test inputs is one
Python integer used as a bit set, so evaluating a comparator costs two big-integer operations.

    python tools/median_search.py merge55        # two sorted 5-lists -> sorted 10-list
    python tools/median_search.py mid6           # two sorted 10-lists -> ranks 7..12 of their union, sorted
    python tools/median_search.py med5x5         # five sorted 5-lists -> their median (no sharing)

Cost = min / max operations left after dead-code elimination (a compare-exchange whose min or max is never read is
one operation, not two).  Simulated annealing over comparator lists; prints the best list found.
"""
import itertools
import math
import random
import sys


def sorted_group_patterns(n):
    """0/1 patterns of a sorted (ascending) group of n: k zeros then n - k ones."""
    return [[0] * (n - k) + [1] * k for k in range(n + 1)]


def make_tests(groups):
    """groups: sizes of the pre-sorted groups.  Returns (wire masks, number of tests, ones count per test)."""
    pats = [sorted_group_patterns(g) for g in groups]
    nw = sum(groups)
    masks = [0] * nw
    ones = []
    t = 0
    for combo in itertools.product(*pats):
        bits = [b for p in combo for b in p]
        for w, b in enumerate(bits):
            if b:
                masks[w] |= 1 << t
        ones.append(sum(bits))
        t += 1
    return masks, t, ones


def rank_mask(ones, total, k):
    """bit set of the tests on which the element of rank k (0-based, ascending) of all `total` inputs is 1."""
    m = 0
    for t, o in enumerate(ones):
        if o >= total - k:
            m |= 1 << t
    return m


def run(comps, masks):
    w = list(masks)
    for a, b in comps:
        if a < 0:
            continue
        x, y = w[a], w[b]
        w[a], w[b] = x & y, x | y
    return w


def live_ops(comps, nw, out_wires):
    """operations left after dead-code elimination; also the list of (a, b, need_min, need_max)."""
    live = [False] * nw
    for o in out_wires:
        live[o] = True
    ops = 0
    kept = []
    for a, b in reversed(comps):
        if a < 0:
            continue
        la, lb = live[a], live[b]
        if la or lb:
            ops += la + lb
            kept.append((a, b, la, lb))
            live[a] = live[b] = True
    kept.reverse()
    return ops, kept


def errors(w, targets):
    e = 0
    for wire, mask in targets:
        e += bin(w[wire] ^ mask).count("1")
    return e


def anneal(masks, nw, targets, start, length, iters, seed, t0=2.0, t1=0.05, err_weight=4.0, verbose=True):
    rnd = random.Random(seed)
    comps = list(start) + [(-1, -1)] * max(0, length - len(start))
    outs = [t[0] for t in targets]

    def cost(cs):
        w = run(cs, masks)
        e = errors(w, targets)
        o, _ = live_ops(cs, nw, outs)
        return o + err_weight * e, o, e

    cur, cur_o, cur_e = cost(comps)
    best = (cur_o if cur_e == 0 else 10 ** 9, list(comps))
    for it in range(iters):
        temp = t0 * (t1 / t0) ** (it / iters)
        i = rnd.randrange(len(comps))
        old = comps[i]
        r = rnd.random()
        if r < 0.25:
            comps[i] = (-1, -1)
        elif r < 0.35 and i + 1 < len(comps):
            comps[i], comps[i + 1] = comps[i + 1], comps[i]
        else:
            a = rnd.randrange(nw)
            b = rnd.randrange(nw)
            if a == b:
                continue
            comps[i] = (a, b)  # (min to a, max to b: either orientation is allowed)
        new, new_o, new_e = cost(comps)
        if new <= cur or rnd.random() < math.exp((cur - new) / temp):
            cur, cur_o, cur_e = new, new_o, new_e
            if new_e == 0 and new_o < best[0]:
                best = (new_o, [c for c in comps])
                if verbose:
                    print(f"  it {it}: {new_o} ops", flush=True)
        else:
            if r < 0.35 and r >= 0.25 and i + 1 < len(comps):
                comps[i], comps[i + 1] = comps[i + 1], comps[i]
            else:
                comps[i] = old
    return best


def greedy_prune(comps, masks, nw, targets):
    """drop comparators one at a time while the program stays correct."""
    comps = [c for c in comps if c[0] >= 0]
    changed = True
    while changed:
        changed = False
        for i in range(len(comps)):
            trial = comps[:i] + comps[i + 1:]
            if errors(run(trial, masks), targets) == 0:
                comps = trial
                changed = True
                break
    return comps


def batcher_merge(lo_a, n_a, lo_b, n_b):
    """odd-even merge of two sorted runs on consecutive wires by padding to a power of two (pruned by the caller)."""
    n = 1
    while n < max(n_a, n_b):
        n *= 2
    # virtual wires 0 .. 2n-1: first run then padding (+inf), second run then padding
    real = {}
    for i in range(n_a):
        real[i] = lo_a + i
    for i in range(n_b):
        real[n + i] = lo_b + i
    comps = []

    def merge(lo, cnt, r):
        step = r * 2
        if step < cnt:
            merge(lo, cnt, step)
            merge(lo + r, cnt, step)
            for i in range(lo + r, lo + cnt - r, step):
                comps.append((i, i + r))
        else:
            comps.append((lo, lo + r))

    merge(0, 2 * n, 1)
    out = []
    for a, b in comps:
        if a in real and b in real:
            out.append((real[a], real[b]))
        # a comparator with a +inf partner leaves the real wire where it is (padding sits at the top of each run)
    return out


SORT5 = [(0, 1), (3, 4), (2, 4), (2, 3), (1, 4), (0, 3), (0, 2), (1, 3), (1, 2)]


def problem(name):
    if name == "merge55":
        groups = [5, 5]
        masks, nt, ones = make_tests(groups)
        # output: fully sorted on wires in the order given by `order`
        return groups, masks, nt, ones, list(range(10))
    raise SystemExit("unknown problem")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "med5x5"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if what == "merge55":
        groups = [5, 5]
        masks, nt, ones = make_tests(groups)
        nw = 10
        # any assignment of ranks to wires is acceptable in principle; fix: rank k ends on wire perm[k] -- searched with
        # the natural target "rank k on wire k" after an odd-even start
        targets = [(k, rank_mask(ones, nw, k)) for k in range(nw)]
        start = batcher_merge(0, 5, 5, 5)
        print("start", len(start), "comparators, errors", errors(run(start, masks), targets))
        length = 20
    elif what == "mid6":
        groups = [10, 10]
        masks, nt, ones = make_tests(groups)
        nw = 20
        targets = [(k, rank_mask(ones, nw, k)) for k in range(7, 13)]
        start = batcher_merge(0, 10, 10, 10)
        print("start", len(start), "comparators, errors", errors(run(start, masks), targets))
        length = 45
    elif what == "med5x5":
        groups = [5] * 5
        masks, nt, ones = make_tests(groups)
        nw = 25
        targets = [(12, rank_mask(ones, nw, 12))]
        start = []
        length = 70
    else:
        raise SystemExit("unknown problem")
    print(f"{what}: {nw} wires, {nt} tests")
    if start and errors(run(start, masks), targets) == 0:
        pr = greedy_prune(start, masks, nw, targets)
        o, _ = live_ops(pr, nw, [t[0] for t in targets])
        print("pruned start:", len(pr), "comparators,", o, "ops")
        start = pr
    best = anneal(masks, nw, targets, start, length, iters, seed)
    comps = [c for c in best[1] if c[0] >= 0]
    comps = greedy_prune(comps, masks, nw, targets) if best[0] < 10 ** 9 else comps
    o, kept = live_ops(comps, nw, [t[0] for t in targets])
    print("best:", o, "ops,", len(kept), "comparators")
    print(kept)


if __name__ == "__main__":
    main()
