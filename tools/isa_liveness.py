"""Approximate VGPR liveness over a kernel's assembly listing (hipcc -S): where is the register peak?

    python tools/isa_liveness.py /tmp/regcheck/one.s [kernel-name-substring] [top]

Backward dataflow over the control-flow graph of the listing (labels, s_branch / s_cbranch_*).  The first operand of
an instruction is taken as its definition unless the mnemonic is a store / compare / export; v_fmac / v_mac / dpp /
writelane destinations are also uses.  Partial-exec definitions are treated as full definitions, so the numbers are a
lower bound -- good enough to see which region of the kernel holds the peak.
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


NO_DEF = ("ds_write", "global_store", "buffer_store", "flat_store", "scratch_store", "v_cmp", "v_cmpx", "s_",
          "global_atomic", "ds_add_u32", "ds_max", "ds_min", "v_readlane", "v_readfirstlane", "ds_append", "exp",
          "buffer_atomic", "ds_or", "ds_and", "ds_inc", "ds_dec", "ds_sub")
ALSO_USE = ("v_fmac", "v_mac", "v_writelane", "v_pk_fmac", "v_dot2c", "v_movrel")


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_point_step"
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(path).read().split("\n")
    start = max(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(want) + r"\S*:", l))
    ins = []  # (text, defs, uses, label or None)
    labels = {}
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        op = t.split()[0]
        ops = t[len(op):].split(",")
        d, u = set(), set()
        if op.startswith(NO_DEF) or "atomic" in op and "ret" not in op:
            for o in ops:
                u |= regs(o)
        else:
            d = regs(ops[0]) if ops else set()
            for o in ops[1:]:
                u |= regs(o)
            if op.startswith(ALSO_USE) or "dpp" in t or "row_" in t or "d16" in op:
                u |= d
        ins.append((t, d, u, op))
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (t, d, u, op) in enumerate(ins):
        if op == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels:
                succ[i].append(labels[tgt])
            continue
        if op in ("s_endpgm",):
            continue
        if i + 1 < n:
            succ[i].append(i + 1)
        if op.startswith("s_cbranch"):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] < n:
                succ[i].append(labels[tgt])
    live_in = [set() for _ in range(n)]
    changed = True
    rounds = 0
    while changed and rounds < 60:
        changed = False
        rounds += 1
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            new = (out - ins[i][1]) | ins[i][2]
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    counts = [len(s) for s in live_in]
    print(f"{n} instructions, {rounds} rounds, peak live VGPRs {max(counts)}")
    # regions of high pressure
    inv = {v: k for k, v in labels.items()}
    thr = max(counts) - 6
    i = 0
    shown = 0
    while i < n and shown < top:
        if counts[i] >= thr:
            j = i
            while j < n and counts[j] >= thr - 6:
                j += 1
            lab = max((v for v in labels.values() if v <= i), default=0)
            print(f"\n[{i}..{j}) peak {max(counts[i:j])} after label {inv.get(lab)} (+{i - lab})")
            k = max(range(i, j), key=lambda q: counts[q])
            for q in range(max(k - 4, 0), min(k + 5, n)):
                print(f"   {counts[q]:4d}  {ins[q][0][:100]}")
            shown += 1
            i = j
        else:
            i += 1
    # coarse profile: max pressure per 2 % of the listing
    step = max(n // 50, 1)
    print("\nprofile (max live per 2% of the listing):")
    print(" ".join(str(max(counts[a:a + step])) for a in range(0, n, step)))


main()
