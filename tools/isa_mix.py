"""Static instruction mix of one kernel in a hipcc -S listing, per basic block.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o glh.s glimpse_amd/csrc/glimpse_hip.hip
    python tools/isa_mix.py glh.s '_ZN3glh12k_point_stepILi512ELi10ELi4ELi1ELb0ELb1EEEvNS_9PointArgsE' [min_block]
"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait/barrier"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        if re.match(r"v_(mov|accvgpr|cndmask|readlane|readfirstlane|writelane|permlane|swap|perm_b32|bfi|alignbit|mov_b64)", op):
            return "v_move/select"
        if op.startswith("v_cmp"):
            return "v_cmp"
        if re.search(r"_f64", op) and not op.startswith("v_cvt"):
            return "v_f64"
        if re.search(r"_f32", op) and not op.startswith("v_cvt"):
            return "v_f32"
        if op.startswith("v_cvt"):
            return "v_cvt"
        if re.search(r"(u64|i64|b64)", op):
            return "v_int64"
        return "v_int32"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_block = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    blocks, cur, label = [], [], "entry"
    total = collections.Counter()
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append((label, cur))
            cur, label = [], m.group(1)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        cur.append((op, t))
        total[classify(op)] += 1
    blocks.append((label, cur))
    n = sum(total.values())
    print(f"{name}: {n} instructions in {len(blocks)} blocks")
    print("  total:", dict(total.most_common()))
    for label, ins in blocks:
        if len(ins) < min_block:
            continue
        c = collections.Counter(classify(op) for op, _ in ins)
        dpp = sum(1 for _, t in ins if "dpp" in t or "row_" in t)
        print(f"\n{label}: {len(ins)}  {dict(c.most_common())}  dpp={dpp}")
        ops = collections.Counter(op for op, _ in ins)
        print("   ", ", ".join(f"{k}:{v}" for k, v in ops.most_common(24)))


main()
