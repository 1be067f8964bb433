"""Table of tools/phase_counts.sh: SQ counters of the LAST k_point_step launch of every cut, and their differences."""
import csv
import glob
import json
import sys
from collections import defaultdict

ORDER = ["0", "15", "16", "17", "18", "19", "1", "2", "3", "4", "5", "6", "10", "11", "12", "7", "8", "9", "full"]
WHAT = {"0": "(entry)", "15": "prologue: arguments, tables, record indices", "16": "A: start-up loads", "17": "A: particle loop",
        "18": "A: wave / block reduction of the boxes", "19": "A: search box (thread 0)", "1": "A: barrier",
        "2": "B: tile prep", "3": "B: SSD", "4": "B: spline fit", "5": "C: sampling (+ weights)", "6": "C: exp (when separate)",
        "10": "D: weight sum", "11": "D: scan", "12": "D: fix-up", "7": "D: search + rank tables", "8": "E: gather",
        "9": "F: moments", "full": "(exit)"}
# particle-frames of one launch / 64: C3 unless the bench arguments say otherwise (--workload W [--points P])
ARGS = sys.argv[1:]
SHAPES = {"C2": (256, 2000), "C3": (4096, 5000), "C4": (1250, 10000), "C5": (512, 5000)}
wname = ARGS[ARGS.index("--workload") + 1] if "--workload" in ARGS else "C3"
P, N = SHAPES[wname]
if "--points" in ARGS:
    P = int(ARGS[ARGS.index("--points") + 1])
units = P * N / 64
rows, times = {}, {}
for k in ORDER:
    tr = []
    for path in glob.glob(f"gpurun_out/pc_{k}/**/s_kernel_trace.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "k_point_step" in row["Kernel_Name"]:
                    tr.append((int(row["Dispatch_Id"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
    if tr:
        times[k] = sorted(tr)[-1][1]
for k in ORDER:
    vals = defaultdict(list)
    for path in glob.glob(f"gpurun_out/pc_{k}/**/s_counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "k_point_step" in row["Kernel_Name"]:
                    vals[row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    rows[k] = {c: sorted(v)[-1][1] for c, v in vals.items()}
out, prev = [], None
print(f"{'cut':>5} {'VALU/64pf':>10} {'d VALU':>8} {'d SALU':>8} {'d LDS':>7} {'d VMEM':>7} {'ms':>8} {'d ms':>8}  phase ending at the cut")
tprev = 0.0
for k in ORDER:
    r = rows.get(k)
    if not r or "SQ_INSTS_VALU" not in r:
        continue
    cur = {c: r.get(c, 0.0) / units for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    cur["VMEM"] = (r.get("SQ_INSTS_VMEM_RD", 0.0) + r.get("SQ_INSTS_VMEM_WR", 0.0)) / units
    d = {c: cur[c] - (prev[c] if prev else 0.0) for c in cur}
    t = times.get(k, float("nan"))
    print(f"{k:>5} {cur['SQ_INSTS_VALU']:10.1f} {d['SQ_INSTS_VALU']:8.1f} {d['SQ_INSTS_SALU']:8.1f} {d['SQ_INSTS_LDS']:7.1f} {d['VMEM']:7.2f} {t:8.4f} {t - tprev:8.4f}  {WHAT[k]}")
    out.append({"cut": k, "phase": WHAT[k], "cumulative_per_64pf": cur, "delta_per_64pf": d, "launch_ms_cut_here": t})
    tprev = t
    prev = cur
json.dump({"unit": f"wave-instructions per 64 particle-frames ({wname}: {P} x {N}), last launch of the sequence", "cuts": out},
          open(f"gpurun_out/phase_counts_{wname}.json", "w"), indent=1)
