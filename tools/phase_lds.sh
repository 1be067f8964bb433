#!/bin/bash
# LDS activity and bank conflicts of the fused kernel phase by phase (same cuts as tools/phase_counts.sh, steady-state C3):
#   tools/phase_lds.sh [bench args]  -> table on stdout
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python3 bench.py --no-cpu-baseline --no-api --no-secondary --burn-in 6 --steps 4 --warmup 2 "$@" > /dev/null 2>&1
for k in 16 17 19 2 3 4 5 7 8 full; do
  if [ $k = full ]; then unset GLH_PT_STOP; else export GLH_PT_STOP=$k:${PC_FRAME:-10}; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES \
    -d gpurun_out/pl_$k -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-api --no-secondary --burn-in 6 --steps 4 --warmup 2 "$@" > gpurun_out/pl_$k.log 2>&1
done
python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
ORDER = ["16", "17", "19", "2", "3", "4", "5", "7", "8", "full"]
WHAT = {"16": "prologue", "17": "A particle loop", "19": "A reductions + box", "2": "B tile prep", "3": "B SSD", "4": "B fit", "5": "C sampling",
        "7": "D", "8": "E gather", "full": "F"}
prev = defaultdict(float)
out = []
print(f"{'cut':>5} {'LDS instr':>10} {'idx active':>11} {'bank confl':>11} {'share':>6} {'addr confl':>11}  phase")
for k in ORDER:
    vals = defaultdict(list)
    for path in glob.glob(f"gpurun_out/pl_{k}/**/s_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "k_point_step" in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    cur = {c: sorted(v)[-1][1] for c, v in vals.items()}
    if not cur:
        continue
    d = {c: cur[c] - prev[c] for c in cur}
    act = d.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"{k:>5} {d.get('SQ_INSTS_LDS', 0)/1e6:10.2f} {act/1e6:11.2f} {d.get('SQ_LDS_BANK_CONFLICT', 0)/1e6:11.2f} "
          f"{(d.get('SQ_LDS_BANK_CONFLICT', 0)/act if act else 0):6.2f} {d.get('SQ_LDS_ADDR_CONFLICT', 0)/1e6:11.2f}  {WHAT[k]}")
    out.append({"cut": k, "phase": WHAT[k], "lds_instructions": d.get("SQ_INSTS_LDS", 0), "idx_active_cycles": act,
                "bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT", 0), "addr_conflict_cycles": d.get("SQ_LDS_ADDR_CONFLICT", 0),
                "bank_conflict_share": (d.get("SQ_LDS_BANK_CONFLICT", 0) / act if act else 0)})
    prev = defaultdict(float, cur)
tot_a = sum(o["idx_active_cycles"] for o in out); tot_c = sum(o["bank_conflict_cycles"] for o in out)
json.dump({"unit": "cycles / instructions per launch (last k_point_step launch of a short steady-state sequence, cut at every "
                   "phase stamp: differences of successive cuts), rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ...",
           "total_idx_active_cycles": tot_a, "total_bank_conflict_cycles": tot_c,
           "bank_conflict_share": tot_c / tot_a if tot_a else 0, "phases": out}, open("gpurun_out/phase_lds.json", "w"), indent=1)
PY
