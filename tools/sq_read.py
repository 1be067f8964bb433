"""Mean SQ counters of k_point_step over the last launches of tools/sq_probe.sh runs: python tools/sq_read.py tag ..."""
import csv
import sys
from collections import defaultdict

for tag in sys.argv[1:]:
    tot = {}
    for d in ("sq_", "sq2_", "sq3_", "sq4_", "sq5_", "sq6_"):
        vals = defaultdict(list)
        try:
            with open(f"gpurun_out/{d}{tag}/s_counter_collection.csv") as f:
                for row in csv.DictReader(f):
                    if "k_point_step" in row["Kernel_Name"]:
                        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
        except OSError:
            continue
        for k, v in vals.items():
            tot[k] = sum(v[-4:]) / len(v[-4:])
    print(tag, {k: round(v / 1e6, 2) for k, v in sorted(tot.items())})
    if "SQ_INSTS_VALU" in tot:
        pf = 4096 * 5000 / 64
        print("   VALU/64pf", round(tot["SQ_INSTS_VALU"] / pf, 1), "cyc/VALU", round(4 * tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_INSTS_VALU"], 2),
              "LDS/64pf", round(tot.get("SQ_INSTS_LDS", 0) / pf, 1), "SALU/64pf", round(tot.get("SQ_INSTS_SALU", 0) / pf, 1),
              "wave kcycles", round(4 * tot["SQ_WAVE_CYCLES"] / tot["SQ_WAVES"] / 1e3, 1),
              "wait_any", round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 3), "wait_inst", round(tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3),
              "valu_active", round(tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_WAVE_CYCLES"], 3))
