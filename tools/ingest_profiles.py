"""Turn the files tools/profile_round.sh left under gpurun_out/prof_<tag>/ into the tracked profiles/<tag>_* set:
bench lines, rocprofv3 kernel stats, PMC traffic summaries (tools/pmc_summary.py) and the C3 SQ counters.

    python tools/ingest_profiles.py gpurun_out/prof_r02b r02
"""
import csv
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"C3": "C3:4096x5000", "C2": "C2:256x2000", "C4": "C4:1250x10000", "C5": "C5:2048x5000"}
PF = {"C3": 4096 * 5000, "C2": 256 * 2000, "C4": 1250 * 10000, "C5": 2048 * 5000}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    for w, key in KEYS.items():
        bench = os.path.join(src, f"{w}_bench.json")
        if not os.path.exists(bench):
            continue
        shutil.copy(bench, os.path.join(prof, f"{tag}_{w}_bench.json"))
        shutil.copy(os.path.join(src, f"{w}_trace", "t_kernel_stats.csv"), os.path.join(prof, f"{tag}_{w}_kernel_stats.csv"))
        line = json.load(open(bench))
        # launches of the dominant kernel before the timed region (warm-up steps + burn-in), as the bench line reports them
        warm = int(line["config"].get("untimed_kernel_launches", line["config"]["untimed_launches"]))
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f"{tag}_{w}", key,
                        os.path.join(src, f"{w}_trace", "t_kernel_trace.csv"),
                        os.path.join(src, f"{w}_fetch", "f_counter_collection.csv"),
                        os.path.join(src, f"{w}_write", "w_counter_collection.csv"), str(warm), str(warm)],
                       check=True, stdout=subprocess.DEVNULL)
        s = json.load(open(os.path.join(prof, f"{tag}_{w}_summary.json")))
        k = s["kernels"]["k_point_step"]
        # the points of one launch (two streams: half the batch): bench.py scales the per-launch traffic by it
        tpath = os.path.join(prof, "pmc_traffic.json")
        table = json.load(open(tpath))
        if key in table and "k_point_step" in table[key]:
            table[key]["k_point_step"]["points_per_launch"] = line["roofline"].get("per_launch", {}).get(
                "points", line["config"]["points_per_gpu"])
            json.dump(table, open(tpath, "w"), indent=1)
        print(f"{w}: bench avg_launch_ms {line['roofline']['avg_launch_ms']:.4f}  rocprof mean_ms_timed {k['mean_ms_timed']:.4f} "
              f"({k['launches']} launches)  frac {line['roofline']['frac']:.4f}  "
              f"traffic {k.get('hbm_bytes_per_launch', 0) / 1e9:.3f} GB vs algorithmic "
              f"{line['roofline']['algorithmic_bytes_per_launch'] / 1e9:.3f} GB")
    one = os.path.join(src, "C3s1_bench.json")  # the headline configuration with one launch per frame (--streams 1)
    if os.path.exists(one):
        shutil.copy(one, os.path.join(prof, f"{tag}_C3_one_stream_bench.json"))
        shutil.copy(os.path.join(src, "C3s1_trace", "t_kernel_stats.csv"), os.path.join(prof, f"{tag}_C3_one_stream_kernel_stats.csv"))
        line = json.load(open(one))
        print(f"C3 one stream: avg_launch_ms {line['roofline']['avg_launch_ms']:.4f} frac {line['roofline']['frac']:.4f} "
              f"first steps {line.get('first_steps_ms')}")
    full = os.path.join(src, "C3_full.json")
    if os.path.exists(full):
        shutil.copy(full, os.path.join(prof, f"{tag}_C3_full_bench.json"))
    sq = os.path.join(src, "C3_sq", "s_counter_collection.csv")
    if os.path.exists(sq):
        vals = defaultdict(list)
        with open(sq) as f:
            for row in csv.DictReader(f):
                if "k_point_step" in row["Kernel_Name"]:
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
        n = min(len(v) for v in vals.values())
        # launches of a frame update (two streams: two launches of half the points each): the figures below are PER FRAME
        # UPDATE, i.e. the sum over a frame's launches
        try:
            line = json.load(open(os.path.join(src, "C3_bench.json")))
            lpf = int(round(line["roofline"]["launches_per_step"] / line["config"]["frame_updates_per_step"]))
        except (OSError, ValueError, KeyError):
            lpf = 1
        last = 99 * lpf
        timed = {k: v[-last:] if n > last else v for k, v in vals.items()}
        out = {k: lpf * sum(v) / len(v) for k, v in sorted(timed.items())}
        out["launches_per_frame_update"] = lpf
        out["note"] = (f"per frame update (sum of its {lpf} launch(es)), mean over the {len(next(iter(timed.values())))} timed launches of: rocprofv3 --kernel-trace --pmc "
                       + " ".join(sorted(vals)) + " -- python3 bench.py --no-cpu-baseline --no-api (C3, 100 frames from the "
                       "prior, GLH_MATH_FAST); SQ_*_CYCLES in quad-cycles")
        out["derived"] = {
            "valu_wave_instructions_per_64_particle_frames": out["SQ_INSTS_VALU"] / (PF["C3"] / 64),
            "valu_active_share_of_wave_cycles": out["SQ_ACTIVE_INST_VALU"] / out["SQ_WAVE_CYCLES"],
            "wait_any_share": out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"],
            "wait_inst_any_share": out["SQ_WAIT_INST_ANY"] / out["SQ_WAVE_CYCLES"],
            "wave_kcycles": 4 * out["SQ_WAVE_CYCLES"] / out["SQ_WAVES"] / 1e3,
        }
        json.dump(out, open(os.path.join(prof, f"{tag}_C3_sq_counters.json"), "w"), indent=1)
        print("SQ:", {k: round(v, 4) for k, v in out["derived"].items()})


main()
