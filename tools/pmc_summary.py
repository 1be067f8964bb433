"""Summarise rocprofv3 runs of bench.py into profiles/ (run here, on files merged back from the GPU box).

    python tools/pmc_summary.py <tag> <workload key, e.g. C3:4096x5000> <kernel_trace.csv> <fetch.csv> <write.csv>
                                [untimed launches in the trace run] [untimed launches in the PMC runs]

* kernel trace  -> mean duration per kernel, and the mean over the TIMED launches only (the bench's
  warm-up launches of the dominant kernel are dropped), to set beside bench.py's HIP-event figure;
* PMC passes (separate runs, one counter each: FETCH_SIZE, WRITE_SIZE) -> HBM bytes per launch,
  corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: counters are in KiB and
  FETCH_SIZE reports half of the bytes of a wide (16 B / lane) coalesced read stream, so
      hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
  and written to profiles/pmc_traffic.json, which bench.py reads for `roofline.traffic`.
"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    name = name.split("(")[0]
    name = name.split("<")[0]
    return name.replace("glh::", "")


def main():
    tag, key, trace, fetch, write = sys.argv[1:6]
    warmup = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    pmc_skip = int(sys.argv[7]) if len(sys.argv) > 7 else 2
    dur = defaultdict(list)
    with open(trace) as f:
        for row in csv.DictReader(f):
            dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    counters = {}
    for path, cname in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        vals = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == cname:
                    vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        counters[cname] = vals
    summary = {"tag": tag, "workload": key, "kernels": {}}
    for k, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if not k.startswith("k_"):
            continue
        timed = d[warmup:] if len(d) > warmup + 1 else d
        e = {"launches": len(d), "mean_ms_all": sum(d) / len(d) / 1e6, "mean_ms_timed": sum(timed) / len(timed) / 1e6}
        fs = counters["FETCH_SIZE"].get(k)
        ws = counters["WRITE_SIZE"].get(k)
        if fs and ws:
            fs_t = fs[pmc_skip:] if len(fs) > pmc_skip + 1 else fs  # the PMC passes' burn-in + warm-up launches
            ws_t = ws[pmc_skip:] if len(ws) > pmc_skip + 1 else ws
            f_kib, w_kib = sum(fs_t) / len(fs_t), sum(ws_t) / len(ws_t)
            e.update({"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                      "hbm_read_bytes_per_launch": 2.0 * f_kib * 1024, "hbm_write_bytes_per_launch": w_kib * 1024,
                      "hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024,
                      "correction": "gfx950: FETCH_SIZE x2 for 16 B/lane streams, KiB units (MI355X_MICROARCH.md, HBM)"})
        summary["kernels"][k] = e
    out = os.path.join(ROOT, "profiles", f"{tag}_summary.json")
    with open(out, "w") as f:
        json.dump(summary, f, indent=1)
    table_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(table_path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        table = {}
    table[key] = {k: {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "source": f"profiles/{tag}_summary.json"}
                  for k, e in summary["kernels"].items() if "hbm_bytes_per_launch" in e}
    # (a launch of k_point_step is what glh_track enqueues: with two streams, half of the points -- bench.py multiplies
    # by the launches of a frame)
    import datetime
    table["_meta"] = {"collected": "round 5, " + datetime.date.today().isoformat(),
                      "launch": "one k_point_step launch as glh_track enqueues it (two streams: half of the points)"}
    with open(table_path, "w") as f:
        json.dump(table, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
