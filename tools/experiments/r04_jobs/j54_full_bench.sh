python bench.py > gpurun_out/r4j54_full.json 2> gpurun_out/r4j54_full.err; tail -2 gpurun_out/r4j54_full.err
python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/r4j54_driver.json 2> gpurun_out/r4j54_driver.err
python - <<'PY'
import json
l=json.load(open("gpurun_out/r4j54_full.json"))
print(l["ms_per_frame"], l["roofline"]["frac"], l.get("api_ms_per_step"), l["cpu_baseline"]["value"])
for k,v in l["secondary"].items():
    print(k, v.get("ms_per_frame"), v.get("roofline_frac"), v.get("error"))
d=json.load(open("gpurun_out/r4j54_driver.json")); print("driver cmd", d["ms_per_step"], d.get("ms_per_frame"), d["roofline"]["frac"])
PY
