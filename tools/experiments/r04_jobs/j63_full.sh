python bench.py > gpurun_out/r4j63_full.json 2> gpurun_out/r4j63_full.err; tail -1 gpurun_out/r4j63_full.err
python - <<'PY'
import json
l=json.load(open("gpurun_out/r4j63_full.json"))
print("C3", l["ms_per_frame"], l["roofline"]["frac"], "api", l.get("api_ms_per_step"))
for k,v in l["secondary"].items():
    print(k, round(v.get("ms_per_frame",0),4), round(v.get("gpu_ms_per_frame",0),4), round(v.get("roofline_frac",0),3))
PY
