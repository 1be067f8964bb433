timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4j49_tests.log 2>&1; tail -2 gpurun_out/r4j49_tests.log
python bench.py --no-cpu-baseline --no-api > gpurun_out/r4j49_bench.json 2> gpurun_out/r4j49_bench.err; tail -3 gpurun_out/r4j49_bench.err
python - <<'PY'
import json
l=json.load(open("gpurun_out/r4j49_bench.json"))
print(l["ms_per_frame"], l["roofline"]["frac"])
for k,v in l["secondary"].items():
    print(k, v.get("ms_per_frame"), v.get("roofline_frac"), v.get("error"))
PY
