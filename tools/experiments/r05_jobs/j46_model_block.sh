# round 5, job 46: Tracker.track(parallel=N) hands its workers parameter tables instead of model objects
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -5
python bench.py --no-cpu-baseline --no-secondary 2> gpurun_out/r5j46.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:v for k,v in d.items() if k.startswith('api')})"
