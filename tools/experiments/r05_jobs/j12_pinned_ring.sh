# round 5, job 12: decoders -> page-locked shared ring -> device (no staging copy): tests, the from-files legs
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_abi.py tests/test_parallel_pool.py -x -q -k "files or parallel or abi or decoder" 2>&1 | tail -5
python bench.py --no-cpu-baseline > gpurun_out/r5j12_bench.json 2> gpurun_out/r5j12_bench.err
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|no RCCL communicator" gpurun_out/r5j12_bench.err | tail -8
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j12_bench.json').read().splitlines()[0])
for k in ('C3_from_files_jpeg','C3_from_files_tiff'):
    v=d['secondary'][k]; print(k, {q:v[q] for q in v if q not in ('note','workload')})
print({k:v for k,v in d['api_parallel_2'].items() if k!='note'})
PY
