# round 5, job 11: the run from files with decoder processes; the parallel call's time, by part; the file-ingest tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "files or parallel" 2>&1 | tail -3
python bench.py --no-cpu-baseline > gpurun_out/r5j11_bench.json 2> gpurun_out/r5j11_bench.err
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5j11_bench.err | tail -12
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j11_bench.json').read().splitlines()[0])
for k in ('C3_from_files_jpeg','C3_from_files_tiff'):
    v=d['secondary'][k]; print(k, {q:v[q] for q in v if q not in ('note','workload')})
print({k:v for k,v in d['api_parallel_2'].items() if k!='note'})
PY
GLH_DECODE_THREADS=1 python bench.py --no-cpu-baseline > gpurun_out/r5j11_bench_threads.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j11_bench_threads.json').read().splitlines()[0])
for k in ('C3_from_files_jpeg','C3_from_files_tiff'):
    v=d['secondary'][k]; print('threads', k, {q:v[q] for q in ('call_seconds','frames_per_s','gpu_idle_share','decode_ms_per_frame_per_core','frame_loop_waited_for_decoders_seconds')})
PY
