# round 5, job 25: the new automatic stream rule: tests, the default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_lifecycle.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5j25_bench.json 2> gpurun_out/r5j25_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j25_bench.json').read().splitlines()[0])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'health', d['health'])
for k,v in d['secondary'].items():
    if 'error' in v: print(k, 'ERROR', v['error'][:300]); continue
    print(k, {q: (round(v[q],4) if isinstance(v[q], float) else v[q]) for q in ('ms_per_frame','roofline_frac','track_streams','frames_per_s','gpu_idle_share') if q in v})
PY
