# round 5, job 9: the whole GPU suite and the default bench line (every secondary leg, the API legs, the CPU baselines)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r5j09_tests.txt 2>&1
tail -6 gpurun_out/r5j09_tests.txt
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5j09_bench.json 2> gpurun_out/r5j09_bench.err
tail -25 gpurun_out/r5j09_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j09_bench.json').read().splitlines()[0])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'spread', d['spread']['ms_per_step_all'], 'profiled', d['spread']['profiled_pass_ms_per_step'])
print('frac', d['roofline']['frac'], 'health', d['health'])
for k,v in d['secondary'].items():
    if 'error' in v: print(k, 'ERROR', v['error'][:300]); continue
    print(k, {q: (round(v[q],4) if isinstance(v[q], float) else v[q]) for q in ('ms_per_frame','roofline_frac','frames_per_s','gpu_idle_share','call_seconds','decode_ms_per_frame_per_core','same_tracks_as_arrays','frame_loop_waited_for_decoders_seconds','decode_threads','upload_staging_GBps') if q in v})
print({k:v for k,v in d.items() if k.startswith('api')})
PY
