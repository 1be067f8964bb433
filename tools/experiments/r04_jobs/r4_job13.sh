#!/bin/bash
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4j15_tests.log 2>&1
tail -5 gpurun_out/r4j15_tests.log
for bits in 32 64; do for mode in 1 0; do
  echo "bits $bits fused $mode:"; python tools/experiments/float_frames_probe.py $bits 1024 $mode 30 2>&1 | tail -2 | head -1
done; done | tee gpurun_out/r4j15_float_probe.txt
python bench.py --no-cpu-baseline --no-api 2>gpurun_out/r4j15_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_frame'],4), round(d['roofline']['frac'],4))
for k,v in d['secondary'].items(): print(k, v.get('ms_per_frame'), v.get('roofline_frac'), v.get('variant'), v.get('error'))
" | tee gpurun_out/r4j15_secondary.txt
