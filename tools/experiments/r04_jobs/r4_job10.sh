#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4j10_tests.log 2>&1
tail -5 gpurun_out/r4j10_tests.log
for w in "--motion tangent_cartesian" "--bits 16" ""; do
  echo "--- $w"; AB_ENVS="prev.so" tools/ab.sh --no-secondary $w 2>/dev/null | grep -v "^base"
done | tee gpurun_out/r4j10_ab_general.txt
python bench.py --no-cpu-baseline --no-api 2>gpurun_out/r4j10_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_frame'],4), round(d['roofline']['frac'],4))
for k,v in d['secondary'].items(): print(k, v.get('ms_per_frame'), v.get('roofline_frac'), v.get('variant'), v.get('error'))
" | tee gpurun_out/r4j10_secondary.txt
