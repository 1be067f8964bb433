#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests/test_gpu_streams.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -3
for w in "" "--workload C4" "--workload C5 --points 2048"; do
  for st in 1 2 3 4; do
    python bench.py --no-cpu-baseline --no-api --no-secondary $w --streams $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w streams $st', round(d['ms_per_frame'],4), 'frac', round(r['frac'],4))"
  done
done 2>&1 | tee gpurun_out/r4j5_streams.txt
python tools/api_timeline.py C3 2>&1 | tail -26 | tee gpurun_out/r4j5_api_timeline.txt
tools/phase_counts.sh --streams 1 > gpurun_out/r4j5_phase_counts.txt 2>&1
tail -25 gpurun_out/r4j5_phase_counts.txt
tools/phase_lds.sh --streams 1 > gpurun_out/r4j5_phase_lds.txt 2>&1
tail -14 gpurun_out/r4j5_phase_lds.txt
