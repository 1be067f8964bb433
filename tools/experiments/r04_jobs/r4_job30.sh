#!/bin/bash
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests/test_gpu_fused.py tests/test_gpu_pinned.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4j30_tests.log 2>&1
tail -3 gpurun_out/r4j30_tests.log
for w in "" "--channels 3" "--streams 1"; do
  echo "--- $w"; AB_ENVS="prev.so" tools/ab.sh --no-secondary $w 2>/dev/null | grep -v "^base"
done | tee gpurun_out/r4j30_ab_hcl.txt
