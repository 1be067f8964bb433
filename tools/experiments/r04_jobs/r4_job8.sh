#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests/test_gpu_fused.py tests/test_gpu_pinned.py tests/test_gpu_streams.py tests/test_gpu_fullsize.py tests/test_gpu_benched_instantiations.py -x -q -m gpu > gpurun_out/r4j8_tests.log 2>&1
tail -3 gpurun_out/r4j8_tests.log
for w in "" "--workload C4" "--workload C5 --points 2048" "--streams 1" "--motion tangent_cartesian"; do
  echo "--- $w"; AB_ENVS="gu2.so" tools/ab.sh --no-secondary $w 2>/dev/null | grep -v "^base"
done | tee gpurun_out/r4j8_ab_gu.txt
