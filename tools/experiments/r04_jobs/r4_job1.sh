#!/bin/bash
# round-4 job 1: parity of the touched instantiations + same-box A/B (base = HEAD of round 3)
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests/test_gpu_fused.py tests/test_gpu_pinned.py tests/test_gpu_benched_instantiations.py tests/test_gpu_fast_math.py -x -q -m gpu > gpurun_out/r4j1_tests.log 2>&1
tail -3 gpurun_out/r4j1_tests.log
echo "--- C5 (2048 points)"; AB_ENVS="GLH_PT_PPT0=1" tools/ab.sh --workload C5 --points 2048 --no-secondary 2>&1 | tee gpurun_out/r4j1_ab_C5.txt
echo "--- C5 shard (512 points)"; tools/ab.sh --workload C5 --points 512 --no-secondary 2>&1 | tee gpurun_out/r4j1_ab_C5s.txt
echo "--- C3"; tools/ab.sh --no-secondary 2>&1 | tee gpurun_out/r4j1_ab_C3.txt
echo "--- C4"; tools/ab.sh --workload C4 --no-secondary 2>&1 | tee gpurun_out/r4j1_ab_C4.txt
