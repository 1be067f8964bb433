#!/bin/bash
# round-4 job 2: shared-row median + unrolled observer loop; parity + same-box A/B against round 3's HEAD (base.so)
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_pinned.py tests/test_gpu_benched_instantiations.py tests/test_gpu_fast_math.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4j2_tests.log 2>&1
tail -3 gpurun_out/r4j2_tests.log
echo "--- C3"; tools/ab.sh --no-secondary 2>&1 | tee gpurun_out/r4j2_ab_C3.txt
echo "--- C5 (2048 points)"; tools/ab.sh --workload C5 --points 2048 --no-secondary 2>&1 | tee gpurun_out/r4j2_ab_C5.txt
echo "--- C4"; tools/ab.sh --workload C4 --no-secondary 2>&1 | tee gpurun_out/r4j2_ab_C4.txt
echo "--- C2"; tools/ab.sh --workload C2 --no-secondary 2>&1 | tee gpurun_out/r4j2_ab_C2.txt
echo "--- C3 rgb"; tools/ab.sh --channels 3 --no-secondary 2>&1 | tee gpurun_out/r4j2_ab_C3rgb.txt
