#!/bin/bash
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests -x -q -m gpu > gpurun_out/r4j4_tests.log 2>&1
tail -3 gpurun_out/r4j4_tests.log
echo "--- C3"; AB_ENVS="--streams=1" tools/ab.sh --no-secondary 2>&1 | tee gpurun_out/r4j4_ab_C3.txt
echo "--- C4"; AB_ENVS="--streams=1" tools/ab.sh --workload C4 --no-secondary 2>&1 | tee gpurun_out/r4j4_ab_C4.txt
echo "--- C5 (2048 points)"; AB_ENVS="--streams=1" tools/ab.sh --workload C5 --points 2048 --no-secondary 2>&1 | tee gpurun_out/r4j4_ab_C5.txt
python bench.py > gpurun_out/r4j4_full.json 2> gpurun_out/r4j4_full.err
tail -c 6000 gpurun_out/r4j4_full.json
