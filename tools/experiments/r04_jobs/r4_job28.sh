#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests -x -q -m gpu > gpurun_out/r4j28_tests.log 2>&1
tail -3 gpurun_out/r4j28_tests.log
tools/profile_round.sh r04 > gpurun_out/r4j28_profile.log 2>&1
tail -3 gpurun_out/r4j28_profile.log
tools/phase_counts.sh --streams 1 > gpurun_out/r4j28_phase_counts.txt 2>&1
cp gpurun_out/phase_counts_C3.json gpurun_out/r4j28_phase_counts_C3_steady.json
PC_FRAME=4 tools/phase_counts.sh --streams 1 --burn-in 0 --frames-per-step 1 > gpurun_out/r4j28_phase_counts_frame4.txt 2>&1
cp gpurun_out/phase_counts_C3.json gpurun_out/r4j28_phase_counts_C3_frame4.json
tools/phase_counts.sh --workload C5 --points 2048 --streams 1 > gpurun_out/r4j28_phase_counts_C5.txt 2>&1
tail -22 gpurun_out/r4j28_phase_counts_C5.txt
tools/phase_lds.sh --streams 1 > gpurun_out/r4j28_phase_lds.txt 2>&1
python bench.py > gpurun_out/r4j28_full.json 2> gpurun_out/r4j28_full.err
tail -c 300 gpurun_out/r4j28_full.json
