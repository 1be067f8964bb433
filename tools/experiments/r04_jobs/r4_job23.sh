#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
tools/phase_counts.sh --streams 1 > gpurun_out/r4j23_phase_counts.txt 2>&1
cp gpurun_out/phase_counts_C3.json gpurun_out/r4j23_phase_counts_C3_steady.json
PC_FRAME=4 tools/phase_counts.sh --streams 1 --burn-in 0 --frames-per-step 1 > gpurun_out/r4j23_phase_counts_frame4.txt 2>&1
cp gpurun_out/phase_counts_C3.json gpurun_out/r4j23_phase_counts_C3_frame4.json
tail -22 gpurun_out/r4j23_phase_counts_frame4.txt
tools/phase_lds.sh --streams 1 > gpurun_out/r4j23_phase_lds.txt 2>&1
tail -12 gpurun_out/r4j23_phase_lds.txt
python bench.py > gpurun_out/r4j23_full.json 2> gpurun_out/r4j23_full.err
tail -c 600 gpurun_out/r4j23_full.json
