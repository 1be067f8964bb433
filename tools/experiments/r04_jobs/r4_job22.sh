#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
tools/profile_round.sh r04 > gpurun_out/r4j22_profile.log 2>&1
tail -3 gpurun_out/r4j22_profile.log
tools/phase_counts.sh --streams 1 > gpurun_out/r4j22_phase_counts.txt 2>&1
tail -22 gpurun_out/r4j22_phase_counts.txt
PC_FRAME=4 tools/phase_counts.sh --streams 1 --burn-in 0 > gpurun_out/r4j22_phase_counts_frame4.txt 2>&1
cp gpurun_out/phase_counts_C3.json gpurun_out/phase_counts_C3_frame4.json 2>/dev/null
tail -22 gpurun_out/r4j22_phase_counts_frame4.txt
