#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests -x -q -m gpu > gpurun_out/r4j16_tests.log 2>&1
tail -3 gpurun_out/r4j16_tests.log
tools/profile_round.sh r04 > gpurun_out/r4j16_profile.log 2>&1
tail -5 gpurun_out/r4j16_profile.log
