#!/bin/bash
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python tools/experiments/two_streams.py C3 40 2 2>&1 | tail -2 | tee gpurun_out/r4j3_two_streams.txt
python tools/experiments/two_streams.py C3 40 4 2>&1 | tail -1 | tee -a gpurun_out/r4j3_two_streams.txt
python tools/experiments/two_streams.py C4 40 2 2>&1 | tail -1 | tee -a gpurun_out/r4j3_two_streams.txt
python tools/api_timeline.py C3 2>&1 | tail -30 | tee gpurun_out/r4j3_api_timeline.txt
