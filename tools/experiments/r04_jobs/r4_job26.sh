#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/api_profile.py C3 2>&1 | head -40 | tee gpurun_out/r4j26_api_profile.txt
python tools/api_timeline.py C3 2>&1 | tail -24 | tee gpurun_out/r4j26_api_timeline.txt
