#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "--- plain"; API_REPS=4 python tools/api_timeline.py C3 2>&1 | grep -E "warm call|everything|set_motion|get_tracks" 
echo "--- gc off"; API_REPS=4 API_GC_OFF=1 python tools/api_timeline.py C3 2>&1 | grep -E "warm call|everything|set_motion|get_tracks"
