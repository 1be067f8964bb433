#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/phase_probe.py C3 4096 5000 30 2>&1 | tail -25 | tee gpurun_out/r4j21_phase_probe_C3.txt
python tools/phase_probe.py C4 1250 10000 30 2>&1 | tail -25 | tee gpurun_out/r4j21_phase_probe_C4.txt
