#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests -x -q -m gpu > gpurun_out/r4j31_tests.log 2>&1
tail -3 gpurun_out/r4j31_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4j31_driver_line.json 2> gpurun_out/r4j31_driver_line.err
python -c "
import json; d=json.load(open('gpurun_out/r4j31_driver_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['health'])"
