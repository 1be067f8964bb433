#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r4j29_tests.log 2>&1
tail -5 gpurun_out/r4j29_tests.log
