timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "last_pixel" > gpurun_out/r4j67_tests.log 2>&1; tail -25 gpurun_out/r4j67_tests.log
