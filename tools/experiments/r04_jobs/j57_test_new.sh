timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "three_observers or every_motion_model" > gpurun_out/r4j57_tests.log 2>&1; tail -15 gpurun_out/r4j57_tests.log
