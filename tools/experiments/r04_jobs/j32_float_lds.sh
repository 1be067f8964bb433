# float32 tiles in LDS + block-parallel NumPy sums + matched values by count: parity, then the phase split
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q -k "float or f32 or f64 or 16 or wide or depth or dtype" > gpurun_out/r4j32_tests.log 2>&1
tail -5 gpurun_out/r4j32_tests.log
for b in 32 16; do GLH_BITS=$b timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j32_phase_b$b.txt 2>&1; grep -n "split\|point_step\|tile_prep " gpurun_out/r4j32_phase_b$b.txt; done
