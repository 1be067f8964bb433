# 16-bit / float frames: template CDF staged with one memory latency (prev.so = before)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q -k "float or f32 or f64 or 16 or wide or depth or dtype or highpass" > gpurun_out/r4j52_tests.log 2>&1
tail -2 gpurun_out/r4j52_tests.log
for cfg in "--bits 16" "--bits 32"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j52_ab_cdf.txt 2>&1
cat gpurun_out/r4j52_ab_cdf.txt
