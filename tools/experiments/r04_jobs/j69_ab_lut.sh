# median + match: the sixteen table look-ups of a task issued together after the network (prev.so = HEAD)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4j69_tests.log 2>&1; tail -2 gpurun_out/r4j69_tests.log
for cfg in "" "--streams 1" "--workload C4" "--workload C5 --points 2048" "--channels 3" "--motion tangent_cartesian"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j69_ab_lut.txt 2>&1
cat gpurun_out/r4j69_ab_lut.txt
