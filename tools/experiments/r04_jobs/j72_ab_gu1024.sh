# the 1 024-thread plain code with three records in flight in the gather (prev.so = two)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_pinned.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r4j72_tests.log 2>&1; tail -2 gpurun_out/r4j72_tests.log
for cfg in "--workload C4" "--workload C4 --streams 1" "--workload C4 --math exact"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j72_ab_gu1024.txt 2>&1
cat gpurun_out/r4j72_ab_gu1024.txt
