# tile fetch: unconditional loads, channel count decided outside the four fetches, RGB as one unaligned dword (prev.so = HEAD)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_pinned.py -m gpu -x -q > gpurun_out/r4j66_tests.log 2>&1; tail -2 gpurun_out/r4j66_tests.log
for cfg in "" "--channels 3" "--streams 1" "--workload C4" "--workload C5 --points 2048" "--workload C2"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j66_ab_fetch.txt 2>&1
cat gpurun_out/r4j66_ab_fetch.txt
