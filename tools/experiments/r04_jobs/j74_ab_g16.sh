# SSD: template rows of small surfaces on 16 lanes (prev.so = at most 8)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_pinned.py -m gpu -x -q > gpurun_out/r4j74_tests.log 2>&1; tail -2 gpurun_out/r4j74_tests.log
for cfg in "" "--workload C5 --points 2048" "--workload C2" "--motion tangent_cartesian"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j74_ab_g16.txt 2>&1
cat gpurun_out/r4j74_ab_g16.txt
