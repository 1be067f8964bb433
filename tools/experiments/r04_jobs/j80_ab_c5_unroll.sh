# C5: the last observer's sampling loop unrolled by two (prev.so = HEAD)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--workload C5 --points 2048" "--workload C5 --points 512"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j80_ab_c5_unroll.txt 2>&1
cat gpurun_out/r4j80_ab_c5_unroll.txt
