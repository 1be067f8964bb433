# observer 0's per-thread coordinates as a vector type (indexed register writes in rolled loops); prev.so = the commit before
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--motion tangent_cartesian" "" "--streams 1" "--workload C5 --points 2048" "--bits 16"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j39_ab_dvec.txt 2>&1
cat gpurun_out/r4j39_ab_dvec.txt
