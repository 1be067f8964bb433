# parked heights for the tangent models over CONSTANT surfaces too, now that the general gather keeps one record in flight
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--motion tangent_cartesian" "--bits 16" "--motion tangent_cylindrical"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j64_ab_zpark1.txt 2>&1
cat gpurun_out/r4j64_ab_zpark1.txt
