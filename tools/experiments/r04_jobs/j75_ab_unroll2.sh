# the general code's particle loop unrolled by two (prev.so = HEAD)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--motion tangent_cartesian" "--bits 16" "--bits 32"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j75_ab_unroll2.txt 2>&1
cat gpurun_out/r4j75_ab_unroll2.txt
