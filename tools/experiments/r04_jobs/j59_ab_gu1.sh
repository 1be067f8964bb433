# general code with ONE record in flight in the gather (no scratch) against two (40 B of scratch): gu1.so
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--motion tangent_cartesian" "--bits 16" "--motion tangent_cartesian --dem gridded" ""; do
  echo "--- $cfg"
  AB_ENVS="gu1.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j59_ab_gu1.txt 2>&1
cat gpurun_out/r4j59_ab_gu1.txt
