# the raster instantiations (one observer) with phase A's loop unrolled too (prev.so = HEAD)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--motion tangent_cartesian --dem gridded" "--dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j78_ab_unroll_grid.txt 2>&1
cat gpurun_out/r4j78_ab_unroll_grid.txt
