# windows of the dem / dem_sigma rasters in LDS (prev.so = the commit before: samples from memory by a guessed interval)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests -m gpu -x -q -k "raster or dem or surface or tangent or cylindrical or viewshed or motion or api" > gpurun_out/r4j44_tests.log 2>&1
tail -3 gpurun_out/r4j44_tests.log
for cfg in "--motion tangent_cartesian --dem gridded" "--dem gridded" "--workload C5 --points 2048 --dem gridded" "--motion tangent_cylindrical --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j44_ab_patch.txt 2>&1
cat gpurun_out/r4j44_ab_patch.txt
