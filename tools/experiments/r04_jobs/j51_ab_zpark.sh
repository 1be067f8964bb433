# tangent models: the evolved height parked by phase A serves the gather's re-evolution (prev.so = the commit before)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests -m gpu -x -q -k "raster or dem or surface or tangent or cylindrical or viewshed or motion or api or fused" > gpurun_out/r4j51_tests.log 2>&1
tail -2 gpurun_out/r4j51_tests.log
for cfg in "--motion tangent_cartesian --dem gridded" "--motion tangent_cartesian" "--motion tangent_cylindrical --dem gridded" "--dem gridded" "--bits 16"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j51_ab_zpark.txt 2>&1
cat gpurun_out/r4j51_ab_zpark.txt
