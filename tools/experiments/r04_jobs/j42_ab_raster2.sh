# raster interval by guess (uniform coordinates), phase A / E in two copies: with and without the raster samples (prev.so = HEAD)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests -m gpu -x -q -k "raster or dem or surface or tangent or cylindrical or viewshed or motion" > gpurun_out/r4j42_tests.log 2>&1
tail -3 gpurun_out/r4j42_tests.log
for cfg in "--motion tangent_cartesian" "--motion tangent_cartesian --dem gridded" "--dem gridded" "--workload C5 --points 2048 --dem gridded" "" "--bits 16"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j42_ab_raster.txt 2>&1
cat gpurun_out/r4j42_ab_raster.txt
