# DEM and DEM uncertainty on one grid: the cell and the weights once for both samples (prev.so = HEAD)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests -m gpu -x -q -k "raster or dem or surface or tangent or cylindrical or viewshed or motion or api" > gpurun_out/r4j55_tests.log 2>&1
tail -2 gpurun_out/r4j55_tests.log
for cfg in "--dem gridded" "--workload C5 --points 2048 --dem gridded" "--motion tangent_cartesian --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j55_ab_pair.txt 2>&1
cat gpurun_out/r4j55_ab_pair.txt
