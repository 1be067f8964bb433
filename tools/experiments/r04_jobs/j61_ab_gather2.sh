# raster instantiations: no sampler in the gather (prev.so = the commit before)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests -m gpu -x -q -k "raster or dem or surface or tangent or cylindrical or viewshed or motion or api or three_observers" > gpurun_out/r4j61_tests.log 2>&1
tail -2 gpurun_out/r4j61_tests.log
for cfg in "--motion tangent_cartesian --dem gridded" "--dem gridded" "--workload C5 --points 2048 --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j61_ab_gather2.txt 2>&1
cat gpurun_out/r4j61_ab_gather2.txt
