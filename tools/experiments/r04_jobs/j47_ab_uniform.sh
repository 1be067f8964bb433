# raster windows: the wave takes the window when all its samples can (uniform branch), branch-free interval step; prev.so = the commit before
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests -m gpu -x -q -k "raster or dem or surface or tangent or cylindrical or viewshed or motion or api" > gpurun_out/r4j47_tests.log 2>&1
tail -2 gpurun_out/r4j47_tests.log
for cfg in "--motion tangent_cartesian --dem gridded" "--dem gridded" "--workload C5 --points 2048 --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j47_ab_uniform.txt 2>&1
cat gpurun_out/r4j47_ab_uniform.txt
