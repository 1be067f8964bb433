# ... and with two observers (prev.so = one-observer raster instantiations unrolled only)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4j79_tests.log 2>&1; tail -2 gpurun_out/r4j79_tests.log
for cfg in "--workload C5 --points 2048 --dem gridded" "--workload C5 --points 2048 --dem gridded --motion tangent_cartesian"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j79_ab_unroll_grid2.txt 2>&1
cat gpurun_out/r4j79_ab_unroll_grid2.txt
