# the general code WITHOUT raster samples with phase A's particle loop fully unrolled, like the plain code's (prev.so = HEAD)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4j76_tests.log 2>&1; tail -2 gpurun_out/r4j76_tests.log
for cfg in "--motion tangent_cartesian" "--bits 16" "--bits 32" "--motion tangent_cartesian --math exact" "--motion tangent_cartesian --workload C4" "--motion tangent_cartesian --workload C5 --points 2048"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j76_ab_unroll.txt 2>&1
cat gpurun_out/r4j76_ab_unroll.txt
