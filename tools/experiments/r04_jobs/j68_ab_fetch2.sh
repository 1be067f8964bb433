# tile fetch of every frame type with its loads issued back to back (prev.so = HEAD before the fetch changes)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4j68_tests.log 2>&1; tail -2 gpurun_out/r4j68_tests.log
for cfg in "--bits 16" "--bits 32" "--channels 3" "--motion tangent_cartesian"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j68_ab_fetch2.txt 2>&1
cat gpurun_out/r4j68_ab_fetch2.txt
