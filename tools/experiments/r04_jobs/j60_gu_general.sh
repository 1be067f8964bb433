# the general code with one record in flight: tests, and `new` against gu1.so (every instantiation at one) / prev.so
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4j60_tests.log 2>&1; tail -2 gpurun_out/r4j60_tests.log
for cfg in "--motion tangent_cartesian" "--bits 32" "--channels 3" "--workload C4"; do
  echo "--- $cfg"
  AB_ENVS="gu1.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j60_ab_gu.txt 2>&1
cat gpurun_out/r4j60_ab_gu.txt
