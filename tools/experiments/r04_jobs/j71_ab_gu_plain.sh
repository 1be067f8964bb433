# the plain code's gather with one / three records in flight against two (gu1.so, gu3.so)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--workload C5 --points 2048" "" "--workload C4" "--workload C5 --points 512"; do
  echo "--- $cfg"
  AB_ENVS="gu1.so gu3.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j71_ab_gu_plain.txt 2>&1
cat gpurun_out/r4j71_ab_gu_plain.txt
