# blocks of points on 2 / 3 / 4 streams
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "" "--workload C4" "--workload C5 --points 2048"; do
  echo "--- $cfg"
  AB_ENVS="--streams=3 --streams=4" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r4j73_streams.txt 2>&1
cat gpurun_out/r4j73_streams.txt
