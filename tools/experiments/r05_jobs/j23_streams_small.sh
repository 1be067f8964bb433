# round 5, job 23: one-round batches on 1 .. 4 streams (a launch waits for its slowest point: smaller groups, smaller maxima?)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
for cfg in "--workload C5" "--workload C2" "--workload C3 --points 512"; do
  echo "--- $cfg"
  AB_ENVS="--streams=2 --streams=3 --streams=4" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j23_streams_small.txt 2>&1
cat gpurun_out/r5j23_streams_small.txt
