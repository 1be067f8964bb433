# round 5, job 24: where two streams start to pay for small batches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
for cfg in "--workload C3 --points 320" "--workload C3 --points 384" "--workload C3 --points 448" "--workload C3 --points 768" "--workload C5 --points 384" "--workload C5 --points 768" "--workload C4 --points 256" "--workload C4 --points 384" "--workload C2 --points 512"; do
  echo "--- $cfg"
  AB_ENVS="--streams=2" bash tools/ab.sh --no-secondary --streams 1 $cfg 2>/dev/null
done > gpurun_out/r5j24_streams_threshold.txt 2>&1
cat gpurun_out/r5j24_streams_threshold.txt
