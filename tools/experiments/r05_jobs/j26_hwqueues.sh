# round 5, job 26: is the cliff at four streams a hardware-queue limit?  (GPU_MAX_HW_QUEUES)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  for cfg in "--workload C5" "--workload C2" "--workload C3"; do
    echo "--- GPU_MAX_HW_QUEUES=$q $cfg"
    AB_ENVS="--streams=3 --streams=4" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
  done
done > gpurun_out/r5j26_hwqueues.txt 2>&1
cat gpurun_out/r5j26_hwqueues.txt
