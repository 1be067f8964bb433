# round 5, job 18: the two barriers at the end of phase A wait for LDS only (the template loads stay in flight)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
for cfg in "--workload C3" "--workload C3 --streams 1" "--workload C4" "--workload C5 --points 2048" "--workload C5" "--workload C2"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j18_ab_lds_barriers.txt 2>&1
cat gpurun_out/r5j18_ab_lds_barriers.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_pinned.py tests/test_gpu_streams.py -x -q -m gpu 2>&1 | tail -3
