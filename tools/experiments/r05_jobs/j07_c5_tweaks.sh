# round 5, job 7: two observers: 16-lane SSD row split (in the tree), two-deep prefetch of observer 1's coordinates (pf2.so);
# the raster windows' origin requested during the prologue; the file-ingest test again
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_fused.py tests/test_gpu_streams.py -x -q -m gpu > gpurun_out/r5j07_tests.txt 2>&1
tail -4 gpurun_out/r5j07_tests.txt
for cfg in "--workload C5 --points 2048" "--workload C5" "--motion tangent_cartesian --dem gridded" "--workload C3"; do
  echo "--- $cfg"
  AB_ENVS="pf2.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j07_ab_c5.txt 2>&1
cat gpurun_out/r5j07_ab_c5.txt
