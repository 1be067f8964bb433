# round 5, job 30: inverses of larger surfaces staged in LDS behind the SSD when region 2 has the room
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
( timeout 2400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_pinned.py tests/test_gpu_fullsize.py tests/test_gpu_random_sweep.py -x -q -m gpu ) > gpurun_out/r5j30_tests.txt 2>&1; tail -3 gpurun_out/r5j30_tests.txt
for cfg in "--workload C2" "--workload C5" "--workload C3 --streams 1" "--workload C3"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j30_ab_inv_lds.txt 2>&1
cat gpurun_out/r5j30_ab_inv_lds.txt
python tools/experiments/slow_points.py C2 256 2000 30 | tail -6
