# round 5, job 29: the spline fit by explicit inverses up to 40 x 40 (was: both inverses in 512 doubles, 16 x 16)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
( timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r5j29_tests.txt 2>&1; tail -4 gpurun_out/r5j29_tests.txt
for cfg in "--workload C2" "--workload C5" "--workload C3" "--workload C5 --points 2048" "--workload C4" "--workload C3 --streams 1"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j29_ab_dense_fit.txt 2>&1
cat gpurun_out/r5j29_ab_dense_fit.txt
python tools/experiments/slow_points.py C2 256 2000 30 | head -8
