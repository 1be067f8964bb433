# round 5, job 14: the windows' origins and cell sizes in scalar registers (RasterWin)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
for cfg in "--motion tangent_cartesian --dem gridded" "--dem gridded" "--workload C5 --points 2048 --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j14_ab_win_sgpr.txt 2>&1
cat gpurun_out/r5j14_ab_win_sgpr.txt
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py -x -q -m gpu -k "raster or tangent or gridded or motion" 2>&1 | tail -3
