# round 5, job 31: the tangent step's two surface samples at once (the second usually in the first one's cell)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py -x -q -m gpu -k "raster or tangent or gridded or motion" 2>&1 | tail -3
for cfg in "--motion tangent_cartesian --dem gridded" "--motion tangent_cylindrical --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j31_ab_sample2.txt 2>&1
cat gpurun_out/r5j31_ab_sample2.txt
