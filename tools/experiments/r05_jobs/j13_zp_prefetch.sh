# round 5, job 13: the gather's parked height requested one iteration early (tangent models over rasters)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
for cfg in "--motion tangent_cartesian --dem gridded" "--motion tangent_cylindrical --dem gridded" "--workload C5 --points 2048 --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j13_ab_zp_prefetch.txt 2>&1
cat gpurun_out/r5j13_ab_zp_prefetch.txt
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "raster or tangent" 2>&1 | tail -3
