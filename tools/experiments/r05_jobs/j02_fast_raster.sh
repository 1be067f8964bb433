# round 5, job 2: surface samples and the tangent models' step in fast arithmetic (raster_bilinear_fast): tests, then A/B
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5j02_tests.txt 2>&1
tail -5 gpurun_out/r5j02_tests.txt
for cfg in "--motion tangent_cartesian --dem gridded" "--workload C5 --points 2048 --dem gridded" "--dem gridded" "--motion tangent_cartesian"; do
  echo "--- $cfg"
  bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j02_ab_fast_raster.txt 2>&1
cat gpurun_out/r5j02_ab_fast_raster.txt
