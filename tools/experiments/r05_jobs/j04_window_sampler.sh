# round 5, job 4: the window-only sampler (raster_sample_window): tests, A/B against round 4's library, phases
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r5j04_tests.txt 2>&1
tail -5 gpurun_out/r5j04_tests.txt
for cfg in "--motion tangent_cartesian --dem gridded" "--workload C5 --points 2048 --dem gridded" "--dem gridded"; do
  echo "--- $cfg"
  bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j04_ab_window.txt 2>&1
cat gpurun_out/r5j04_ab_window.txt
for cfg in "cartesian gridded" "tangent_cartesian gridded"; do
  set -- $cfg
  echo "=== motion $1 dem $2"
  GLH_MOTION=$1 GLH_DEM=$2 python tools/phase_probe.py C3 4096 5000 12 2>&1 | grep "A split\|A evolve\|E gather\|last step"
done > gpurun_out/r5j04_phases.txt 2>&1
cat gpurun_out/r5j04_phases.txt
