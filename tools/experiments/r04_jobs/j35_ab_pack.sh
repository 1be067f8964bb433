# 16-bit / float frames: (count, median count) packed into the tile, mapped by every thread (prev.so = the commit before)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q -k "float or f32 or f64 or 16 or wide or depth or dtype or highpass" > gpurun_out/r4j35_tests.log 2>&1
tail -3 gpurun_out/r4j35_tests.log
for cfg in "--bits 16" "--bits 32"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j35_ab_pack.txt 2>&1
cat gpurun_out/r4j35_ab_pack.txt
for b in 16 32; do GLH_BITS=$b timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j35_phase_b$b.txt 2>&1; grep -n "split\|point_step\|tile_prep " gpurun_out/r4j35_phase_b$b.txt; done
