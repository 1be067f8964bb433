# same-box A/B of the float32-in-LDS tile stage (prev.so = HEAD before it); interval table as before
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--bits 32" "--bits 16"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j34_ab_float.txt 2>&1
cat gpurun_out/r4j34_ab_float.txt
GLH_BITS=32 timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j34_phase_b32.txt 2>&1; grep -n "split\|point_step\|tile_prep " gpurun_out/r4j34_phase_b32.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4j34_tests.log 2>&1; tail -3 gpurun_out/r4j34_tests.log
