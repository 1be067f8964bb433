# fast arithmetic: bilinear weights of a raster sample by reciprocal + Newton (prev.so = IEEE divisions)
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4j45_tests.log 2>&1
tail -3 gpurun_out/r4j45_tests.log
for cfg in "--motion tangent_cartesian --dem gridded" "--dem gridded" "--workload C5 --points 2048 --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="prev.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j45_ab_fastdiv.txt 2>&1
cat gpurun_out/r4j45_ab_fastdiv.txt
