# experiment: small batches (one workgroup per CU or fewer) on the 1 024-thread instantiation
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--workload C2" "--workload C3 --points 256" "--workload C5 --points 256"; do
  echo "--- $cfg"
  AB_ENVS="GLH_PT_FORCE_BIG=1" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r4j36_force_big.txt 2>&1
cat gpurun_out/r4j36_force_big.txt
