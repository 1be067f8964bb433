# round 5, job 6: the window sampler with one rare branch per axis and interleaved axis tables: tests, A/B, counts
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r5j06_tests.txt 2>&1
tail -5 gpurun_out/r5j06_tests.txt
for cfg in "--motion tangent_cartesian --dem gridded" "--workload C5 --points 2048 --dem gridded" "--dem gridded"; do
  echo "--- $cfg"
  bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j06_ab_window2.txt 2>&1
cat gpurun_out/r5j06_ab_window2.txt
for cfg in "tangent_cartesian gridded"; do
  set -- $cfg
  args="--no-cpu-baseline --no-api --no-secondary --burn-in 6 --steps 4 --warmup 2 --repeats 1 --motion $1 --dem $2"
  rm -rf gpurun_out/pc_*
  python3 bench.py $args > /dev/null 2>&1
  for k in 16 17 7 8; do
    export GLH_PT_STOP=$k:10
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
      -d gpurun_out/pc_$k -o s --output-format csv -- python3 bench.py $args > gpurun_out/pc_$k.log 2>&1
    unset GLH_PT_STOP
  done
  echo "=== motion $1 dem $2"
  python3 tools/phase_counts.py
done > gpurun_out/r5j06_counts.txt 2>&1
cat gpurun_out/r5j06_counts.txt
rm -rf gpurun_out/pc_*
