# round 5, job 5: (1) the recompute variant of the two-observer code (rec.so, -DGLH_PT_RECOMP=1): tests, A/B at C5;
# (2) dynamic instruction counts of phase A's loop and of the gather over rasters (cuts 16 / 17 / 7 / 8)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
GLH_LIB=$PWD/glimpse_amd/lib/rec.so timeout 1200 python -m pytest tests/test_gpu_pinned.py tests/test_gpu_fused.py "tests/test_gpu_fullsize.py::test_full_size_properties" -x -q -m gpu -k "not C3 and not C4 and not C2" > gpurun_out/r5j05_tests_rec.txt 2>&1
tail -5 gpurun_out/r5j05_tests_rec.txt
for cfg in "--workload C5 --points 2048" "--workload C5" "--workload C5 --points 2048 --streams 1"; do
  echo "--- $cfg"
  AB_ENVS="rec.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j05_ab_recomp.txt 2>&1
cat gpurun_out/r5j05_ab_recomp.txt
for cfg in "tangent_cartesian constant" "tangent_cartesian gridded" "cartesian gridded"; do
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
done > gpurun_out/r5j05_counts.txt 2>&1
cat gpurun_out/r5j05_counts.txt
rm -rf gpurun_out/pc_*
