# round 5, job 17: C5 with observer 1 stored (the tree) and recomputed (rec.so): instruction counts by phase, HBM traffic
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
args="--no-cpu-baseline --no-api --no-secondary --workload C5 --points 2048 --repeats 1"
for lib in libglimpse_hip rec; do
  export GLH_LIB=$PWD/glimpse_amd/lib/$lib.so
  rm -rf gpurun_out/pc_*
  python3 bench.py $args --burn-in 6 --steps 4 --warmup 2 > /dev/null 2>&1
  for k in 17 1 5 7 8 full; do
    if [ $k = full ]; then unset GLH_PT_STOP; else export GLH_PT_STOP=$k:10; fi
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
      -d gpurun_out/pc_$k -o s --output-format csv -- python3 bench.py $args --burn-in 6 --steps 4 --warmup 2 > gpurun_out/pc_$k.log 2>&1
  done
  unset GLH_PT_STOP
  echo "=== C5 2048 x 5000, $lib.so (cuts: 17 end of phase A's loop, 1 end of A, 5 end of both observers' B + C, 7 end of D, 8 end of E)"
  python3 tools/phase_counts.py --workload C5 --points 2048
  mkdir -p gpurun_out/c5_$lib
  python3 bench.py $args > gpurun_out/c5_$lib/bench.json 2> /dev/null
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/c5_$lib/trace -o t --output-format csv -- python3 bench.py $args > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/c5_$lib/fetch -o f --output-format csv -- python3 bench.py $args > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/c5_$lib/write -o w --output-format csv -- python3 bench.py $args > /dev/null 2>&1
done > gpurun_out/r5j17_c5_variants.txt 2>&1
cat gpurun_out/r5j17_c5_variants.txt
rm -rf gpurun_out/pc_*
find gpurun_out/c5_libglimpse_hip gpurun_out/c5_rec -name "*.csv" | head -20
du -sh gpurun_out/c5_*
