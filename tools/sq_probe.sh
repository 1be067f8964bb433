#!/bin/bash
# SQ counters of the fused kernel for one or more builds (lib names under glimpse_amd/lib/), steady-state C3.
# usage: tools/sq_probe.sh name1.so name2.so ...   -> gpurun_out/sq_<name>/s_counter_collection.csv
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
for so in "$@"; do
  tag=${so%.so}
  export GLH_LIB=$GRAFT_REPO_ROOT/glimpse_amd/lib/$so
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU \
    -d gpurun_out/sq_$tag -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-api --burn-in 6 --steps 4 --warmup 2 > gpurun_out/sq_$tag.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA \
    -d gpurun_out/sq2_$tag -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-api --burn-in 6 --steps 4 --warmup 2 > gpurun_out/sq2_$tag.log 2>&1
done
