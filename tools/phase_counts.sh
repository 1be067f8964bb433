#!/bin/bash
# Dynamic instruction counts of the fused kernel phase by phase, steady-state C3: the last launch of a short sequence is
# cut at every phase stamp in turn (GLH_PT_STOP=stamp:frame, glh_point.h: PT_STAMP) and its SQ counters are read; the
# differences between successive cuts are the phases.   usage (on the GPU box): tools/phase_counts.sh [bench args]
#   -> gpurun_out/phase_counts_<workload>.json, table on stdout (tools/phase_counts.py); PC_FRAME=n cuts frame n instead of the
#      last one (with --burn-in 0 --steps 4: n = 4, the widest search tiles after the prior)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python3 bench.py --no-cpu-baseline --no-api --no-secondary --burn-in 6 --steps 4 --warmup 2 "$@" > /dev/null 2>&1  # (fills the frame cache)
for k in 0 15 16 17 18 19 1 2 3 4 5 6 10 11 12 7 8 9 full; do
  if [ $k = full ]; then unset GLH_PT_STOP; else export GLH_PT_STOP=$k:${PC_FRAME:-10}; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
    -d gpurun_out/pc_$k -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-api --no-secondary --burn-in 6 --steps 4 --warmup 2 "$@" > gpurun_out/pc_$k.log 2>&1
done
python3 tools/phase_counts.py "$@"
