#!/bin/bash
# more SQ counter passes of the fused kernel (steady-state C3): instruction cache, VALU mix, memory latency levels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
so=${1:-libglimpse_hip.so}
tag=${so%.so}
export GLH_LIB=$GRAFT_REPO_ROOT/glimpse_amd/lib/$so
B="python3 bench.py --no-cpu-baseline --no-api --burn-in 6 --steps 4 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d gpurun_out/sq3_$tag -o s --output-format csv -- $B > gpurun_out/sq3_$tag.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d gpurun_out/sq4_$tag -o s --output-format csv -- $B > gpurun_out/sq4_$tag.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d gpurun_out/sq5_$tag -o s --output-format csv -- $B > gpurun_out/sq5_$tag.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE -d gpurun_out/sq6_$tag -o s --output-format csv -- $B > gpurun_out/sq6_$tag.log 2>&1
