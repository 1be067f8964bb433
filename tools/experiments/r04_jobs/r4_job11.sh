#!/bin/bash
cd $GRAFT_REPO_ROOT
for bits in 32 64; do for mode in 1 0; do
  echo "bits $bits fused $mode:"; python tools/experiments/float_frames_probe.py $bits 1024 $mode 30 2>&1 | tail -2
done; done | tee gpurun_out/r4j11_float_probe.txt
