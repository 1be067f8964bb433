#!/bin/bash
cd $GRAFT_REPO_ROOT
for lib in libglimpse_hip fp1 fp2 fp3; do
  for bits in 32 64; do
    echo "$lib bits $bits:"; GLH_LIB=$PWD/glimpse_amd/lib/$lib.so python tools/experiments/float_frames_probe.py $bits 1024 1 30 2>&1 | tail -2 | head -1
  done
done | tee gpurun_out/r4j12_float_probe.txt
echo "uint8:"; python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
from glimpse_amd import _lib as lib, workloads
T=30
wl = workloads.Workload("C3", n_frames=T, n_points=1024, n_particles=5000)
with lib.Context(wl.P, wl.N, 1, max_search_dim=160, max_frames=T) as ctx:
    workloads.setup_context(ctx, wl)
    ctx.set_math("fast"); ctx.set_frame(0); ctx.init_particles(seed=3); ctx.init_templates(0,0); ctx.record_moments(0)
    for i in range(1,T):
        if i == T-3: ctx.sync(); t0=time.perf_counter()
        ctx.step(i,1.0,[i],seed=3)
    ctx.sync(); print("uint8 ms/frame", 1e3*(time.perf_counter()-t0)/3)
PY
