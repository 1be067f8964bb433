#!/bin/bash
# Registers / scratch / code size of ONE instantiation of the fused kernel in seconds (the full library takes 1.5 min):
# a scratch translation unit that includes the kernel headers and instantiates only the requested variant.
#   tools/regcheck.sh [TB PPT NOBS SURF FAST CONTRACT]      (default: 512 10 1 false true true = C3; SURF: false / true / 2)
#   REGCHECK_MINW: waves per SIMD of __launch_bounds__ (default 4); REGCHECK_FLAGS: extra compiler flags (e.g. "-mllvm -disable-machine-licm")
TB=${1:-512}; PPT=${2:-10}; NOBS=${3:-1}; SURF=${4:-false}; FAST=${5:-true}; CON=${6:-$FAST}
case $SURF in false) SURF=0 ;; true) SURF=1 ;; esac   # (the surface code: 0 plain, 1 general, 2 general with rasters)
OUT=${REGCHECK_OUT:-/tmp/regcheck}
mkdir -p $OUT
cat > $OUT/one.hip <<EOT
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>
#define GLH_POINT_TU 1
#include "$(cd "$(dirname "$0")/.." && pwd)/glimpse_amd/csrc/glh_point.h"
template __global__ void glh::k_point_step<$TB, $PPT, ${REGCHECK_MINW:-4}, $NOBS, $SURF, $FAST, $CON>(glh::PointArgs);
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function \
  $REGCHECK_FLAGS -I"$(cd "$(dirname "$0")/.." && pwd)/include" --cuda-device-only -S -o $OUT/one.s $OUT/one.hip 2>&1 | grep -i "error" -A4
grep -E "^; (NumVgprs|ScratchSize|codeLenInByte|NumSgprs|Occupancy)" $OUT/one.s | tr '\n' ' '; echo
