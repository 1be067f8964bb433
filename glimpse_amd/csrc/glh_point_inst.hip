// glh_point_inst.hip -- ONE instantiation of the fused frame step per translation unit (glh_point_variants.h):
//   hipcc -c -DPT_TB=512 -DPT_PPT=10 -DPT_NOBS=1 -DPT_SURF=0 -DPT_FAST=1 -DPT_CON=1 glh_point_inst.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#define GLH_POINT_TU 1
#include "glh_point.h"
#include "glh_point_variants.h"

#if !defined(PT_TB) || !defined(PT_PPT) || !defined(PT_NOBS) || !defined(PT_SURF) || !defined(PT_FAST) || !defined(PT_CON)
#error "define PT_TB, PT_PPT, PT_NOBS, PT_SURF, PT_FAST, PT_CON"
#endif

#ifndef PT_MINW
#define PT_MINW 4  // waves per SIMD the register allocation aims at (128 VGPRs; 6 -- an experiment -- would be 80)
#endif

#define GLH_PT_NAME_X(TB, PPT, NOBS, S, F, C) GLH_PT_NAME(TB, PPT, NOBS, S, F, C)

namespace glh {
const void* GLH_PT_NAME_X(PT_TB, PT_PPT, PT_NOBS, PT_SURF, PT_FAST, PT_CON)() {
  return (const void*)k_point_step<PT_TB, PT_PPT, PT_MINW, PT_NOBS, PT_SURF, (bool)PT_FAST, (bool)PT_CON>;
}
}  // namespace glh
