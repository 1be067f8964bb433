// glh_point_variants.h -- the instantiations of the fused frame step (glh_point.h: k_point_step) the library carries.
// Every instantiation is its own translation unit (glh_point_inst.hip compiled once per entry with -DPT_*: the build
// runs them in parallel), exporting the kernel's host handle through pt_kernel_<TB>_<PPT>_<NOBS>_<SURF><FAST><CON>();
// glimpse_hip.hip looks the handle up by shape and code and launches it with hipLaunchKernel.
// glimpse_amd/build.py reads the two lists below (keep the X(...) entries on these lines).
#pragma once

// shapes: threads per workgroup, particles of observer 0 kept in registers per thread, observers
#define GLH_PT_SHAPES(X) X(512, 0, 1) X(512, 0, 2) X(512, 0, 3) X(512, 0, 4) X(512, 4, 1) X(512, 10, 1) X(512, 10, 2) X(1024, 0, 1) X(1024, 0, 2) X(1024, 0, 3) X(1024, 0, 4) X(1024, 10, 1) X(1024, 10, 2)
// codes: SURF (0 the plain code; 1 the general code: every motion model, every frame type, constant surfaces; 2 the
// general code with the context's rasters: gridded dem / dem_sigma, viewshed), FAST (fast arithmetic), CON (compile-time contract)
#define GLH_PT_CODES(X, TB, PPT, NOBS) \
  X(TB, PPT, NOBS, 0, 0, 0) X(TB, PPT, NOBS, 1, 0, 0) X(TB, PPT, NOBS, 0, 1, 1) X(TB, PPT, NOBS, 1, 1, 0) X(TB, PPT, NOBS, 1, 1, 1) \
  X(TB, PPT, NOBS, 2, 0, 0) X(TB, PPT, NOBS, 2, 1, 0) X(TB, PPT, NOBS, 2, 1, 1)

#define GLH_PT_NAME(TB, PPT, NOBS, S, F, C) pt_kernel_##TB##_##PPT##_##NOBS##_##S##F##C
