// One translation unit per curve configuration: explicit instantiation of every curve-templated kernel
// (hipcc -DMSM_CURVE_TU=CvBls377 | CvBls381 | CvPallas).  See kernel_inst.h.
#include <hip/hip_runtime.h>
#ifndef MSM_CURVE_TU
#error "compile with -DMSM_CURVE_TU=<curve configuration>"
#endif
#include "kernel_inst.h"
