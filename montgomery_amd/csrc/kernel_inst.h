// The curve-templated kernels are compiled once per curve in their own translation unit (kernels_curve.hip,
// -DMSM_CURVE_TU=<config>), in parallel; msm_api.hip sees them as extern templates.  X(name, argument types).
#pragma once
#include "msm_kernels.h"
#include "msm_gen_kernels.h"

#define MSM_COMMA ,
#define MSM_CURVE_KERNELS(X, CV)                                                                                          \
  X(msm::k_points_from_wire<CV>, (uint32_t*, const uint32_t*, uint64_t, int, uint32_t*))                                 \
  X(msm::k_table_next<CV>, (uint32_t*, const uint32_t*, uint64_t, int))                                                   \
  X(msm::k_digits<CV>, (uint32_t*, const uint32_t*, uint32_t, int, int, int, int, int, int, uint32_t*, uint32_t, uint32_t*, uint32_t, \
                       uint64_t, uint32_t, uint32_t, uint32_t, uint32_t))                   \
  X(msm::k_batch_add<CV MSM_COMMA msm::MODE_GATHER>, (msm::BatchArgs))                                                   \
  X(msm::k_batch_add<CV MSM_COMMA msm::MODE_REGULAR>, (msm::BatchArgs))                                                  \
  X(msm::k_batch_add<CV MSM_COMMA msm::MODE_SEARCH>, (msm::BatchArgs))                                                   \
  X(msm::k_bucket_finish<CV>, (uint32_t*, const uint4*, uint64_t, const uint32_t*, uint32_t, const uint32_t*))                     \
  X(msm::k_bucket_reduce<CV>, (uint32_t*, uint32_t*, const uint4*, uint64_t, const uint32_t*, const uint32_t*, uint32_t, \
                               uint32_t, uint32_t, uint32_t))                                                            \
  X(msm::k_window_sum<CV>, (uint32_t*, const uint32_t*, uint32_t))                                                       \
  X(msm::k_column_tree<CV>, (uint32_t*, const uint32_t*, uint32_t, uint32_t))                                            \
  X(msm::k_bit_tree<CV>, (uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, int, int, uint32_t))          \
  X(msm::k_test_fp<CV>, (uint32_t*, const uint32_t*, const uint32_t*, uint32_t, int))                                    \
  X(msm::k_test_batch_inverse<CV>, (uint32_t*, const uint32_t*, uint32_t, uint32_t))                                     \
  X(msm::k_test_glv<CV>, (uint32_t*, const uint32_t*, uint32_t))                                                         \
  X(msm::k_test_fp_raw<CV>, (uint32_t*, const uint32_t*, const uint32_t*, uint32_t, int))                                \
  X(msm::k_test_curve_op<CV>, (uint32_t*, const uint32_t*, const uint32_t*, uint32_t, int))                              \
  X(msm_gen::k_gen_points<CV>, (uint32_t*, const uint32_t*, uint64_t, uint64_t))

#define MSM_EXTERN_KERNEL(name, args) extern template __global__ void name args;
#define MSM_DEFINE_KERNEL(name, args) template __global__ void name args;

#ifdef MSM_CURVE_TU
MSM_CURVE_KERNELS(MSM_DEFINE_KERNEL, msm::MSM_CURVE_TU)
#else
MSM_CURVE_KERNELS(MSM_EXTERN_KERNEL, msm::CvBls377)
MSM_CURVE_KERNELS(MSM_EXTERN_KERNEL, msm::CvBls381)
MSM_CURVE_KERNELS(MSM_EXTERN_KERNEL, msm::CvPallas)
#endif
