// Shared helpers of the C-ABI translation units (error state, device check, params widening).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/sfb.h"
#include "qp_dense_kernel.h"

namespace sfb {

sfb_status fail(sfb_status st, const std::string &msg);
sfb_status hip_fail(hipError_t e, const char *what);
sfb_status require_device();
DenseKernelParams make_kernel_params(const sfb_qp_params *prm, int n, int m);
struct SparsePlanHost;
const SparsePlanHost &plan_host(const sfb_sparse_qp_plan *plan);  // capi_sparse.hip: the pattern the kernel works on
const SparsePlanHost &plan_io(const sfb_sparse_qp_plan *plan);    // the caller's pattern (== plan_host unless pruned)

}  // namespace sfb
