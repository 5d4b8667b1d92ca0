// Shared helpers of the C-ABI translation units (error state, device check, params widening).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/sfb.h"
#include "qp_dense_kernel.h"

namespace sfb {

sfb_status fail(sfb_status st, const std::string &msg);
sfb_status hip_fail(hipError_t e, const char *what);
sfb_status require_device();
DenseKernelParams make_kernel_params(const sfb_qp_params *prm, int n, int m);
// QPSolverParams::verbose on the host-pointer entry points (qp_solver.hpp:409-420, :550-565 print a per-phase time
// breakdown and the outcome): one summary of the call -- phase times, status histogram, iteration statistics.
void verbose_report(const char *what, int64_t batch, int n, int m, double h2d_ms, double solve_ms, double d2h_ms,
                    const int32_t *code, const uint32_t *iter);
// ... and, for ONE problem, the per-iteration table itself in the reference's format (:409-420, :490-501): trace = rows x 5
// (ITER, OBJ, PRI_RES, DUA_RES, TIME us; ITER < 0 ends the table), as written by sfb_sparse_qp_solve_batch_trace.
void verbose_table(const char *kind, int n, int m, const double *trace, int rows, const char *note = nullptr);
// ... and its closing summary (:550-565) from the six per-phase times of the *_phases entry points
void verbose_summary(int32_t code, uint32_t iter, const double *phase_us);
// rows a table needs for these parameters (capped)
int verbose_table_rows(const sfb_qp_params *prm);
// Multi-device entry points (sfb_*_multi): the device list of sfb_set_devices (default: every visible device), and a
// helper that cuts [0, batch) into one contiguous shard per list entry and runs `fn(device, first, count)` for every
// non-empty shard on its own host thread with that device current.  Returns the first failure (its message becomes
// this thread's sfb_last_error).
std::vector<int> device_list();
sfb_status run_sharded(int64_t batch, const std::function<sfb_status(int device, int64_t first, int64_t count)> &fn);
struct SparsePlanHost;
const SparsePlanHost &plan_host(const sfb_sparse_qp_plan *plan);  // capi_sparse.hip: the pattern the kernel works on
const SparsePlanHost &plan_io(const sfb_sparse_qp_plan *plan);    // the caller's pattern (== plan_host unless pruned)

}  // namespace sfb
