// Host<->kernel interface of the dense QP kernel (internal; the public boundary is include/sfb.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sfb {

// QPSolverParams (qp_solver.hpp:29-68) with the float members already widened to double exactly as
// the reference does at :353-356 / :587 / :593 / :605 / :630.
struct DenseKernelParams {
  int n, m;
  double alpha, alpha_comp, rho_bar, sigma;
  double eps_abs, eps_rel, eps_pinf, eps_dinf, delta;
  uint32_t max_iter;  // effective bound (prm.max_iter or SFB_QP_DEVICE_ITER_CAP)
  long long max_time_ns;  // qp_solver.hpp:504-507, < 0 = unset; measured per item on the device clock (see sfb.h)
  uint32_t stop_check_iter;
  uint32_t polish_iter;
  int scaling, polish;
  int reuse;  // sfb_qp_params::reuse_factor (shared-pattern sparse kernel only)
};

// Batch-major device arrays of one call (include/sfb.h: sfb_qp_dense_solve_batch)
struct QpBatch {
  const double *P, *q, *A, *l, *u, *wx, *wy;
  double *x, *y, *obj;
  uint32_t *iter;
  int32_t *code;
};

size_t qp_dense_lds_bytes(int n, int m);
size_t qp_dense4_lds_bytes(int n, int m);

// k = n+m <= 32: four QPs per wavefront, persistent grid with a device-side queue (qp_dense4.hip)
// workspace: qp_dense4_ws_bytes(n, m, batch) bytes of device memory, or nullptr = stream-ordered allocation per launch
hipError_t qp_dense4_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, hipStream_t stream,
                            void *workspace = nullptr);
size_t qp_dense4_ws_bytes(int n, int m, int64_t batch);

// 64 < n+m <= 1024: one QP per wavefront, KKT matrix and pivoted LDL' in the QP's HBM workspace (qp_dense_big.hip);
// workspace: batch * qp_dense_big_ws_doubles(n, m) doubles
size_t qp_dense_big_ws_doubles(int n, int m);
hipError_t qp_dense_big_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, double *workspace, hipStream_t stream);
constexpr int kDenseBigMaxK = 1024;

// 32 < n+m <= 128: one QP per wavefront, everything on chip (qp_dense_mid.hip): packed factor in LDS, block sweeps with
// the pivot broadcast fused into the FP64 FMA, the matrix written in Eigen's pivot order before an unpivoted factorisation
constexpr int kDenseMidMaxK = 128;
size_t qp_dense_mid_lds_bytes(int n, int m);
// workspace: qp_dense_mid_ws_bytes bytes of device memory (0 for batches the chip holds at once), or nullptr = stream-ordered allocation
size_t qp_dense_mid_ws_bytes(const DenseKernelParams &kp, int64_t batch);
hipError_t qp_dense_mid_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, hipStream_t stream, void *workspace = nullptr);
// the same solve through the TRACE instance (one block per QP, any n + m <= 128): a row (ITER, OBJ, PRI_RES, DUA_RES, TIME us) of the
// reference's verbose table per stopping check into trace [batch][trace_cap][5] (device memory, rows preset by the caller)
// phase_us (device, nullable): batch x 6 per-phase microseconds (sfb.h, sfb_qp_dense_solve_batch_phases) FOLLOWED by batch x 10
// doubles of scratch for the kernel's stamps -- the caller allocates batch x 16 doubles
hipError_t qp_dense_mid_trace_launch(const DenseKernelParams &kp, int64_t batch, const QpBatch &g, hipStream_t stream, double *trace, int trace_cap,
                                     double *phase_us = nullptr);

hipError_t qp_dense_launch(const DenseKernelParams &kp, int64_t batch, const double *P, const double *q,
                           const double *A, const double *l, const double *u, const double *wx, const double *wy,
                           double *x, double *y, double *obj, uint32_t *iter, int32_t *code, hipStream_t stream,
                           void *workspace = nullptr);  // (workspace: see qp_dense4_launch; unused by the one-per-wave kernels)

}  // namespace sfb
