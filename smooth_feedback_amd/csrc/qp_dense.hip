// Dense QP path, n + m <= 128: which kernel takes a batch.
//
//   n + m <= 32          qp_dense4.hip     four QPs per wavefront (one per 16-lane row), factor in VGPRs
//   32 < n + m <= 128    qp_dense_mid.hip  one QP per wavefront, registers-only engine up to 64, LDS block engine beyond
//   a time limit (QPSolverParams::max_time, qp_solver.hpp:504-507) on n + m <= 32: the 32 < n + m <= 128 kernel, whose
//   smallest instance serves every n + m <= 48 and compares the wall clock at every stopping check that leaves the status
//   open; the four-per-wave kernel time-slices its slots and carries no per-QP clock.
// Every route is bit-identical to oracle/qp_oracle.c, so the routing never shows in a result.
// (Rounds 1-3 had one-QP-per-wave kernels for n + m <= 64 here -- LDS factor, v_readlane sweeps; superseded by the two
// files above in rounds 2 and 4 and removed in round 5.)
#include <hip/hip_runtime.h>

#include "../../include/sfb.h"
#include "qp_dense_kernel.h"

namespace sfb {

// LDS of the setup / finish kernels of the four-per-wave route (one QP per wavefront there): the packed KKT triangle,
// copies of P and A, the vectors of the scaling / polish / report code in qp_dense_common.h
size_t qp_dense_lds_bytes(int n, int m)
{
  const int k = n + m;
  const size_t doubles = ((size_t)k * (k + 1)) / 2 + (size_t)n * n + (size_t)m * n + 5 * (size_t)n + 8 * (size_t)m + (size_t)k;
  const size_t ints    = (size_t)k + (size_t)m;
  return doubles * sizeof(double) + ((ints * sizeof(int) + 15) / 16) * 16;
}

hipError_t qp_dense_launch(const DenseKernelParams &kp, int64_t batch, const double *P, const double *q,
                           const double *A, const double *l, const double *u, const double *wx, const double *wy,
                           double *x, double *y, double *obj, uint32_t *iter, int32_t *code, hipStream_t stream,
                           void *workspace)
{
  const int k = kp.n + kp.m;
  if (k < 1 || k > kDenseMidMaxK) return hipErrorInvalidValue;
  const QpBatch g{P, q, A, l, u, wx, wy, x, y, obj, iter, code};
  if (k <= 32 && kp.max_time_ns < 0) return qp_dense4_launch(kp, batch, g, stream, workspace);
  return qp_dense_mid_launch(kp, batch, g, stream, workspace);
}

}  // namespace sfb
