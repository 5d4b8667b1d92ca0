// Host<->kernel interface of the batched EKF covariance kernels.  Internal.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sfb {

struct EkfArgs {
  int64_t batch;
  // predict
  const double *A, *Q, *dt;
  int q_shared, dt_shared;
  // runge_kutta4 predict only: the linearisation at the stage times t + dt/2 and t + dt (nullptr: A at all stages)
  const double *A_mid, *A_end;
  // update
  const double *H, *R, *r;
  int r_shared;
  double *delta;
  int32_t *info;
  // state
  double *P;
};

// largest dof / ny of the generic kernel (one filter per wavefront, matrices in LDS); the register-resident
// one-filter-per-lane kernels cover dof in {2,3,4,6,7} x ny in {1,2,3}, (4,4), (6,6), and the update for dof 8..10
constexpr int kEkfMaxDim = 16;
bool ekf_supported(int dof, int ny, bool update);
hipError_t ekf_launch(const EkfArgs &a, int dof, int ny, bool predict, bool update, hipStream_t stream);
// predict with one runge_kutta4 step instead of one Euler step (uses A, Q, dt, P of the arguments)
hipError_t ekf_rk4_launch(const EkfArgs &a, int dof, hipStream_t stream);

}  // namespace sfb
