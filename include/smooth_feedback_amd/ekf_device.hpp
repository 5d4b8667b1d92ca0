// A swarm of Lie-group EKFs that lives on the GPU (HIP only: include from a translation unit compiled by hipcc).
// The host front (ekf.hpp, EKF<G>) keeps what needs the user's callbacks on the CPU -- the linearisation of f and h at
// the estimate, the state step, g (+) delta -- and sends the covariance algebra to sfb_ekf_*_batch.  For a model whose
// functors are device-callable all of it can stay where the covariances are: one GPU thread per filter runs the very
// same helper functions (ekf.hpp: ekf_linearise_dyn, ekf_linearise_meas, ekf_rk4_state; lie.hpp is __host__
// __device__), writes A / H / r next to the resident P, and the batched covariance kernels follow on the same stream.
// Per call only the measurements go up; estimates and covariances are downloaded when asked for.
//
//   Dyn : Tangent operator()(double t, const G & g) const      body velocity, ekf.hpp:62-63
//   Meas: Vec<Ny> operator()(const G & g) const                measurement in R^Ny, ekf.hpp:116-121
// both with `__host__ __device__` call operators.  predict / update are smooth::feedback::EKF::predict / ::update
// (ekf.hpp:79-103, :116-139) for every filter of the swarm; step() is one predict substep followed by the update with
// the two covariance passes in one launch (sfb_ekf_predict_update_batch).
#pragma once
#ifndef __HIPCC__
#error "ekf_device.hpp needs hipcc"
#endif
#include <hip/hip_runtime.h>

#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "detail/ekf_lane.hpp"
#include "ekf.hpp"

namespace smooth_feedback_amd {

namespace detail {

// linearisation of the dynamics at the estimate and the state step of one (sub)step of length h starting at time t
template<class G, class Dyn, bool RK4>
__global__ void __launch_bounds__(64) ekf_predict_lin_kernel(const int64_t B, const Dyn f, const double t, const double h, G * __restrict__ g,
                                                             double * __restrict__ A, double * __restrict__ Am, double * __restrict__ Ae)
{
  constexpr int N = G::Dof;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const G x = g[b];
  typename G::Tangent fv;
  const Mat<N, N> A0 = ekf_linearise_dyn<G>([&](const G & y) { return f(t, y); }, x, fv);
  for (int e = 0; e < N * N; ++e) A[b * N * N + e] = A0.a[e];
  if constexpr (RK4) {  // the stage matrices of cov_ode (ekf.hpp:84-89) at the frozen estimate
    typename G::Tangent fs;
    const Mat<N, N> A1 = ekf_linearise_dyn<G>([&](const G & y) { return f(t + 0.5 * h, y); }, x, fs);
    const Mat<N, N> A2 = ekf_linearise_dyn<G>([&](const G & y) { return f(t + h, y); }, x, fs);
    for (int e = 0; e < N * N; ++e) { Am[b * N * N + e] = A1.a[e]; Ae[b * N * N + e] = A2.a[e]; }
    g[b] = ekf_rk4_state(f, t, h, x, fv);
  } else {
    for (auto & v : fv) v *= h;
    g[b] = rplus(x, fv);  // ekf.hpp:97
  }
}

template<class G, class Meas, int Ny>
__global__ void __launch_bounds__(64) ekf_update_lin_kernel(const int64_t B, const Meas hfn, const G * __restrict__ g,
                                                            const double * __restrict__ y, double * __restrict__ H, double * __restrict__ r)
{
  constexpr int N = G::Dof;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  Vec<Ny> yb, rb;
  for (int i = 0; i < Ny; ++i) yb[i] = y[b * Ny + i];
  Mat<Ny, N> Hm{};
  ekf_linearise_meas<Ny>(hfn, g[b], yb, Hm, rb);
  for (int e = 0; e < Ny * N; ++e) H[b * Ny * N + e] = Hm.a[e];
  for (int i = 0; i < Ny; ++i) r[b * Ny + i] = rb[i];
}

template<class G>
__global__ void __launch_bounds__(64) ekf_apply_kernel(const int64_t B, G * __restrict__ g, const double * __restrict__ delta)
{
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  typename G::Tangent d;
  for (int i = 0; i < G::Dof; ++i) d[i] = delta[b * G::Dof + i];
  g[b] = rplus(g[b], d);  // ekf.hpp:137
}

// One predict substep (Euler, ekf.hpp:84-97) followed by the update (:116-139) of every filter in ONE launch: a lane
// linearises the dynamics at its estimate, steps its covariance (registers), steps the estimate, linearises the
// measurement at the new estimate, runs the Kalman update and applies g (+) delta -- the covariance crosses HBM once in
// each direction, A / H / r / delta never leave the registers.  Same helper functions and the same per-lane covariance
// arithmetic (detail/ekf_lane.hpp) as the separate launches: identical results.
template<class G, class Dyn, class Meas, int Ny>
__global__ void __launch_bounds__(64) ekf_step_fused_kernel(const int64_t B, const Dyn f, const Meas hfn, const double t, const double h,
                                                            const Mat<G::Dof, G::Dof> Q, const Mat<Ny, Ny> R, G * __restrict__ g,
                                                            const double * __restrict__ y, double * __restrict__ Pg,
                                                            int32_t * __restrict__ info)
{
  namespace L = sfb::ekf_lane;
  constexpr int N = G::Dof, NN = N * N, NP = NN | 1;
  __shared__ double lds[NP * L::kLanes];
  const int lane      = threadIdx.x;
  const int64_t item0 = (int64_t)blockIdx.x * L::kLanes, b = item0 + lane;
  const bool live     = b < B;
  double P[NN];
  L::tile_load<NN>(Pg, item0, B, lds, lane);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < NN; ++e) P[e] = live ? lds[lane * NP + e] : 0.0;
  __syncthreads();
  G x = live ? g[b] : G::Identity();
  bool ok = true;
  {
    typename G::Tangent fv;
    const Mat<N, N> A0 = ekf_linearise_dyn<G>([&](const G & xx) { return f(t, xx); }, x, fv);
    double A[NN];
#pragma unroll
    for (int e = 0; e < NN; ++e) A[e] = A0.a[e];
    L::ekf_lane_predict<N>(P, A, [&](const int i, const int j) { return Q.a[i + j * N]; }, live ? h : 0.0);
    for (auto & v : fv) v *= h;
    x = rplus(x, fv);  // ekf.hpp:97
  }
  {
    Vec<Ny> yb, rb;
    for (int i = 0; i < Ny; ++i) yb[i] = live ? y[b * Ny + i] : 0.0;
    Mat<Ny, N> Hm{};
    ekf_linearise_meas<Ny>(hfn, x, yb, Hm, rb);
    double H[Ny * N], rv[Ny], delta[N];
#pragma unroll
    for (int e = 0; e < Ny * N; ++e) H[e] = Hm.a[e];
#pragma unroll
    for (int i = 0; i < Ny; ++i) rv[i] = rb[i];
    ok = L::ekf_lane_update<N, Ny>(P, H, [&](const int a, const int c) { return (a <= c) ? R.a[a + c * Ny] : 0.0; }, rv, delta);
    typename G::Tangent d;
    for (int i = 0; i < N; ++i) d[i] = delta[i];
    x = rplus(x, d);  // ekf.hpp:137
  }
  if (live) {
    g[b]    = x;
    info[b] = ok ? 0 : 1;
#pragma unroll
    for (int e = 0; e < NN; ++e) lds[lane * NP + e] = P[e];
  }
  __syncthreads();
  L::tile_store<NN>(Pg, item0, B, lds, lane);
}

inline void ekf_hip_check(hipError_t e, const char * what)
{
  if (e != hipSuccess) throw std::runtime_error(std::string("ekf_device: ") + what + ": " + hipGetErrorString(e));
}
}  // namespace detail

template<class G, class Dyn, class Meas, int Ny, EKFStepper Stp = EKFStepper::Euler>
class EKFSwarmDevice {
public:
  static constexpr int N = G::Dof;
  using CovT = Mat<N, N>;

  EKFSwarmDevice(Dyn f, Meas h, int64_t filters) : f_(f), h_(h), B_(filters)
  {
    if (B_ < 1) throw std::invalid_argument("EKFSwarmDevice: at least one filter");
    const size_t B = (size_t)B_, nn = (size_t)N * N;
    const size_t doubles = B * (nn * (Stp == EKFStepper::RK4 ? 4 : 2) + (size_t)Ny * N + 2 * (size_t)Ny + N) + nn + (size_t)Ny * Ny + 2;
    detail::ekf_hip_check(hipMalloc(reinterpret_cast<void **>(&mem_), doubles * 8 + B * (sizeof(G) + 4) + 64), "hipMalloc");
    double * d = mem_;
    P_ = d; d += B * nn;  A_ = d; d += B * nn;
    if constexpr (Stp == EKFStepper::RK4) { Am_ = d; d += B * nn; Ae_ = d; d += B * nn; }
    H_ = d; d += B * Ny * N;  y_ = d; d += B * Ny;  r_ = d; d += B * Ny;  delta_ = d; d += B * N;
    Q_ = d; d += nn;  R_ = d; d += Ny * Ny;  dt_ = d; d += 2;
    g_    = reinterpret_cast<G *>(d);
    info_ = reinterpret_cast<int32_t *>(g_ + B);
    std::vector<G> g0(B, G::Identity());  // EKF's defaults: identity estimate, identity covariance (ekf.hpp:141-144)
    std::vector<CovT> P0(B, CovT::Identity());
    reset(g0, P0);
  }
  EKFSwarmDevice(const EKFSwarmDevice &)             = delete;
  EKFSwarmDevice & operator=(const EKFSwarmDevice &) = delete;
  ~EKFSwarmDevice()
  {
    if (mem_) (void)hipFree(mem_);
  }

  int64_t size() const { return B_; }

  /// ekf.hpp:52-56 for every filter
  void reset(const std::vector<G> & g, const std::vector<CovT> & P)
  {
    if ((int64_t)g.size() != B_ || (int64_t)P.size() != B_) throw std::invalid_argument("EKFSwarmDevice: one state and covariance per filter");
    reset(g.data(), P.data());
  }
  /// ekf.hpp:61 / :66 (device -> host)
  std::vector<G> estimates() const
  {
    std::vector<G> out((size_t)B_);
    estimates(out.data());
    return out;
  }
  std::vector<CovT> covariances() const
  {
    std::vector<CovT> out((size_t)B_);
    covariances(out.data());
    return out;
  }
  /// 1 where the LDL' of the innovation covariance failed in the last update (the reference does not check)
  std::vector<int32_t> update_info() const
  {
    std::vector<int32_t> out((size_t)B_);
    update_info(out.data());
    return out;
  }
  /// the same on arrays of size() entries (a shard of a larger swarm: multi_device.hpp)
  void reset(const G * g, const CovT * P)
  {
    detail::ekf_hip_check(hipMemcpy(g_, g, (size_t)B_ * sizeof(G), hipMemcpyHostToDevice), "hipMemcpy(states)");
    detail::ekf_hip_check(hipMemcpy(P_, P, (size_t)B_ * sizeof(CovT), hipMemcpyHostToDevice), "hipMemcpy(covariances)");
  }
  void estimates(G * out) const { detail::ekf_hip_check(hipMemcpy(out, g_, (size_t)B_ * sizeof(G), hipMemcpyDeviceToHost), "hipMemcpy(states)"); }
  void covariances(CovT * out) const
  {
    detail::ekf_hip_check(hipMemcpy(out, P_, (size_t)B_ * sizeof(CovT), hipMemcpyDeviceToHost), "hipMemcpy(covariances)");
  }
  void update_info(int32_t * out) const { detail::ekf_hip_check(hipMemcpy(out, info_, (size_t)B_ * 4, hipMemcpyDeviceToHost), "hipMemcpy(info)"); }
  void upload_measurements(const Vec<Ny> * y)
  {
    detail::ekf_hip_check(hipMemcpy(y_, y, (size_t)B_ * sizeof(Vec<Ny>), hipMemcpyHostToDevice), "hipMemcpy(measurements)");
  }
  /// resident buffers, for callers that produce measurements or consume estimates on the device
  G * device_estimates() { return g_; }
  double * device_covariances() { return P_; }
  double * device_measurements() { return y_; }

  /// ekf.hpp:79-103 for every filter: propagate by tau in substeps of at most dt (default: one step), re-linearising
  /// before each substep, covariance first
  void predict(const CovT & Q, double tau, std::optional<double> dt = {})
  {
    upload_small(Q_, Q.a.data(), (size_t)N * N);
    double t          = 0;
    const double dt_v = dt.value_or(2 * tau);
    while (t + dt_v < tau) {
      substep(t, dt_v);
      t += dt_v;
    }
    substep(t, tau - t);
  }

  /// ekf.hpp:116-139 for every filter; y: one measurement per filter
  void update(const std::vector<Vec<Ny>> & y, const Mat<Ny, Ny> & R)
  {
    upload_measurements(y);
    update_resident(R);
  }
  /// the same with the measurements already in device_measurements()
  void update_resident(const Mat<Ny, Ny> & R)
  {
    upload_small(R_, R.a.data(), (size_t)Ny * Ny);
    linearise_meas();
    sfb_check_(sfb_ekf_update_batch(B_, N, Ny, H_, R_, 1, r_, P_, delta_, info_, nullptr));
    apply();
  }

  /// step(): everything of a round in ONE launch (default), or linearisation / covariance / state launches one after
  /// the other on resident buffers (the path predict() and update() take); same results
  void one_launch(bool on) { one_launch_ = on; }
  /// predict(Q, tau) with ONE substep followed by update(y, R): same results, the covariance is read and written once
  void step(const CovT & Q, double tau, const std::vector<Vec<Ny>> & y, const Mat<Ny, Ny> & R)
  {
    upload_measurements(y);
    step_resident(Q, tau, R);
  }
  void step_resident(const CovT & Q, double tau, const Mat<Ny, Ny> & R)
  {
    if constexpr (Stp != EKFStepper::Euler) {
      predict(Q, tau);
      update_resident(R);
    } else if (one_launch_) {
      hipLaunchKernelGGL((detail::ekf_step_fused_kernel<G, Dyn, Meas, Ny>), grid(), dim3(64), 0, nullptr, B_, f_, h_, 0.0, tau, Q, R, g_, y_, P_,
                         info_);
      detail::ekf_hip_check(hipGetLastError(), "ekf_step_fused_kernel");
    } else {
      upload_small(Q_, Q.a.data(), (size_t)N * N);
      upload_small(R_, R.a.data(), (size_t)Ny * Ny);
      upload_small(dt_, &tau, 1);
      linearise_dyn(0.0, tau);
      linearise_meas();  // at the predicted estimate, as update() after predict()
      sfb_check_(sfb_ekf_predict_update_batch(B_, N, Ny, A_, Q_, 1, dt_, 1, H_, R_, 1, r_, P_, delta_, info_, nullptr));
      apply();
    }
  }

private:
  static void sfb_check_(sfb_status st) { detail::ekf_check(st); }
  void upload_small(double * dst, const double * src, size_t n)
  {
    detail::ekf_hip_check(hipMemcpy(dst, src, n * 8, hipMemcpyHostToDevice), "hipMemcpy");
  }
  void upload_measurements(const std::vector<Vec<Ny>> & y)
  {
    if ((int64_t)y.size() != B_) throw std::invalid_argument("EKFSwarmDevice: one measurement per filter");
    upload_measurements(y.data());
  }
  dim3 grid() const { return dim3((unsigned)((B_ + 63) / 64)); }
  void linearise_dyn(double t, double h)
  {
    hipLaunchKernelGGL((detail::ekf_predict_lin_kernel<G, Dyn, Stp == EKFStepper::RK4>), grid(), dim3(64), 0, nullptr, B_, f_, t, h, g_, A_, Am_,
                       Ae_);
    detail::ekf_hip_check(hipGetLastError(), "ekf_predict_lin_kernel");
  }
  void linearise_meas()
  {
    hipLaunchKernelGGL((detail::ekf_update_lin_kernel<G, Meas, Ny>), grid(), dim3(64), 0, nullptr, B_, h_, g_, y_, H_, r_);
    detail::ekf_hip_check(hipGetLastError(), "ekf_update_lin_kernel");
  }
  void apply()
  {
    hipLaunchKernelGGL((detail::ekf_apply_kernel<G>), grid(), dim3(64), 0, nullptr, B_, g_, delta_);
    detail::ekf_hip_check(hipGetLastError(), "ekf_apply_kernel");
  }
  void substep(double t, double h)
  {
    upload_small(dt_, &h, 1);
    linearise_dyn(t, h);  // A (and the stage matrices) at the estimate BEFORE its step; the covariance step uses them (:94-96)
    if constexpr (Stp == EKFStepper::Euler)
      sfb_check_(sfb_ekf_predict_batch(B_, N, A_, Q_, 1, dt_, 1, P_, nullptr));
    else
      sfb_check_(sfb_ekf_predict_rk4_batch(B_, N, A_, Am_, Ae_, Q_, 1, dt_, 1, P_, nullptr));
  }

  Dyn f_;
  Meas h_;
  int64_t B_;
  bool one_launch_ = true;
  double * mem_ = nullptr;
  double *P_ = nullptr, *A_ = nullptr, *Am_ = nullptr, *Ae_ = nullptr, *H_ = nullptr, *y_ = nullptr, *r_ = nullptr, *delta_ = nullptr;
  double *Q_ = nullptr, *R_ = nullptr, *dt_ = nullptr;
  G * g_          = nullptr;
  int32_t * info_ = nullptr;
};

}  // namespace smooth_feedback_amd
