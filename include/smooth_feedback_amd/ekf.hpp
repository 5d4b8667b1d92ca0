// Host side of the EKF path: same interface as smooth::feedback::EKF<G> (reference ekf.hpp:39-149)
// for one filter; ekf_device.hpp (hipcc) runs a swarm of filters with device-callable models entirely on the GPU.  The Lie-group work that needs the user's callbacks
// (linearisation of f and h at the estimate, state propagation, g (+) delta) stays here on the host;
// the covariance algebra runs on the GPU through sfb_ekf_*_batch (include/sfb.h).
//
// Derivatives: analytic if the callable offers jacobian(...), else forward differences with step
// sqrt(eps) (the reference's default without the autodiff header, SURVEY.md section 8 notes).
// Stepper (ekf.hpp:27-31, the Stp template argument): explicit Euler (the reference's default) or
// runge_kutta4 (the one tests/test_ekf.cpp:113-115 instantiates); predict(f, Q, tau, dt) runs ceil(tau/dt)
// substeps and re-linearises before each one, covariance first (ekf.hpp:93-102).  The state is stepped on
// the group with the same tableau (stage states g (+) dt a_ij k_j, result g (+) dt sum b_i k_i).
#pragma once
#include <sfb.h>

#include <cmath>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "lie.hpp"

namespace smooth_feedback_amd {

namespace detail {
inline void ekf_check(sfb_status st)
{
  if (st != SFB_OK) throw std::runtime_error(std::string("sfb: ") + sfb_last_error());
}

// forward-difference step of the host and device fronts: sqrt(DBL_EPSILON)
SFB_LIE_HD constexpr double ekf_fd_step() { return 1.4901161193847656e-08; }

// A = -ad(f(g)) + d^r f/dg  (ekf.hpp:86-87), right-derivative by forward differences
template<class G, class F>
SFB_LIE_HD Mat<G::Dof, G::Dof> ekf_linearise_dyn(F && f, const G & g, typename G::Tangent & fv)
{
  constexpr int N = G::Dof;
  fv = f(g);
  Mat<N, N> dr{};
  const double h = ekf_fd_step();
  for (int c = 0; c < N; ++c) {
    typename G::Tangent e{};
    e[c]          = h;
    const auto f2 = f(rplus(g, e));
    for (int r = 0; r < N; ++r) dr(r, c) = (f2[r] - fv[r]) / h;
  }
  const Mat<N, N> adf = G::ad(fv);
  Mat<N, N> A{};
  for (int i = 0; i < N * N; ++i) A.a[i] = -adf.a[i] + dr.a[i];
  return A;
}

// H = d^r h/dg by forward differences and the innovation r = y - h(g)  (ekf.hpp:119-121, measurements in R^Ny)
template<int Ny, class G, class HF>
SFB_LIE_HD void ekf_linearise_meas(HF && h, const G & g, const Vec<Ny> & y, Mat<Ny, G::Dof> & Hm, Vec<Ny> & r)
{
  constexpr int N    = G::Dof;
  const Vec<Ny> hval = h(g);
  const double eps   = ekf_fd_step();
  for (int c = 0; c < N; ++c) {
    typename G::Tangent e{};
    e[c]          = eps;
    const auto h2 = h(rplus(g, e));
    for (int rr = 0; rr < Ny; ++rr) Hm(rr, c) = (h2[rr] - hval[rr]) / eps;
  }
  for (int i = 0; i < Ny; ++i) r[i] = y[i] - hval[i];
}

// runge_kutta4 on the group: stage states g (+) dt a_ij k_j, result g (+) dt sum b_i k_i; k1 = f(t, g) given
template<class G, class F>
SFB_LIE_HD G ekf_rk4_state(F && f, double t, double h, const G & g, const typename G::Tangent & k1)
{
  auto scaled = [](typename G::Tangent k, double c) { for (auto & v : k) v *= c; return k; };
  const auto k2 = f(t + 0.5 * h, rplus(g, scaled(k1, 0.5 * h)));
  const auto k3 = f(t + 0.5 * h, rplus(g, scaled(k2, 0.5 * h)));
  const auto k4 = f(t + h, rplus(g, scaled(k3, h)));
  typename G::Tangent d{};
  for (int i = 0; i < G::Dof; ++i) d[i] = h * (1.0 / 6.0) * k1[i] + h * (1.0 / 3.0) * k2[i] + h * (1.0 / 3.0) * k3[i] + h * (1.0 / 6.0) * k4[i];
  return rplus(g, d);
}
}  // namespace detail

enum class EKFStepper { Euler = SFB_EKF_EULER, RK4 = SFB_EKF_RK4 };

/// smooth::feedback::EKF<G, DT, Stp>, ekf.hpp:39-149 (measurement space R^Ny)
template<class G, EKFStepper Stp = EKFStepper::Euler>
class EKF {
public:
  static constexpr int N = G::Dof;
  using CovT = Mat<N, N>;

  void reset(const G & g, const CovT & P) { g_hat_ = g; P_ = P; }   // ekf.hpp:52-56
  G estimate() const { return g_hat_; }                               // :61
  CovT covariance() const { return P_; }                              // :66

  /// ekf.hpp:79-103.  f(t, g) -> Tangent, Q: process covariance (upper triangle used)
  template<class F>
  void predict(F && f, const CovT & Q, double tau, std::optional<double> dt = {})
  {
    double t          = 0;
    const double dt_v = dt.value_or(2 * tau);
    auto step = [&](double h) {
      typename G::Tangent fv;
      const CovT A = detail::ekf_linearise_dyn<G>([&](const G & x) { return f(t, x); }, g_hat_, fv);
      // covariance first: it depends on g_hat_ (:94-96)
      if constexpr (Stp == EKFStepper::Euler) {
        detail::ekf_check(sfb_ekf_predict_stepper_batch_host(SFB_EKF_EULER, 1, N, A.a.data(), Q.a.data(), 1, &h, 1, P_.a.data()));
      } else {
        // runge_kutta4 calls cov_ode at t, t + h/2 (twice) and t + h, and cov_ode linearises f(t_stage, .) at the
        // (frozen) estimate each time (:84-89): for dynamics that depend on t the three matrices differ
        typename G::Tangent fs;
        const CovT Am = detail::ekf_linearise_dyn<G>([&](const G & x) { return f(t + 0.5 * h, x); }, g_hat_, fs);
        const CovT Ae = detail::ekf_linearise_dyn<G>([&](const G & x) { return f(t + h, x); }, g_hat_, fs);
        detail::ekf_check(sfb_ekf_predict_rk4_batch_host(1, N, A.a.data(), Am.a.data(), Ae.a.data(), Q.a.data(), 1, &h, 1,
                                                         P_.a.data()));
      }
      if constexpr (Stp == EKFStepper::Euler) {
        for (auto & v : fv) v *= h;
        g_hat_ = rplus(g_hat_, fv);  // euler on the group: g <- g (+) dt f (:97)
      } else {
        g_hat_ = detail::ekf_rk4_state(f, t, h, g_hat_, fv);
      }
    };
    while (t + dt_v < tau) {
      step(dt_v);
      t += dt_v;
    }
    step(tau - t);
  }

  /// ekf.hpp:116-139.  h(g) -> Vec<Ny> (measurement in R^Ny), y measured value, R covariance
  template<int Ny, class H>
  void update(H && h, const Vec<Ny> & y, const Mat<Ny, Ny> & R)
  {
    Mat<Ny, N> Hm{};
    Vec<Ny> r{};
    detail::ekf_linearise_meas<Ny>(h, g_hat_, y, Hm, r);
    typename G::Tangent delta{};
    int32_t info = 0;
    detail::ekf_check(sfb_ekf_step_batch_host(1, N, Ny, nullptr, nullptr, 0, nullptr, 0, Hm.a.data(), R.a.data(), 1,
                                              r.data(), P_.a.data(), delta.data(), &info));
    g_hat_ = rplus(g_hat_, delta);  // :137
  }

private:
  G g_hat_  = G::Identity();
  CovT P_   = CovT::Identity();
};

}  // namespace smooth_feedback_amd
