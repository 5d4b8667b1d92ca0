// Host side of the active-set-invariance (ASI) safety filter: same interface as
// smooth::feedback::{ASIFProblem, ASIFtoQPParams, asif_to_qp_allocate, asif_to_qp_update, asif_to_qp,
// ASIFilterParams, ASIFilter} (reference asif_func.hpp:40-262, asif.hpp:17-110).  The closed-loop
// backup trajectory and its sensitivity are integrated here on the host (explicit Euler, as the
// reference does with odeint, asif_func.hpp:121-122,175-179); the resulting small dense QP
// (n = nu + 1, m = K nh + nu_ineq + 1) is solved on the GPU through the C-ABI:
//   sfb_qp_dense_solve_batch_host: the dense kernels for n + m <= 64 (the defaults K = 10, nh = 1 give k = 14);
//   behind the same call, larger problems (n = 3, m = 203 in examples/mpc_asif_vehicle.cpp:105-129) run on the
//   shared-pattern sparse kernel with a full pattern -- same ADMM, different (non-pivoted) factorisation order,
//   so parity with the dense oracle is to tolerance there.
// ASIFSwarm filters a batch of agents with ONE launch.
//
// Derivatives of f, h and bu: forward differences with step sqrt(eps) (the reference differentiates
// with its default diff::Type, autodiff when that header is present).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <memory>
#include <optional>
#include <type_traits>
#include <utility>
#include <thread>
#include <vector>

#include "lie.hpp"
#include "qp.hpp"

namespace smooth_feedback_amd {

/// common.hpp:18-30: { A (m - c) in [l, u] }, A is rows x Dof<M> column-major
template<class M>
struct ManifoldBounds {
  int rows = 0;
  std::vector<double> A{};
  M c = M::Identity();
  std::vector<double> l{}, u{};
};

namespace detail {
template<int N>
SFB_LIE_HD Vec<N> ones()
{
  Vec<N> v{};
  for (int i = 0; i < N; ++i) v[i] = 1.0;
  return v;
}
}  // namespace detail

/// asif_func.hpp:40-53
template<class X, class U>
struct ASIFProblem {
  double T{1};
  X x0 = X::Identity();
  U u_des = U::Identity();
  Vec<U::Dof> W_u = detail::ones<U::Dof>();
  ManifoldBounds<U> ulim{};
};

/// asif_func.hpp:58-68
struct ASIFtoQPParams {
  std::size_t K{10};
  double alpha{1};
  double dt{0.1};
  double relax_cost{100};
};

/// asif_func.hpp:78-99: M = K nh + nu_ineq + 1 rows, N = nu + 1 variables, all zero
template<class X, class U>
void asif_to_qp_allocate(QuadraticProgram<> & qp, std::size_t K, std::size_t nu_ineq, std::size_t nh)
{
  const int M = int(K * nh + nu_ineq + 1), N = U::Dof + 1;
  qp.n = N;
  qp.m = M;
  qp.A.assign((size_t)M * N, 0.0);
  qp.l.assign(M, 0.0);
  qp.u.assign(M, 0.0);
  qp.P.assign((size_t)N * N, 0.0);
  qp.q.assign(N, 0.0);
}

namespace detail {
SFB_LIE_HD inline double fd_step() { return 1.4901161193847656e-08; }  // sqrt(DBL_EPSILON) = 2^-26, exactly

// right derivative of g -> fun(g) (values in R^NO) at g, by forward differences
template<int NO, class G, class Fun>
SFB_LIE_HD Mat<NO, G::Dof> dr_fd(Fun && fun, const G & g, const Vec<NO> & f0)
{
  Mat<NO, G::Dof> J{};
  const double h = fd_step();
  for (int c = 0; c < G::Dof; ++c) {
    typename G::Tangent e{};
    e[c]          = h;
    const auto f1 = fun(rplus(g, e));
    for (int r = 0; r < NO; ++r) J(r, c) = (f1[r] - f0[r]) / h;
  }
  return J;
}
// fn(agent, t, x) seen as fn(t, x) for one agent; forwards fn.jacobian(agent, t, x, J) when the callable has one
template<class Fn, class G>
struct AgentFn {
  const Fn & fn;
  std::size_t agent;
  SFB_LIE_HD auto operator()(double t, const G & x) const { return fn(agent, t, x); }
  template<class M, class F2 = Fn>  // (F2: keeps the member lookup dependent, so a callable without one is no error)
  SFB_LIE_HD auto jacobian(double t, const G & x, M & J) const -> decltype(std::declval<const F2 &>().jacobian(agent, t, x, J))
  {
    return fn.jacobian(agent, t, x, J);
  }
};
}  // namespace detail

/// The problem of asif_to_qp_update as plain data (usable in device code): ASIFProblem with the input bounds as
/// pointers, and the parameters of ASIFtoQPParams.
template<class X, class U>
struct ASIFProblemView {
  double T;
  X x0;
  U u_des;
  Vec<U::Dof> W_u;
  int ulim_rows;
  const double *ulim_A, *ulim_l, *ulim_u;  // ManifoldBounds: A rows x Dof<U> column-major, l, u [rows]
  U ulim_c;
  int K;
  double alpha, dt, relax_cost;
};

/// asif_func.hpp:104-199 on raw arrays: P [N*N], q [N], A [M*N] column-major, l, u [M] with N = nu + 1,
/// M = K nh + nu_ineq + 1, all-zero on entry as asif_to_qp_allocate leaves them (only the entries the reference writes
/// are written).  f(x, u) -> Tangent<X>, h(t, x) -> Vec<nh>, bu(t, x) -> U.  Host and device (asif_device.hpp) run
/// this very function, so their QPs differ only by what sin / cos / atan2 of the two maths libraries differ.
template<class X, class U, class F, class H, class BU>
SFB_LIE_HD void asif_fill(double * qP, double * qq, double * qA, double * ql, double * qu, const ASIFProblemView<X, U> & pbm,
                          const F & f, const H & h, const BU & bu)
{
  constexpr int nx = X::Dof, nu = U::Dof;
  using HVal       = std::decay_t<decltype(h(0.0, pbm.x0))>;
  constexpr int nh = int(std::tuple_size<HVal>::value);
  const int nu_ineq = pbm.ulim_rows;
  const int K = pbm.K, M = K * nh + nu_ineq + 1;
  const double inf = std::numeric_limits<double>::infinity();
  auto A = [&](int r, int c) -> double & { return qA[(size_t)r + (size_t)c * M]; };

  // iteration variables :139-143
  const double tau = pbm.T / static_cast<double>(K);
  const double dt  = (pbm.dt < tau) ? pbm.dt : tau;
  double t         = 0;
  X x              = pbm.x0;
  Mat<nx, nx> S    = Mat<nx, nx>::Identity();  // dx/dx0

  // value of the dynamics at call time and its derivative w.r.t. u :155-157
  // Derivatives: the reference differentiates f, h and bu with smooth::diff (autodiff when available).  Here a functor
  // may carry analytic right-Jacobians -- f.jacobian(x, u, dfdx, dfdu) (the MPC front's convention), h.jacobian(t, x,
  // dhdx), bu.jacobian(t, x, dbudx) --; whatever is missing is replaced by forward differences (step sqrt(eps)).
  constexpr bool f_an = requires(Mat<nx, nx> & a, Mat<nx, nu> & b) { f.jacobian(pbm.x0, pbm.u_des, a, b); };
  const Vec<nx> f0 = f(x, pbm.u_des);
  Mat<nx, nu> d_f0_du{};
  if constexpr (f_an) {
    Mat<nx, nx> unused{};
    f.jacobian(x, pbm.u_des, unused, d_f0_du);
  } else {
    d_f0_du = detail::dr_fd<nx>([&](const U & vu) { return f(x, vu); }, pbm.u_des, f0);
  }

  for (int k = 0; k != K; ++k) {
    // barrier function and its derivatives w.r.t. (t, x) :161-166
    const HVal hval = h(t, x);
    const double e  = detail::fd_step();
    const HVal ht   = h(t + e, x);
    Mat<nh, nx> dh_dx{};
    if constexpr (requires { h.jacobian(t, x, dh_dx); }) h.jacobian(t, x, dh_dx);
    else dh_dx = detail::dr_fd<nh>([&](const X & vx) { return h(t, vx); }, x, hval);
    // barrier constraint :168-172
    const Mat<nh, nx> dh_dx0 = dh_dx * S;
    const Mat<nh, nu> Ak     = dh_dx0 * d_f0_du;
    const Vec<nh> dhf        = dh_dx0 * f0;
    for (int r = 0; r < nh; ++r) {
      for (int c = 0; c < nu; ++c) A(k * nh + r, c) = Ak(r, c);
      ql[k * nh + r] = -((ht[r] - hval[r]) / e) - pbm.alpha * hval[r] - dhf[r];
      qu[k * nh + r] = inf;
    }
    // integrate system and sensitivity until the next constraint :175-180.  As in the reference the
    // step is fixed per interval, the state is stepped first and the sensitivity ODE is evaluated at
    // the stepped state with the old time.
    const double rest = tau * double(k + 1) - t, dt_act = (rest < dt) ? rest : dt;
    while (t < tau * double(k + 1)) {
      {
        auto dx = f(x, bu(t, x));
        for (int i = 0; i < nx; ++i) dx[i] *= dt_act;
        x = rplus(x, dx);
      }
      {
        auto fcl_fun = [&](const X & vx) { return f(vx, bu(t, vx)); };
        Vec<nx> fcl{};
        Mat<nx, nx> dS{};
        if constexpr (f_an) {  // d/dx f(x, bu(x)) = df/dx + df/du dbu/dx
          const U ucl = bu(t, x);
          fcl         = f(x, ucl);
          Mat<nx, nu> dfu{};
          f.jacobian(x, ucl, dS, dfu);
          Mat<nu, nx> dbx{};
          if constexpr (requires { bu.jacobian(t, x, dbx); }) {
            bu.jacobian(t, x, dbx);
          } else {
            const double hh = detail::fd_step();
            for (int c = 0; c < nx; ++c) {
              typename X::Tangent e{};
              e[c]          = hh;
              const auto du = rminus(bu(t, rplus(x, e)), ucl);
              for (int r = 0; r < nu; ++r) dbx(r, c) = du[r] / hh;
            }
          }
          dS = dS + dfu * dbx;
        } else {
          fcl = fcl_fun(x);
          dS  = detail::dr_fd<nx>(fcl_fun, x, fcl);
        }
        const Mat<nx, nx> adf = X::ad(fcl);
        for (int i = 0; i < nx * nx; ++i) dS.a[i] -= adf.a[i];
        S = S + dt_act * (dS * S);
      }
      t += dt_act;
    }
  }

  // relaxation of the barrier constraints :183
  for (int r = 0; r < K * nh; ++r) A(r, nu) = 1.0;
  // input bounds :186-188
  {
    const auto d = rminus(pbm.u_des, pbm.ulim_c);
    for (int r = 0; r < nu_ineq; ++r) {
      double Ad = 0.0;
      for (int c = 0; c < nu; ++c) {
        const double a     = pbm.ulim_A[(size_t)r + (size_t)c * nu_ineq];
        A(K * nh + r, c) = a;
        Ad += a * d[c];
      }
      ql[K * nh + r] = pbm.ulim_l[r] - Ad;
      qu[K * nh + r] = pbm.ulim_u[r] - Ad;
    }
  }
  // bounds on the relaxation delta :191-193
  A(K * nh + nu_ineq, nu) = 1.0;
  ql[K * nh + nu_ineq]     = 0.0;
  qu[K * nh + nu_ineq]     = inf;
  // cost :195-198
  const int N = nu + 1;
  for (int i = 0; i < nu; ++i) qP[(size_t)i + (size_t)i * N] = pbm.W_u[i];
  qP[(size_t)nu + (size_t)nu * N] = pbm.relax_cost;
  qq[nu]                          = 0.0;
}

/// asif_func.hpp:104-199.  f(x, u) -> Tangent<X>, h(t, x) -> Vec<nh>, bu(t, x) -> U
template<class X, class U, class F, class H, class BU>
void asif_to_qp_update(QuadraticProgram<> & qp, const ASIFProblem<X, U> & pbm, const ASIFtoQPParams & prm, F && f, H && h,
                       BU && bu)
{
  const ASIFProblemView<X, U> v{pbm.T, pbm.x0, pbm.u_des, pbm.W_u, pbm.ulim.rows, pbm.ulim.A.data(), pbm.ulim.l.data(),
                                pbm.ulim.u.data(), pbm.ulim.c, int(prm.K), prm.alpha, prm.dt, prm.relax_cost};
  asif_fill<X, U>(qp.P.data(), qp.q.data(), qp.A.data(), qp.l.data(), qp.u.data(), v, f, h, bu);
}


/// asif_func.hpp:245-260
template<class X, class U, class F, class H, class BU>
QuadraticProgram<> asif_to_qp(const ASIFProblem<X, U> & pbm, const ASIFtoQPParams & prm, F && f, H && h, BU && bu)
{
  using HVal       = std::decay_t<decltype(h(0.0, pbm.x0))>;
  constexpr int nh = int(std::tuple_size<HVal>::value);
  QuadraticProgram<> qp;
  asif_to_qp_allocate<X, U>(qp, prm.K, pbm.ulim.rows, nh);
  asif_to_qp_update<X, U>(qp, pbm, prm, std::forward<F>(f), std::forward<H>(h), std::forward<BU>(bu));
  return qp;
}

/// asif.hpp:17-32
template<class U>
struct ASIFilterParams {
  double T{1};
  std::size_t nh{1};
  Vec<U::Dof> u_weight = detail::ones<U::Dof>();
  ManifoldBounds<U> ulim{};
  ASIFtoQPParams asif{};
  QPSolverParams qp{};
};

namespace detail {
/// Dense QPs of one shape on the GPU through the dense entry point: the register/LDS-resident dense kernels
/// for n + m <= 64, the shared-pattern sparse kernel with a full pattern behind the same call for larger ones.
class DenseQPBackend {
public:
  DenseQPBackend(int n, int m, const QPSolverParams & prm) : n_(n), m_(m), prm_(prm) {}

  /// B problems, batch-major dense column-major arrays as in sfb_qp_dense_solve_batch_host
  void solve_batch(int64_t B, const double * P, const double * q, const double * A, const double * l, const double * u,
                   const double * wx, const double * wy, double * x, double * y, double * obj, uint32_t * iter,
                   int32_t * code)
  {
    const sfb_qp_params c = prm_.to_c();
    sfb_check(sfb_qp_dense_solve_batch_host(&c, B, n_, m_, P, q, A, l, u, wx, wy, x, y, obj, iter, code));
  }

private:
  int n_, m_;
  QPSolverParams prm_;
};
}  // namespace detail

/// asif.hpp:41-110
template<class G, class U, class Dyn>
class ASIFilter {
public:
  explicit ASIFilter(Dyn f, ASIFilterParams<U> prm = {}) : f_(std::move(f)), prm_(std::move(prm))
  {
    asif_to_qp_allocate<G, U>(qp_, prm_.asif.K, prm_.ulim.rows, prm_.nh);  // :59-60
    backend_ = std::make_unique<detail::DenseQPBackend>(qp_.n, qp_.m, prm_.qp);
  }

  /// asif.hpp:82-102: {u, code}
  template<class H, class BU>
  std::pair<U, QPSolutionStatus> operator()(const G & g, const U & u_des, H && h, BU && bu)
  {
    ASIFProblem<G, U> pbm{prm_.T, g, u_des, prm_.u_weight, prm_.ulim};
    asif_to_qp_update<G, U>(qp_, pbm, prm_.asif, f_, std::forward<H>(h), std::forward<BU>(bu));
    QPSolution<> sol;
    sol.primal.resize(qp_.n);
    sol.dual.resize(qp_.m);
    int32_t code = 6;
    backend_->solve_batch(1, qp_.P.data(), qp_.q.data(), qp_.A.data(), qp_.l.data(), qp_.u.data(),
                          warmstart_ ? warmstart_->primal.data() : nullptr, warmstart_ ? warmstart_->dual.data() : nullptr,
                          sol.primal.data(), sol.dual.data(), &sol.objective, &sol.iter, &code);
    sol.code = static_cast<QPSolutionStatus>(code);
    if (sol.code == QPSolutionStatus::Optimal) warmstart_ = sol;  // :99
    typename U::Tangent du{};
    for (int i = 0; i < U::Dof; ++i) du[i] = sol.primal[i];
    last_ = sol;
    return {rplus(u_des, du), sol.code};  // :101
  }

  const QuadraticProgram<> & qp() const { return qp_; }
  const QPSolution<> & last_solution() const { return last_; }

private:
  Dyn f_;
  QuadraticProgram<> qp_;
  ASIFilterParams<U> prm_;
  std::optional<QPSolution<>> warmstart_;
  QPSolution<> last_;
  std::unique_ptr<detail::DenseQPBackend> backend_;
};

/// Batched safety filtering: B agents with the same dynamics and parameters, one QP launch per tick.
template<class G, class U, class Dyn>
class ASIFSwarm {
public:
  ASIFSwarm(Dyn f, std::size_t agents, ASIFilterParams<U> prm = {}) : f_(std::move(f)), B_(agents), prm_(std::move(prm))
  {
    asif_to_qp_allocate<G, U>(qp_, prm_.asif.K, prm_.ulim.rows, prm_.nh);
    n_ = qp_.n;
    m_ = qp_.m;
    backend_ = std::make_unique<detail::DenseQPBackend>(n_, m_, prm_.qp);
    P_.assign(B_ * n_ * n_, 0.0); q_.assign(B_ * n_, 0.0); A_.assign(B_ * m_ * n_, 0.0);
    l_.assign(B_ * m_, 0.0); u_.assign(B_ * m_, 0.0);
    x_.assign(B_ * n_, 0.0); y_.assign(B_ * m_, 0.0); wx_ = x_; wy_ = y_;
    iter_.assign(B_, 0); code_.assign(B_, 6);
  }

  /// h(b, t, x), bu(b, t, x): per-agent safe set and backup controller.  Returns the filtered inputs.
  template<class H, class BU>
  std::vector<U> operator()(const std::vector<G> & g, const std::vector<U> & u_des, H && h, BU && bu)
  {
    const auto T0 = std::chrono::steady_clock::now();
    // assembly (sensitivity ODE of every agent, asif_func.hpp:145-179) on the host cores, one scratch QP per thread;
    // h, bu and the dynamics must be callable concurrently for different agents
    {
      // one thread per physical core (the ODE is floating-point bound: SMT siblings only add overhead, measured);
      // SFB_ASIF_THREADS overrides
      const char * tenv = getenv("SFB_ASIF_THREADS");
      const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
      const int T = (int)std::min<std::size_t>(tenv ? (unsigned)std::max(1, atoi(tenv)) : (hw >= 16 ? hw / 2 : hw), B_);
      std::vector<std::thread> th;
      for (int k = 0; k < T; ++k)
        th.emplace_back([&, k] {
          QuadraticProgram<> qp = qp_;  // same layout, private values
          for (std::size_t b = B_ * k / T; b < B_ * (k + 1) / T; ++b) {
            ASIFProblem<G, U> pbm{prm_.T, g[b], u_des[b], prm_.u_weight, prm_.ulim};
            asif_to_qp_update<G, U>(qp, pbm, prm_.asif, f_, detail::AgentFn<std::remove_reference_t<H>, G>{h, b},
                                    detail::AgentFn<std::remove_reference_t<BU>, G>{bu, b});
            std::copy(qp.P.begin(), qp.P.end(), P_.begin() + b * n_ * n_);
            std::copy(qp.q.begin(), qp.q.end(), q_.begin() + b * n_);
            std::copy(qp.A.begin(), qp.A.end(), A_.begin() + b * m_ * n_);
            std::copy(qp.l.begin(), qp.l.end(), l_.begin() + b * m_);
            std::copy(qp.u.begin(), qp.u.end(), u_.begin() + b * m_);
          }
        });
      for (auto & t : th) t.join();
    }
    const auto T1 = std::chrono::steady_clock::now();
    if (getenv("SFB_ASIF_TIMING")) fprintf(stderr, "[asif swarm] assembly of %zu agents %.1f ms\n", B_, std::chrono::duration<double, std::milli>(T1 - T0).count());
    backend_->solve_batch((int64_t)B_, P_.data(), q_.data(), A_.data(), l_.data(), u_.data(),
                          have_warm_ ? wx_.data() : nullptr, have_warm_ ? wy_.data() : nullptr, x_.data(), y_.data(),
                          nullptr, iter_.data(), code_.data());
    const auto T2 = std::chrono::steady_clock::now();
    if (getenv("SFB_ASIF_TIMING")) fprintf(stderr, "[asif swarm] solve (host entry point) %.1f ms\n", std::chrono::duration<double, std::milli>(T2 - T1).count());
    std::vector<U> out(B_);
    for (std::size_t b = 0; b < B_; ++b) {
      typename U::Tangent du{};
      for (int i = 0; i < U::Dof; ++i) du[i] = x_[b * n_ + i];
      out[b] = rplus(u_des[b], du);
      if (code_[b] == 0) {  // asif.hpp:99: only an Optimal solution becomes the next warm start
        std::copy(x_.begin() + b * n_, x_.begin() + (b + 1) * n_, wx_.begin() + b * n_);
        std::copy(y_.begin() + b * m_, y_.begin() + (b + 1) * m_, wy_.begin() + b * m_);
      }
    }
    have_warm_ = true;  // agents that never were Optimal keep the zero start (== no warm start)
    return out;
  }

  const std::vector<int32_t> & codes() const { return code_; }
  const std::vector<uint32_t> & iterations() const { return iter_; }
  /// the QPs of the last call (batch-major, column-major matrices) and their primal / dual solutions
  void copy_problem(double * P, double * q, double * A, double * l, double * u, double * x, double * y) const
  {
    std::copy(P_.begin(), P_.end(), P); std::copy(q_.begin(), q_.end(), q); std::copy(A_.begin(), A_.end(), A);
    std::copy(l_.begin(), l_.end(), l); std::copy(u_.begin(), u_.end(), u);
    std::copy(x_.begin(), x_.end(), x); std::copy(y_.begin(), y_.end(), y);
  }
  /// the warm start the NEXT call will use (zeros before the first call)
  void copy_warm_start(double * wx, double * wy) const
  {
    std::copy(wx_.begin(), wx_.end(), wx);
    std::copy(wy_.begin(), wy_.end(), wy);
  }

private:
  Dyn f_;
  std::size_t B_;
  ASIFilterParams<U> prm_;
  QuadraticProgram<> qp_;
  int n_ = 0, m_ = 0;
  std::unique_ptr<detail::DenseQPBackend> backend_;
  std::vector<double> P_, q_, A_, l_, u_, x_, y_, wx_, wy_;
  std::vector<uint32_t> iter_;
  std::vector<int32_t> code_;
  bool have_warm_ = false;
};

}  // namespace smooth_feedback_amd
