// C++ front of the QP path: same names, template parameters and semantics as smooth::feedback (reference qp.hpp,
// qp_solver.hpp), storage in plain std::vector (no Eigen in this tree), numerics in libsfb.so (HIP).
//
//   QuadraticProgram<M, N, Scalar>       qp.hpp:31-45    M, N static or -1 (dynamic), mixed allowed
//   QuadraticProgramSparse<Scalar>       qp.hpp:60-79    P CSC as stored, A CSR
//   QPSolutionStatus, QPSolution<M,N,S>  qp.hpp:82-108
//   QPSolverParams                       qp_solver.hpp:29-68
//   QPSolver<Pbm>::{analyze, solve, sol} qp_solver.hpp:242-757   copyable and movable like the reference's
//   solve_qp(pbm, prm, warmstart)        qp_solver.hpp:779-787
// Scalar must be double (the device path is fp64).  include/smooth/feedback/*.hpp forward to these headers and alias
// the namespace, so reference-style `smooth::feedback::QPSolver<...>` spellings compile against this tree.
#pragma once
#include <sfb.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace smooth_feedback_amd {

/// qp.hpp:82-92
enum class QPSolutionStatus { Optimal, PolishFailed, PrimalInfeasible, DualInfeasible, MaxIterations, MaxTime, Unknown };

/// qp_solver.hpp:29-68
struct QPSolverParams {
  bool verbose = false;
  float alpha = 1.6f, rho = 0.1f, sigma = 1e-6f;
  bool scaling = true;
  float eps_abs = 1e-3f, eps_rel = 1e-3f, eps_primal_inf = 1e-4f, eps_dual_inf = 1e-4f;
  std::optional<uint32_t> max_iter            = {};
  std::optional<std::chrono::nanoseconds> max_time = {};
  uint32_t stop_check_iter = 25;
  bool polish              = true;
  uint32_t polish_iter     = 5;
  float delta              = 1e-6f;
  /// extension (sfb.h, sfb_qp_params::reuse_factor): P and A are bit-identical to the previous solve on this solver's
  /// workspace -- scaling and factorisation are kept where provably unchanged; results do not depend on the flag
  bool reuse_factor = false;

  sfb_qp_params to_c() const
  {
    sfb_qp_params c;
    sfb_qp_params_default(&c);
    c.verbose = verbose; c.alpha = alpha; c.rho = rho; c.sigma = sigma; c.scaling = scaling;
    c.eps_abs = eps_abs; c.eps_rel = eps_rel; c.eps_primal_inf = eps_primal_inf; c.eps_dual_inf = eps_dual_inf;
    c.max_iter = max_iter ? int64_t(*max_iter) : -1;
    c.max_time_ns = max_time ? int64_t(max_time->count()) : -1;
    c.stop_check_iter = stop_check_iter; c.polish = polish; c.polish_iter = polish_iter; c.delta = delta;
    c.reuse_factor = reuse_factor;
    return c;
  }
};

/// qp.hpp:31-45.  P (n x n) and A (m x n) column-major like Eigen's default; M, N >= 0 fix a size at compile time
/// (checked against the data at solve time), -1 leaves it to n / m.
template<int M = -1, int N = -1, class Scalar = double>
struct QuadraticProgram {
  static_assert(std::is_same_v<Scalar, double>, "the device path is fp64");
  static constexpr int RowsAtCompileTime = M, ColsAtCompileTime = N;
  int n = N >= 0 ? N : 0, m = M >= 0 ? M : 0;
  std::vector<double> P, q, A, l, u;
  QuadraticProgram() { if (M >= 0 && N >= 0) resize(N, M); }
  void resize(int n_, int m_)
  {
    n = n_; m = m_;
    P.assign((size_t)n * n, 0.0); q.assign(n, 0.0); A.assign((size_t)m * n, 0.0); l.assign(m, 0.0); u.assign(m, 0.0);
  }
};

/// qp.hpp:60-79: P CSC (Eigen::SparseMatrix default), A CSR (Eigen::RowMajor)
template<class Scalar = double>
struct QuadraticProgramSparse {
  static_assert(std::is_same_v<Scalar, double>, "the device path is fp64");
  int n = 0, m = 0;
  std::vector<int32_t> P_colptr, P_rowind;
  std::vector<double> P_val;
  std::vector<double> q;
  std::vector<int32_t> A_rowptr, A_colind;
  std::vector<double> A_val;
  std::vector<double> l, u;
};

/// qp.hpp:95-108
template<int M = -1, int N = -1, class Scalar = double>
struct QPSolution {
  QPSolutionStatus code = QPSolutionStatus::Unknown;
  uint32_t iter         = 0;
  std::vector<double> primal, dual;
  double objective = 0.;
};

inline void sfb_check(sfb_status st)
{
  if (st != SFB_OK) throw std::runtime_error(std::string("sfb: ") + sfb_last_error());
}

namespace detail {
template<class T> struct is_sparse_qp : std::false_type {};
template<class S> struct is_sparse_qp<QuadraticProgramSparse<S>> : std::true_type {};

/// Owner of the symbolic analysis of a sparse pattern.  Copy semantics of the reference's detail::LDLTWrapper
/// (qp_solver.hpp:209-231): a COPY starts without an analysis (it is redone at the next solve), a move takes it over.
struct PlanHolder {
  sfb_sparse_qp_plan * plan = nullptr;
  PlanHolder() = default;
  PlanHolder(const PlanHolder &) {}
  PlanHolder(PlanHolder && o) noexcept : plan(o.plan) { o.plan = nullptr; }
  PlanHolder & operator=(const PlanHolder & o)
  {
    if (this != &o) reset();
    return *this;
  }
  PlanHolder & operator=(PlanHolder && o) noexcept
  {
    if (this != &o) { reset(); plan = o.plan; o.plan = nullptr; }
    return *this;
  }
  ~PlanHolder() { reset(); }
  void reset()
  {
    sfb_sparse_qp_plan_destroy(plan);
    plan = nullptr;
  }
};
/// pins travel with the analysis: a copy starts unpinned (it has no analysis), a move takes them over
struct PinCount {
  int v = 0;
  PinCount() = default;
  PinCount(const PinCount &) {}
  PinCount(PinCount && o) noexcept : v(o.v) { o.v = 0; }
  PinCount & operator=(const PinCount &) { return *this; }
  PinCount & operator=(PinCount && o) noexcept { if (this != &o) { v = o.v; o.v = 0; } return *this; }
};
}  // namespace detail

/// QPSolver<Pbm>, qp_solver.hpp:242-757.  Pbm = QuadraticProgram<M, N> (dense kernels) or QuadraticProgramSparse<>
/// (analyze() = symbolic analysis once per pattern, solve() / solve_batch() for problems that share it).
template<class Pbm>
class QPSolver {
public:
  static constexpr bool sparse = detail::is_sparse_qp<Pbm>::value;
  using Solution               = QPSolution<>;

  QPSolver(const QPSolverParams & prm = {}) : prm_(prm) {}                         // :267
  QPSolver(const Pbm & pbm, const QPSolverParams & prm = {}) : prm_(prm) { analyze(pbm); }  // :276
  QPSolver(const QPSolver &)                = default;
  QPSolver(QPSolver &&) noexcept            = default;
  QPSolver & operator=(const QPSolver &)    = default;
  QPSolver & operator=(QPSolver &&) noexcept = default;
  ~QPSolver()                               = default;

  /// :292
  const Solution & sol() const { return sol_; }

  /// :297-338.  Dense: sizes the solution.  Sparse: + SimplicialLDLT::analyzePattern (:424): KKT pattern, elimination
  /// order, pattern of L.  user_perm / stage / A_keep: see sfb_sparse_qp_plan_create_pruned (A_keep, one byte per
  /// stored entry of A: 0 = the entry is zero in every problem solved with this analysis -- checked per item on the
  /// device, a wrong declaration costs time, not correctness).
  void analyze(const Pbm & pbm, const int32_t * user_perm = nullptr, const int32_t * stage = nullptr,
               const uint8_t * A_keep = nullptr)
  {
    n_ = pbm.n; m_ = pbm.m;
    sol_.primal.assign(n_, 0.0);
    sol_.dual.assign(m_, 0.0);
    if constexpr (sparse) {
      if (pins_.v > 0) throw std::logic_error("QPSolver::analyze: the analysis is in use by a device-resident swarm");
      holder_.reset();
      last_P_.clear(); last_A_.clear();
      sfb_check(sfb_sparse_qp_plan_create_pruned(pbm.n, pbm.m, pbm.P_colptr.data(), pbm.P_rowind.data(), pbm.A_rowptr.data(),
                                                 pbm.A_colind.data(), 1, user_perm, stage, A_keep, &holder_.plan));
    } else {
      (void)user_perm; (void)stage; (void)A_keep;
      if ((Pbm::RowsAtCompileTime >= 0 && Pbm::RowsAtCompileTime != pbm.m) ||
          (Pbm::ColsAtCompileTime >= 0 && Pbm::ColsAtCompileTime != pbm.n))
        throw std::invalid_argument("QPSolver: problem does not have its static size");
    }
    analyzed_ = true;
  }
  bool analyzed() const
  {
    if constexpr (sparse) return holder_.plan != nullptr;
    else return analyzed_;
  }
  /// forget the analysis (the next solve analyses again)
  void reset()
  {
    if (pins_.v > 0) throw std::logic_error("QPSolver::reset: the analysis is in use by a device-resident swarm");
    holder_.reset();
    analyzed_ = false;
  }

  /// :343-344.  The returned reference is valid until the next solve (as in the reference).
  const Solution & solve(const Pbm & pbm, std::optional<std::reference_wrapper<const Solution>> warmstart = {})
  {
    return solve(pbm, warmstart ? &warmstart->get() : nullptr);
  }
  const Solution & solve(const Pbm & pbm, const Solution * warmstart)
  {
    if (!analyzed() || n_ != pbm.n || m_ != pbm.m) analyze(pbm);
    int32_t code    = 6;
    sfb_qp_params c = prm_.to_c();
    if constexpr (sparse) {
      // A time-invariant problem family (a linear MPC: only q, l, u move between ticks) presents the very same
      // matrices again: tell the kernel, which then keeps the scaling and the factor it still holds for them
      // (sfb_qp_params::reuse_factor -- same results, bit for bit).
      const bool same = last_P_.size() == pbm.P_val.size() && last_A_.size() == pbm.A_val.size() && !last_A_.empty() &&
                        std::memcmp(last_P_.data(), pbm.P_val.data(), last_P_.size() * sizeof(double)) == 0 &&
                        std::memcmp(last_A_.data(), pbm.A_val.data(), last_A_.size() * sizeof(double)) == 0;
      if (same) { c.reuse_factor = 1; ++reuse_count_; }
      else { last_P_ = pbm.P_val; last_A_ = pbm.A_val; }
    }
    const double * wx = warmstart ? warmstart->primal.data() : nullptr;
    const double * wy = warmstart ? warmstart->dual.data() : nullptr;
    if constexpr (sparse) {
      sfb_check(sfb_sparse_qp_solve_batch_host(holder_.plan, &c, 1, pbm.P_val.data(), pbm.q.data(), pbm.A_val.data(),
                                               pbm.l.data(), pbm.u.data(), wx, wy, sol_.primal.data(), sol_.dual.data(),
                                               &sol_.objective, &sol_.iter, &code));
    } else {
      sfb_check(sfb_qp_dense_solve_batch_host(&c, 1, pbm.n, pbm.m, pbm.P.data(), pbm.q.data(), pbm.A.data(), pbm.l.data(),
                                              pbm.u.data(), wx, wy, sol_.primal.data(), sol_.dual.data(), &sol_.objective,
                                              &sol_.iter, &code));
    }
    sol_.code = static_cast<QPSolutionStatus>(code);
    return sol_;
  }

  /// Batched solve on host buffers for problems sharing the analysed pattern (sparse): Px [B][nnzP], q [B][n],
  /// Ax [B][nnzA], l, u [B][m]; dense: P [B][n*n], A [B][m*n] column-major.
  void solve_batch(int64_t B, const double * Px, const double * q, const double * Ax, const double * l, const double * u,
                   const double * warm_x, const double * warm_y, double * x, double * y, double * obj, uint32_t * iter,
                   int32_t * code)
  {
    const sfb_qp_params c = prm_.to_c();
    if constexpr (sparse) {
      if (!holder_.plan) throw std::logic_error("QPSolver::solve_batch: analyze() first");
      // the batch call shares the plan's host-side device workspace with solve(): whatever factor solve() left there is
      // gone, so the next solve() must not vouch for it (reuse_factor) even if it sees its old matrices again
      last_P_.clear(); last_A_.clear();
      sfb_check((multi_device_ ? sfb_sparse_qp_solve_batch_host_multi : sfb_sparse_qp_solve_batch_host)(
        holder_.plan, &c, B, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code));
    } else {
      sfb_check((multi_device_ ? sfb_qp_dense_solve_batch_host_multi : sfb_qp_dense_solve_batch_host)(
        &c, B, n_, m_, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code));
    }
  }
  /// solve_batch() on every device of the process (sfb_set_devices, sfb.h): the batch is cut into contiguous shards,
  /// one per device, each solved by its own host thread; same results as on one device.  The reference has no
  /// counterpart (its batches are a sequential loop, benchmarks/bench_types.hpp:93).
  void shard_over_devices(bool on) { multi_device_ = on; }
  bool sharded_over_devices() const { return multi_device_; }

  /// solves that found the previous solve's matrices again and were flagged reuse_factor
  int64_t factor_reuse_count() const { return reuse_count_; }

  int64_t nnzL() const
  {
    int64_t v = 0;
    if constexpr (sparse) sfb_sparse_qp_plan_info(holder_.plan, nullptr, &v, nullptr);
    return v;
  }
  const QPSolverParams & params() const { return prm_; }
  sfb_sparse_qp_plan * plan() { return holder_.plan; }
  /// Objects that keep the raw plan pointer beyond a call (device-resident swarms: sfb_mpc_swarm holds it, sfb.h) pin
  /// the analysis for their lifetime; while pinned, reset() and analyze() refuse to destroy it.
  void pin_plan() { ++pins_.v; }
  void unpin_plan() { if (pins_.v > 0) --pins_.v; }
  bool plan_pinned() const { return pins_.v > 0; }

private:
  QPSolverParams prm_{};
  detail::PlanHolder holder_;
  bool analyzed_ = false;
  std::vector<double> last_P_, last_A_;  // sparse: the matrices of the previous solve() (factor reuse)
  int64_t reuse_count_ = 0;
  detail::PinCount pins_;
  bool multi_device_ = false;
  int n_ = 0, m_ = 0;
  Solution sol_;
};

template<class Pbm> QPSolver(const Pbm &, const QPSolverParams &) -> QPSolver<Pbm>;
template<class Pbm> QPSolver(const Pbm &) -> QPSolver<Pbm>;

using SparseQPSolver = QPSolver<QuadraticProgramSparse<double>>;

/// solve_qp, qp_solver.hpp:779-787: n + m <= 64 runs on the register / LDS-resident dense kernels, up to 1024 on the
/// pivoted dense kernel with the factor in HBM, sparse problems on the shared-pattern kernel (include/sfb.h)
template<class Pbm>
QPSolution<> solve_qp(const Pbm & pbm, const QPSolverParams & prm = {},
                      std::optional<std::reference_wrapper<const QPSolution<>>> warmstart = {})
{
  QPSolver<Pbm> solver(pbm, prm);
  return solver.solve(pbm, warmstart);
}
template<class Pbm>
QPSolution<> solve_qp(const Pbm & pbm, const QPSolverParams & prm, const QPSolution<> * warmstart)
{
  QPSolver<Pbm> solver(pbm, prm);
  return solver.solve(pbm, warmstart);
}

/// The parameter mapping of the reference's OSQP comparator (compat/osqp.hpp:54-80): what solve_qp_osqp sets on an
/// OSQPSettings before osqp_setup.  OSQP itself is not part of this tree; field names are OSQP's.
struct OsqpSettingsView {
  int verbose, scaling, check_termination, polish, polish_refine_iter, adaptive_rho, scaled_termination;
  double sigma, alpha, rho, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, delta, time_limit;
  long long max_iter;
};
inline OsqpSettingsView osqp_settings_from(const QPSolverParams & prm)
{
  OsqpSettingsView s{};
  s.verbose = prm.verbose; s.sigma = prm.sigma; s.alpha = prm.alpha; s.rho = prm.rho;
  s.eps_abs = prm.eps_abs; s.eps_rel = prm.eps_rel; s.eps_prim_inf = prm.eps_primal_inf; s.eps_dual_inf = prm.eps_dual_inf;
  s.scaling = prm.scaling; s.check_termination = (int)prm.stop_check_iter; s.polish = prm.polish;
  s.polish_refine_iter = (int)prm.polish_iter; s.delta = prm.delta;
  s.adaptive_rho       = 0;  // the reference's solver keeps rho fixed per constraint class (:68)
  s.scaled_termination = 0;  // stopping tests on the unscaled problem
  s.max_iter   = prm.max_iter ? (long long)*prm.max_iter : std::numeric_limits<long long>::max();
  s.time_limit = prm.max_time ? std::chrono::duration<double>(*prm.max_time).count() : 0.0;
  return s;
}

}  // namespace smooth_feedback_amd
