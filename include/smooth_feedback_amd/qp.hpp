// C++ front of the QP path: same names and meaning as smooth::feedback (reference qp.hpp,
// qp_solver.hpp), storage in plain std::vector (no Eigen here), numerics in libsfb.so (HIP).
#pragma once
#include <sfb.h>

#include <chrono>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
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

  sfb_qp_params to_c() const
  {
    sfb_qp_params c;
    sfb_qp_params_default(&c);
    c.verbose = verbose; c.alpha = alpha; c.rho = rho; c.sigma = sigma; c.scaling = scaling;
    c.eps_abs = eps_abs; c.eps_rel = eps_rel; c.eps_primal_inf = eps_primal_inf; c.eps_dual_inf = eps_dual_inf;
    c.max_iter = max_iter ? int64_t(*max_iter) : -1;
    c.max_time_ns = max_time ? int64_t(max_time->count()) : -1;
    c.stop_check_iter = stop_check_iter; c.polish = polish; c.polish_iter = polish_iter; c.delta = delta;
    return c;
  }
};

/// qp.hpp:31-45 (dynamic sizes; P n x n and A m x n column-major like Eigen's default)
struct QuadraticProgram {
  int n = 0, m = 0;
  std::vector<double> P, q, A, l, u;
};

/// qp.hpp:60-79: P CSC, A CSR
struct QuadraticProgramSparse {
  int n = 0, m = 0;
  std::vector<int32_t> P_colptr, P_rowind;
  std::vector<double> P_val;
  std::vector<double> q;
  std::vector<int32_t> A_rowptr, A_colind;
  std::vector<double> A_val;
  std::vector<double> l, u;
};

/// qp.hpp:95-108
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

/// solve_qp for dense problems (qp_solver.hpp:779-787); n + m <= 64 runs on the dense kernels, larger problems on
/// the sparse kernel with a full pattern (include/sfb.h, SFB_QP_DENSE_MAX_K)
inline QPSolution solve_qp(const QuadraticProgram & pbm, const QPSolverParams & prm = {},
                           const QPSolution * warmstart = nullptr)
{
  QPSolution sol;
  sol.primal.resize(pbm.n);
  sol.dual.resize(pbm.m);
  int32_t code       = 6;
  const sfb_qp_params c = prm.to_c();
  sfb_check(sfb_qp_dense_solve_batch_host(&c, 1, pbm.n, pbm.m, pbm.P.data(), pbm.q.data(), pbm.A.data(), pbm.l.data(),
                                          pbm.u.data(), warmstart ? warmstart->primal.data() : nullptr,
                                          warmstart ? warmstart->dual.data() : nullptr, sol.primal.data(),
                                          sol.dual.data(), &sol.objective, &sol.iter, &code));
  sol.code = static_cast<QPSolutionStatus>(code);
  return sol;
}

/// QPSolver<QuadraticProgramSparse>: analyze() once per pattern, solve() for batches sharing it
/// (qp_solver.hpp:242-757, sparse instantiation).
class SparseQPSolver {
public:
  SparseQPSolver() = default;
  explicit SparseQPSolver(const QPSolverParams & prm) : prm_(prm) {}
  SparseQPSolver(const SparseQPSolver &)             = delete;
  SparseQPSolver & operator=(const SparseQPSolver &) = delete;
  ~SparseQPSolver() { sfb_sparse_qp_plan_destroy(plan_); }

  /// qp_solver.hpp:297-338 (+ SimplicialLDLT::analyzePattern :424)
  /// A_keep (nullable, one byte per stored entry of A): 0 = the entry is zero in every problem solved with this
  /// analysis (explicit zeros of dense Jacobian blocks); see sfb_sparse_qp_plan_create_pruned -- checked per item
  /// on the device, a wrong declaration costs time, not correctness.
  void analyze(const QuadraticProgramSparse & pbm, const int32_t * user_perm = nullptr,
               const int32_t * stage = nullptr, const uint8_t * A_keep = nullptr)
  {
    sfb_sparse_qp_plan_destroy(plan_);
    plan_ = nullptr;
    n_ = pbm.n; m_ = pbm.m;
    nnzP_ = (int)pbm.P_val.size(); nnzA_ = (int)pbm.A_val.size();
    sfb_check(sfb_sparse_qp_plan_create_pruned(pbm.n, pbm.m, pbm.P_colptr.data(), pbm.P_rowind.data(),
                                               pbm.A_rowptr.data(), pbm.A_colind.data(), 1, user_perm, stage, A_keep,
                                               &plan_));
  }
  bool analyzed() const { return plan_ != nullptr; }
  /// forget the analysis (the next solve analyses again)
  void reset()
  {
    sfb_sparse_qp_plan_destroy(plan_);
    plan_ = nullptr;
  }
  int64_t nnzL() const
  {
    int64_t v = 0;
    sfb_sparse_qp_plan_info(plan_, nullptr, &v, nullptr);
    return v;
  }

  /// batched solve on host buffers: Px [B][nnzP], q [B][n], Ax [B][nnzA], l,u [B][m]
  void solve_batch(int64_t B, const double * Px, const double * q, const double * Ax, const double * l, const double * u,
                   const double * warm_x, const double * warm_y, double * x, double * y, double * obj, uint32_t * iter,
                   int32_t * code)
  {
    const sfb_qp_params c = prm_.to_c();
    sfb_check(sfb_sparse_qp_solve_batch_host(plan_, &c, B, Px, q, Ax, l, u, warm_x, warm_y, x, y, obj, iter, code));
  }

  /// qp_solver.hpp:343-568 for one problem with the analysed pattern
  QPSolution solve(const QuadraticProgramSparse & pbm, const QPSolution * warmstart = nullptr)
  {
    if (!plan_) analyze(pbm);
    QPSolution sol;
    sol.primal.resize(n_);
    sol.dual.resize(m_);
    int32_t code = 6;
    solve_batch(1, pbm.P_val.data(), pbm.q.data(), pbm.A_val.data(), pbm.l.data(), pbm.u.data(),
                warmstart ? warmstart->primal.data() : nullptr, warmstart ? warmstart->dual.data() : nullptr,
                sol.primal.data(), sol.dual.data(), &sol.objective, &sol.iter, &code);
    sol.code = static_cast<QPSolutionStatus>(code);
    return sol;
  }

  const QPSolverParams & params() const { return prm_; }
  sfb_sparse_qp_plan * plan() { return plan_; }

private:
  QPSolverParams prm_{};
  sfb_sparse_qp_plan * plan_ = nullptr;
  int n_ = 0, m_ = 0, nnzP_ = 0, nnzA_ = 0;
};

}  // namespace smooth_feedback_amd
