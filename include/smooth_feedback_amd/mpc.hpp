// Host side of the MPC path: re-linearise around (xdes, udes), transcribe to a sparse QP in error
// coordinates on an LGR mesh, solve the QP(s) on the GPU through the C-ABI, return udes(0) (+) du_0.
//
// Mirrors smooth::feedback::MPC (reference mpc.hpp:405-519) and the transcription it calls:
//   ocp_to_qp_allocate        ocp_to_qp.hpp:40-114   variable/row layout and sparsity pattern
//   ocp_to_qp_update_cost     :117-195 (+ mesh_integrate, mesh_function.hpp:273-419)  -- ctor only
//   ocp_to_qp_update_dyn      :198-276
//   ocp_to_qp_update_cr       :279-323
//   ocp_to_qp_update_ce       :326-373 with MPCCE (mpc.hpp:275-302)
// Deviations, all on the host: Jacobians of user functions are analytic when the functor has
// `jacobian(...)`, else forward differences with step sqrt(eps) (the reference's default without the
// autodiff header).  The template parameters are the reference's, in its order (mpc.hpp:372-380):
// MPC<T, X, U, F, CR, Kmesh> with T any type with a time_trait (time.hpp: double seconds, std::chrono time points /
// durations as in the reference's time.hpp:25-89) and the number of running constraints read off CR's result type.
// set_weights() stores the weights like the reference does and, like the reference at v1, does NOT
// re-transcribe the cost: P is built once in the constructor (mpc.hpp:423 vs :593-598).
#pragma once
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <atomic>
#include <algorithm>
#include <cmath>
#include <functional>
#include <optional>
#include <limits>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

#include "lie.hpp"
#include "mesh.hpp"
#include "qp.hpp"
#include "time.hpp"

namespace smooth_feedback_amd {

namespace detail {
/// value and right-Jacobians of fn(x, u) -> Vec<NO>: the functor's `jacobian(x, u, dx, du)` when it has one, else forward
/// differences with step sqrt(eps) (the reference's default without the autodiff header).  Host and device.
template<int NO, class Fn, class X, class U>
SFB_LIE_HD void xu_jacobian(const Fn & fn, const X & x, const U & u, Vec<NO> & val, Mat<NO, X::Dof> & dx, Mat<NO, U::Dof> & du)
{
  val = fn(x, u);
  if constexpr (requires { fn.jacobian(x, u, dx, du); }) {
    fn.jacobian(x, u, dx, du);
  } else {
    const double h = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON) = 2^-26
    for (int c = 0; c < X::Dof; ++c) {
      typename X::Tangent e{};
      e[c]          = h;
      const auto v2 = fn(rplus(x, e), u);
      for (int r = 0; r < NO; ++r) dx(r, c) = (v2[r] - val[r]) / h;
    }
    for (int c = 0; c < U::Dof; ++c) {
      typename U::Tangent e{};
      e[c]          = h;
      const auto v2 = fn(x, rplus(u, e));
      for (int r = 0; r < NO; ++r) du(r, c) = (v2[r] - val[r]) / h;
    }
  }
}
/// number of entries of a functor's result (Vec<N> = std::array<double, N>): Ncr of mpc.hpp:383
template<class V>
inline constexpr int result_size_v = (int)std::tuple_size_v<std::remove_cvref_t<V>>;

/// Desired trajectories (detail::XDes / UDes, mpc.hpp:32-56).  Held through a shared_ptr: copies of an MPC share them
/// (mpc.hpp:407, 607-608), so set_xdes / set_udes on a copy are seen by the original.  `generation` counts the setter
/// calls: every controller looks at a new trajectory once (refresh_structure).
template<class T, class X, class U>
struct MPCDes {
  std::function<X(T)> xdes = [](T) { return X::Identity(); };
  std::function<typename X::Tangent(T)> dxdes = [](T) { return typename X::Tangent{}; };
  std::function<U(T)> udes = [](T) { return U::Identity(); };
  uint64_t generation = 0;
};

/// MPCDyn / MPCCR (mpc.hpp:122-147, 230-263): a functor with set_time(T) is told the absolute time of the node first
template<class T, class Fn>
inline void set_time_if(Fn & fn, const T & t)
{
  if constexpr (requires(Fn & fvar, T tvar) { fvar.set_time(tvar); }) fn.set_time(t);
}
template<class T, class Fn>
inline constexpr bool has_set_time_v = requires(std::remove_reference_t<Fn> & fvar, T tvar) { fvar.set_time(tvar); };
}  // namespace detail

/// mpc.hpp:309-333
struct MPCParams {
  std::size_t K{10};
  double tf{1};
  bool warmstart{true};
  QPSolverParams qp{};
  /// (not in the reference) Analyse the QP on the entries of A that are non-zero at the probed linearisations
  /// instead of on everything ocp_to_qp stores (dense Jacobian blocks: block_add, utils/sparse.hpp:33-50).  Exact
  /// (explicit zeros contribute exact zeros) and verified per solve on the device, see analyze_solver().
  bool prune_explicit_zeros{true};
};

/// mpc.hpp:344-356
template<class X, class U>
struct MPCWeights {
  Mat<X::Dof, X::Dof> Q   = Mat<X::Dof, X::Dof>::Identity();
  Mat<X::Dof, X::Dof> Qtf = Mat<X::Dof, X::Dof>::Identity();
  Mat<U::Dof, U::Dof> R   = Mat<U::Dof, U::Dof>::Identity();
};

/// smooth::feedback::MPC, mpc.hpp:372-636: the reference's template parameters in the reference's order (its trailing
/// diff::Type is not a parameter here: Jacobians are the functor's own `jacobian` or forward differences, see the top
/// of this file).  F and CR may be reference types (tests/test_mpc.cpp:69: `MPC<T, X, U, MyDynamics &, ...>`).
/// Copies (mpc.hpp:437-445): a copy is a controller of its own -- own QP, own warm start, own solver that analyses the
/// pattern again at its first solve like a copied QPSolver does (qp_solver.hpp:209-231) -- and SHARES the desired
/// trajectories with the original (mpc.hpp:407, 607-608).  Swarms share ONE analysis explicitly (MPCSwarm* take the
/// prototype by reference).
template<Time T, class X, class U, class F, class CR, std::size_t Kmesh_ = 4>
class MPC {
public:
  static constexpr int Kmesh = (int)Kmesh_;
  static constexpr int Nx = X::Dof, Nu = U::Dof;
  /// mpc.hpp:383
  static constexpr int Ncr = detail::result_size_v<std::invoke_result_t<std::remove_reference_t<CR> &, const X &, const U &>>;
  using TangentX = Vec<Nx>;
  using TimeT    = T;
  /// absolute time t plus a horizon offset in seconds
  static T tplus(const T & t, double s) { return time_trait<T>::plus(t, s); }

  /// mpc.hpp:405-434 (by value: serves the reference's rvalue and lvalue constructors alike).  `w` (not in the
  /// reference's constructor): the weights to transcribe -- the reference transcribes its defaults here and
  /// set_weights() never reaches the QP (see set_weights).
  MPC(F f, CR cr, Vec<Ncr> crl, Vec<Ncr> cru, MPCParams prm = {}, MPCWeights<X, U> w = {})
      : f_(std::forward<F>(f)), cr_(std::forward<CR>(cr)), crl_(crl), cru_(cru), prm_(std::move(prm)),
        mesh_(int((prm_.K + Kmesh - 1) / Kmesh), Kmesh), des_(std::make_shared<Des>()), solver_(prm_.qp), weights_(w)
  {
    allocate(w);
  }
  /// mpc.hpp:435-447: default constructor, default copies and moves
  MPC()                        = default;
  MPC(const MPC &)             = default;
  MPC(MPC &&)                  = default;
  MPC & operator=(const MPC &) = default;
  MPC & operator=(MPC &&)      = default;
  ~MPC()                       = default;

  // ---- desired trajectories, mpc.hpp:520-586 ----
  /// absolute time: x_des(t) and its body velocity dx_des(t)
  void set_xdes(std::function<X(T)> x_des, std::function<TangentX(T)> dx_des)
  {
    des_->xdes  = std::move(x_des);
    des_->dxdes = std::move(dx_des);
    structure_changed();
  }
  /// absolute time, derivative by central differences of x(t) (the reference autodiffs / finite-differences x(t))
  void set_xdes(std::function<X(T)> x_des)
  {
    auto xd     = x_des;
    des_->dxdes = [xd](T t) {
      const double h = 1e-6;
      TangentX d     = rminus(xd(tplus(t, h)), xd(tplus(t, -h)));
      for (auto & v : d) v /= (2 * h);
      return d;
    };
    des_->xdes = std::move(x_des);
    structure_changed();
  }
  void set_udes(std::function<U(T)> u_des)
  {
    des_->udes = std::move(u_des);
    structure_changed();
  }
  /// relative time, mpc.hpp:539-545: u_des(t) = f(t - t0) with f: double (seconds) -> U
  template<class Fun>
    requires std::is_same_v<std::invoke_result_t<Fun, double>, U>
  void set_udes_rel(Fun && f, T t0 = T(0))
  {
    set_udes([t0, f = std::forward<Fun>(f)](T t_abs) -> U { return f(time_trait<T>::minus(t_abs, t0)); });
  }
  /// relative time, mpc.hpp:572-586: x_des(t) = f(t - t0); the body velocity is the derivative of f (here: central
  /// differences with step 1e-6 s, or f.velocity(t_rel) when the functor provides it)
  template<class Fun>
    requires std::is_same_v<std::invoke_result_t<Fun, double>, X>
  void set_xdes_rel(Fun && f, T t0 = T(0))
  {
    std::function<X(T)> xd = [t0, f](T t_abs) -> X { return f(time_trait<T>::minus(t_abs, t0)); };
    std::function<TangentX(T)> dxd = [t0, f](T t_abs) -> TangentX {
      const double tr = time_trait<T>::minus(t_abs, t0);
      if constexpr (requires { { f.velocity(tr) } -> std::convertible_to<TangentX>; }) {
        return f.velocity(tr);
      } else {
        const double h = 1e-6;
        TangentX d     = rminus(f(tr + h), f(tr - h));
        for (auto & v : d) v /= (2 * h);
        return d;
      }
    };
    set_xdes(std::move(xd), std::move(dxd));
  }
  /// mpc.hpp:593-598.  As in the reference at v1 the new weights are stored but never reach the QP: the cost block P is
  /// transcribed once, in the constructor (mpc.hpp:423), and operator() only updates the dynamics / constraint rows.
  /// Construct a new MPC (or pass the weights to the constructor) to control with other weights.
  void set_weights(const MPCWeights<X, U> & w) { weights_ = w; }
  const MPCWeights<X, U> & weights() const { return weights_; }
  void reset_warmstart() { warm_.reset(); }

  // ---- sizes / pattern ----
  int N() const { return mesh_.N_colloc(); }
  int nvar() const { return Nx * (N() + 1) + Nu * N(); }
  int ncon() const { return Nx * N() + Ncr * N() + Nx; }
  int uvar_B() const { return Nx * (N() + 1); }
  const QuadraticProgramSparse<> & qp() const { return qp_; }
  const Mesh & mesh() const { return mesh_; }
  const MPCParams & params() const { return prm_; }
  SparseQPSolver & solver() { return solver_; }
  /// Elimination stages for the solver's constrained minimum-degree order (unknowns of a lower stage are eliminated
  /// before any of a higher stage).  Stage 0: everything interior to a mesh interval -- the intervals decouple once
  /// the states they share (x at nodes 0, Kmesh, 2 Kmesh, ..., N: the separators) are held back.  The separators,
  /// a block-tridiagonal chain after the interiors are gone, are ordered by NESTED DISSECTION of the chain (middle
  /// separator last, recursively): the dependent chain of the triangular sweeps then grows with the logarithm of
  /// the number of intervals instead of linearly (headline problem: 161 + 119 sweep steps instead of 213 + 189 for
  /// 8 % more entries in L).  The rows pinning x_0 go with separator 0.  nested == false: all separators in one
  /// stage (eliminated as a chain).
  std::vector<int32_t> elimination_stage(bool nested = true) const
  {
    std::vector<int32_t> st(nvar() + ncon(), 0);
    const int nsep = N() / Kmesh + 1;
    std::vector<int32_t> lv(nsep, 1);
    if (nested) {
      int depth = 1;
      while ((1 << depth) - 1 < nsep) ++depth;  // levels of the dissection tree
      std::function<void(int, int, int)> rec = [&](int lo, int hi, int level) {
        if (lo > hi) return;
        const int mid = (lo + hi) / 2;
        lv[mid]       = level;
        rec(lo, mid - 1, level - 1);
        rec(mid + 1, hi, level - 1);
      };
      rec(0, nsep - 1, depth);
    }
    for (int s = 0; s < nsep; ++s)
      for (int c = 0; c < Nx; ++c) st[s * Kmesh * Nx + c] = lv[s];
    for (int d = 0; d < Nx; ++d) st[nvar() + cecon_B() + d] = lv[0];
    return st;
  }
  /// OR the non-zero entries of A at (t, x) into keep (one byte per stored entry of A).
  void probe_structure(const T & t, const X & x, std::vector<uint8_t> & keep) const
  {
    std::vector<double> Av(qp_.A_val.size()), lv(qp_.m), uv(qp_.m);
    assemble(t, x, Av.data(), lv.data(), uv.data());
    keep.resize(Av.size(), 0);
    for (size_t e = 0; e < Av.size(); ++e) keep[e] |= !(Av[e] == 0.0);
  }
  /// probe at a few ticks after t from perturbed states on the desired trajectory (deterministic)
  void probe_default(const T & t, std::vector<uint8_t> & keep) const
  {
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (int s = 0; s < 6; ++s) {
      const T ts = tplus(t, s * prm_.tf / double(N()));
      TangentX xi{};
      for (auto & v : xi) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        v   = (double(lcg >> 11) / 9007199254740992.0 - 0.5);
      }
      probe_structure(ts, rplus(des_->xdes(ts), xi), keep);
    }
  }
  void probe_values(const double * Aval, std::vector<uint8_t> & keep) const
  {
    keep.resize(qp_.A_val.size(), 0);
    for (size_t e = 0; e < keep.size(); ++e) keep[e] |= !(Aval[e] == 0.0);
  }
  /// Symbolic analysis of the QP (once).  `keep` (nullable): the stored entries of A seen non-zero in a sample of
  /// the linearisations this controller will solve (probe_structure); everything else is declared an explicit
  /// zero and left out of the KKT pattern.  The declaration is checked on the device for every solve, and a
  /// problem that violates it is solved on the whole pattern (sfb_sparse_qp_plan_create_pruned).
  void analyze_solver(const std::vector<uint8_t> * keep = nullptr)
  {
    if (!solver_.analyzed()) {
      const auto st = elimination_stage();
      solver_.analyze(qp_, nullptr, st.data(), (keep && prm_.prune_explicit_zeros) ? keep->data() : nullptr);
      if (keep && prm_.prune_explicit_zeros) keep_analysed_ = *keep;
      else keep_analysed_.clear();
      seen_generation_ = des_->generation;
    }
  }
  /// After set_xdes / set_udes the linearisations may have non-zeros where the analysed ones had explicit zeros.  The
  /// device checks the declaration for every solve anyway (a violating problem is solved on the whole pattern: slower,
  /// never wrong), so nothing HAS to happen; this looks at the new trajectory once, at the next solve, and analyses
  /// again only if it really leaves the analysed structure -- and never while a device-resident swarm holds the plan
  /// (QPSolver::pin_plan: the swarm keeps the raw plan pointer, sfb.h).  Aval (nullable): values of A to probe as well.
  void refresh_structure(const T & t, const double * Aval = nullptr)
  {
    if (!structure_dirty() || !solver_.analyzed()) return;
    seen_generation_ = des_->generation;
    if (!prm_.prune_explicit_zeros || keep_analysed_.empty() || solver_.plan_pinned()) return;
    std::vector<uint8_t> keep;
    if (Aval) probe_values(Aval, keep);
    probe_default(t, keep);
    bool inside = true;
    for (size_t e = 0; e < keep.size() && inside; ++e) inside = !keep[e] || keep_analysed_[e];
    if (inside) return;
    for (size_t e = 0; e < keep.size(); ++e) keep[e] |= keep_analysed_[e];
    solver_.reset();
    analyze_solver(&keep);
  }

  /// Numeric part of MPC::operator() before the solve (mpc.hpp:473-486): writes the values of A (in
  /// the pattern of qp().A_*), l and u for current time t and state x.  Thread-safe (const).
  /// Functors with set_time(T) (MPCDyn / MPCCR, mpc.hpp:131, 247) are told the absolute time of every node: here on
  /// COPIES of the functors (const, callable from many threads); operator() tells the controller's own functors.
  void assemble(const T & t, const X & x, double * Aval, double * l, double * u) const
  {
    if constexpr (kTimedFunctors) {
      std::remove_cvref_t<F> fc   = f_;
      std::remove_cvref_t<CR> crc = cr_;
      assemble_with(fc, crc, t, x, Aval, l, u);
    } else {
      assemble_with(f_, cr_, t, x, Aval, l, u);
    }
  }

private:
  template<class Fn, class CRn>
  void assemble_with(Fn & fn, CRn & crn, const T & t, const X & x, double * Aval, double * l, double * u) const
  {
    const int Nn = N();
    const double tf = prm_.tf;
    for (int p = 0; p < (int)qp_.A_val.size(); ++p) Aval[p] = 0.0;  // set_zero(A.middleRows(...)), :236
    // --- ocp_to_qp_update_dyn :240-275 ---
    for (int s = 0, M = 0; s < mesh_.N_ivals(); M += Kmesh, ++s) {
      const double alpha = mesh_.alpha(s);
      for (int i = 0; i < Kmesh; ++i) {
        const int node   = M + i;
        const double t_i = tf * mesh_.node(node);
        const X xl       = des_->xdes(tplus(t, t_i));
        const TangentX dxl = des_->dxdes(tplus(t, t_i));
        const U ul       = des_->udes(tplus(t, t_i));
        Vec<Nx> fv;
        Mat<Nx, Nx> dfdx;
        Mat<Nx, Nu> dfdu;
        detail::set_time_if<T>(fn, tplus(t, t_i));
        detail::xu_jacobian<Nx>(fn, xl, ul, fv, dfdx, dfdu);
        Mat<Nx, Nx> adm{};
        if constexpr (!X::IsCommutative) {
          TangentX s2;
          for (int d = 0; d < Nx; ++d) s2[d] = fv[d] + dxl[d];
          adm = X::ad(s2);
        }
        for (int d = 0; d < Nx; ++d) {
          const int row = dcon_B() + node * Nx + d;
          int p         = qp_.A_rowptr[row];
          for (int j = 0; j <= Kmesh; ++j) {  // x blocks of the interval, ascending column order
            const double dc = alpha * mesh_.D(j, i);
            if (j == i) {
              for (int c = 0; c < Nx; ++c) {
                double v = 0.0;
                v += tf * dfdx(d, c);                                       // :258
                if constexpr (!X::IsCommutative) v += (-tf / 2) * adm(d, c);  // :262-264
                if (c == d) v -= dc;                                        // :266-270
                Aval[p++] = v;
              }
            } else {
              double v = 0.0;
              v -= dc;
              Aval[p++] = v;
            }
          }
          for (int c = 0; c < Nu; ++c) Aval[p++] = 0.0 + tf * dfdu(d, c);  // :259
          l[row] = -tf * (fv[d] - dxl[d]);                                // :272-273
          u[row] = l[row];
        }
      }
    }
    // --- ocp_to_qp_update_cr :279-323 (always evaluated here; the reference skips it when cr has no
    //     set_time, which leaves the constructor-time values -- identical for time-invariant cr) ---
    for (int node = 0; node < Nn; ++node) {
      const double t_i = tf * mesh_.node(node);
      const X xl       = des_->xdes(tplus(t, t_i));
      const U ul       = des_->udes(tplus(t, t_i));
      Vec<Ncr> cv;
      Mat<Ncr, Nx> dcdx;
      Mat<Ncr, Nu> dcdu;
      detail::set_time_if<T>(crn, tplus(t, t_i));
      detail::xu_jacobian<Ncr>(crn, xl, ul, cv, dcdx, dcdu);
      for (int d = 0; d < Ncr; ++d) {
        const int row = crcon_B() + node * Ncr + d;
        int p         = qp_.A_rowptr[row];
        for (int c = 0; c < Nx; ++c) Aval[p++] = dcdx(d, c);
        for (int c = 0; c < Nu; ++c) Aval[p++] = dcdu(d, c);
        l[row] = crl_[d] - cv[d];  // :321
        u[row] = cru_[d] - cv[d];  // :322
      }
    }
    // --- ocp_to_qp_update_ce :326-373 with MPCCE: ce = x0 (-) x0_fix linearised at xl(0) ---
    {
      const X xl0      = des_->xdes(t);
      const TangentX e = rminus(xl0, x);  // MPCCE::operator(), mpc.hpp:288-291
      const auto J     = X::dr_expinv(e); // MPCCE::jacobian,  mpc.hpp:293-301
      for (int d = 0; d < Nx; ++d) {
        const int row = cecon_B() + d;
        int p         = qp_.A_rowptr[row];
        for (int c = 0; c < Nx; ++c) Aval[p++] = J(d, c);
        l[row] = 0.0 - e[d];  // :371  cel - ceval
        u[row] = 0.0 - e[d];  // :372
      }
    }
  }

public:
  /// Description of this transcription for the device-side assembly (sfb_mpc_layout, sfb.h); owns the arrays.
  struct DeviceLayout {
    std::vector<double> alpha, D, crl, cru;
    std::vector<int32_t> kind, dof;
    std::vector<uint8_t> jac_keep;  // empty: unpacked records
    sfb_mpc_layout c{};
  };
  /// jac_keep (nullable): flags of the Jacobian entries the records carry (RecordPacking::keep)
  std::unique_ptr<DeviceLayout> device_layout(const std::vector<uint8_t> * jac_keep = nullptr) const
  {
    auto L = std::make_unique<DeviceLayout>();
    if (jac_keep) L->jac_keep = *jac_keep;
    for (int s = 0; s < mesh_.N_ivals(); ++s) L->alpha.push_back(mesh_.alpha(s));
    L->D.resize((size_t)(Kmesh + 1) * Kmesh);
    for (int j = 0; j <= Kmesh; ++j)
      for (int i = 0; i < Kmesh; ++i) L->D[(size_t)j * Kmesh + i] = mesh_.D(j, i);
    for (int d = 0; d < Ncr; ++d) { L->crl.push_back(crl_[d]); L->cru.push_back(cru_[d]); }
    if constexpr (!X::IsCommutative) LieParts<X>::append(L->kind, L->dof);
    L->c = sfb_mpc_layout{Nx, Nu, Ncr, Kmesh, mesh_.N_ivals(), prm_.tf, L->alpha.data(), L->D.data(),
                          (int32_t)L->kind.size(), L->kind.data(), L->dof.data(), L->crl.data(), L->cru.data(),
                          L->jac_keep.empty() ? nullptr : L->jac_keep.data()};
    return L;
  }

  /// Packed records (sfb_mpc_layout::jac_keep): the dense Jacobian blocks of a bundle state are mostly structural
  /// zeros, and the record is what a device-resident swarm moves through PCIe every tick.  keep: one flag per entry of
  /// [dfdx Nx*Nx | dfdu Nx*Nu | dcdx Ncr*Nx | dcdu Ncr*Nu | J Nx*Nx], row-major.
  struct RecordPacking {
    static constexpr int kFlags = 2 * Nx * Nx + Nx * Nu + Ncr * Nx + Ncr * Nu;
    std::vector<uint8_t> keep = std::vector<uint8_t>(kFlags, 0);
    int n_fx = 0, n_fu = 0, n_cx = 0, n_cu = 0, n_J = 0;
    void count()
    {
      auto cnt = [&](int off, int len) { int c = 0; for (int e = 0; e < len; ++e) c += keep[off + e] != 0; return c; };
      n_fx = cnt(0, Nx * Nx); n_fu = cnt(Nx * Nx, Nx * Nu); n_cx = cnt(Nx * Nx + Nx * Nu, Ncr * Nx);
      n_cu = cnt(Nx * Nx + Nx * Nu + Ncr * Nx, Ncr * Nu); n_J = cnt(Nx * Nx + Nx * Nu + Ncr * Nx + Ncr * Nu, Nx * Nx);
    }
    int64_t doubles(int Nn) const { return (int64_t)Nn * (2 * Nx + n_fx + n_fu + Ncr + n_cx + n_cu) + Nx + n_J; }
  };
  /// marks the Jacobian entries that are non-zero in the full record `rec`
  void probe_record(const double * rec, RecordPacking & pk) const
  {
    const int Nn = N();
    const double *dfx = rec + 2 * Nn * Nx, *dfu = dfx + Nn * Nx * Nx, *dcx = dfu + Nn * Nx * Nu + Nn * Ncr,
                 *dcu = dcx + Nn * Ncr * Nx, *Jm = dcu + Nn * Ncr * Nu + Nx;
    uint8_t * k = pk.keep.data();
    auto mark = [&](const double * blk, int blocks, int len) {
      for (int b = 0; b < blocks; ++b)
        for (int e = 0; e < len; ++e) k[e] |= !(blk[(size_t)b * len + e] == 0.0);
      k += len;
    };
    mark(dfx, Nn, Nx * Nx); mark(dfu, Nn, Nx * Nu); mark(dcx, Nn, Ncr * Nx); mark(dcu, Nn, Ncr * Nu); mark(Jm, 1, Nx * Nx);
    pk.count();
  }
  /// like probe_default: a few ticks after t, perturbed states on the desired trajectory (deterministic)
  RecordPacking probe_record_default(const T & t) const
  {
    RecordPacking pk;
    std::vector<double> rec((size_t)record_doubles(N()));
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (int s = 0; s < 6; ++s) {
      const T ts = tplus(t, s * prm_.tf / double(N()));
      TangentX xi{};
      for (auto & v : xi) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        v   = (double(lcg >> 11) / 9007199254740992.0 - 0.5);
      }
      fill_record(ts, rplus(des_->xdes(ts), xi), rec.data());
      probe_record(rec.data(), pk);
    }
    return pk;
  }
  /// full record -> packed record; false if an entry outside the flags is not zero (the packing does not fit this
  /// linearisation: the caller falls back to unpacked records)
  bool pack_record(const double * rec, double * out, const RecordPacking & pk) const
  {
    const int Nn = N();
    bool ok = true;
    const double * src = rec;
    const uint8_t * k  = pk.keep.data();
    auto copy = [&](int len) { for (int e = 0; e < len; ++e) *out++ = *src++; };
    auto pack = [&](int blocks, int len) {
      for (int b = 0; b < blocks; ++b)
        for (int e = 0; e < len; ++e, ++src) {
          if (k[e]) *out++ = *src;
          else ok = ok && (*src == 0.0);
        }
      k += len;
    };
    copy(2 * Nn * Nx);                                   // f, dxdes
    pack(Nn, Nx * Nx); pack(Nn, Nx * Nu);                // dfdx, dfdu
    copy(Nn * Ncr);                                      // c
    pack(Nn, Ncr * Nx); pack(Nn, Ncr * Nu);              // dcdx, dcdu
    copy(Nx);                                            // e
    pack(1, Nx * Nx);                                    // J
    return ok;
  }
  static constexpr int64_t record_doubles(int Nn)
  {
    return (int64_t)Nn * (2 * Nx + Nx * Nx + Nx * Nu + Ncr + Ncr * Nx + Ncr * Nu) + Nx + Nx * Nx;
  }
  /// The part of assemble() that needs the user's callbacks: linearisation of dynamics, running constraint
  /// and initial-state constraint at (xdes, udes) for time t and state x, as one record of
  /// sfb_mpc_assemble_batch (layout in sfb.h; matrices row-major).  Thread-safe (const).
  void fill_record(const T & t, const X & x, double * rec) const
  {
    if constexpr (kTimedFunctors) {
      std::remove_cvref_t<F> fc   = f_;
      std::remove_cvref_t<CR> crc = cr_;
      fill_record_with(fc, crc, t, x, rec);
    } else {
      fill_record_with(f_, cr_, t, x, rec);
    }
  }

private:
  template<class Fn, class CRn>
  void fill_record_with(Fn & fn, CRn & crn, const T & t, const X & x, double * rec) const
  {
    const int Nn = N();
    const double tf = prm_.tf;
    double *f = rec, *dx = f + Nn * Nx, *dfx = dx + Nn * Nx, *dfu = dfx + Nn * Nx * Nx, *cc = dfu + Nn * Nx * Nu,
           *dcx = cc + Nn * Ncr, *dcu = dcx + Nn * Ncr * Nx, *e0 = dcu + Nn * Ncr * Nu, *Jm = e0 + Nx;
    for (int node = 0; node < Nn; ++node) {
      const double t_i   = tf * mesh_.node(node);
      const X xl         = des_->xdes(tplus(t, t_i));
      const TangentX dxl = des_->dxdes(tplus(t, t_i));
      const U ul         = des_->udes(tplus(t, t_i));
      Vec<Nx> fv;
      Mat<Nx, Nx> dfdx;
      Mat<Nx, Nu> dfdu;
      detail::set_time_if<T>(fn, tplus(t, t_i));
      detail::xu_jacobian<Nx>(fn, xl, ul, fv, dfdx, dfdu);
      Vec<Ncr> cv;
      Mat<Ncr, Nx> dcdx;
      Mat<Ncr, Nu> dcdu;
      detail::set_time_if<T>(crn, tplus(t, t_i));
      detail::xu_jacobian<Ncr>(crn, xl, ul, cv, dcdx, dcdu);
      for (int d = 0; d < Nx; ++d) {
        f[node * Nx + d]  = fv[d];
        dx[node * Nx + d] = dxl[d];
        for (int c = 0; c < Nx; ++c) dfx[(node * Nx + d) * Nx + c] = dfdx(d, c);
        for (int c = 0; c < Nu; ++c) dfu[(node * Nx + d) * Nu + c] = dfdu(d, c);
      }
      for (int d = 0; d < Ncr; ++d) {
        cc[node * Ncr + d] = cv[d];
        for (int c = 0; c < Nx; ++c) dcx[(node * Ncr + d) * Nx + c] = dcdx(d, c);
        for (int c = 0; c < Nu; ++c) dcu[(node * Ncr + d) * Nu + c] = dcdu(d, c);
      }
    }
    const X xl0      = des_->xdes(t);
    const TangentX e = rminus(xl0, x);
    const auto J     = X::dr_expinv(e);
    for (int d = 0; d < Nx; ++d) {
      e0[d] = e[d];
      for (int c = 0; c < Nx; ++c) Jm[d * Nx + c] = J(d, c);
    }
  }

public:
  /// mpc.hpp:518   udes(0) (+) primal[uvar_B : +Nu]
  U input_from_primal(const T & t, const double * primal) const
  {
    typename U::Tangent du{};
    for (int c = 0; c < Nu; ++c) du[c] = primal[uvar_B() + c];
    return rplus(des_->udes(t), du);
  }

  U input_from_du0(const T & t, const double * du0) const
  {
    typename U::Tangent du{};
    for (int c = 0; c < Nu; ++c) du[c] = du0[c];
    return rplus(des_->udes(t), du);
  }

  /// MPC::operator(), mpc.hpp:458-519, with the reference's signature
  std::pair<U, QPSolutionStatus> operator()(const T & t, const X & x,
                                            std::optional<std::reference_wrapper<std::vector<U>>> u_traj = std::nullopt,
                                            std::optional<std::reference_wrapper<std::vector<X>>> x_traj = std::nullopt)
  {
    return solve_tick(t, x, u_traj ? &u_traj->get() : nullptr, x_traj ? &x_traj->get() : nullptr);
  }
  /// the same with nullable pointers (u_traj has no default here: `mpc(t, x)` is the reference's overload)
  std::pair<U, QPSolutionStatus> operator()(const T & t, const X & x, std::vector<U> * u_traj, std::vector<X> * x_traj = nullptr)
  {
    return solve_tick(t, x, u_traj, x_traj);
  }

private:
  std::pair<U, QPSolutionStatus> solve_tick(const T & t, const X & x, std::vector<U> * u_traj, std::vector<X> * x_traj)
  {
    assemble_with(f_, cr_, t, x, qp_.A_val.data(), qp_.l.data(), qp_.u.data());  // the controller's own functors see set_time
    refresh_structure(t, qp_.A_val.data());
    if (!solver_.analyzed()) {
      std::vector<uint8_t> keep;
      probe_values(qp_.A_val.data(), keep);
      probe_default(t, keep);
      analyze_solver(&keep);
    }
    const QPSolution<> sol = solver_.solve(qp_, warm_ ? &*warm_ : nullptr);  // :491
    const int Nn = N();
    if (u_traj) {  // :494-500
      u_traj->resize(Nn);
      for (int i = 0; i < Nn; ++i) {
        typename U::Tangent du{};
        for (int c = 0; c < Nu; ++c) du[c] = sol.primal[uvar_B() + i * Nu + c];
        (*u_traj)[i] = rplus(des_->udes(tplus(t, prm_.tf * mesh_.node(i))), du);
      }
    }
    if (x_traj) {  // :501-507
      x_traj->resize(Nn + 1);
      for (int i = 0; i <= Nn; ++i) {
        TangentX dx{};
        for (int c = 0; c < Nx; ++c) dx[c] = sol.primal[i * Nx + c];
        (*x_traj)[i] = rplus(des_->xdes(tplus(t, prm_.tf * mesh_.node(i))), dx);
      }
    }
    if (prm_.warmstart &&
        (sol.code == QPSolutionStatus::Optimal || sol.code == QPSolutionStatus::MaxTime ||
         sol.code == QPSolutionStatus::MaxIterations))
      warm_ = sol;  // :510-516
    return {input_from_primal(t, sol.primal.data()), sol.code};
  }

  // a new linearisation trajectory may have other explicit zeros: looked at by refresh_structure() at the next solve
  // of every controller that shares it
  void structure_changed() { ++des_->generation; }
  bool structure_dirty() const { return des_ && des_->generation != seen_generation_; }
  int dcon_B() const { return 0; }
  int crcon_B() const { return Nx * N(); }
  int cecon_B() const { return Nx * N() + Ncr * N(); }

  // ocp_to_qp_allocate (:40-114) + constructor-time cost (:117-195; MPCIntegrand/MPCObj hessians,
  // mpc.hpp:110-114, :198-227)
  void allocate(const MPCWeights<X, U> & w)
  {
    const int Nn = N();
    qp_.n = nvar();
    qp_.m = ncon();
    qp_.q.assign(qp_.n, 0.0);  // dF of the integrand and dtheta/dx vanish at the linearisation point
    qp_.l.assign(qp_.m, 0.0);
    qp_.u.assign(qp_.m, 0.0);
    // A pattern, CSR, rows [dyn | cr | ce], columns [x_0..x_N | u_0..u_{N-1}]  (:56, :63-69)
    qp_.A_rowptr.assign(1, 0);
    for (int s = 0, M = 0; s < mesh_.N_ivals(); M += Kmesh, ++s)
      for (int i = 0; i < Kmesh; ++i)
        for (int d = 0; d < Nx; ++d) {
          for (int j = 0; j <= Kmesh; ++j) {
            if (j == i)
              for (int c = 0; c < Nx; ++c) qp_.A_colind.push_back((M + j) * Nx + c);
            else
              qp_.A_colind.push_back((M + j) * Nx + d);
          }
          for (int c = 0; c < Nu; ++c) qp_.A_colind.push_back(uvar_B() + (M + i) * Nu + c);
          qp_.A_rowptr.push_back((int)qp_.A_colind.size());
        }
    for (int node = 0; node < Nn; ++node)
      for (int d = 0; d < Ncr; ++d) {
        for (int c = 0; c < Nx; ++c) qp_.A_colind.push_back(node * Nx + c);
        for (int c = 0; c < Nu; ++c) qp_.A_colind.push_back(uvar_B() + node * Nu + c);
        qp_.A_rowptr.push_back((int)qp_.A_colind.size());
      }
    for (int d = 0; d < Nx; ++d) {
      for (int c = 0; c < Nx; ++c) qp_.A_colind.push_back(c);
      qp_.A_rowptr.push_back((int)qp_.A_colind.size());
    }
    qp_.A_val.assign(qp_.A_colind.size(), 0.0);

    // P (upper triangle, CSC): per node  w_i*tf*Q on x_i, w_i*tf*R on u_i (R entries guarded by
    // Q(i,j) != 0 exactly like mpc.hpp:219-223); x_0 block += 0.5*Qtf (MPCObj writes Qtf into the
    // x0 block, mpc.hpp:110-114, ocp_to_qp.hpp:189); x_N has no stored entry.
    std::vector<std::vector<std::pair<int, double>>> cols(qp_.n);
    auto add = [&](int r, int c, double v) {
      if (r > c) return;  // upper_only
      for (auto & e : cols[c])
        if (e.first == r) { e.second += v; return; }
      cols[c].push_back({r, v});
    };
    for (int i = 0; i < Nn; ++i) {
      const double sc = (mesh_.weight(i) * 1.0) * (prm_.tf - 0.0);  // wl * (tf - t0)
      for (int c = 0; c < Nx; ++c)
        for (int r = 0; r < Nx; ++r)
          if (w.Q(r, c) != 0) add(i * Nx + r, i * Nx + c, sc * w.Q(r, c));
      // (the reference guards the R entries with Q(r, c) != 0, mpc.hpp:219-223: kept for bit parity of the pattern;
      //  where Q has no such entry -- Nu > Nx -- the R entry is kept instead of reading out of range)
      for (int c = 0; c < Nu; ++c)
        for (int r = 0; r < Nu; ++r)
          if (r >= Nx || c >= Nx || w.Q(r, c) != 0) add(uvar_B() + i * Nu + r, uvar_B() + i * Nu + c, sc * w.R(r, c));
    }
    for (int c = 0; c < Nx; ++c)
      for (int r = 0; r < Nx; ++r)
        if (w.Qtf(r, c) != 0) add(r, c, 0.5 * w.Qtf(r, c));
    qp_.P_colptr.assign(1, 0);
    for (int c = 0; c < qp_.n; ++c) {
      std::sort(cols[c].begin(), cols[c].end());
      for (auto & e : cols[c]) {
        qp_.P_rowind.push_back(e.first);
        qp_.P_val.push_back(e.second);
      }
      qp_.P_colptr.push_back((int)qp_.P_rowind.size());
    }
  }

  using Des = detail::MPCDes<T, X, U>;
  static constexpr bool kTimedFunctors = detail::has_set_time_v<T, F> || detail::has_set_time_v<T, CR>;

  F f_;
  CR cr_;
  Vec<Ncr> crl_{}, cru_{};
  MPCParams prm_{};
  Mesh mesh_{};
  std::shared_ptr<Des> des_ = std::make_shared<Des>();  // shared by copies, mpc.hpp:407
  QuadraticProgramSparse<> qp_;
  SparseQPSolver solver_;     // a copy drops the analysis (qp.hpp PlanHolder, qp_solver.hpp:209-231)
  MPCWeights<X, U> weights_{};
  std::vector<uint8_t> keep_analysed_;  // the A_keep mask of the current analysis (empty: whole pattern)
  uint64_t seen_generation_ = 0;        // des_->generation this controller's analysis has looked at
  std::optional<QPSolution<>> warm_;
};

/// A swarm of agents running the same MPC (same model, horizon and weights => same QP pattern),
/// each with its own time and state: host threads assemble, ONE batched GPU solve.
template<class MPCT>
class MPCSwarm {
public:
  explicit MPCSwarm(MPCT & proto, int64_t agents, int threads = 0)
      : mpc_(proto), B_(agents), threads_(threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency()))
  {
    const auto & qp = mpc_.qp();
    nA_ = (int)qp.A_val.size();
    nP_ = (int)qp.P_val.size();
    Px_.resize((size_t)B_ * nP_);
    q_.assign((size_t)B_ * qp.n, 0.0);
    Ax_.resize((size_t)B_ * nA_);
    l_.resize((size_t)B_ * qp.m);
    u_.resize((size_t)B_ * qp.m);
    x_.resize((size_t)B_ * qp.n);
    y_.resize((size_t)B_ * qp.m);
    iter_.resize(B_);
    code_.resize(B_);
    for (int64_t b = 0; b < B_; ++b) std::copy(qp.P_val.begin(), qp.P_val.end(), Px_.begin() + (size_t)b * nP_);
  }

  /// one control tick for all agents: returns inputs and per-agent status codes
  template<class XT, class UT>
  void step(const std::vector<typename MPCT::TimeT> & t, const std::vector<XT> & xs, std::vector<UT> & us,
            std::vector<QPSolutionStatus> & codes)
  {
    const auto & qp = mpc_.qp();
    if (B_ > 0) mpc_.refresh_structure(t[0]);  // after set_xdes / set_udes on the prototype
    parallel_for([&](int64_t b) {
      mpc_.assemble(t[b], xs[b], &Ax_[(size_t)b * nA_], &l_[(size_t)b * qp.m], &u_[(size_t)b * qp.m]);
    });
    if (!mpc_.solver().analyzed()) {  // structure from a sample of this batch (up to 64 agents, evenly spaced)
      std::vector<uint8_t> keep;
      const int64_t S = std::min<int64_t>(B_, 64);
      for (int64_t s = 0; s < S; ++s) mpc_.probe_values(&Ax_[(size_t)(s * B_ / S) * nA_], keep);
      mpc_.analyze_solver(&keep);
    }
    // an agent without a stored solution starts from zeros, which IS the cold start (qp_solver.hpp:436-445)
    const bool warm = mpc_.params().warmstart;
    if (warm && wx_.empty()) {
      wx_.assign((size_t)B_ * qp.n, 0.0);
      wy_.assign((size_t)B_ * qp.m, 0.0);
    }
    mpc_.solver().solve_batch(B_, Px_.data(), q_.data(), Ax_.data(), l_.data(), u_.data(), warm ? wx_.data() : nullptr,
                              warm ? wy_.data() : nullptr, x_.data(), y_.data(), nullptr, iter_.data(), code_.data());
    us.resize(B_);
    codes.resize(B_);
    for (int64_t b = 0; b < B_; ++b) {
      us[b]    = mpc_.input_from_primal(t[b], &x_[(size_t)b * qp.n]);
      codes[b] = static_cast<QPSolutionStatus>(code_[b]);
      if (warm && (code_[b] == 0 || code_[b] == 4 || code_[b] == 5)) {  // mpc.hpp:510-516; otherwise the older one stays
        std::copy(x_.begin() + (size_t)b * qp.n, x_.begin() + (size_t)(b + 1) * qp.n, wx_.begin() + (size_t)b * qp.n);
        std::copy(y_.begin() + (size_t)b * qp.m, y_.begin() + (size_t)(b + 1) * qp.m, wy_.begin() + (size_t)b * qp.m);
      }
    }
  }
  const std::vector<uint32_t> & iterations() const { return iter_; }
  const std::vector<double> & Ax() const { return Ax_; }
  const std::vector<double> & l() const { return l_; }
  const std::vector<double> & u() const { return u_; }
  const std::vector<double> & Px() const { return Px_; }
  const std::vector<double> & primal() const { return x_; }

private:
  template<class Fn>
  void parallel_for(Fn && fn)
  {
    const int T = (int)std::min<int64_t>(threads_, B_);
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k)
      th.emplace_back([&, k] {
        for (int64_t b = B_ * k / T; b < B_ * (k + 1) / T; ++b) fn(b);
      });
    for (auto & t : th) t.join();
  }
  MPCT & mpc_;
  int64_t B_;
  int threads_, nA_ = 0, nP_ = 0;
  std::vector<double> Px_, q_, Ax_, l_, u_, x_, y_, wx_, wy_;
  std::vector<uint32_t> iter_;
  std::vector<int32_t> code_;
};

/// The same swarm resident on the device (sfb_mpc_swarm, sfb.h): the host only linearises (fill_record);
/// assembly of A, l, u, the solve and the warm starts stay in HBM, and only du_0, code and iter come back.
template<class MPCT>
class MPCSwarmDevice {
public:
  /// t_probe: a time in the range the agents will run at (the structure of the linearisation is probed there)
  explicit MPCSwarmDevice(MPCT & proto, int64_t agents, int threads = 0, typename MPCT::TimeT t_probe = {})
      : mpc_(proto), B_(agents), threads_(threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency()))
  {
    if (!mpc_.solver().analyzed()) {
      std::vector<uint8_t> keep;
      mpc_.probe_default(t_probe, keep);
      mpc_.analyze_solver(&keep);
    }
    // packed records: structure of the Jacobians probed like the structure of A, checked per record when packing
    pack_ = mpc_.probe_record_default(t_probe);
    if (const char * v = std::getenv("SFB_MPC_PACK_PROBE_EMPTY"); v && v[0] == '1') {  // tests: a probe that saw nothing
      pack_ = typename MPCT::RecordPacking{};
      pack_.count();
    }
    packed_ = mpc_.params().prune_explicit_zeros && pack_.doubles(mpc_.N()) < MPCT::record_doubles(mpc_.N());
    layout_ = mpc_.device_layout(packed_ ? &pack_.keep : nullptr);
    recd_   = packed_ ? pack_.doubles(mpc_.N()) : MPCT::record_doubles(mpc_.N());
    du0_.resize((size_t)B_ * MPCT::Nu);
    iter_.resize(B_);
    code_.resize(B_);
    const auto & qp = mpc_.qp();
    sfb_check(sfb_mpc_swarm_create(mpc_.solver().plan(), &layout_->c, qp.P_val.data(), qp.q.data(), B_, &swarm_));
    mpc_.solver().pin_plan();  // the swarm holds the raw plan pointer (sfb.h: the plan must outlive the swarm)
    sfb_check(sfb_mpc_swarm_host_records(swarm_, &rec_));  // pinned, owned by the swarm
  }
  MPCSwarmDevice(const MPCSwarmDevice &)             = delete;
  MPCSwarmDevice & operator=(const MPCSwarmDevice &) = delete;
  ~MPCSwarmDevice()
  {
    sfb_mpc_swarm_destroy(swarm_);
    mpc_.solver().unpin_plan();
  }

  void reset_warmstart() { sfb_check(sfb_mpc_swarm_reset_warmstart(swarm_)); }

  /// one control tick for all agents; primal / dual (nullable) receive the full solutions
  template<class XT, class UT>
  void step(const std::vector<typename MPCT::TimeT> & t, const std::vector<XT> & xs, std::vector<UT> & us,
            std::vector<QPSolutionStatus> & codes, std::vector<double> * primal = nullptr,
            std::vector<double> * dual = nullptr)
  {
    const auto & qp = mpc_.qp();
    // the records are written straight into the swarm's pinned buffer, chunk by chunk; the DMA of a finished chunk
    // (sfb_mpc_swarm_upload) overlaps with the linearisation of the next one
    const int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(kUploadChunks, B_ / std::max(1, threads_)));
    const auto tfill0 = std::chrono::steady_clock::now();
    {
      const int T = (int)std::min<int64_t>(threads_, B_ / chunks);
      std::vector<std::atomic<int>> done(chunks);
      for (auto & d : done) d.store(0);
      std::atomic<int> failed{0}, misfit{0};
      const bool packed = packed_;
      std::vector<std::thread> th;
      for (int k = 0; k < T; ++k)
        th.emplace_back([&, k] {
          std::vector<double> scratch(packed ? (size_t)MPCT::record_doubles(mpc_.N()) : 0);
          for (int64_t c0 = 0; c0 < chunks; ++c0) {  // every thread takes its share of chunk after chunk
            const int64_t b0 = B_ * c0 / chunks, b1 = B_ * (c0 + 1) / chunks, cnt = b1 - b0;
            for (int64_t b = b0 + cnt * k / T; b < b0 + cnt * (k + 1) / T; ++b) {
              if (!packed) {
                mpc_.fill_record(t[b], xs[b], rec_ + (size_t)b * recd_);
              } else {
                mpc_.fill_record(t[b], xs[b], scratch.data());
                if (!mpc_.pack_record(scratch.data(), rec_ + (size_t)b * recd_, pack_)) misfit.store(1);
              }
            }
            if (done[c0].fetch_add(1) + 1 == T && sfb_mpc_swarm_upload(swarm_, b0, cnt) != SFB_OK) failed.store(1);  // last one in
          }
        });
      for (auto & x : th) x.join();
      if (failed.load()) sfb_check(SFB_ERR_HIP);
      if (misfit.load()) {
        // a linearisation has a non-zero where the probe saw none: unpacked records from now on (same results; warm
        // starts and solver memory of the swarm are untouched), and this tick's records once more
        packed_ = false;
        recd_   = MPCT::record_doubles(mpc_.N());
        sfb_check(sfb_mpc_swarm_set_jac_keep(swarm_, nullptr, nullptr));
        step_fill_unpacked(t, xs);
      }
    }
    if (const char * tv = std::getenv("SFB_MPC_TIMING"); tv && tv[0] == '1')
      std::fprintf(stderr, "[MPCSwarmDevice] linearise  %8.3f ms (uploads of finished chunks in flight)\n",
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tfill0).count());
    if (primal) primal->resize((size_t)B_ * qp.n);
    if (dual) dual->resize((size_t)B_ * qp.m);
    const sfb_qp_params c = mpc_.solver().params().to_c();
    sfb_check(sfb_mpc_swarm_step_host(swarm_, &c, rec_, nullptr, mpc_.params().warmstart ? 1 : 0, du0_.data(),
                                      iter_.data(), code_.data(), primal ? primal->data() : nullptr,
                                      dual ? dual->data() : nullptr));
    us.resize(B_);
    codes.resize(B_);
    for (int64_t b = 0; b < B_; ++b) {
      us[b]    = mpc_.input_from_du0(t[b], &du0_[(size_t)b * MPCT::Nu]);
      codes[b] = static_cast<QPSolutionStatus>(code_[b]);
    }
  }
  const std::vector<uint32_t> & iterations() const { return iter_; }
  /// the linearisation records of the last tick ([agents][record_doubles], in the swarm's pinned buffer)
  const double * records() const { return rec_; }
  int64_t record_doubles() const { return recd_; }
  bool packed_records() const { return packed_; }
  sfb_mpc_swarm * handle() { return swarm_; }

private:
  static constexpr int64_t kUploadChunks = 8;
  template<class XT>
  void step_fill_unpacked(const std::vector<typename MPCT::TimeT> & t, const std::vector<XT> & xs)
  {
    parallel_for(0, B_, [&](int64_t b) { mpc_.fill_record(t[b], xs[b], rec_ + (size_t)b * recd_); });
  }
  template<class Fn>
  void parallel_for(int64_t b0, int64_t b1, Fn && fn)
  {
    const int64_t cnt = b1 - b0;
    const int T       = (int)std::min<int64_t>(threads_, cnt);
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k)
      th.emplace_back([&, k] {
        for (int64_t b = b0 + cnt * k / T; b < b0 + cnt * (k + 1) / T; ++b) fn(b);
      });
    for (auto & t : th) t.join();
  }
  MPCT & mpc_;
  int64_t B_, recd_ = 0;
  int threads_;
  std::unique_ptr<typename MPCT::DeviceLayout> layout_;
  sfb_mpc_swarm * swarm_ = nullptr;
  double * rec_ = nullptr;  // the swarm's pinned record buffer
  typename MPCT::RecordPacking pack_;
  bool packed_ = false;
  std::vector<double> du0_;
  std::vector<uint32_t> iter_;
  std::vector<int32_t> code_;
};

}  // namespace smooth_feedback_amd
