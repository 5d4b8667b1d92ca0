// Generic optimal-control-problem -> sparse QP front (host side; the QP goes to the shared-pattern sparse kernel
// through QPSolver<QuadraticProgramSparse<>> / solve_qp, include/smooth_feedback_amd/qp.hpp).
//
// Mirrors the reference's
//   OCP<X, U, Theta, F, G, CR, CE>      ocp.hpp:47-112
//   OCPSolution                         ocp.hpp:124-166
//   ocp_to_qp()                         ocp_to_qp.hpp:421-435  (= allocate :40-114 + update_cost :117-195 +
//                                       update_dyn :198-276 + update_cr :279-323 + update_ce :326-373)
//   qpsol_to_ocpsol()                   ocp_to_qp.hpp:452-499
// Same variable order [x_0 .. x_N | u_0 .. u_{N-1}], row order [dyn Nx N | cr Ncr N | ce Nce], same formulas and the
// same order of the additions into an entry; P holds its upper triangle only (block_add(..., upper_only = true)).
// Differences, stated once:
//   * derivatives: the reference differentiates the user's lambdas with smooth::diff (autodiff when available, else
//     numerical).  Here a functor may carry analytic derivatives (members `jacobian` / `hessian`, signatures below);
//     otherwise forward differences with step sqrt(eps) on the group (x (+) h e_i), second derivatives as
//     differences of first derivatives -- the definition smooth's dr<2> uses (derivative of the derivative).
//   * the linearisation trajectory's time derivative: `dxl_fun` if given, else central differences of xl_fun.
//   * stored pattern: every entry the reference's allocation reserves is stored (explicit zeros included), so the
//     pattern depends on the sizes only and one symbolic analysis serves every re-linearisation.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "lie.hpp"
#include "mesh.hpp"
#include "qp.hpp"

namespace smooth_feedback_amd {

/// ocp.hpp:47-112.  theta(tf, x0, xf, q) -> double; f(t, x, u) -> Vec<Nx>; g(t, x, u) -> Vec<Nq>;
/// cr(t, x, u) -> Vec<Ncr>; ce(tf, x0, xf, q) -> Vec<Nce>.
template<class X_, class U_, int Nq_, int Ncr_, int Nce_, class Theta, class F, class G, class CR, class CE>
struct OCP {
  using X = X_;
  using U = U_;
  static constexpr int Nx = X::Dof, Nu = U::Dof, Nq = Nq_, Ncr = Ncr_, Nce = Nce_;
  static_assert(Nx > 0 && Nu > 0 && Nq > 0 && Ncr > 0 && Nce > 0, "Static size required");  // ocp.hpp:70-74
  Theta theta;
  F f;
  G g;
  CR cr;
  Vec<Ncr> crl{}, cru{};
  CE ce;
  Vec<Nce> cel{}, ceu{};
};

/// deduces Nq, Ncr, Nce from the functors' return types like ocp.hpp:62-66
template<class X, class U, class Theta, class F, class G, class CR, class CE>
auto make_ocp(Theta theta, F f, G g, CR cr, decltype(cr(0.0, X{}, U{})) crl, decltype(cr(0.0, X{}, U{})) cru, CE ce,
              decltype(ce(0.0, X{}, X{}, g(0.0, X{}, U{}))) cel, decltype(ce(0.0, X{}, X{}, g(0.0, X{}, U{}))) ceu)
{
  constexpr int Nq  = (int)std::tuple_size_v<decltype(g(0.0, X{}, U{}))>;
  constexpr int Ncr = (int)std::tuple_size_v<decltype(cr(0.0, X{}, U{}))>;
  constexpr int Nce = (int)std::tuple_size_v<decltype(ce(0.0, X{}, X{}, g(0.0, X{}, U{})))>;
  return OCP<X, U, Nq, Ncr, Nce, Theta, F, G, CR, CE>{theta, f, g, cr, crl, cru, ce, cel, ceu};
}

/// ocp.hpp:124-166
template<class X, class U>
struct OCPSolution {
  double t0 = 0., tf = 1.;
  std::function<U(double)> u;
  std::function<X(double)> x;
};

namespace detail {

inline double ocp_fd_step() { return std::sqrt(std::numeric_limits<double>::epsilon()); }

template<int N>
Vec<N> unit(int i, double h)
{
  Vec<N> e{};
  e[i] = h;
  return e;
}

/// value and right-Jacobians of fn(t, x, u) -> Vec<M> w.r.t. x and u (t is not a QP variable)
template<int M, class Fn, class X, class U>
void jac_xu(const Fn & fn, double t, const X & x, const U & u, Vec<M> & val, Mat<M, X::Dof> & dx, Mat<M, U::Dof> & du)
{
  if constexpr (requires { fn.jacobian(t, x, u, dx, du); }) {
    val = fn(t, x, u);
    fn.jacobian(t, x, u, dx, du);
  } else {
    const double h = ocp_fd_step();
    val            = fn(t, x, u);
    for (int c = 0; c < X::Dof; ++c) {
      const Vec<M> v = fn(t, rplus(x, unit<X::Dof>(c, h)), u);
      for (int r = 0; r < M; ++r) dx(r, c) = (v[r] - val[r]) / h;
    }
    for (int c = 0; c < U::Dof; ++c) {
      const Vec<M> v = fn(t, x, rplus(u, unit<U::Dof>(c, h)));
      for (int r = 0; r < M; ++r) du(r, c) = (v[r] - val[r]) / h;
    }
  }
}

/// Hessian of the scalar fn(t, x, u)[0] w.r.t. (x, u): blocks xx, xu, uu (derivative of the right-Jacobian)
template<class Fn, class X, class U>
void hess_xu(const Fn & fn, double t, const X & x, const U & u, Mat<X::Dof, X::Dof> & hxx, Mat<X::Dof, U::Dof> & hxu,
             Mat<U::Dof, U::Dof> & huu)
{
  constexpr int Nx = X::Dof, Nu = U::Dof;
  if constexpr (requires { fn.hessian(t, x, u, hxx, hxu, huu); }) {
    fn.hessian(t, x, u, hxx, hxu, huu);
  } else {
    const double h = std::cbrt(std::numeric_limits<double>::epsilon());  // second differences: a larger step
    auto grad = [&](const X & xx, const U & uu, Vec<Nx> & gx, Vec<Nu> & gu) {
      const double v0 = fn(t, xx, uu)[0];
      for (int c = 0; c < Nx; ++c) gx[c] = (fn(t, rplus(xx, unit<Nx>(c, h)), uu)[0] - v0) / h;
      for (int c = 0; c < Nu; ++c) gu[c] = (fn(t, xx, rplus(uu, unit<Nu>(c, h)))[0] - v0) / h;
    };
    Vec<Nx> gx0{}, gx1{};
    Vec<Nu> gu0{}, gu1{};
    grad(x, u, gx0, gu0);
    for (int c = 0; c < Nx; ++c) {
      grad(rplus(x, unit<Nx>(c, h)), u, gx1, gu1);
      for (int r = 0; r < Nx; ++r) hxx(r, c) = (gx1[r] - gx0[r]) / h;
    }
    for (int c = 0; c < Nu; ++c) {
      grad(x, rplus(u, unit<Nu>(c, h)), gx1, gu1);
      for (int r = 0; r < Nx; ++r) hxu(r, c) = (gx1[r] - gx0[r]) / h;
      for (int r = 0; r < Nu; ++r) huu(r, c) = (gu1[r] - gu0[r]) / h;
    }
  }
}

/// value and right-Jacobians of fn(tf, x0, xf, q) -> Vec<M> (or double for M = 0) w.r.t. x0, xf, q
template<int M, int Nq, class Fn, class X>
void jac_end(const Fn & fn, double tf, const X & x0, const X & xf, const Vec<Nq> & q, Vec<(M > 0 ? M : 1)> & val,
             Mat<(M > 0 ? M : 1), X::Dof> & d0, Mat<(M > 0 ? M : 1), X::Dof> & df, Mat<(M > 0 ? M : 1), Nq> & dq)
{
  constexpr int R = M > 0 ? M : 1, Nx = X::Dof;
  auto eval = [&](const X & a, const X & b, const Vec<Nq> & qq) {
    Vec<R> v{};
    if constexpr (M > 0) v = fn(tf, a, b, qq);
    else v[0] = fn(tf, a, b, qq);
    return v;
  };
  val = eval(x0, xf, q);
  if constexpr (requires { fn.jacobian(tf, x0, xf, q, d0, df, dq); }) {
    fn.jacobian(tf, x0, xf, q, d0, df, dq);
  } else {
    const double h = ocp_fd_step();
    for (int c = 0; c < Nx; ++c) {
      const Vec<R> a = eval(rplus(x0, unit<Nx>(c, h)), xf, q), b = eval(x0, rplus(xf, unit<Nx>(c, h)), q);
      for (int r = 0; r < R; ++r) {
        d0(r, c) = (a[r] - val[r]) / h;
        df(r, c) = (b[r] - val[r]) / h;
      }
    }
    for (int c = 0; c < Nq; ++c) {
      Vec<Nq> qq = q;
      qq[c] += h;
      const Vec<R> a = eval(x0, xf, qq);
      for (int r = 0; r < R; ++r) dq(r, c) = (a[r] - val[r]) / h;
    }
  }
}

/// Hessian blocks of the scalar theta w.r.t. (x0, x0), (x0, xf), (xf, xf)
template<int Nq, class Fn, class X>
void hess_end(const Fn & fn, double tf, const X & x0, const X & xf, const Vec<Nq> & q, Mat<X::Dof, X::Dof> & h00,
              Mat<X::Dof, X::Dof> & h0f, Mat<X::Dof, X::Dof> & hff)
{
  constexpr int Nx = X::Dof;
  if constexpr (requires { fn.hessian(tf, x0, xf, q, h00, h0f, hff); }) {
    fn.hessian(tf, x0, xf, q, h00, h0f, hff);
  } else {
    const double h = std::cbrt(std::numeric_limits<double>::epsilon());
    auto grad = [&](const X & a, const X & b, Vec<Nx> & g0, Vec<Nx> & gf) {
      const double v0 = fn(tf, a, b, q);
      for (int c = 0; c < Nx; ++c) {
        g0[c] = (fn(tf, rplus(a, unit<Nx>(c, h)), b, q) - v0) / h;
        gf[c] = (fn(tf, a, rplus(b, unit<Nx>(c, h)), q) - v0) / h;
      }
    };
    Vec<Nx> g0{}, gf{}, a0{}, af{};
    grad(x0, xf, g0, gf);
    for (int c = 0; c < Nx; ++c) {
      grad(rplus(x0, unit<Nx>(c, h)), xf, a0, af);
      for (int r = 0; r < Nx; ++r) h00(r, c) = (a0[r] - g0[r]) / h;
      grad(x0, rplus(xf, unit<Nx>(c, h)), a0, af);
      for (int r = 0; r < Nx; ++r) {
        h0f(r, c) = (a0[r] - g0[r]) / h;
        hff(r, c) = (af[r] - gf[r]) / h;
      }
    }
  }
}

/// coeffRef of the row-major A / of the upper triangle of the column-major P: the entry must be in the pattern
inline double & csr_at(const std::vector<int32_t> & ptr, const std::vector<int32_t> & ind, std::vector<double> & val, int major,
                       int minor)
{
  const auto b = ind.begin() + ptr[major], e = ind.begin() + ptr[major + 1];
  const auto it = std::lower_bound(b, e, minor);
  if (it == e || *it != minor) throw std::logic_error("ocp_to_qp: entry outside the allocated pattern");
  return val[(size_t)(it - ind.begin())];
}

}  // namespace detail

/// ocp_to_qp_allocate, ocp_to_qp.hpp:40-114: sizes and the stored pattern (depends on the sizes only)
template<class Ocp>
void ocp_to_qp_allocate(QuadraticProgramSparse<> & qp, const Ocp &, const Mesh & mesh)
{
  constexpr int Nx = Ocp::Nx, Nu = Ocp::Nu, Ncr = Ocp::Ncr, Nce = Ocp::Nce;
  const int N = mesh.N_colloc(), K = mesh.K;
  const int xvar_L = Nx * (N + 1), uvar_L = Nu * N;                 // :58-59
  const int dcon_L = Nx * N, crcon_L = Ncr * N, cecon_L = Nce;      // :61-63
  const int crcon_B = dcon_L, cecon_B = crcon_B + crcon_L, uvar_B = xvar_L;
  qp.n = xvar_L + uvar_L;
  qp.m = dcon_L + crcon_L + cecon_L;
  qp.q.assign(qp.n, 0.0);
  qp.l.assign(qp.m, 0.0);
  qp.u.assign(qp.m, 0.0);
  // A, row-major (:81-88): a dyn row of node i (interval first node M) has the x_i block row, the Ki + 1 collocation
  // entries on its own state component (one of them inside the x_i block), and the u_i block row
  std::vector<std::vector<int32_t>> rows(qp.m);
  for (int i = 0; i < N; ++i) {
    const int M = (i / K) * K;
    for (int d = 0; d < Nx; ++d) {
      auto & r = rows[i * Nx + d];
      for (int c = 0; c < Nx; ++c) r.push_back(i * Nx + c);
      for (int j = 0; j <= K; ++j)
        if (M + j != i) r.push_back((M + j) * Nx + d);
      for (int c = 0; c < Nu; ++c) r.push_back(uvar_B + i * Nu + c);
    }
    for (int d = 0; d < Ncr; ++d) {
      auto & r = rows[crcon_B + i * Ncr + d];
      for (int c = 0; c < Nx; ++c) r.push_back(i * Nx + c);
      for (int c = 0; c < Nu; ++c) r.push_back(uvar_B + i * Nu + c);
    }
  }
  for (int d = 0; d < Nce; ++d) {
    auto & r = rows[cecon_B + d];
    for (int c = 0; c < Nx; ++c) r.push_back(c);
    for (int c = 0; c < Nx; ++c) r.push_back(xvar_L - Nx + c);
  }
  qp.A_rowptr.assign(qp.m + 1, 0);
  qp.A_colind.clear();
  for (int r = 0; r < qp.m; ++r) {
    std::sort(rows[r].begin(), rows[r].end());
    rows[r].erase(std::unique(rows[r].begin(), rows[r].end()), rows[r].end());
    qp.A_colind.insert(qp.A_colind.end(), rows[r].begin(), rows[r].end());
    qp.A_rowptr[r + 1] = (int32_t)qp.A_colind.size();
  }
  qp.A_val.assign(qp.A_colind.size(), 0.0);
  // P, column-major upper triangle (:91-100): x_i x_i upper blocks, the x_0 x_N block, x_i u_i blocks, u_i u_i upper
  std::vector<std::vector<int32_t>> cols(qp.n);
  for (int i = 0; i <= N; ++i)
    for (int c = 0; c < Nx; ++c) {
      auto & col = cols[i * Nx + c];
      if (i == N)
        for (int r = 0; r < Nx; ++r) col.push_back(r);  // d2 theta / dx0 dxf
      for (int r = 0; r <= c; ++r) col.push_back(i * Nx + r);
    }
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < Nu; ++c) {
      auto & col = cols[uvar_B + i * Nu + c];
      for (int r = 0; r < Nx; ++r) col.push_back(i * Nx + r);
      for (int r = 0; r <= c; ++r) col.push_back(uvar_B + i * Nu + r);
    }
  qp.P_colptr.assign(qp.n + 1, 0);
  qp.P_rowind.clear();
  for (int c = 0; c < qp.n; ++c) {
    std::sort(cols[c].begin(), cols[c].end());
    cols[c].erase(std::unique(cols[c].begin(), cols[c].end()), cols[c].end());
    qp.P_rowind.insert(qp.P_rowind.end(), cols[c].begin(), cols[c].end());
    qp.P_colptr[c + 1] = (int32_t)qp.P_rowind.size();
  }
  qp.P_val.assign(qp.P_rowind.size(), 0.0);
}

/// ocp_to_qp_update, ocp_to_qp.hpp:396-399 (cost :117-195, dyn :198-276, cr :279-323, ce :326-373).
/// xl_fun(t) -> X, ul_fun(t) -> U: linearisation trajectory on [0, tf]; dxl_fun(t) -> Vec<Nx>: its body velocity
/// (nullptr-like empty std::function: central differences of xl_fun).
template<class Ocp, class XL, class UL>
void ocp_to_qp_update(QuadraticProgramSparse<> & qp, const Ocp & ocp, const Mesh & mesh, double tf, const XL & xl_fun,
                      const UL & ul_fun, const std::function<Vec<Ocp::Nx>(double)> & dxl_fun = {})
{
  using X = typename Ocp::X;
  using U = typename Ocp::U;
  constexpr int Nx = Ocp::Nx, Nu = Ocp::Nu, Nq = Ocp::Nq, Ncr = Ocp::Ncr, Nce = Ocp::Nce;
  static_assert(Nq == 1, "exactly one integral supported in ocp_to_qp");  // :134
  const int N = mesh.N_colloc(), K = mesh.K;
  const int xvar_L = Nx * (N + 1), dcon_L = Nx * N, crcon_L = Ncr * N;
  const int uvar_B = xvar_L, crcon_B = dcon_L, cecon_B = crcon_B + crcon_L;
  const double t0 = 0.;
  auto A = [&](int r, int c) -> double & { return detail::csr_at(qp.A_rowptr, qp.A_colind, qp.A_val, r, c); };
  auto P = [&](int r, int c) -> double & { return detail::csr_at(qp.P_colptr, qp.P_rowind, qp.P_val, c, r); };  // r <= c
  std::fill(qp.P_val.begin(), qp.P_val.end(), 0.0);  // :153-154
  std::fill(qp.q.begin(), qp.q.end(), 0.0);
  std::fill(qp.A_val.begin(), qp.A_val.end(), 0.0);  // :236 (cr and ce rows are block_write-n below)

  const X xl0 = xl_fun(0.), xlf = xl_fun(tf);  // :160-161
  const Vec<Nq> ql{1.};                        // :166

  // ---- cost :172-194 ----
  Vec<1> th{};
  Mat<1, Nx> dth0{}, dthf{};
  Mat<1, Nq> dthq{};
  detail::jac_end<0, Nq>(ocp.theta, tf, xl0, xlf, ql, th, dth0, dthf, dthq);
  const double qo_q = dthq(0, 0);  // :176
  for (int i = 0; i < N; ++i) {    // mesh_integrate<2>(g) restricted to what :181-184 read: w_i (tf - t0) * (dg, d2g) per node
    const double t_i = t0 + (tf - t0) * mesh.node(i), w = mesh.weight(i) * (tf - t0);
    const X xl_i = xl_fun(t_i);
    const U ul_i = ul_fun(t_i);
    Vec<Nq> gv{};
    Mat<Nq, Nx> gx{};
    Mat<Nq, Nu> gu{};
    detail::jac_xu<Nq>(ocp.g, t_i, xl_i, ul_i, gv, gx, gu);
    Mat<Nx, Nx> hxx{};
    Mat<Nx, Nu> hxu{};
    Mat<Nu, Nu> huu{};
    detail::hess_xu(ocp.g, t_i, xl_i, ul_i, hxx, hxu, huu);
    for (int c = 0; c < Nx; ++c)
      for (int r = 0; r <= c; ++r) P(i * Nx + r, i * Nx + c) += qo_q * (w * hxx(r, c));  // :181 (upper only)
    for (int c = 0; c < Nu; ++c) {
      for (int r = 0; r < Nx; ++r) P(i * Nx + r, uvar_B + i * Nu + c) += qo_q * (w * hxu(r, c));
      for (int r = 0; r <= c; ++r) P(uvar_B + i * Nu + r, uvar_B + i * Nu + c) += qo_q * (w * huu(r, c));
    }
    for (int c = 0; c < Nx; ++c) qp.q[i * Nx + c] = qo_q * (w * gx(0, c));          // :183
    for (int c = 0; c < Nu; ++c) qp.q[uvar_B + i * Nu + c] = qo_q * (w * gu(0, c));  // :184
  }
  {
    Mat<Nx, Nx> h00{}, h0f{}, hff{};
    detail::hess_end<Nq>(ocp.theta, tf, xl0, xlf, ql, h00, h0f, hff);
    for (int c = 0; c < Nx; ++c)
      for (int r = 0; r <= c; ++r) P(r, c) += 0.5 * h00(r, c);                       // :191
    for (int c = 0; c < Nx; ++c)
      for (int r = 0; r < Nx; ++r) P(r, Nx * N + c) += 0.5 * h0f(r, c);              // :192
    for (int c = 0; c < Nx; ++c)
      for (int r = 0; r <= c; ++r) P(Nx * N + r, Nx * N + c) += 0.5 * hff(r, c);     // :193
    for (int c = 0; c < Nx; ++c) {
      qp.q[c] += dth0(0, c);            // :195
      qp.q[Nx * N + c] += dthf(0, c);   // :196
    }
  }

  // ---- collocation constraints :242-275 ----
  for (int ival = 0, M = 0; ival < mesh.N_ivals(); M += K, ++ival) {
    const double alpha = mesh.alpha(ival);
    for (int i = 0; i < K; ++i) {
      const double t_i = t0 + (tf - t0) * mesh.node(M + i);
      const X xl_i     = xl_fun(t_i);
      Vec<Nx> dxl_i{};
      if (dxl_fun) dxl_i = dxl_fun(t_i);
      else {
        const double h = 1e-6 * (tf - t0 > 0 ? tf - t0 : 1.0);
        const auto d   = rminus(xl_fun(t_i + h), xl_fun(t_i - h));
        for (int c = 0; c < Nx; ++c) dxl_i[c] = d[c] / (2 * h);
      }
      const U ul_i = ul_fun(t_i);
      Vec<Nx> f_i{};
      Mat<Nx, Nx> dfx{};
      Mat<Nx, Nu> dfu{};
      detail::jac_xu<Nx>(ocp.f, t_i, xl_i, ul_i, f_i, dfx, dfu);
      const int row0 = (M + i) * Nx;
      for (int c = 0; c < Nx; ++c)
        for (int r = 0; r < Nx; ++r) A(row0 + r, (M + i) * Nx + c) += tf * dfx(r, c);           // :257
      for (int c = 0; c < Nu; ++c)
        for (int r = 0; r < Nx; ++r) A(row0 + r, uvar_B + (M + i) * Nu + c) += tf * dfu(r, c);  // :258
      if constexpr (!X::IsCommutative) {                                                         // :261-263
        Vec<Nx> s{};
        for (int c = 0; c < Nx; ++c) s[c] = f_i[c] + dxl_i[c];
        const auto adm = X::ad(s);
        for (int c = 0; c < Nx; ++c)
          for (int r = 0; r < Nx; ++r) A(row0 + r, (M + i) * Nx + c) += (-tf / 2) * adm(r, c);
      }
      for (int j = 0; j <= K; ++j)  // :265-269
        for (int d = 0; d < Nx; ++d) A(row0 + d, (M + j) * Nx + d) -= alpha * mesh.D(j, i);
      for (int d = 0; d < Nx; ++d) {  // :271-272
        qp.l[row0 + d] = -tf * (f_i[d] - dxl_i[d]);
        qp.u[row0 + d] = qp.l[row0 + d];
      }
    }
  }

  // ---- running constraints :311-322 ----
  for (int i = 0; i < N; ++i) {
    const double t_i = t0 + (tf - t0) * mesh.node(i);
    const X xl_i = xl_fun(t_i);
    const U ul_i = ul_fun(t_i);
    Vec<Ncr> cv{};
    Mat<Ncr, Nx> cx{};
    Mat<Ncr, Nu> cu{};
    detail::jac_xu<Ncr>(ocp.cr, t_i, xl_i, ul_i, cv, cx, cu);
    for (int d = 0; d < Ncr; ++d) {
      const int row = crcon_B + i * Ncr + d;
      for (int c = 0; c < Nx; ++c) A(row, i * Nx + c) = cx(d, c);
      for (int c = 0; c < Nu; ++c) A(row, uvar_B + i * Nu + c) = cu(d, c);
      qp.l[row] = ocp.crl[d] - cv[d];
      qp.u[row] = ocp.cru[d] - cv[d];
    }
  }

  // ---- end constraints :362-372 ----
  {
    Vec<Nce> cev{};
    Mat<Nce, Nx> d0{}, df{};
    Mat<Nce, Nq> dq{};
    detail::jac_end<Nce, Nq>(ocp.ce, tf, xl0, xlf, ql, cev, d0, df, dq);
    for (int d = 0; d < Nce; ++d) {
      for (int c = 0; c < Nx; ++c) A(cecon_B + d, c) = d0(d, c);
      for (int c = 0; c < Nx; ++c) A(cecon_B + d, xvar_L - Nx + c) = df(d, c);
      qp.l[cecon_B + d] = ocp.cel[d] - cev[d];
      qp.u[cecon_B + d] = ocp.ceu[d] - cev[d];
    }
  }
}

/// ocp_to_qp(), ocp_to_qp.hpp:421-435
template<class Ocp, class XL, class UL>
QuadraticProgramSparse<> ocp_to_qp(const Ocp & ocp, const Mesh & mesh, double tf, const XL & xl_fun, const UL & ul_fun,
                                   const std::function<Vec<Ocp::Nx>(double)> & dxl_fun = {})
{
  QuadraticProgramSparse<> qp;
  ocp_to_qp_allocate(qp, ocp, mesh);
  ocp_to_qp_update(qp, ocp, mesh, tf, xl_fun, ul_fun, dxl_fun);
  return qp;
}

namespace detail {
/// Mesh::eval (collocation/mesh.hpp:428-470), p = 0: Lagrange interpolation of the values r (one Vec<D> per node,
/// N + 1 of them when `extend`, else N) in the interval that contains t in [0, 1].
template<int D>
Vec<D> mesh_eval(const Mesh & mesh, double t, const std::vector<Vec<D>> & r, bool extend)
{
  const int K = mesh.K, nI = mesh.N_ivals();
  int ival = (t <= 0) ? 0 : (t >= 1 ? nI - 1 : std::min(nI - 1, (int)(t * nI)));
  const double tau0 = mesh.interval_start(ival), tauf = (ival + 1 < nI) ? mesh.interval_start(ival + 1) : 1.0;
  const double u    = 2 * (t - tau0) / (tauf - tau0) - 1;
  const int npts    = (extend || ival + 1 < nI) ? K + 1 : K;  // the next interval's first node closes the interval
  Vec<D> ret{};
  for (int j = 0; j < npts; ++j) {
    double w = 1.0;
    for (int k2 = 0; k2 < npts; ++k2)
      if (k2 != j) w *= (u - mesh.tau[k2]) / (mesh.tau[j] - mesh.tau[k2]);
    const auto & v = r[(size_t)ival * K + j];
    for (int d = 0; d < D; ++d) ret[d] += w * v[d];
  }
  return ret;
}
}  // namespace detail

/// qpsol_to_ocpsol(), ocp_to_qp.hpp:452-499
template<class Ocp, class XL, class UL>
OCPSolution<typename Ocp::X, typename Ocp::U> qpsol_to_ocpsol(const Ocp &, const Mesh & mesh, const QPSolution<> & qpsol, double tf,
                                                              XL xl_fun, UL ul_fun)
{
  using X = typename Ocp::X;
  using U = typename Ocp::U;
  constexpr int Nx = Ocp::Nx, Nu = Ocp::Nu;
  const int N = mesh.N_colloc();
  if ((int)qpsol.primal.size() != Nx * (N + 1) + Nu * N) throw std::invalid_argument("qpsol_to_ocpsol: solution of another problem");
  std::vector<Vec<Nx>> Xmat(N + 1);
  std::vector<Vec<Nu>> Umat(N);
  for (int i = 0; i <= N; ++i)
    for (int d = 0; d < Nx; ++d) Xmat[i][d] = qpsol.primal[(size_t)i * Nx + d];
  for (int i = 0; i < N; ++i)
    for (int d = 0; d < Nu; ++d) Umat[i][d] = qpsol.primal[(size_t)Nx * (N + 1) + (size_t)i * Nu + d];
  OCPSolution<X, U> sol;
  sol.t0 = 0.;
  sol.tf = tf;
  sol.x  = [tf, mesh, Xmat = std::move(Xmat), xl_fun](double t) -> X {
    return rplus(xl_fun(t), detail::mesh_eval<Nx>(mesh, t / tf, Xmat, true));
  };
  sol.u = [tf, mesh, Umat = std::move(Umat), ul_fun](double t) -> U {
    return rplus(ul_fun(t), detail::mesh_eval<Nu>(mesh, t / tf, Umat, false));
  };
  return sol;
}

}  // namespace smooth_feedback_amd
