// Example / test harness: concrete MPC models behind a C interface (see models.h).
#include "models.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <optional>
#include <random>
#include <thread>
#include <vector>

#include <smooth_feedback_amd/asif.hpp>
#include <smooth_feedback_amd/ekf.hpp>
#include <smooth_feedback_amd/mpc.hpp>
#include <smooth/feedback/mpc.hpp>  // the reference's include path and namespace (sfbx_test_mpc_api)

#include "vehicle_model.h"

using namespace smooth_feedback_amd;
using sfbx::U2;
using sfbx::VehicleBU;
using sfbx::VehicleDyn6;
using sfbx::VehicleH;
using sfbx::X6;

namespace {

// ---- the vehicle models (states, dynamics, running constraint, desired trajectories, barrier, backup controller):
// vehicle_model.h, shared with the device-side harness ----
using sfbx::InputBox;
using sfbx::MPC12;
using sfbx::MPC6;
using sfbx::VehicleDyn12;
using sfbx::X12;
X6 xdes6(double t) { return sfbx::VehicleModel6{}.xdes(t); }
X12 xdes12(double t) { return sfbx::VehicleModel12{}.xdes(t); }
MPC6 make6(int K, double tf) { return sfbx::make_vehicle_mpc<MPC6, sfbx::VehicleModel6>(K, tf); }
MPC12 make12(int K, double tf) { return sfbx::make_vehicle_mpc<MPC12, sfbx::VehicleModel12>(K, tf); }

Vec<6> mpc_dyn(MPC6 &, const X6 & x, const U2 & u) { return VehicleDyn6{}(x, u); }
Vec<12> mpc_dyn(MPC12 &, const X12 & x, const U2 & u) { return VehicleDyn12{}(x, u); }

template<class X>
X perturbed(const X & x0, uint64_t seed)
{
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> d(-0.5, 0.5);
  typename X::Tangent xi{};
  for (auto & v : xi) v = d(rng);
  return rplus(x0, xi);
}

template<class M, class XF>
int assemble_batch(M & mpc, XF xdes, int64_t batch, uint64_t seed, double * Aval, double * l, double * u, int threads)
{
  const int nA = (int)mpc.qp().A_val.size(), m = mpc.qp().m;
  const int T  = std::max(1, std::min<int>(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), (int)std::max<int64_t>(1, batch)));
  std::vector<std::thread> th;
  for (int k = 0; k < T; ++k)
    th.emplace_back([&, k] {
      for (int64_t b = batch * k / T; b < batch * (k + 1) / T; ++b) {
        const double t = 0.025 * double(b % 400);
        mpc.assemble(t, perturbed(xdes(t), seed + (uint64_t)b), Aval + (size_t)b * nA, l + (size_t)b * m, u + (size_t)b * m);
      }
    });
  for (auto & t : th) t.join();
  return 0;
}

template<class M>
int fill_pattern(const M & mpc, int32_t * Pp, int32_t * Pi, double * Pval, int32_t * Ap, int32_t * Aj)
{
  const auto & qp = mpc.qp();
  std::copy(qp.P_colptr.begin(), qp.P_colptr.end(), Pp);
  std::copy(qp.P_rowind.begin(), qp.P_rowind.end(), Pi);
  std::copy(qp.P_val.begin(), qp.P_val.end(), Pval);
  std::copy(qp.A_rowptr.begin(), qp.A_rowptr.end(), Ap);
  std::copy(qp.A_colind.begin(), qp.A_colind.end(), Aj);
  return 0;
}

std::vector<double> g_tick_seconds;  // wall time of every swarm.step() of the last swarm_step call

bool g_swarm_multi_device = false;  // sfbx_mpc_swarm_step_multi: shard the swarm's solves over the device list

template<class M, class XF, class X, class Swarm = MPCSwarm<M>>
int swarm_step(M & mpc, XF xdes, int64_t batch, uint64_t seed, int ticks, double * u0, int32_t * codes, uint32_t * iters)
{
  Swarm swarm(mpc, batch);
  mpc.solver().shard_over_devices(g_swarm_multi_device);
  std::vector<double> t(batch);
  std::vector<X> xs(batch);
  for (int64_t b = 0; b < batch; ++b) {
    t[b]  = 0.025 * double(b % 400);
    xs[b] = perturbed(xdes(t[b]), seed + (uint64_t)b);
  }
  std::vector<U2> us;
  std::vector<QPSolutionStatus> cs;
  g_tick_seconds.assign(ticks, 0.0);
  for (int k = 0; k < ticks; ++k) {
    const auto t0 = std::chrono::steady_clock::now();
    swarm.step(t, xs, us, cs);
    g_tick_seconds[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // crude closed loop: integrate the state with the applied input for one 25 ms tick
    for (int64_t b = 0; b < batch; ++b) {
      auto f = mpc_dyn(mpc, xs[b], us[b]);
      for (auto & v : f) v *= 0.025;
      xs[b] = rplus(xs[b], f);
      t[b] += 0.025;
    }
  }
  for (int64_t b = 0; b < batch; ++b) {
    u0[2 * b] = us[b].v[0]; u0[2 * b + 1] = us[b].v[1];
    codes[b] = (int32_t)cs[b];
    iters[b] = swarm.iterations()[b];
  }
  return 0;
}
template<class M, class XF>
int records_batch(M & mpc, XF xdes, int64_t batch, uint64_t seed, double * rec, int threads)
{
  const int64_t rd = M::record_doubles(mpc.N());
  const int T = std::max(1, std::min<int>(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), (int)std::max<int64_t>(1, batch)));
  std::vector<std::thread> th;
  for (int k = 0; k < T; ++k)
    th.emplace_back([&, k] {
      for (int64_t b = batch * k / T; b < batch * (k + 1) / T; ++b) {
        const double t = 0.025 * double(b % 400);
        mpc.fill_record(t, perturbed(xdes(t), seed + (uint64_t)b), rec + (size_t)b * rd);
      }
    });
  for (auto & t : th) t.join();
  return 0;
}

}  // namespace

extern "C" {

int sfbx_mpc_dims(int variant, int K, int * n, int * m, int * nnzP, int * nnzA, int * Nx, int * Nu, int * N)
{
  auto fill = [&](const auto & mpc, int nx) {
    *n = mpc.qp().n; *m = mpc.qp().m; *nnzP = (int)mpc.qp().P_val.size(); *nnzA = (int)mpc.qp().A_val.size();
    *Nx = nx; *Nu = 2; *N = mpc.N();
  };
  if (variant == 6) { auto mpc = make6(K, 5.0); fill(mpc, 6); return 0; }
  if (variant == 12) { auto mpc = make12(K, 5.0); fill(mpc, 12); return 0; }
  return -1;
}

int sfbx_mpc_pattern(int variant, int K, double tf, int32_t * Pp, int32_t * Pi, double * Pval, int32_t * Ap, int32_t * Aj)
{
  if (variant == 6) { auto mpc = make6(K, tf); return fill_pattern(mpc, Pp, Pi, Pval, Ap, Aj); }
  if (variant == 12) { auto mpc = make12(K, tf); return fill_pattern(mpc, Pp, Pi, Pval, Ap, Aj); }
  return -1;
}

int sfbx_mpc_stage(int variant, int K, int32_t * stage)
{
  auto fill = [&](const auto & mpc) {
    const char * flat = getenv("SFB_MPC_STAGE");  // A/B knob: "flat" = separators eliminated as a chain
    const auto st = mpc.elimination_stage(!(flat && flat[0] == 'f'));
    std::copy(st.begin(), st.end(), stage);
    return 0;
  };
  if (variant == 6) { auto mpc = make6(K, 5.0); return fill(mpc); }
  if (variant == 12) { auto mpc = make12(K, 5.0); return fill(mpc); }
  return -1;
}

int sfbx_mpc_assemble_batch(int variant, int K, double tf, int64_t batch, uint64_t seed, double * Aval, double * l,
                            double * u, int threads)
{
  if (variant == 6) { auto mpc = make6(K, tf); return assemble_batch(mpc, xdes6, batch, seed, Aval, l, u, threads); }
  if (variant == 12) { auto mpc = make12(K, tf); return assemble_batch(mpc, xdes12, batch, seed, Aval, l, u, threads); }
  return -1;
}

int sfbx_mpc_layout(int variant, int K, double tf, int32_t * dims, double * alpha, double * D, int32_t * kind,
                    int32_t * dof, double * crl, double * cru)
{
  auto fill = [&](const auto & mpc) {
    auto L = mpc.device_layout();
    dims[0] = L->c.nx; dims[1] = L->c.nu; dims[2] = L->c.ncr; dims[3] = L->c.kmesh; dims[4] = L->c.nivals; dims[5] = L->c.nparts;
    std::copy(L->alpha.begin(), L->alpha.end(), alpha);
    std::copy(L->D.begin(), L->D.end(), D);
    std::copy(L->kind.begin(), L->kind.end(), kind);
    std::copy(L->dof.begin(), L->dof.end(), dof);
    std::copy(L->crl.begin(), L->crl.end(), crl);
    std::copy(L->cru.begin(), L->cru.end(), cru);
    return 0;
  };
  if (variant == 6) { auto mpc = make6(K, tf); return fill(mpc); }
  if (variant == 12) { auto mpc = make12(K, tf); return fill(mpc); }
  return -1;
}

int sfbx_mpc_records(int variant, int K, double tf, int64_t batch, uint64_t seed, double * rec, int threads)
{
  if (variant == 6) { auto mpc = make6(K, tf); return records_batch(mpc, xdes6, batch, seed, rec, threads); }
  if (variant == 12) { auto mpc = make12(K, tf); return records_batch(mpc, xdes12, batch, seed, rec, threads); }
  return -1;
}

int sfbx_last_tick_seconds(double * out, int n)
{
  for (int k = 0; k < n && k < (int)g_tick_seconds.size(); ++k) out[k] = g_tick_seconds[k];
  return (int)g_tick_seconds.size();
}

int sfbx_mpc_swarm_device_step(int variant, int K, double tf, int64_t batch, uint64_t seed, int ticks, double * u0,
                               int32_t * codes, uint32_t * iters)
{
  try {
    if (variant == 6) { auto mpc = make6(K, tf); return swarm_step<MPC6, decltype(&xdes6), X6, MPCSwarmDevice<MPC6>>(mpc, xdes6, batch, seed, ticks, u0, codes, iters); }
    if (variant == 12) { auto mpc = make12(K, tf); return swarm_step<MPC12, decltype(&xdes12), X12, MPCSwarmDevice<MPC12>>(mpc, xdes12, batch, seed, ticks, u0, codes, iters); }
  } catch (const std::exception &) {
    return -2;
  }
  return -1;
}

int sfbx_mpc_swarm_step(int variant, int K, double tf, int64_t batch, uint64_t seed, int ticks, double * u0,
                        int32_t * codes, uint32_t * iters)
{
  try {
    if (variant == 6) { auto mpc = make6(K, tf); return swarm_step<MPC6, decltype(&xdes6), X6>(mpc, xdes6, batch, seed, ticks, u0, codes, iters); }
    if (variant == 12) { auto mpc = make12(K, tf); return swarm_step<MPC12, decltype(&xdes12), X12>(mpc, xdes12, batch, seed, ticks, u0, codes, iters); }
  } catch (const std::exception &) {
    return -2;
  }
  return -1;
}

int sfbx_mpc_swarm_step_multi(int variant, int K, double tf, int64_t batch, uint64_t seed, int ticks, const int * devices,
                              int ndev, double * u0, int32_t * codes, uint32_t * iters)
{
  // MPCSwarm with its batched solves sharded over `devices` from this one process (sfb_set_devices +
  // QPSolver::shard_over_devices): SURVEY.md section 8(b) / 8(e) for the caller the reference actually has, C++.
  if (sfb_set_devices(devices, ndev) != SFB_OK) return -3;
  g_swarm_multi_device = true;
  const int rc = sfbx_mpc_swarm_step(variant, K, tf, batch, seed, ticks, u0, codes, iters);
  g_swarm_multi_device = false;
  (void)sfb_set_devices(nullptr, 0);
  return rc;
}

}  // extern "C"

#include <smooth/feedback/ocp_to_qp.hpp>
#include <smooth/feedback/qp_solver.hpp>  // the reference's include path and namespace (forwarding header)

namespace {
// tests/test_qp.cpp:37-52 StaticProperties
static_assert(std::is_copy_assignable_v<smooth::feedback::QPSolver<smooth::feedback::QuadraticProgram<-1, -1>>>);
static_assert(std::is_copy_constructible_v<smooth::feedback::QPSolver<smooth::feedback::QuadraticProgram<-1, -1>>>);
static_assert(std::is_move_assignable_v<smooth::feedback::QPSolver<smooth::feedback::QuadraticProgram<-1, -1>>>);
static_assert(std::is_move_constructible_v<smooth::feedback::QPSolver<smooth::feedback::QuadraticProgram<-1, -1>>>);
static_assert(std::is_copy_assignable_v<smooth::feedback::QPSolver<smooth::feedback::QuadraticProgramSparse<double>>>);
static_assert(std::is_copy_constructible_v<smooth::feedback::QPSolver<smooth::feedback::QuadraticProgramSparse<double>>>);
static_assert(std::is_move_assignable_v<smooth::feedback::QPSolver<smooth::feedback::QuadraticProgramSparse<double>>>);
static_assert(std::is_move_constructible_v<smooth::feedback::QPSolver<smooth::feedback::QuadraticProgramSparse<double>>>);

template<class Pbm>
int five_copies(const Pbm & problem, double * primal_out)
{
  // tests/test_qp.cpp:338-372 / :374-415: a solver, its copy, a copy-assigned, a moved-to and a move-assigned one
  const smooth::feedback::QPSolverParams test_prm{.verbose = false, .polish = true};
  smooth::feedback::QPSolver solver1(problem, test_prm);
  auto sol1 = solver1.solve(problem);
  auto solver2 = solver1;
  auto sol2    = solver2.solve(problem);
  smooth::feedback::QPSolver<Pbm> solver3;
  solver3   = solver1;
  auto sol3 = solver3.solve(problem);
  auto solver4 = std::move(solver1);
  auto sol4    = solver4.solve(problem);
  smooth::feedback::QPSolver<Pbm> solver5;
  solver5   = std::move(solver2);
  auto sol5 = solver5.solve(problem);
  int k = 0;
  for (const auto * s : {&sol1, &sol2, &sol3, &sol4, &sol5}) {
    if (s->code != smooth::feedback::QPSolutionStatus::Optimal) return 10 + k;
    for (double v : s->primal) primal_out[k++] = v;
  }
  return 0;
}
}  // namespace

extern "C" {

int sfbx_test_qp_solver_api(double * primal_dense, double * primal_sparse, double * primal_partial)
{
  try {
    // TwoDimensional data of tests/test_qp.cpp:340-346
    const double inf = std::numeric_limits<double>::infinity();
    smooth::feedback::QuadraticProgram<2, 2> problem;
    problem.P = {0.0100131, 0, 0, 0.01};
    problem.q = {-0.329554, 0.536459};
    problem.A = {-0.0639209, -0.467, -0.168, 0};  // column-major [[-0.0639209, -0.168], [-0.467, 0]]
    problem.l = {-inf, -inf};
    problem.u = {-0.034974, 0.46571};
    if (int rc = five_copies(problem, primal_dense)) return rc;
    smooth::feedback::QuadraticProgramSparse sp;  // :383-388 sparseView(): structural zeros dropped
    sp.n = 2; sp.m = 2;
    sp.P_colptr = {0, 1, 2}; sp.P_rowind = {0, 1}; sp.P_val = {0.0100131, 0.01};
    sp.A_rowptr = {0, 2, 3}; sp.A_colind = {0, 1, 0}; sp.A_val = {-0.0639209, -0.168, -0.467};
    sp.q = problem.q; sp.l = problem.l; sp.u = problem.u;
    if (int rc = five_copies(sp, primal_sparse)) return 100 + rc;
    // PartialDynamic (:124-147): static N, dynamic M -- P = I, q = (-4, .25), A = I, -1 <= x <= 1
    smooth::feedback::QuadraticProgram<-1, 2> pd;
    pd.resize(2, 2);
    pd.P = {1, 0, 0, 1}; pd.q = {-4, 0.25}; pd.A = {1, 0, 0, 1}; pd.l = {-1, -1}; pd.u = {1, 1};
    const auto sol = smooth::feedback::solve_qp(pd, smooth::feedback::QPSolverParams{.polish = true});
    if (sol.code != smooth::feedback::QPSolutionStatus::Optimal) return 200;
    primal_partial[0] = sol.primal[0]; primal_partial[1] = sol.primal[1]; primal_partial[2] = sol.objective;
    // hot start from its own solution (:69-72)
    const auto hs = smooth::feedback::solve_qp(pd, smooth::feedback::QPSolverParams{.polish = true}, sol);
    if (hs.code != smooth::feedback::QPSolutionStatus::Optimal) return 201;
    primal_partial[3] = hs.primal[0]; primal_partial[4] = hs.primal[1];
    // a static size that does not match the data is refused
    smooth::feedback::QuadraticProgram<3, 2> bad;
    bad.resize(2, 2);
    try {
      smooth::feedback::QPSolver<decltype(bad)> s(bad);
      return 202;
    } catch (const std::invalid_argument &) {}
    // the OSQP comparator's parameter mapping (compat/osqp.hpp:54-80)
    const auto os = smooth::feedback::osqp_settings_from(smooth::feedback::QPSolverParams{.max_iter = 77, .stop_check_iter = 10});
    if (os.max_iter != 77 || os.check_termination != 10 || os.adaptive_rho != 0 || os.scaled_termination != 0 || os.time_limit != 0.0)
      return 203;
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}

int sfbx_test_solve_after_solve_batch(double * out)
{
  // solve(A1) -> solve_batch(B = 1, A2) -> solve(A1): the batch call shares the plan's device workspace with solve(), so
  // the third call must NOT vouch for the factor of the first (reuse_factor) although it sees the same matrices again.
  // Scaling off: c = 1 for both problems, i.e. the kernel's own re-check of c cannot tell them apart.
  try {
    QuadraticProgramSparse<> sp;
    sp.n = 2; sp.m = 2;
    sp.P_colptr = {0, 1, 2}; sp.P_rowind = {0, 1}; sp.P_val = {1.0, 2.0};
    sp.A_rowptr = {0, 2, 3}; sp.A_colind = {0, 1, 0}; sp.A_val = {1.0, 1.0, 1.0};
    sp.q = {-1.0, -1.0}; sp.l = {-0.5, -0.25}; sp.u = {0.5, 0.25};
    QPSolverParams prm;
    prm.scaling = false; prm.polish = false;
    SparseQPSolver solver(sp, prm);
    const auto first = solver.solve(sp);
    const int64_t reuse0 = solver.factor_reuse_count();
    auto other = sp;
    other.A_val = {3.0, -2.0, 0.5};
    double x[2], y[2], obj;
    uint32_t it;
    int32_t code;
    solver.solve_batch(1, other.P_val.data(), other.q.data(), other.A_val.data(), other.l.data(), other.u.data(), nullptr, nullptr,
                       x, y, &obj, &it, &code);
    const auto third = solver.solve(sp);
    out[0] = (third.primal == first.primal && third.dual == first.dual && third.iter == first.iter && third.code == first.code) ? 1.0 : 0.0;
    out[1] = double(solver.factor_reuse_count() - reuse0);  // 0: the third call was not flagged
    const auto fourth = solver.solve(sp);                    // now the previous call on the workspace WAS solve(sp)
    out[2] = double(solver.factor_reuse_count() - reuse0);  // 1
    out[3] = (fourth.primal == first.primal && fourth.iter == first.iter) ? 1.0 : 0.0;
    out[4] = first.primal[0]; out[5] = first.primal[1]; out[6] = x[0]; out[7] = x[1];
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}

int sfbx_test_ocp_to_qp_basic(double * out, int solve)
{
  // tests/test_ocp_to_qp.cpp:41-107 (OcpToQp.Basic) with the generic front, through the reference's include path:
  // theta = |xf|^2 + 2 q, f = (v, u), g = u^2, cr = u in [-1, 1], ce = xf in [-5, 5]^2, Mesh<5,5> refined to two
  // intervals of 5 nodes, tf = 2, linearised around xl(t) = (0.05 t^2, 0.1 t), ul = 0.1.
  namespace sf = smooth::feedback;
  using X2 = sf::Rn<2>;
  using U1 = sf::Rn<1>;
  try {
    const auto theta = [](double, const X2 &, const X2 & xf, const sf::Vec<1> & q) { return xf.v[0] * xf.v[0] + xf.v[1] * xf.v[1] + 2 * q[0]; };
    const auto f     = [](double, const X2 & x, const U1 & u) { return sf::Vec<2>{x.v[1], u.v[0]}; };
    const auto g     = [](double, const X2 &, const U1 & u) { return sf::Vec<1>{u.v[0] * u.v[0]}; };
    const auto cr    = [](double, const X2 &, const U1 & u) { return sf::Vec<1>{u.v[0]}; };
    const auto ce    = [](double, const X2 &, const X2 & xf, const sf::Vec<1> &) { return sf::Vec<2>{xf.v[0], xf.v[1]}; };
    const auto ocp   = sf::make_ocp<X2, U1>(theta, f, g, cr, {-1.0}, {1.0}, ce, {-5.0, -5.0}, {5.0, 5.0});
    const sf::Mesh mesh(2, 5);  // Mesh<5,5> + refine_ph(0, 10)
    constexpr double tf = 2.;
    const auto xl_fun = [](double t) { X2 x; x.v = {0.05 * t * t, 0.1 * t}; return x; };
    const auto ul_fun = [](double) { U1 u; u.v = {0.1}; return u; };
    const auto qp = sf::ocp_to_qp(ocp, mesh, tf, xl_fun, ul_fun);
    const int N = mesh.N_colloc();
    // :81-86 sizes
    out[0] = qp.n; out[1] = qp.m; out[2] = (double)qp.q.size(); out[3] = (double)qp.l.size(); out[4] = (double)qp.u.size();
    out[5] = (double)(qp.P_colptr.size() - 1); out[6] = (double)(qp.A_rowptr.size() - 1);
    // :90-106 the exact trajectory satisfies the constraints
    const double x0 = 3, v0 = -0.3, u0 = 0.1;
    std::vector<double> var(qp.n);
    for (int i = 0; i <= N; ++i) {
      const double t = tf * mesh.node(i);
      var[2 * i] = x0 + v0 * t + u0 * t * t / 2; var[2 * i + 1] = v0 + u0 * t;
    }
    for (int i = 0; i < N; ++i) var[2 * (N + 1) + i] = u0;
    double lo = 1e300, hi = 1e300;
    for (int r = 0; r < qp.m; ++r) {
      double s = 0.0;
      for (int q = qp.A_rowptr[r]; q < qp.A_rowptr[r + 1]; ++q) s += qp.A_val[q] * var[qp.A_colind[q]];
      lo = std::min(lo, s - qp.l[r]);
      hi = std::min(hi, qp.u[r] - s);
    }
    out[7] = lo; out[8] = hi;
    // cost entries: d2(theta)/dxf^2 / 2 = I on the x_N block, qo_q w_i tf d2g/du^2 = 2 * w_i tf * 2 on u_i
    out[9]  = qp.P_val[qp.P_colptr[2 * N + 1] - 1];                    // P(x_N[0], x_N[0])
    out[10] = qp.P_val[qp.P_colptr[2 * (N + 1) + 1] - 1] / (mesh.weight(0) * tf);  // P(u_0, u_0) / (w_0 tf) = 4
    out[11] = qp.q[2 * (N + 1)] / (mesh.weight(0) * tf);               // qo_q dg/du = 2 * 2 ul = 0.4
    if (solve) {  // examples/ocp_se2_qp.cpp:33-41: solve_qp + qpsol_to_ocpsol
      sf::QPSolverParams prm;
      prm.max_iter = 4000;
      const auto sol  = sf::solve_qp(qp, prm);
      const auto osol = sf::qpsol_to_ocpsol(ocp, mesh, sol, tf, xl_fun, ul_fun);
      out[12] = (double)(int)sol.code;
      out[13] = osol.u(0.5).v[0];
      out[14] = osol.x(tf).v[0];
      out[15] = osol.x(tf).v[1];
      out[16] = osol.x(0.0).v[0];
      // the interpolated state equals linearisation + node value at a node
      out[17] = osol.x(tf * mesh.node(3)).v[1] - (xl_fun(tf * mesh.node(3)).v[1] + sol.primal[2 * 3 + 1]);
    }
    return 0;
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sfbx_test_ocp_to_qp_basic: %s\n", e.what());
    return -1;
  }
}

int sfbx_test_ocp_to_qp_parabola(double * out)
{
  // tests/test_ocp_to_qp.cpp:41-107: double integrator x = (p, v), f = (v, u), Mesh<5,5> refined to two intervals
  // (K = 10), tf = 2, linearised around xl(t) = (0.05 t^2, 0.1 t), ul = 0.1; the exact trajectory
  // x(t) = (3 - 0.3 t + 0.05 t^2, -0.3 + 0.1 t), u = 0.1 must satisfy l <= A var <= u to 1e-8 (:105-106).
  // Here through the MPC transcription (same collocation rows; the MPC pins x_0 where that OCP boxes x_f).
  using X2 = Rn<2>;
  using U1 = Rn<1>;
  struct Dyn {
    Vec<2> operator()(const X2 & x, const U1 & u) const { return {x.v[1], u.v[0]}; }
  };
  struct Cr {
    Vec<1> operator()(const X2 &, const U1 & u) const { return {u.v[0]}; }
  };
  MPCParams p;
  p.K = 10; p.tf = 2.0;
  MPC<double, X2, U1, Dyn, Cr, 5> mpc(Dyn{}, Cr{}, {-1.0}, {1.0}, p);
  mpc.set_xdes([](double t) { X2 x; x.v = {0.05 * t * t, 0.1 * t}; return x; }, [](double t) { return Vec<2>{0.1 * t, 0.1}; });
  mpc.set_udes([](double) { U1 u; u.v = {0.1}; return u; });
  auto xtraj = [](double t) { X2 x; x.v = {3.0 - 0.3 * t + 0.05 * t * t, -0.3 + 0.1 * t}; return x; };
  const auto & qp = mpc.qp();
  std::vector<double> Av(qp.A_val.size()), l(qp.m), u(qp.m), var(qp.n);
  mpc.assemble(0.0, xtraj(0.0), Av.data(), l.data(), u.data());
  const int N = mpc.N();
  for (int i = 0; i <= N; ++i) {
    const auto x = xtraj(p.tf * mpc.mesh().node(i));
    var[2 * i] = x.v[0]; var[2 * i + 1] = x.v[1];
  }
  for (int i = 0; i < N; ++i) var[mpc.uvar_B() + i] = 0.1;
  // NOTE the MPC works in error coordinates around (xdes, udes): the exact trajectory enters as its deviation
  for (int i = 0; i <= N; ++i) {
    const double t = p.tf * mpc.mesh().node(i);
    var[2 * i] -= 0.05 * t * t; var[2 * i + 1] -= 0.1 * t;
  }
  for (int i = 0; i < N; ++i) var[mpc.uvar_B() + i] -= 0.1;
  double lo = 1e300, hi = 1e300;  // min(A var - l), min(u - A var) over all rows (dyn, cr, ce)
  for (int r = 0; r < qp.m; ++r) {
    double s = 0.0;
    for (int q = qp.A_rowptr[r]; q < qp.A_rowptr[r + 1]; ++q) s += Av[q] * var[qp.A_colind[q]];
    lo = std::min(lo, s - l[r]);
    hi = std::min(hi, u[r] - s);
  }
  out[0] = lo; out[1] = hi; out[2] = (double)N; out[3] = (double)qp.n; out[4] = (double)qp.m; out[5] = (double)mpc.mesh().N_ivals();
  return 0;
}

int sfbx_test_mpc_doubleintegrator(int ticks, double * u_out, uint32_t * iters, int32_t * codes, double * u_ref, uint32_t * iters_ref,
                                   int64_t * reuse_count, double * seconds)
{
  // examples/mpc_doubleintegrator.cpp:31-101: x = (p, v), f = (v, u), -0.5 <= u <= 0.5, K = 20, tf = 5,
  // xdes(t) = (-0.5 sin 0.3 t, 0), 50 ms ticks, closed loop (here integrated exactly: u is constant over a tick).
  // A linear system: the QP matrices are the same at every tick, only l and u move -- the solver front recognises
  // that and the kernel keeps scaling and factor (sfb_qp_params::reuse_factor).  Second closed loop with the
  // reuse switched off (SFB_QP_NO_REUSE=1): inputs and iteration counts must be identical.
  using X2 = Rn<2>;
  using U1 = Rn<1>;
  struct Dyn {
    Vec<2> operator()(const X2 & x, const U1 & u) const { return {x.v[1], u.v[0]}; }
  };
  struct Cr {
    Vec<1> operator()(const X2 &, const U1 & u) const { return {u.v[0]}; }
  };
  try {
    for (int pass = 0; pass < 2; ++pass) {
      setenv("SFB_QP_NO_REUSE", pass ? "1" : "0", 1);
      MPCParams p;
      p.K = 20; p.tf = 5.0;
      MPC<double, X2, U1, Dyn, Cr> mpc(Dyn{}, Cr{}, {-0.5}, {0.5}, p);
      mpc.set_xdes([](double t) { X2 x; x.v = {-0.5 * std::sin(0.3 * t), 0.0}; return x; },
                   [](double t) { return Vec<2>{-0.15 * std::cos(0.3 * t), 0.0}; });
      mpc.set_udes([](double) { U1 u; u.v = {0.0}; return u; });
      X2 x; x.v = {0.6, -0.2};
      const double dt = 0.05;
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < ticks; ++k) {
        const auto [u, code] = mpc(k * dt, x);
        (pass ? u_ref : u_out)[k]      = u.v[0];
        (pass ? iters_ref : iters)[k]  = mpc.solver().sol().iter;
        if (!pass) codes[k] = (int32_t)code;
        x.v = {x.v[0] + dt * x.v[1] + 0.5 * dt * dt * u.v[0], x.v[1] + dt * u.v[0]};
      }
      seconds[pass] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (!pass) *reuse_count = mpc.solver().factor_reuse_count();
    }
    unsetenv("SFB_QP_NO_REUSE");
    return 0;
  } catch (const std::exception & e) {
    unsetenv("SFB_QP_NO_REUSE");
    std::fprintf(stderr, "sfbx_test_mpc_doubleintegrator: %s\n", e.what());
    return -1;
  }
}

int sfbx_test_mpc_time_and_setters(double * out)
{
  // (1) set_xdes_rel / set_udes_rel (mpc.hpp:539-586) against the absolute-time setters, Time = double
  using sfbx::VehicleModel6;
  const VehicleModel6 mdl{};
  MPCParams p;
  p.K = 10; p.tf = 2.5;
  const double t0 = 0.75, t = 1.3;
  MPC6 a(mdl.f, mdl.cr, {-0.5, -0.5}, {0.5, 0.5}, p), b(mdl.f, mdl.cr, {-0.5, -0.5}, {0.5, 0.5}, p);
  a.set_xdes([mdl, t0](double ta) { return mdl.xdes(ta - t0); }, [mdl, t0](double ta) { return mdl.dxdes(ta - t0); });
  a.set_udes([mdl, t0](double ta) { return mdl.udes(ta - t0); });
  b.set_xdes_rel([mdl](double tr) { return mdl.xdes(tr); }, t0);  // velocity by central differences
  b.set_udes_rel([mdl](double tr) { return mdl.udes(tr); }, t0);
  const X6 x = perturbed(mdl.xdes(t - t0), 17);
  const int nA = (int)a.qp().A_val.size(), m = a.qp().m;
  std::vector<double> Aa(nA), la(m), ua(m), Ab(nA), lb(m), ub(m);
  a.assemble(t, x, Aa.data(), la.data(), ua.data());
  b.assemble(t, x, Ab.data(), lb.data(), ub.data());
  double d = 0;
  for (int e = 0; e < nA; ++e) d = std::max(d, std::fabs(Aa[e] - Ab[e]));
  for (int e = 0; e < m; ++e) d = std::max({d, std::fabs(la[e] - lb[e]), std::fabs(ua[e] - ub[e])});
  out[0] = d;  // finite-difference velocity vs the analytic one: ~1e-9
  // (2) the Time concept (time.hpp:25-89): the same controller on a std::chrono clock
  using Clock = std::chrono::steady_clock;
  using TP    = Clock::time_point;
  using MPC6c = MPC<TP, X6, U2, VehicleDyn6, sfbx::InputBox<X6>>;
  static_assert(Time<TP> && Time<std::chrono::nanoseconds> && Time<double>);
  const TP epoch = TP{} + std::chrono::hours(1000);
  MPC6c c(mdl.f, mdl.cr, {-0.5, -0.5}, {0.5, 0.5}, p);
  c.set_xdes_rel([mdl](double tr) { return mdl.xdes(tr); }, time_trait<TP>::plus(epoch, t0));
  c.set_udes_rel([mdl](double tr) { return mdl.udes(tr); }, time_trait<TP>::plus(epoch, t0));
  std::vector<double> Ac(nA), lc(m), uc(m);
  c.assemble(time_trait<TP>::plus(epoch, t), x, Ac.data(), lc.data(), uc.data());
  d = 0;
  for (int e = 0; e < nA; ++e) d = std::max(d, std::fabs(Ac[e] - Ab[e]));
  for (int e = 0; e < m; ++e) d = std::max({d, std::fabs(lc[e] - lb[e]), std::fabs(uc[e] - ub[e])});
  out[1] = d;  // nanosecond clock resolution against double seconds: ~1e-8
  // (3) set_weights (mpc.hpp:593-598): stored, and -- as in the reference at v1 -- not transcribed into P
  const std::vector<double> P0 = a.qp().P_val;
  MPCWeights<X6, U2> w;
  w.Q(0, 0) = 7.0; w.R(1, 1) = 3.0; w.Qtf(2, 2) = 5.0;
  a.set_weights(w);
  out[2] = (a.qp().P_val == P0) ? 1.0 : 0.0;
  out[3] = (a.weights().Q(0, 0) == 7.0 && a.weights().R(1, 1) == 3.0 && a.weights().Qtf(2, 2) == 5.0) ? 1.0 : 0.0;
  MPC6 wctor(mdl.f, mdl.cr, {-0.5, -0.5}, {0.5, 0.5}, p, w);  // the constructor is what transcribes weights
  out[4] = (wctor.qp().P_val != P0) ? 1.0 : 0.0;
  // (4) a new desired trajectory after the analysis: looked at lazily, re-analysed only when it leaves the analysed
  //     structure, never while a device-resident swarm pins the plan (the symbolic analysis is host code: no GPU)
  std::vector<uint8_t> keep;
  b.probe_default(t, keep);
  b.analyze_solver(&keep);
  const sfb_sparse_qp_plan * plan0 = b.solver().plan();
  b.set_udes_rel([mdl](double tr) { return mdl.udes(tr); }, t0 + 0.1);  // same structure
  b.refresh_structure(t);
  out[5] = (b.solver().plan() == plan0) ? 1.0 : 0.0;
  // a trajectory with another structure: a straight line along x leaves the SE2 adjoint block emptier / other entries
  b.set_xdes([](double ta) { X6 g; g.part<0>() = SE2{0.3 * ta, 1.0, 1.0, 0.0}; g.part<1>().v = {0.3, 0.0, 0.0}; return g; },
             [](double) { return Vec<6>{0.3, 0, 0, 0, 0, 0}; });
  b.solver().pin_plan();
  b.refresh_structure(t);
  out[6] = (b.solver().plan() == plan0) ? 1.0 : 0.0;  // pinned: untouched
  bool threw = false;
  try { b.solver().reset(); } catch (const std::logic_error &) { threw = true; }
  out[7] = threw ? 1.0 : 0.0;
  b.solver().unpin_plan();
  b.set_udes_rel([mdl](double tr) { return mdl.udes(tr); }, t0);  // marks the structure dirty again
  b.refresh_structure(t);
  out[8] = b.solver().analyzed() ? 1.0 : 0.0;  // still analysed (same plan or a new one that covers both structures)
  return 0;
}

int sfbx_test_mpc_se2(double * u_out, int32_t * codes, int32_t * traj_sizes)
{
  // tests/test_mpc.cpp:34-58,77-117
  struct Dyn {
    Vec<3> operator()(const SE2 &, const U2 & u) const { return {u.v[0], 0.0, u.v[1]}; }
  };
  struct Cr {
    Vec<2> operator()(const SE2 &, const U2 & u) const { return {u.v[0], u.v[1]}; }
  };
  try {
    for (int pass = 0; pass < 2; ++pass) {
      MPCParams p;  // defaults: K = 10, tf = 1 (tests/test_mpc.cpp:77)
      p.warmstart = (pass == 0);
      MPC<double, SE2, U2, Dyn, Cr> mpc(Dyn{}, Cr{}, {-1, -1}, {1, 1}, p);
      mpc.set_udes([](double) { U2 u; u.v = {1.0, 1.0}; return u; });                      // :91
      mpc.set_xdes([](double) { return SE2::Identity(); }, [](double) { return Vec<3>{}; });  // :92
      const SE2 x = rplus(SE2::Identity(), SE2::Tangent{0.3, -0.2, 0.25});
      std::vector<U2> ut;
      std::vector<SE2> xt;
      for (int k = 0; k < 3; ++k) {  // t = 2, 3, 4 (:95-108)
        auto [u, code] = mpc(2.0 + k, x, &ut, &xt);
        u_out[(pass * 3 + k) * 2] = u.v[0]; u_out[(pass * 3 + k) * 2 + 1] = u.v[1];
        codes[pass * 3 + k] = (int32_t)code;
      }
      traj_sizes[pass * 2] = (int32_t)ut.size(); traj_sizes[pass * 2 + 1] = (int32_t)xt.size();
    }
  } catch (const std::exception &) {
    return -2;
  }
  return 0;
}

// ---- tests/test_mpc.cpp:34-155 written against <smooth/feedback/mpc.hpp> with only the Lie types renamed
// (smooth::SE2d -> SE2, Eigen::Vector2d -> Rn<2> / Vec<2>, generic `template<typename S>` functors -> double) ----
namespace ref_mpc_test {
using T = double;
using X = SE2;
using U = U2;

struct MyDynamics {  // tests/test_mpc.cpp:34-45
  Vec<3> operator()(const X &, const U & u) const { return {u.v[0], 0.0, u.v[1]}; }
  inline void set_time(double t) { t_ = t; }
  double t_{0};
};
struct MyRunningConstraints {  // :47-58
  Vec<2> operator()(const X &, const U & u) const { return {u.v[0], u.v[1]}; }
  inline void set_time(double t) { t_ = t; }
  double t_{0};
};
using MPC_t = smooth::feedback::MPC<T, X, U, MyDynamics, MyRunningConstraints>;  // :60
// StaticProperties, :62-68
static_assert(std::is_copy_constructible_v<MPC_t>);
static_assert(std::is_copy_assignable_v<MPC_t>);
static_assert(std::is_move_constructible_v<MPC_t>);
static_assert(std::is_move_assignable_v<MPC_t>);
// special type that maintains references to f and cr, :70-71
using MPC_reft = smooth::feedback::MPC<T, X, U, MyDynamics &, MyRunningConstraints &>;
static_assert(MPC_t::Ncr == 2 && MPC_reft::Ncr == 2);  // mpc.hpp:383: read off CR's result

double rel_diff(const U & a, const U & b)
{
  const double d = std::hypot(a.v[0] - b.v[0], a.v[1] - b.v[1]);
  return d / std::max(1e-300, std::min(std::hypot(a.v[0], a.v[1]), std::hypot(b.v[0], b.v[1])));
}
}  // namespace ref_mpc_test

int sfbx_test_mpc_front_host(double * out)
{
  // The host-only half of the reference-shaped front (no solve, no GPU): what tests/test_mpc.cpp relies on before it ever calls
  // operator() -- template parameters in the reference's order with Ncr read off CR, default construction, copies that share
  // the desired trajectories (mpc.hpp:407, 607-608) and own everything else, functors by reference.
  using namespace ref_mpc_test;
  const X x = rplus(X::Identity(), X::Tangent{0.3, -0.2, 0.25});
  MyDynamics f{};
  MyRunningConstraints cr{};
  Vec<2> crl{1, 1};
  MPC_t mpc{f, cr, Vec<2>{-1, -1}, crl};
  MPC_t copy = mpc;  // copy construction
  MPC_t assigned;    // default construction, copy assignment
  assigned = mpc;
  const int nA = (int)mpc.qp().A_val.size(), m = mpc.qp().m;
  std::vector<double> A0(nA), l0(m), u0(m), A1(nA), l1(m), u1(m), A2(nA), l2(m), u2(m);
  mpc.assemble(0.5, x, A0.data(), l0.data(), u0.data());
  copy.set_udes([](T) -> U { U u; u.v = {0.25, -0.5}; return u; });  // a setter on the COPY ...
  mpc.assemble(0.5, x, A1.data(), l1.data(), u1.data());              // ... is seen by the original
  assigned.assemble(0.5, x, A2.data(), l2.data(), u2.data());         // ... and by the other copy
  double d01 = 0, d12 = 0;
  for (int e = 0; e < m; ++e) { d01 = std::max(d01, std::fabs(l0[e] - l1[e])); d12 = std::max(d12, std::fabs(l1[e] - l2[e])); }
  out[0] = d01;  // > 0: the running-constraint bounds moved with udes
  out[1] = d12;  // == 0
  out[2] = (copy.qp().A_val.data() != mpc.qp().A_val.data()) ? 1.0 : 0.0;  // own QP storage
  out[3] = (!copy.solver().analyzed() && !assigned.solver().analyzed()) ? 1.0 : 0.0;  // no analysis travels with a copy
  // functors held BY REFERENCE see set_time through operator()'s assembly path only; the const assembly used by swarms tells copies
  MPC_reft byref{f, cr, Vec<2>{-1, -1}, crl};
  byref.assemble(7.0, x, A0.data(), l0.data(), u0.data());
  out[4] = f.t_;  // untouched by the const path: 0
  out[5] = (double)MPC_t::Ncr + 10.0 * (double)MPC_reft::Nx + 100.0 * (double)MPC_reft::Nu;  // 2 + 30 + 200
  out[6] = (std::is_default_constructible_v<MPC_t> && std::is_copy_assignable_v<MPC_t> && std::is_move_constructible_v<MPC_t>) ? 1.0 : 0.0;
  return 0;
}

int sfbx_test_mpc_api(double * out, int32_t * codes)
{
  using namespace ref_mpc_test;
  try {
    const X x = rplus(X::Identity(), X::Tangent{0.3, -0.2, 0.25});  // X::Random()
    {  // TEST(Mpc, Api), :73-117
      MyDynamics f{};
      MyRunningConstraints cr{};
      Vec<2> crl{1, 1};
      MPC_reft mpc{f, cr, Vec<2>{-1, -1}, crl};

      // nothing set
      auto [u0, code0] = mpc(1, x);
      codes[0] = (int32_t)code0;

      mpc.reset_warmstart();

      mpc.set_weights({
        .Q   = Mat<3, 3>::Identity(),
        .Qtf = Mat<3, 3>::Identity(),
        .R   = Mat<2, 2>::Identity(),
      });

      mpc.set_udes([](T) -> U { U u; u.v = {1.0, 1.0}; return u; });
      mpc.set_xdes_rel([](double) -> X { return X::Identity(); });

      // no warmstart
      auto [u1, code1] = mpc(2, x);
      codes[1] = (int32_t)code1;

      // with warmstart
      auto [u2, code2] = mpc(3, x);
      codes[2] = (int32_t)code2;

      out[0] = rel_diff(u1, u2);

      // output stuff
      std::vector<X> xs;
      std::vector<U> us;
      auto [u3, code3] = mpc(4, x, us, xs);
      codes[3] = (int32_t)code3;

      out[1] = rel_diff(u3, u1);
      out[2] = (us.size() + 1 == xs.size()) ? 1.0 : 0.0;
      out[3] = f.t_;   // ASSERT_GE(f.t_, 4)
      out[4] = cr.t_;  // ASSERT_GE(cr.t_, 4)
      // (not in the reference's test) the nullable-pointer overload gives the same answer
      std::vector<U> us2;
      auto [u3p, code3p] = mpc(4, x, &us2);
      out[5] = (code3p == code3 && us2.size() == us.size()) ? rel_diff(u3p, u3) : 1.0;
    }
    {  // TEST(Mpc, Constructors), :119-155
      MyDynamics f{};
      MyRunningConstraints cr{};
      Vec<2> crl{1, 1};
      MPC_t mpc{f, cr, Vec<2>{-1, -1}, crl};

      mpc.reset_warmstart();

      mpc.set_weights({
        .Q   = Mat<3, 3>::Identity(),
        .Qtf = Mat<3, 3>::Identity(),
        .R   = Mat<2, 2>::Identity(),
      });

      mpc.set_udes([](T) -> U { U u; u.v = {1.0, 1.0}; return u; });
      mpc.set_xdes_rel([](double) -> X { return X::Identity(); });

      auto [u1, code1] = mpc(0, x);

      // copy construction
      auto mpc2        = mpc;
      auto [u2, code2] = mpc2(0, x);

      // copy assignment
      MPC_t mpc3;
      mpc3             = mpc;
      auto [u3, code3] = mpc3(0, x);

      // (not in the reference's test) mpc.hpp:407, 607-608: copies SHARE the desired trajectories -- a setter on the copy
      // is seen by the original -- and own their solvers (qp_solver.hpp:209-231: a copy analyses again)
      out[11] = (mpc2.solver().plan() != nullptr && mpc2.solver().plan() != mpc.solver().plan() &&
                 mpc3.solver().plan() != mpc.solver().plan()) ? 1.0 : 0.0;
      mpc3.set_udes([](T) -> U { U u; u.v = {0.25, -0.5}; return u; });
      auto [u6, code6] = mpc(0, x);  // the ORIGINAL, after the copy's setter
      MPC_t fresh{f, cr, Vec<2>{-1, -1}, crl};
      fresh.set_udes([](T) -> U { U u; u.v = {0.25, -0.5}; return u; });
      fresh.set_xdes_rel([](double) -> X { return X::Identity(); });
      auto [u7, code7] = fresh(0, x);
      out[12] = rel_diff(u6, u7);
      out[13] = rel_diff(u6, u1);  // and it did change the answer
      codes[9] = (int32_t)code6; codes[10] = (int32_t)code7;
      mpc3.set_udes([](T) -> U { U u; u.v = {1.0, 1.0}; return u; });
      mpc.reset_warmstart();

      // move construction
      auto mpc4        = std::move(mpc);
      auto [u4, code4] = mpc4(0, x);

      // move assignment
      MPC_t mpc5;
      mpc5             = std::move(mpc2);
      auto [u5, code5] = mpc5(0, x);

      out[6] = rel_diff(u1, u2);
      out[7] = rel_diff(u1, u3);
      out[8] = rel_diff(u1, u4);
      out[9] = rel_diff(u1, u5);
      codes[4] = (int32_t)code1; codes[5] = (int32_t)code2; codes[6] = (int32_t)code3; codes[7] = (int32_t)code4;
      codes[8] = (int32_t)code5;
    }
    {  // examples/mpc_asif_vehicle.cpp:27-64, 73-96: the controller's declaration with Time = std::chrono::duration<double>
      using Time = std::chrono::duration<double>;
      using Xd   = X6;
      using Ud   = U2;
      auto f = [](const Xd & x, const Ud & u) -> Vec<6> {
        const auto & v = x.part<1>().v;
        return {v[0], v[1], v[2], -0.2 * v[0] + u.v[0], 0.0, -0.4 * v[2] + u.v[1]};
      };
      auto cr = [](const Xd &, const Ud & u) -> Vec<2> { return {u.v[0], u.v[1]}; };
      Vec<2> crl{-0.5, -0.5};
      Vec<2> cru{0.5, 0.5};
      smooth::feedback::MPC<Time, Xd, Ud, decltype(f), decltype(cr)> mpc{
        f,
        cr,
        crl,
        cru,
        {.K = 30, .tf = 5},
      };
      auto xdes = [](double t) -> Xd { return sfbx::VehicleModel6{}.xdes(t); };
      mpc.set_weights({
        .Q   = Mat<6, 6>::Identity(),
        .Qtf = 0.1 * Mat<6, 6>::Identity(),
        .R   = Mat<2, 2>::Identity(),
      });
      mpc.set_xdes_rel(xdes);
      mpc.set_udes_rel([](double) -> Ud { return Ud{}; });
      using namespace std::chrono_literals;
      auto [u, code] = mpc(Time(25ms), perturbed(xdes(0.025), 3));
      codes[11] = (int32_t)code;
      out[10]   = std::max(std::fabs(u.v[0]), std::fabs(u.v[1]));  // inside cr's box
    }
    return 0;
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sfbx_test_mpc_api: %s\n", e.what());
    return -1;
  }
}

double sfbx_lie_selftest(void)
{
  double err = 0.0;
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> d(-1.5, 1.5);
  for (int it = 0; it < 200; ++it) {
    const SE2::Tangent a{d(rng), d(rng), d(rng)}, b{d(rng), d(rng), d(rng)};
    const SE2 g = SE2::exp(a);
    // log(exp(a)) == a
    const auto a2 = g.log();
    for (int i = 0; i < 3; ++i) err = std::max(err, std::fabs(a2[i] - a[i]));
    // rminus(rplus(g, b), g) == b
    const auto b2 = rminus(rplus(g, b), g);
    for (int i = 0; i < 3; ++i) err = std::max(err, std::fabs(b2[i] - b[i]));
    // g * g^-1 == identity
    const SE2 e = g * g.inverse();
    err = std::max({err, std::fabs(e.x), std::fabs(e.y), std::fabs(e.c - 1), std::fabs(e.s)});
    // dr_expinv(a) is the derivative of  h -> rminus(rplus(exp(a), h), identity)... check numerically:
    // log(exp(a) exp(h)) ~= a + dr_expinv(a) h
    const double hstep = 1e-6;
    const auto J = SE2::dr_expinv(a);
    for (int c = 0; c < 3; ++c) {
      SE2::Tangent h{};
      h[c] = hstep;
      const auto l2 = (SE2::exp(a) * SE2::exp(h)).log();
      for (int r = 0; r < 3; ++r) err = std::max(err, std::fabs((l2[r] - a[r]) / hstep - J(r, c)) * 1e-3);
    }
    // Bundle consistency
    X6 x;
    x.part<0>() = g;
    x.part<1>().v = {b[0], b[1], b[2]};
    X6::Tangent t6{a[0], a[1], a[2], b[0], b[1], b[2]};
    const auto t62 = rminus(rplus(x, t6), x);
    for (int i = 0; i < 6; ++i) err = std::max(err, std::fabs(t62[i] - t6[i]));
  }
  return err;
}

int sfbx_test_ekf(double * err)
{
  try {
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> d(-1.0, 1.0);
    // PredictTimeCut (tests/test_ekf.cpp:155-180): constant velocity b, tau = 0.7, dt = 0.5 -> xhat + b tau
    {
      EKF<Rn<2>> ekf;
      Rn<2> xhat; xhat.v = {d(rng), d(rng)};
      Mat<2, 2> P{}; P(0, 0) = d(rng) + 1.1; P(1, 1) = d(rng) + 1.1;
      ekf.reset(xhat, P);
      const Vec<2> b{d(rng), d(rng)};
      Mat<2, 2> Q{}; Q(0, 0) = d(rng) + 1.1; Q(1, 1) = d(rng) + 1.1;
      ekf.predict([&](double, const Rn<2> &) { return b; }, Q, 0.7, 0.5);
      const auto e = ekf.estimate();
      err[0] = std::max(std::fabs(e.v[0] - (xhat.v[0] + 0.7 * b[0])), std::fabs(e.v[1] - (xhat.v[1] + 0.7 * b[1])));
    }
    // UpdateLinear (tests/test_ekf.cpp:50-103), Nx = Ny = 3, diagonal P and R
    err[1] = err[2] = 0.0;
    for (int it = 0; it < 10; ++it) {
      EKF<Rn<3>> ekf;
      Rn<3> xhat; Vec<3> x, hoff;
      Mat<3, 3> P{}, H{}, R{};
      for (int i = 0; i < 3; ++i) { xhat.v[i] = d(rng); x[i] = d(rng); hoff[i] = d(rng); P(i, i) = d(rng) + 1.1; R(i, i) = d(rng) + 1.1; }
      for (auto & v : H.a) v = d(rng);
      ekf.reset(xhat, P);
      const Vec<3> y = [&] { Vec<3> t = H * x; for (int i = 0; i < 3; ++i) t[i] += hoff[i]; return t; }();
      ekf.update<3>([&](const Rn<3> & g) { Vec<3> t = H * g.v; for (int i = 0; i < 3; ++i) t[i] += hoff[i]; return t; }, y, R);
      // textbook: S = H P H' + R (diag P), K = P H' S^-1 by solving S z = r
      Mat<3, 3> S = R;
      for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) for (int k2 = 0; k2 < 3; ++k2) S(a, b2) += H(a, k2) * P(k2, k2) * H(b2, k2);
      Vec<3> r; { Vec<3> hx = H * xhat.v; for (int i = 0; i < 3; ++i) r[i] = y[i] - (hx[i] + hoff[i]); }
      // solve S z = r (Gaussian elimination, 3x3)
      double M[3][4];
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) M[i][j] = S(i, j); M[i][3] = r[i]; }
      for (int c = 0; c < 3; ++c) {
        int piv = c; for (int i = c + 1; i < 3; ++i) if (std::fabs(M[i][c]) > std::fabs(M[piv][c])) piv = i;
        for (int j = 0; j < 4; ++j) std::swap(M[c][j], M[piv][j]);
        for (int i = c + 1; i < 3; ++i) { const double fct = M[i][c] / M[c][c]; for (int j = c; j < 4; ++j) M[i][j] -= fct * M[c][j]; }
      }
      Vec<3> z;
      for (int i = 2; i >= 0; --i) { double sacc = M[i][3]; for (int j = i + 1; j < 3; ++j) sacc -= M[i][j] * z[j]; z[i] = sacc / M[i][i]; }
      const auto est = ekf.estimate();
      for (int i = 0; i < 3; ++i) {
        double xe = xhat.v[i];
        for (int a = 0; a < 3; ++a) xe += P(i, i) * H(a, i) * z[a];
        err[1] = std::max(err[1], std::fabs(est.v[i] - xe));
      }
      // covariance: trace must not increase and stays symmetric
      const auto Pn = ekf.covariance();
      double tr0 = 0, tr1 = 0;
      for (int i = 0; i < 3; ++i) { tr0 += P(i, i); tr1 += Pn(i, i); for (int j = 0; j < 3; ++j) err[2] = std::max(err[2], std::fabs(Pn(i, j) - Pn(j, i))); }
      if (tr1 > tr0 + 1e-12) err[2] = 1.0;
    }
    // SE2 smoke (tests/test_ekf.cpp:31-48 NoCrash, on SE2): predict with substeps, then a position measurement
    {
      EKF<SE2> ekf;
      ekf.reset(SE2::Identity(), Mat<3, 3>::Identity());
      ekf.predict([](double, const SE2 &) { return Vec<3>{1.0, 0.0, 0.3}; }, Mat<3, 3>::Identity(), 1.0, 0.6);
      ekf.update<2>([](const SE2 & g) { return Vec<2>{g.x, g.y}; }, Vec<2>{1.0, 0.2}, Mat<2, 2>::Identity());
      ekf.predict([](double, const SE2 &) { return Vec<3>{1.0, 0.0, 0.3}; }, Mat<3, 3>::Identity(), 1.0, 0.1);
      const auto Pn = ekf.covariance();
      for (const double v : Pn.a) if (!std::isfinite(v)) return -3;
    }
  } catch (const std::exception &) {
    return -2;
  }
  return 0;
}

int sfbx_mesh(int n_ivals, int K, double * nodes, double * weights, double * Dus)
{
  Mesh m(n_ivals, K);
  for (int i = 0; i <= m.N_colloc(); ++i) {
    nodes[i] = m.node(i);
    weights[i] = m.weight(i);
  }
  std::copy(m.Dus.begin(), m.Dus.end(), Dus);
  return 0;
}

}  // extern "C"

// ---- ASIF ----
namespace {
// safe set and backup controller of examples/mpc_asif_vehicle.cpp:95-104: stay 0.7 away from (0, -2.3); the
// direction is evaluated at the query point and treated as constant by the differentiation, as in the example
Vec<1> vehicle_h(double t, const X6 & x) { return VehicleH{}(t, x); }
U2 vehicle_bu(double t, const X6 & x) { return VehicleBU{}(t, x); }
ASIFilterParams<U2> vehicle_asif_params(int K) { return sfbx::vehicle_asif_params(K); }
void copy_qp(const QuadraticProgram<> & qp, double * P, double * q, double * A, double * l, double * u)
{
  std::copy(qp.P.begin(), qp.P.end(), P);
  std::copy(qp.q.begin(), qp.q.end(), q);
  std::copy(qp.A.begin(), qp.A.end(), A);
  std::copy(qp.l.begin(), qp.l.end(), l);
  std::copy(qp.u.begin(), qp.u.end(), u);
}
}  // namespace

int sfbx_asif_basic_qp(const double * x0, const double * udes, double * P, double * q, double * A, double * l, double * u)
{
  ASIFProblem<SE2, U2> pbm;
  pbm.x0      = SE2::FromAngle(x0[0], x0[1], x0[2]);
  pbm.u_des.v = {udes[0], udes[1]};
  pbm.ulim.rows = 2;
  pbm.ulim.A    = {1, 0, 0, 1};
  pbm.ulim.l    = {-1, -1};
  pbm.ulim.u    = {1, 1};
  ASIFtoQPParams prm;
  prm.K = 3;
  const auto qp = asif_to_qp<SE2, U2>(
    pbm, prm, [](const SE2 &, const U2 & uu) { return Vec<3>{uu.v[0], 0.0, uu.v[1]}; },
    [](double, const SE2 & g) { return Vec<2>{g.x, g.y}; },
    [](double, const SE2 &) { U2 b; b.v = {-0.1, 1.0}; return b; });
  if (qp.n != 3 || qp.m != 9) return 1;
  copy_qp(qp, P, q, A, l, u);
  return 0;
}

int sfbx_test_asif(int which, double * u_out, int32_t * code, uint32_t * iter, int32_t * dims, double * P, double * q,
                   double * A, double * l, double * u, double * x, double * y)
{
  try {
    auto report = [&](const QuadraticProgram<> & qp, const QPSolution<> & sol, QPSolutionStatus c) {
      dims[0] = qp.n; dims[1] = qp.m;
      copy_qp(qp, P, q, A, l, u);
      std::copy(sol.primal.begin(), sol.primal.end(), x);
      std::copy(sol.dual.begin(), sol.dual.end(), y);
      *code = (int32_t)c;
      *iter = sol.iter;
    };
    if (which == 0) {
      using U3 = Rn<3>;
      auto f   = [](const SO3 &, const U3 & uu) { return uu.v; };
      ASIFilterParams<U3> prm;
      prm.nh     = 3;
      prm.asif.K = 100;
      ASIFilter<SO3, U3, decltype(f)> asif(f, prm);
      const SO3 g = SO3::exp({0.3, -0.5, 0.8});
      const auto [ua, c] = asif(g, U3{}, [](double, const SO3 & gg) { return gg.log(); },
                                [](double, const SO3 &) { U3 b; b.v = {1, 1, 1}; return b; });
      for (int i = 0; i < 3; ++i) u_out[i] = ua.v[i];
      report(asif.qp(), asif.last_solution(), c);
    } else {
      VehicleDyn6 f;
      ASIFilter<X6, U2, VehicleDyn6> asif(f, vehicle_asif_params(which == 1 ? 200 : 10));
      X6 xx = xdes6(3.7);
      xx.part<0>() = SE2::FromAngle(-2.0, 0.9, -1.9);  // heading towards the obstacle, 1 m away
      U2 ud; ud.v = {0.4, 0.1};
      const auto [ua, c] = asif(xx, ud, VehicleH{}, VehicleBU{});
      u_out[0] = ua.v[0]; u_out[1] = ua.v[1]; u_out[2] = 0.0;
      report(asif.qp(), asif.last_solution(), c);
    }
    return 0;
  } catch (const std::exception &) {
    return 1;
  }
}

/* one host EKF<> object per filter (every predict substep and update is a batch-of-one GPU call): the reference semantics
 * sfbx_ekf_swarm_device (models_device.hip) is checked against */
int sfbx_ekf_swarm_host(int64_t batch, int steps, int rk4, double tau, double dt, const double * states, const double * P0,
                        const double * y, double * states_out, double * P_out)
{
  try {
    const auto Q = sfbx::vehicle_ekf_Q();
    const auto R = sfbx::vehicle_ekf_R();
    auto run = [&](auto & ekf, int64_t b) {
      Mat<6, 6> P;
      std::copy(P0 + 36 * b, P0 + 36 * (b + 1), P.a.begin());
      ekf.reset(sfbx::vehicle_state(states + 7 * b), P);
      for (int k = 0; k < steps; ++k) {
        ekf.predict(sfbx::VehicleEkfDyn{}, Q, tau, dt > 0 ? std::optional<double>(dt) : std::nullopt);
        const double * yk = y + ((size_t)k * batch + b) * 3;
        ekf.template update<3>(sfbx::VehicleEkfMeas{}, Vec<3>{yk[0], yk[1], yk[2]}, R);
      }
      sfbx::vehicle_state_out(ekf.estimate(), states_out + 7 * b);
      const auto Pn = ekf.covariance();
      std::copy(Pn.a.begin(), Pn.a.end(), P_out + 36 * b);
    };
    for (int64_t b = 0; b < batch; ++b) {
      if (rk4) { EKF<X6, EKFStepper::RK4> e; run(e, b); }
      else { EKF<X6> e; run(e, b); }
    }
    return 0;
  } catch (const std::exception &) {
    return -2;
  }
}

int sfbx_asif_swarm_states(int64_t batch, uint64_t seed, double * states, double * udes)
{
  for (int64_t b = 0; b < batch; ++b) {
    const X6 g = perturbed(xdes6(0.025 * double(b % 400)), seed + (uint64_t)b);
    std::mt19937_64 rng(seed + 7919u * (uint64_t)b);
    std::uniform_real_distribution<double> d(-0.5, 0.5);
    const auto & p = g.part<0>();
    const auto & v = g.part<1>().v;
    double * s = states + 7 * b;
    s[0] = p.x; s[1] = p.y; s[2] = p.c; s[3] = p.s; s[4] = v[0]; s[5] = v[1]; s[6] = v[2];
    udes[2 * b] = d(rng);
    udes[2 * b + 1] = d(rng);
  }
  return 0;
}

int sfbx_asif_swarm_step(int64_t batch, uint64_t seed, int K, int ticks, double * u_out, int32_t * codes, uint32_t * iters,
                         double * P, double * q, double * A, double * l, double * u, double * x, double * y, double * wx,
                         double * wy)
{
  try {
    const int n = 3, m = K + 3;
    ASIFSwarm<X6, U2, VehicleDyn6> swarm(VehicleDyn6{}, (size_t)batch, vehicle_asif_params(K));
    std::vector<X6> g((size_t)batch);
    std::vector<U2> ud((size_t)batch);
    std::vector<double> tb((size_t)batch);
    for (int64_t b = 0; b < batch; ++b) {
      tb[b] = 0.025 * double(b % 400);
      g[b]  = perturbed(xdes6(tb[b]), seed + (uint64_t)b);
      std::mt19937_64 rng(seed + 7919u * (uint64_t)b);
      std::uniform_real_distribution<double> d(-0.5, 0.5);
      ud[b].v = {d(rng), d(rng)};
    }
    const VehicleH hb{};    // (SFB_ASIF_FD=1: plain lambdas instead -- forward differences, as before)
    const VehicleBU bub{};
    std::vector<U2> out;
    for (int tick = 0; tick < ticks; ++tick) {
      if (tick > 0) {  // move every vehicle 25 ms along its filtered input (explicit Euler)
        for (int64_t b = 0; b < batch; ++b) {
          auto dx = VehicleDyn6{}(g[b], out[b]);
          for (auto & v : dx) v *= 0.025;
          g[b] = rplus(g[b], dx);
        }
      }
      if (tick == ticks - 1) swarm.copy_warm_start(wx, wy);
      static const bool fd = [] { const char * v = std::getenv("SFB_ASIF_FD"); return v && v[0] == '1'; }();
      if (fd)
        out = swarm(g, ud, [](size_t, double t, const X6 & xx) { return vehicle_h(t, xx); },
                    [](size_t, double t, const X6 & xx) { return vehicle_bu(t, xx); });
      else
        out = swarm(g, ud, hb, bub);
    }
    for (int64_t b = 0; b < batch; ++b) { u_out[2 * b] = out[b].v[0]; u_out[2 * b + 1] = out[b].v[1]; }
    std::copy(swarm.codes().begin(), swarm.codes().end(), codes);
    std::copy(swarm.iterations().begin(), swarm.iterations().end(), iters);
    swarm.copy_problem(P, q, A, l, u, x, y);
    (void)n; (void)m;
    return 0;
  } catch (const std::exception &) {
    return 1;
  }
}

namespace {
template<int Nx>
void predict_linear_case(const double * Ac, const double * Fc, double * err)
{
  Mat<Nx, Nx> A{}, F{};
  for (int i = 0; i < Nx * Nx; ++i) { A.a[i] = Ac[i]; F.a[i] = Fc[i]; }
  std::mt19937_64 rng(11 + Nx);
  std::uniform_real_distribution<double> d(-1.0, 1.0);
  EKF<Rn<Nx>, EKFStepper::RK4> ekf;
  Rn<Nx> xhat;
  Mat<Nx, Nx> P{};
  for (int i = 0; i < Nx; ++i) { xhat.v[i] = d(rng); P(i, i) = d(rng) + 1.1; }
  ekf.reset(xhat, P);
  ekf.predict([&](double, const Rn<Nx> & x) { return A * x.v; }, Mat<Nx, Nx>::Zero(), 0.7, 1e-3);
  const Vec<Nx> xe = F * xhat.v;
  Mat<Nx, Nx> Ft{};
  for (int r = 0; r < Nx; ++r) for (int c = 0; c < Nx; ++c) Ft(r, c) = F(c, r);
  const Mat<Nx, Nx> Pe = F * P * Ft;
  double nx = 0, dx = 0, np = 0, dp = 0;
  const auto est = ekf.estimate();
  const auto cov = ekf.covariance();
  for (int i = 0; i < Nx; ++i) { nx = std::max(nx, std::fabs(xe[i])); dx = std::max(dx, std::fabs(xe[i] - est.v[i])); }
  for (int i = 0; i < Nx * Nx; ++i) { np = std::max(np, std::fabs(Pe.a[i])); dp = std::max(dp, std::fabs(Pe.a[i] - cov.a[i])); }
  err[0] = std::max(err[0], dx / nx);
  err[1] = std::max(err[1], dp / np);
}
}  // namespace

int sfbx_test_ekf_predict_linear(const double * A3, const double * F3, const double * A6, const double * F6, double * err)
{
  try {
    err[0] = err[1] = 0.0;
    predict_linear_case<3>(A3, F3, err);
    predict_linear_case<6>(A6, F6, err);
    return 0;
  } catch (const std::exception &) {
    return 1;
  }
}

int sfbx_test_ekf_predict_linear9(const double * A9, const double * F9, double * err)
{
  try {
    err[0] = err[1] = 0.0;
    predict_linear_case<9>(A9, F9, err);
    return 0;
  } catch (const std::exception &) {
    return 1;
  }
}
