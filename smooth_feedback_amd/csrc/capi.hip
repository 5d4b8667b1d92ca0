// C-ABI shim (include/sfb.h) over the HIP kernels.  No CPU fallback: every compute entry point
// needs a HIP device and fails with SFB_ERR_NO_DEVICE / SFB_ERR_HIP otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "knobs.h"
#include "../../include/sfb.h"
#include "capi_common.h"
#include "qp_dense_kernel.h"

namespace sfb {

namespace { thread_local std::string g_last_error; }

sfb_status fail(sfb_status st, const std::string &msg)
{
  g_last_error = msg;
  return st;
}

sfb_status hip_fail(hipError_t e, const char *what)
{
  // clear the sticky error so that later calls report their own failure
  (void)hipGetLastError();
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
    return fail(SFB_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
  return fail(SFB_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

sfb_status require_device()
{
  int cnt       = 0;
  hipError_t e  = hipGetDeviceCount(&cnt);
  if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
  if (cnt <= 0) return fail(SFB_ERR_NO_DEVICE, "no HIP device visible; the sfb library has no CPU fallback");
  return SFB_OK;
}

void verbose_report(const char *what, int64_t batch, int n, int m, double h2d_ms, double solve_ms, double d2h_ms,
                    const int32_t *code, const uint32_t *iter)
{
  static const char *names[7] = {"Optimal", "PolishFailed", "PrimalInfeasible", "DualInfeasible", "MaxIterations", "MaxTime", "Unknown"};
  long long hist[7] = {0, 0, 0, 0, 0, 0, 0};
  uint32_t imin = 0xFFFFFFFFu, imax = 0;
  double isum = 0.0;
  for (int64_t b = 0; b < batch; ++b) {
    if (code[b] >= 0 && code[b] < 7) hist[code[b]]++;
    if (iter) { imin = std::min(imin, iter[b]); imax = std::max(imax, iter[b]); isum += iter[b]; }
  }
  std::printf("[sfb] %s: %lld problem(s), n = %d, m = %d\n", what, (long long)batch, n, m);
  if (h2d_ms < 0.0)  // a sharded call: the shards' uploads, solves and downloads overlap -- one wall-clock figure for the call
    std::printf("[sfb]   time: %.3f ms for the whole call (upload, solve on the devices, download; the shards run side by side)\n", solve_ms);
  else
    std::printf("[sfb]   time: upload %.3f ms | solve on device (scaling, factorisation, ADMM, polish) %.3f ms | download %.3f ms\n",
                h2d_ms, solve_ms, d2h_ms);
  std::printf("[sfb]   status:");
  for (int c = 0; c < 7; ++c)
    if (hist[c]) std::printf(" %s %lld", names[c], hist[c]);
  if (iter && batch > 0) std::printf("\n[sfb]   iterations: min %u, mean %.1f, max %u", imin, isum / (double)batch, imax);
  std::printf("\n");
  std::fflush(stdout);
}

// The reference's closing summary of a verbose solve (qp_solver.hpp:550-565), from the per-phase times of the device
// (phase_us: scaling + pre-check | matrix filling | factorisation | iteration | polish | un-scale and report, microseconds).
// Its "Total time" runs from t0 (:376, after the scaling) to the end of the polish; the two parts outside are added as lines.
void verbose_summary(int32_t code, uint32_t iter, const double *phase_us)
{
  std::printf("QP solver summary:\n");
  std::printf("Result %d\n", (int)code);
  std::printf("%-25s%10lld\n", "Iterations", (long long)iter - 1);  // (the reference prints iter - 1, :556)
  std::printf("%-26s%10.0f\n", "Total time (\xC2\xB5s)", phase_us[1] + phase_us[2] + phase_us[3] + phase_us[4]);
  std::printf("%-25s%10.0f\n", "  Matrix filling", phase_us[1]);
  std::printf("%-25s%10.0f\n", "  Factorization", phase_us[2]);
  std::printf("%-25s%10.0f\n", "  Iteration", phase_us[3]);
  std::printf("%-25s%10.0f\n", "  Polish", phase_us[4]);
  std::printf("%-25s%10.0f\n", "(before t0: scaling)", phase_us[0]);
  std::printf("%-25s%10.0f\n", "(after: un-scale, report)", phase_us[5]);
  std::printf("=============================================================\n");
  std::fflush(stdout);
}

void verbose_table(const char *kind, int n, int m, const double *trace, int rows, const char *note)
{
  std::printf("========================= QP Solver =========================\n");
  std::printf("Solving %s QP with n=%d, m=%d\n", kind, n, m);
  if (note) std::printf("%s\n", note);
  std::printf("%8s%14s%14s%14s%10s\n", "ITER", "OBJ", "PRI_RES", "DUA_RES", "TIME");
  for (int r = 0; r < rows && trace[(size_t)r * 5] >= 0.0; ++r)
    std::printf("%7.0f:%14.6e%14.6e%14.6e%10.0f\n", trace[(size_t)r * 5], trace[(size_t)r * 5 + 1], trace[(size_t)r * 5 + 2],
                trace[(size_t)r * 5 + 3], trace[(size_t)r * 5 + 4]);
  std::fflush(stdout);
}

int verbose_table_rows(const sfb_qp_params *prm)
{
  const uint64_t sci = prm->stop_check_iter > 0 ? (uint64_t)prm->stop_check_iter : 1u, cap = 4096;
  const uint64_t mi  = prm->max_iter >= 0 ? (uint64_t)prm->max_iter : cap * sci;  // (negative: no limit)
  return (int)std::min<uint64_t>(cap, mi / sci + 2);
}

DenseKernelParams make_kernel_params(const sfb_qp_params *prm, int n, int m)
{
  DenseKernelParams kp;
  kp.n          = n;
  kp.m          = m;
  kp.alpha      = static_cast<double>(prm->alpha);  // qp_solver.hpp:354
  kp.alpha_comp = 1.0 - kp.alpha;                   // :355
  kp.rho_bar    = static_cast<double>(prm->rho);    // :353
  kp.sigma      = static_cast<double>(prm->sigma);  // :356
  kp.eps_abs    = static_cast<double>(prm->eps_abs);
  kp.eps_rel    = static_cast<double>(prm->eps_rel);
  kp.eps_pinf   = static_cast<double>(prm->eps_primal_inf);
  kp.eps_dinf   = static_cast<double>(prm->eps_dual_inf);
  kp.delta      = static_cast<double>(prm->delta);
  kp.max_iter   = prm->max_iter < 0 ? (uint32_t)SFB_QP_DEVICE_ITER_CAP : (uint32_t)prm->max_iter;
  kp.max_time_ns = prm->max_time_ns < 0 ? -1 : prm->max_time_ns;
  kp.stop_check_iter = prm->stop_check_iter;
  kp.polish_iter     = prm->polish_iter;
  kp.scaling         = prm->scaling ? 1 : 0;
  kp.polish          = prm->polish ? 1 : 0;
  kp.reuse           = prm->reuse_factor ? 1 : 0;
  return kp;
}


}  // namespace sfb

namespace {
using sfb::fail;
using sfb::hip_fail;
using sfb::require_device;
using sfb::make_kernel_params;

sfb_status check_qp_args(const sfb_qp_params *prm, int64_t batch, int n, int m, const void *P, const void *q,
                         const void *A, const void *l, const void *u, const void *wx, const void *wy, const void *x,
                         const void *y, const void *code)
{
  if (!prm) return fail(SFB_ERR_INVALID_ARG, "prm is NULL");
  if (batch < 0) return fail(SFB_ERR_INVALID_ARG, "batch < 0");
  if (n < 1 || m < 1) return fail(SFB_ERR_INVALID_ARG, "n and m must be >= 1");
  if (batch > 0 && (!P || !q || !A || !l || !u || !x || !y || !code))
    return fail(SFB_ERR_INVALID_ARG, "NULL problem / solution pointer");
  if ((wx == nullptr) != (wy == nullptr))
    return fail(SFB_ERR_INVALID_ARG, "warm_x and warm_y must both be given or both be NULL");
  if (prm->max_iter > 0xFFFFFFFFll) return fail(SFB_ERR_INVALID_ARG, "max_iter exceeds uint32");
  if (batch > 0x7FFFFFFFll) return fail(SFB_ERR_UNSUPPORTED, "batch exceeds 2^31-1 per call");
  return SFB_OK;
}

// ---- dense problems with SFB_QP_DENSE_MAX_K < n+m <= 1024: the pivoted dense LDL' with the factor in HBM ----
// (bit-identical to the dense oracle, i.e. to the reference's dense branch as far as that is pinned)
sfb_status dense_big(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P, const double *q, const double *A,
                     const double *l, const double *u, const double *wx, const double *wy, double *x, double *y, double *obj,
                     uint32_t *iter, int32_t *code, hipStream_t stream, void *workspace)
{
  if (n + m <= sfb::kDenseMidMaxK) {  // up to 128: everything on chip, no workspace (qp_dense_mid.hip)
    const sfb::DenseKernelParams kpm = make_kernel_params(prm, n, m);
    const sfb::QpBatch gm{P, q, A, l, u, wx, wy, x, y, obj, iter, code};
    const hipError_t em = sfb::qp_dense_mid_launch(kpm, batch, gm, stream, workspace);
    if (em != hipSuccess) return hip_fail(em, "qp_dense_mid_kernel launch");
    return SFB_OK;
  }
  const size_t bytes = (size_t)batch * sfb::qp_dense_big_ws_doubles(n, m) * sizeof(double);
  char *buf          = static_cast<char *>(workspace);  // the caller's (sfb_workspace), or per call below
  bool async_alloc   = true;
  hipError_t e       = hipSuccess;
  if (!workspace) {
    e = hipMallocAsync(reinterpret_cast<void **>(&buf), bytes, stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      async_alloc = false;
      e           = hipMalloc(reinterpret_cast<void **>(&buf), bytes);
      if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    }
  }
  const sfb::DenseKernelParams kp = make_kernel_params(prm, n, m);
  const sfb::QpBatch g{P, q, A, l, u, wx, wy, x, y, obj, iter, code};
  e = sfb::qp_dense_big_launch(kp, batch, g, reinterpret_cast<double *>(buf), stream);
  if (workspace) {
  } else if (async_alloc) {
    (void)hipFreeAsync(buf, stream);
  } else {
    (void)hipStreamSynchronize(stream);
    (void)hipFree(buf);
  }
  if (e != hipSuccess) return hip_fail(e, "qp_dense_big_kernel launch");
  return SFB_OK;
}

// ---- dense problems beyond that: the shared-pattern sparse kernel with a FULL pattern ----
// P (n x n, column-major, all entries stored) IS the CSC value array of a full pattern, so the stopping tests
// and the objective see the same matrix entries in the same order as the dense kernels; only A has to be
// re-laid out by rows.  What differs from the reference's dense solver is the factorisation order (fill-reducing
// elimination order without pivoting instead of pivoted dense LDL'), i.e. rounding.
__global__ void __launch_bounds__(64) dense_A_to_rows_kernel(const double *__restrict__ A, double *__restrict__ Ax,
                                                             const int n, const int m)
{
  const size_t off = (size_t)blockIdx.x * (size_t)(m * n);
  for (int e = threadIdx.x; e < m * n; e += 64) {
    const int r = e / n, c = e - r * n;
    Ax[off + e] = A[off + r + (size_t)c * m];
  }
}

sfb_status dense_full_pattern_plan(int n, int m, sfb_sparse_qp_plan **out)
{
  static std::mutex mu;
  static std::map<std::pair<int, int>, sfb_sparse_qp_plan *> plans;  // one symbolic analysis per shape, kept
  sfb_sparse_qp_plan *plan = nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = plans.find({n, m});
    if (it == plans.end()) {
      std::vector<int32_t> Pp(n + 1), Pi((size_t)n * n), Ap(m + 1), Aj((size_t)m * n);
      for (int c = 0; c <= n; ++c) Pp[c] = c * n;
      for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) Pi[(size_t)c * n + r] = r;
      for (int r = 0; r <= m; ++r) Ap[r] = r * n;
      for (int r = 0; r < m; ++r)
        for (int c = 0; c < n; ++c) Aj[(size_t)r * n + c] = c;
      const sfb_status st = sfb_sparse_qp_plan_create(n, m, Pp.data(), Pi.data(), Ap.data(), Aj.data(), 1, nullptr, &plan);
      if (st != SFB_OK) return st;
      plans[{n, m}] = plan;
    } else {
      plan = it->second;
    }
  }
  *out = plan;
  return SFB_OK;
}

// bytes of one dense_via_sparse call: A by rows, then the sparse kernel's workspace
sfb_status dense_via_sparse_bytes(sfb_sparse_qp_plan *plan, int64_t batch, int n, int m, size_t *abytes, size_t *bytes)
{
  int64_t wsb   = 0;
  sfb_status st = sfb_sparse_qp_plan_workspace_bytes(plan, batch, &wsb);
  if (st != SFB_OK) return st;
  *abytes = ((size_t)batch * (size_t)m * n * sizeof(double) + 255) / 256 * 256;
  *bytes  = *abytes + (size_t)wsb;
  return SFB_OK;
}

sfb_status dense_via_sparse(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P, const double *q,
                            const double *A, const double *l, const double *u, const double *wx, const double *wy,
                            double *x, double *y, double *obj, uint32_t *iter, int32_t *code, hipStream_t stream,
                            void *workspace)
{
  sfb_sparse_qp_plan *plan = nullptr;
  sfb_status st            = dense_full_pattern_plan(n, m, &plan);
  if (st != SFB_OK) return st;
  size_t abytes = 0, bytes = 0;
  st = dense_via_sparse_bytes(plan, batch, n, m, &abytes, &bytes);
  if (st != SFB_OK) return st;
  char *buf        = static_cast<char *>(workspace);
  bool async_alloc = true;
  hipError_t e     = hipSuccess;
  if (!workspace) {
    e = hipMallocAsync(reinterpret_cast<void **>(&buf), bytes, stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      async_alloc = false;
      e           = hipMalloc(reinterpret_cast<void **>(&buf), bytes);
      if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    }
  }
  double *Ax = reinterpret_cast<double *>(buf);
  hipLaunchKernelGGL(dense_A_to_rows_kernel, dim3((unsigned)batch), dim3(64), 0, stream, A, Ax, n, m);
  if ((e = hipGetLastError()) != hipSuccess) {
    if (workspace) {
    } else if (async_alloc) (void)hipFreeAsync(buf, stream);
    else (void)hipFree(buf);
    return hip_fail(e, "dense_A_to_rows_kernel launch");
  }
  st = sfb_sparse_qp_solve_batch(plan, prm, batch, P, q, Ax, l, u, wx, wy, x, y, obj, iter, code, buf + abytes, stream);
  if (workspace) {
  } else if (async_alloc) {
    (void)hipFreeAsync(buf, stream);
  } else {
    (void)hipStreamSynchronize(stream);
    (void)hipFree(buf);
  }
  return st;
}

// verbose, ONE dense problem with n + m <= 128 (device pointers of the call): the per-iteration table of qp_solver.hpp:490-501
// from the TRACE instance of the on-chip dense kernel (qp_dense_mid.hip) -- the same pivoted arithmetic as the solve whose
// results the caller receives (every dense route is bit-identical to the dense oracle), so the rows ARE that solve's iterates.
// The closing summary (:550-565) follows from the same launch's phase stamps.  The re-solve runs WITHOUT the caller's time
// limit and for at most the iterations the returned solve took (real_iter, nullable): a solve that ended with MaxTime has
// its table end at the same iteration instead of wherever the second run's clock would cut it.
void dense_native_table(const sfb_qp_params *prm, int n, int m, const double *P, const double *q, const double *A, const double *l,
                        const double *u, const double *wx, const double *wy, const uint32_t *real_iter, const int32_t *real_code)
{
  const int rows = sfb::verbose_table_rows(prm);
  std::vector<double> tr((size_t)rows * 5, 0.0);
  for (int r = 0; r < rows; ++r) tr[(size_t)r * 5] = -1.0;
  char *buf = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&buf), ((size_t)n + m + 2 + tr.size() + 16) * 8 + 16) != hipSuccess) { (void)hipGetLastError(); return; }
  double *tx = reinterpret_cast<double *>(buf), *ty = tx + n, *tobj = ty + m, *dtr = tobj + 1, *dph = dtr + tr.size();
  uint32_t *titer = reinterpret_cast<uint32_t *>(dph + 16);
  int32_t *tcode  = reinterpret_cast<int32_t *>(titer + 1);
  sfb_qp_params p2 = *prm;
  p2.verbose       = 0;
  p2.max_time_ns   = -1;
  if (real_iter != nullptr && (p2.max_iter < 0 || (int64_t)*real_iter < p2.max_iter)) p2.max_iter = (int64_t)*real_iter;
  const sfb::DenseKernelParams kp = make_kernel_params(&p2, n, m);
  const sfb::QpBatch g{P, q, A, l, u, wx, wy, tx, ty, tobj, titer, tcode};
  double ph[6] = {0, 0, 0, 0, 0, 0};
  uint32_t it2 = 0;
  int32_t code2 = -1;
  // (the phase scratch is zeroed: whatever stamps a kernel variant leaves out read as 0, not as what the allocation held)
  const bool ok = hipMemcpy(dtr, tr.data(), tr.size() * 8, hipMemcpyHostToDevice) == hipSuccess && hipMemset(dph, 0, 16 * 8) == hipSuccess &&
                  sfb::qp_dense_mid_trace_launch(kp, 1, g, nullptr, dtr, rows, dph) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
                  hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess &&
                  hipMemcpy(ph, dph, sizeof(ph), hipMemcpyDeviceToHost) == hipSuccess &&
                  hipMemcpy(&it2, titer, 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(&code2, tcode, 4, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(buf);
  if (ok) {
    sfb::verbose_table("dense", n, m, tr.data(), rows, nullptr);
    sfb::verbose_summary(real_code ? *real_code : code2, real_iter ? *real_iter : it2, ph);
    if (real_iter != nullptr && real_code != nullptr && (it2 != *real_iter || (code2 != *real_code && *real_code != SFB_QP_MAX_TIME)))
      std::printf("[sfb] verbose: NOTE the table's solve ended after %u iterations with code %d, the returned solve after %u iterations with code %d\n",
                  it2, (int)code2, *real_iter, (int)*real_code);
  } else (void)hipGetLastError();
}

// verbose, ONE dense problem (device pointers of the call): the per-iteration table of qp_solver.hpp:490-501.  The dense
// kernels carry no trace; the table is produced by a second, diagnostic solve of the same problem through the sparse
// kernel's TRACE instance on the full pattern -- no pivoting there, so its iterates equal the dense kernel's up to rounding
// (the header says so); the results the caller receives are the dense kernel's.
void dense_verbose_table(const sfb_qp_params *prm, int n, int m, const double *P, const double *q, const double *A, const double *l,
                         const double *u, const double *wx, const double *wy, const uint32_t *real_iter, const int32_t *real_code)
{
  sfb_sparse_qp_plan *plan = nullptr;
  if (dense_full_pattern_plan(n, m, &plan) != SFB_OK) return;
  size_t abytes = 0, bytes = 0;
  if (dense_via_sparse_bytes(plan, 1, n, m, &abytes, &bytes) != SFB_OK) return;
  const int rows     = sfb::verbose_table_rows(prm);
  const size_t extra = ((size_t)n + m + 2 + (size_t)rows * 5) * sizeof(double) + 16;
  char *buf          = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&buf), bytes + extra) != hipSuccess) { (void)hipGetLastError(); return; }
  double *Ax = reinterpret_cast<double *>(buf), *tx = reinterpret_cast<double *>(buf + bytes), *ty = tx + n, *tobj = ty + m,
         *dtrace = tobj + 1;
  uint32_t *titer = reinterpret_cast<uint32_t *>(dtrace + (size_t)rows * 5);
  int32_t *tcode  = reinterpret_cast<int32_t *>(titer + 1);
  std::vector<double> tr((size_t)rows * 5, 0.0);
  for (int r = 0; r < rows; ++r) tr[(size_t)r * 5] = -1.0;
  sfb_qp_params p2 = *prm;
  p2.verbose       = 0;
  bool ok = hipMemcpy(dtrace, tr.data(), tr.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(dense_A_to_rows_kernel, dim3(1), dim3(64), 0, nullptr, A, Ax, n, m);
    ok = hipGetLastError() == hipSuccess &&
         sfb_sparse_qp_solve_batch_trace(plan, &p2, 1, P, q, Ax, l, u, wx, wy, tx, ty, tobj, titer, tcode, buf + abytes, dtrace, rows,
                                         nullptr) == SFB_OK &&
         hipDeviceSynchronize() == hipSuccess && hipMemcpy(tr.data(), dtrace, tr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess;
  }
  uint32_t titer_h = 0;
  int32_t tcode_h  = -1;
  if (ok) ok = hipMemcpy(&titer_h, titer, 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(&tcode_h, tcode, 4, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(buf);
  if (!ok) {
    (void)hipGetLastError();
    std::printf("[sfb] verbose: the diagnostic solve behind the per-iteration table failed (%s); no table\n", sfb_last_error());
    return;
  }
  sfb::verbose_table("dense", n, m, tr.data(), rows,
                     "(table: diagnostic solve without pivoting, equal to the solve's iterates up to rounding)");
  // the solve whose results the caller receives is the pivoted dense kernel's: say so when the two ended differently
  if (real_iter != nullptr && real_code != nullptr && (titer_h != *real_iter || tcode_h != *real_code))
    std::printf("[sfb] verbose: NOTE the table's diagnostic solve ended after %u iterations with code %d, the returned (pivoted) solve after "
                "%u iterations with code %d\n", titer_h, (int)tcode_h, *real_iter, (int)*real_code);
}

}  // namespace

namespace sfb {
namespace {
std::mutex g_devices_mu;
std::vector<int> g_devices;  // empty: all visible devices
}  // namespace

std::vector<int> device_list()
{
  {
    std::lock_guard<std::mutex> lk(g_devices_mu);
    if (!g_devices.empty()) return g_devices;
  }
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) {
    (void)hipGetLastError();
    cnt = 0;
  }
  std::vector<int> all((size_t)std::max(cnt, 0));
  for (int d = 0; d < cnt; ++d) all[(size_t)d] = d;
  return all;
}

sfb_status run_sharded(int64_t batch, const std::function<sfb_status(int, int64_t, int64_t)> &fn)
{
  const std::vector<int> devs = device_list();
  if (devs.empty()) return fail(SFB_ERR_NO_DEVICE, "no HIP device (the sfb library has no CPU fallback)");
  const int64_t G = (int64_t)devs.size();
  std::vector<sfb_status> st((size_t)G, SFB_OK);
  std::vector<std::string> msg((size_t)G);
  std::vector<std::thread> th;
  bool spawn_failed = false;
  for (int64_t g = 0; g < G && !spawn_failed; ++g) {
    const int64_t b0 = batch * g / G, b1 = batch * (g + 1) / G;  // contiguous shards
    if (b1 == b0) continue;
    try {
      th.emplace_back([&, g, b0, b1] {
        try {
          hipError_t e = hipSetDevice(devs[(size_t)g]);
          if (e != hipSuccess) st[(size_t)g] = hip_fail(e, "hipSetDevice");
          else st[(size_t)g] = fn(devs[(size_t)g], b0, b1 - b0);
        } catch (const std::exception &ex) {  // (nothing may leave a thread of an extern "C" entry point)
          st[(size_t)g] = fail(SFB_ERR_HIP, std::string("exception in a shard: ") + ex.what());
        } catch (...) {
          st[(size_t)g] = fail(SFB_ERR_HIP, "exception in a shard");
        }
        if (st[(size_t)g] != SFB_OK) msg[(size_t)g] = sfb_last_error();  // (thread-local: carried over by hand)
      });
    } catch (const std::exception &) {  // std::system_error of the thread constructor: finish what was started, then report
      spawn_failed = true;
    }
  }
  for (auto &t : th) t.join();
  if (spawn_failed) return fail(SFB_ERR_HIP, "could not start a host thread for a device shard");
  for (int64_t g = 0; g < G; ++g)
    if (st[(size_t)g] != SFB_OK) return fail(st[(size_t)g], "device " + std::to_string(devs[(size_t)g]) + ": " + msg[(size_t)g]);
  return SFB_OK;
}
}  // namespace sfb

extern "C" {

const char *sfb_version(void) { return "smooth_feedback_amd 0.1.0 (gfx950)"; }

sfb_status sfb_set_devices(const int *devices, int count)
{
  if (count < 0 || (count > 0 && !devices)) return fail(SFB_ERR_INVALID_ARG, "bad device list");
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess) {
    (void)hipGetLastError();
    visible = 0;
  }
  for (int i = 0; i < count; ++i)
    if (devices[i] < 0 || devices[i] >= visible) return fail(SFB_ERR_INVALID_ARG, "device ordinal out of range");
  std::lock_guard<std::mutex> lk(sfb::g_devices_mu);
  sfb::g_devices.assign(devices, devices + count);
  return SFB_OK;
}

sfb_status sfb_get_devices(int *devices, int capacity, int *count)
{
  if (!count) return fail(SFB_ERR_INVALID_ARG, "count is NULL");
  const std::vector<int> d = sfb::device_list();
  *count = (int)d.size();
  for (int i = 0; devices && i < capacity && i < (int)d.size(); ++i) devices[i] = d[(size_t)i];
  return SFB_OK;
}

const char *sfb_last_error(void) { return sfb::g_last_error.c_str(); }

sfb_status sfb_debug_set(const char *name, const char *value)
{
  if (sfb::knob_set(name, value) != 0) return sfb::fail(SFB_ERR_INVALID_ARG, std::string("sfb_debug_set: unknown knob ") + (name ? name : "(null)"));
  return SFB_OK;
}

sfb_status sfb_device_count(int *count)
{
  if (!count) return fail(SFB_ERR_INVALID_ARG, "count is NULL");
  int cnt      = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    cnt = 0;
  }
  *count = cnt;
  return SFB_OK;
}

void sfb_qp_params_default(sfb_qp_params *p)
{
  if (!p) return;
  p->alpha           = 1.6f;
  p->rho             = 0.1f;
  p->sigma           = 1e-6f;
  p->scaling         = 1;
  p->eps_abs         = 1e-3f;
  p->eps_rel         = 1e-3f;
  p->eps_primal_inf  = 1e-4f;
  p->eps_dual_inf    = 1e-4f;
  p->max_iter        = -1;
  p->max_time_ns     = -1;
  p->stop_check_iter = 25;
  p->polish          = 1;
  p->polish_iter     = 5;
  p->delta           = 1e-6f;
  p->verbose         = 0;
  p->reuse_factor    = 0;
}

struct sfb_workspace {
  void *mem    = nullptr;
  size_t bytes = 0;
  int device   = -1;
};

// device workspace one dense call needs beyond its arguments (0: the LDS-resident one-QP-per-wave kernels)
static sfb_status dense_workspace_need(const sfb_qp_params *prm, int64_t batch, int n, int m, size_t *need)
{
  const int k = n + m;
  *need       = 0;
  if (k > SFB_QP_DENSE_MAX_K) {
    const char *const bv = sfb::knob("SFB_QP_DENSE_BIG");  // 0 (tests): sizes beyond 128 through the un-pivoted sparse kernel
    const bool big_off   = bv && bv[0] == '0';
    if (k <= sfb::kDenseMidMaxK || (k <= sfb::kDenseBigMaxK && !big_off)) {
      *need = k <= sfb::kDenseMidMaxK ? sfb::qp_dense_mid_ws_bytes(make_kernel_params(prm, n, m), batch)
                                      : (size_t)batch * sfb::qp_dense_big_ws_doubles(n, m) * sizeof(double);
      return SFB_OK;
    }
    sfb_sparse_qp_plan *plan = nullptr;
    sfb_status st            = dense_full_pattern_plan(n, m, &plan);
    if (st != SFB_OK) return st;
    size_t abytes = 0;
    return dense_via_sparse_bytes(plan, batch, n, m, &abytes, need);
  }
  // (a time limit sends k <= 32 to the 32 < k <= 128 kernel, which implements it: the larger of the two needs covers both)
  const size_t need_mid = sfb::qp_dense_mid_ws_bytes(make_kernel_params(prm, n, m), batch);
  *need = k <= 32 ? std::max(sfb::qp_dense4_ws_bytes(n, m, batch), prm->max_time_ns >= 0 ? need_mid : (size_t)0) : need_mid;
  return SFB_OK;
}

static sfb_status dense_solve_impl(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P, const double *q,
                                   const double *A, const double *l, const double *u, const double *warm_x,
                                   const double *warm_y, double *x, double *y, double *obj, uint32_t *iter, int32_t *code,
                                   sfb_workspace *ws, bool ws_given, void *stream)
{
  sfb_status st = check_qp_args(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, code);
  if (st != SFB_OK) return st;
  if (ws_given && !ws) return fail(SFB_ERR_INVALID_ARG, "workspace is NULL");
  st = require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  void *wsmem = nullptr;
  if (ws) {
    size_t need = 0;
    st          = dense_workspace_need(prm, batch, n, m, &need);
    if (st != SFB_OK) return st;
    if (ws->bytes < need) return fail(SFB_ERR_INVALID_ARG, "workspace too small (sfb_qp_dense_workspace_bytes)");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != ws->device)
      return fail(SFB_ERR_INVALID_ARG, "workspace belongs to another device than the current one");
    wsmem = need ? ws->mem : nullptr;
  }
  if (n + m > SFB_QP_DENSE_MAX_K) {
    // SFB_QP_DENSE_BIG=0 (A/B, tests): route these sizes to the un-pivoted sparse kernel as well
    const char *const bv = sfb::knob("SFB_QP_DENSE_BIG");
    const bool big_off   = bv && bv[0] == '0';
    if (n + m <= sfb::kDenseMidMaxK || (n + m <= sfb::kDenseBigMaxK && !big_off))
      return dense_big(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, obj, iter, code, static_cast<hipStream_t>(stream),
                       wsmem);
    return dense_via_sparse(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, obj, iter, code,
                            static_cast<hipStream_t>(stream), wsmem);
  }
  const sfb::DenseKernelParams kp = make_kernel_params(prm, n, m);
  hipError_t e = sfb::qp_dense_launch(kp, batch, P, q, A, l, u, warm_x, warm_y, x, y, obj, iter, code,
                                      static_cast<hipStream_t>(stream), wsmem);
  if (e != hipSuccess) return hip_fail(e, "qp_dense_kernel launch");
  return SFB_OK;
}

sfb_status sfb_qp_dense_solve_batch(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                    const double *q, const double *A, const double *l, const double *u,
                                    const double *warm_x, const double *warm_y, double *x, double *y, double *obj,
                                    uint32_t *iter, int32_t *code, void *stream)
{
  return dense_solve_impl(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, obj, iter, code, nullptr, false, stream);
}

sfb_status sfb_qp_dense_solve_batch_trace(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P, const double *q,
                                          const double *A, const double *l, const double *u, const double *warm_x, const double *warm_y,
                                          double *x, double *y, double *obj, uint32_t *iter, int32_t *code, double *trace,
                                          int32_t trace_rows, void *stream)
{
  return sfb_qp_dense_solve_batch_phases(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, obj, iter, code, trace, trace_rows, nullptr, stream);
}

sfb_status sfb_qp_dense_solve_batch_phases(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P, const double *q,
                                           const double *A, const double *l, const double *u, const double *warm_x, const double *warm_y,
                                           double *x, double *y, double *obj, uint32_t *iter, int32_t *code, double *trace,
                                           int32_t trace_rows, double *phase_us, void *stream)
{
  sfb_status st = check_qp_args(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, code);
  if (st != SFB_OK) return st;
  if (trace_rows < 0 || (trace_rows > 0 && !trace)) return fail(SFB_ERR_INVALID_ARG, "trace is NULL or trace_rows < 0");
  if (n + m > sfb::kDenseMidMaxK) return fail(SFB_ERR_UNSUPPORTED, "the per-iteration table of a dense problem needs n + m <= 128");
  st = require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  const sfb::DenseKernelParams kp = make_kernel_params(prm, n, m);
  const sfb::QpBatch g{P, q, A, l, u, warm_x, warm_y, x, y, obj, iter, code};
  const hipError_t e = sfb::qp_dense_mid_trace_launch(kp, batch, g, static_cast<hipStream_t>(stream), trace_rows > 0 ? trace : nullptr, trace_rows, phase_us);
  if (e != hipSuccess) return hip_fail(e, "qp_dense_mid_trace_kernel launch");
  return SFB_OK;
}

sfb_status sfb_qp_dense_solve_batch_ws(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                       const double *q, const double *A, const double *l, const double *u,
                                       const double *warm_x, const double *warm_y, double *x, double *y, double *obj,
                                       uint32_t *iter, int32_t *code, sfb_workspace *workspace, void *stream)
{
  return dense_solve_impl(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, obj, iter, code, workspace, true, stream);
}

sfb_status sfb_qp_dense_workspace_bytes(const sfb_qp_params *prm, int64_t batch, int n, int m, int64_t *bytes)
{
  if (!prm || !bytes) return fail(SFB_ERR_INVALID_ARG, "NULL argument");
  if (batch < 0 || n < 1 || m < 1) return fail(SFB_ERR_INVALID_ARG, "batch < 0 or n, m < 1");
  size_t need   = 0;
  sfb_status st = dense_workspace_need(prm, batch, n, m, &need);
  if (st != SFB_OK) return st;
  *bytes = (int64_t)need;
  return SFB_OK;
}

sfb_status sfb_workspace_create(int64_t bytes, sfb_workspace **out)
{
  if (!out || bytes < 0) return fail(SFB_ERR_INVALID_ARG, "out is NULL or bytes < 0");
  *out          = nullptr;
  sfb_status st = require_device();
  if (st != SFB_OK) return st;
  auto *w       = new sfb_workspace();
  hipError_t e  = hipGetDevice(&w->device);
  if (e == hipSuccess && bytes > 0) e = hipMalloc(&w->mem, (size_t)bytes);
  if (e != hipSuccess) {
    delete w;
    return hip_fail(e, "hipMalloc(workspace)");
  }
  w->bytes = (size_t)bytes;
  *out     = w;
  return SFB_OK;
}

void sfb_workspace_destroy(sfb_workspace *ws)
{
  if (!ws) return;
  if (ws->mem) (void)hipFree(ws->mem);
  delete ws;
}

sfb_status sfb_workspace_info(const sfb_workspace *ws, void **device_ptr, int64_t *bytes)
{
  if (!ws) return fail(SFB_ERR_INVALID_ARG, "workspace is NULL");
  if (device_ptr) *device_ptr = ws->mem;
  if (bytes) *bytes = (int64_t)ws->bytes;
  return SFB_OK;
}

}  // extern "C"

namespace {
struct HostStage {
  char *mem;
  size_t bytes;
};
std::mutex g_stage_mu;
std::map<int, HostStage> g_stage;  // per device: the staging buffer kept between host-pointer calls
HostStage host_stage_take(int dev)
{
  std::lock_guard<std::mutex> lk(g_stage_mu);
  HostStage s = {nullptr, 0};
  auto it = g_stage.find(dev);
  if (it != g_stage.end()) { s = it->second; g_stage.erase(it); }
  return s;
}
void host_stage_give(int dev, HostStage s)
{
  if (!s.mem) return;
  HostStage drop = {nullptr, 0};
  {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    auto it = g_stage.find(dev);
    if (it == g_stage.end()) g_stage[dev] = s;
    else if (it->second.bytes < s.bytes) { drop = it->second; it->second = s; }  // keep the larger one
    else drop = s;
  }
  if (drop.mem) (void)hipFree(drop.mem);
}
}  // namespace

extern "C" {

sfb_status sfb_qp_dense_solve_batch_host_multi(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                               const double *q, const double *A, const double *l, const double *u,
                                               const double *warm_x, const double *warm_y, double *x, double *y,
                                               double *obj, uint32_t *iter, int32_t *code)
{
  sfb_status st = check_qp_args(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, code);
  if (st != SFB_OK) return st;
  if (batch == 0) return require_device();
  const size_t N = (size_t)n, M = (size_t)m;
  return sfb::run_sharded(batch, [&](int, int64_t b0, int64_t cnt) {
    const size_t o = (size_t)b0;
    return sfb_qp_dense_solve_batch_host(prm, cnt, n, m, P + o * N * N, q + o * N, A + o * M * N, l + o * M, u + o * M,
                                         warm_x ? warm_x + o * N : nullptr, warm_y ? warm_y + o * M : nullptr, x + o * N,
                                         y + o * M, obj ? obj + o : nullptr, iter ? iter + o : nullptr, code + o);
  });
}

void sfb_host_staging_trim(void)
{
  std::map<int, HostStage> old;
  {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    old.swap(g_stage);
  }
  for (auto &kv : old)
    if (kv.second.mem) (void)hipFree(kv.second.mem);
}

sfb_status sfb_qp_dense_solve_batch_host(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P,
                                         const double *q, const double *A, const double *l, const double *u,
                                         const double *warm_x, const double *warm_y, double *x, double *y,
                                         double *obj, uint32_t *iter, int32_t *code)
{
  sfb_status st = check_qp_args(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, code);
  if (st != SFB_OK) return st;
  st = require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;

  const size_t B = (size_t)batch, N = (size_t)n, M = (size_t)m;
  const size_t in_d  = B * (N * N + N + M * N + 2 * M) + (warm_x ? B * (N + M) : 0);
  const size_t out_d = B * (N + M + 1);
  // Staging memory of the host-pointer entry point: ONE buffer per device is kept between calls (ASIFilter solves one
  // small QP per tick through here, and a hipMalloc / hipFree pair per call would dominate its latency).  The lock is
  // held only while the buffer is taken or handed back: concurrent callers do not serialise -- one of them gets the
  // kept buffer, the others allocate their own for the call.  sfb_host_staging_trim() frees what is kept.
  const size_t bytes = (in_d + out_d) * sizeof(double) + B * (sizeof(uint32_t) + sizeof(int32_t));
  int devid          = 0;
  hipError_t e       = hipGetDevice(&devid);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  HostStage mine = host_stage_take(devid);
  if (mine.bytes < bytes) {
    if (mine.mem) (void)hipFree(mine.mem);
    mine = {nullptr, 0};
    e    = hipMalloc(reinterpret_cast<void **>(&mine.mem), bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    mine.bytes = bytes;
  }
  struct Return {  // hand the buffer back on every exit path
    int dev;
    HostStage &s;
    ~Return() { host_stage_give(dev, s); }
  } ret{devid, mine};
  char *dev = mine.mem;

  double *dP = reinterpret_cast<double *>(dev);
  double *dq = dP + B * N * N;
  double *dA = dq + B * N;
  double *dl = dA + B * M * N;
  double *du = dl + B * M;
  double *dwx = du + B * M, *dwy = nullptr;
  double *dx = dwx;
  if (warm_x) {
    dwy = dwx + B * N;
    dx  = dwy + B * M;
  } else {
    dwx = nullptr;
  }
  double *dy     = dx + B * N;
  double *dobj   = dy + B * M;
  uint32_t *dit  = reinterpret_cast<uint32_t *>(dobj + B);
  int32_t *dcode = reinterpret_cast<int32_t *>(dit + B);

  auto H2D = [&](void *d, const void *h, size_t nb) { return hipMemcpy(d, h, nb, hipMemcpyHostToDevice); };
  auto D2H = [&](void *h, const void *d, size_t nb) { return hipMemcpy(h, d, nb, hipMemcpyDeviceToHost); };
  st = SFB_OK;
  using clk = std::chrono::steady_clock;
  auto ms   = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto tv0 = clk::now();
  auto tv1 = tv0, tv2 = tv0;
  do {
    if ((e = H2D(dP, P, B * N * N * 8)) != hipSuccess) break;
    if ((e = H2D(dq, q, B * N * 8)) != hipSuccess) break;
    if ((e = H2D(dA, A, B * M * N * 8)) != hipSuccess) break;
    if ((e = H2D(dl, l, B * M * 8)) != hipSuccess) break;
    if ((e = H2D(du, u, B * M * 8)) != hipSuccess) break;
    if (warm_x) {
      if ((e = H2D(dwx, warm_x, B * N * 8)) != hipSuccess) break;
      if ((e = H2D(dwy, warm_y, B * M * 8)) != hipSuccess) break;
    }
    tv1 = clk::now();
    st = sfb_qp_dense_solve_batch(prm, batch, n, m, dP, dq, dA, dl, du, dwx, dwy, dx, dy, dobj, dit, dcode, nullptr);
    if (st != SFB_OK) break;
    if ((e = hipDeviceSynchronize()) != hipSuccess) break;
    tv2 = clk::now();
    if ((e = D2H(x, dx, B * N * 8)) != hipSuccess) break;
    if ((e = D2H(y, dy, B * M * 8)) != hipSuccess) break;
    if (obj && (e = D2H(obj, dobj, B * 8)) != hipSuccess) break;
    if (iter && (e = D2H(iter, dit, B * 4)) != hipSuccess) break;
    if ((e = D2H(code, dcode, B * 4)) != hipSuccess) break;
  } while (false);
  if (e != hipSuccess) st = hip_fail(e, "sfb_qp_dense_solve_batch_host");
  const auto tv3 = clk::now();
  if (st == SFB_OK && prm->verbose && batch == 1) {  // (inputs still on the device)
    uint32_t it1 = 0;
    const bool have_it = iter ? (it1 = iter[0], true) : hipMemcpy(&it1, dit, 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (n + m <= sfb::kDenseMidMaxK) dense_native_table(prm, n, m, dP, dq, dA, dl, du, dwx, dwy, have_it ? &it1 : nullptr, code);  // the solve's own iterates
    else dense_verbose_table(prm, n, m, dP, dq, dA, dl, du, dwx, dwy, have_it ? &it1 : nullptr, code);
  }
  if (st == SFB_OK && prm->verbose) {
    std::vector<uint32_t> itv;
    if (!iter) {
      itv.resize(B);
      if (hipMemcpy(itv.data(), dit, B * 4, hipMemcpyDeviceToHost) != hipSuccess) itv.clear();
    }
    sfb::verbose_report("dense QP batch", batch, n, m, ms(tv0, tv1), ms(tv1, tv2), ms(tv2, tv3), code,
                        iter ? iter : (itv.empty() ? nullptr : itv.data()));
  }
  return st;
}

sfb_status sfb_qp_dense_solve_batch_host_trace(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P, const double *q,
                                               const double *A, const double *l, const double *u, const double *warm_x, const double *warm_y,
                                               double *x, double *y, double *obj, uint32_t *iter, int32_t *code, double *trace,
                                               int32_t trace_rows)
{
  return sfb_qp_dense_solve_batch_host_phases(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, obj, iter, code, trace, trace_rows, nullptr);
}

sfb_status sfb_qp_dense_solve_batch_host_phases(const sfb_qp_params *prm, int64_t batch, int n, int m, const double *P, const double *q,
                                                const double *A, const double *l, const double *u, const double *warm_x, const double *warm_y,
                                                double *x, double *y, double *obj, uint32_t *iter, int32_t *code, double *trace,
                                                int32_t trace_rows, double *phase_us)
{
  sfb_status st = check_qp_args(prm, batch, n, m, P, q, A, l, u, warm_x, warm_y, x, y, code);
  if (st != SFB_OK) return st;
  if (trace_rows < 0 || (trace_rows > 0 && !trace)) return fail(SFB_ERR_INVALID_ARG, "trace is NULL or trace_rows < 0");
  if (n + m > sfb::kDenseMidMaxK) return fail(SFB_ERR_UNSUPPORTED, "the per-iteration table of a dense problem needs n + m <= 128");
  st = require_device();
  if (st != SFB_OK) return st;
  if (batch == 0) return SFB_OK;
  const size_t B = (size_t)batch, N = (size_t)n, M = (size_t)m, TR = B * (size_t)trace_rows * 5, PH = phase_us ? B * 16 : 0;
  const size_t doubles = B * (N * N + N + M * N + 2 * M) + (warm_x ? B * (N + M) : 0) + B * (N + M + 1) + TR + PH;
  char *mem    = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&mem), doubles * 8 + B * 8);
  if (e != hipSuccess) return hip_fail(e, "hipMalloc");
  double *dP = reinterpret_cast<double *>(mem), *dq = dP + B * N * N, *dA = dq + B * N, *dl = dA + B * M * N, *du = dl + B * M, *nx = du + B * M;
  double *dwx = nullptr, *dwy = nullptr;
  if (warm_x) { dwx = nx; dwy = dwx + B * N; nx = dwy + B * M; }
  double *dx = nx, *dy = dx + B * N, *dobj = dy + B * M, *dtr = dobj + B;
  double *dph    = dtr + TR;
  uint32_t *dit  = reinterpret_cast<uint32_t *>(dph + PH);
  int32_t *dcode = reinterpret_cast<int32_t *>(dit + B);
  auto H2D = [&](void *d, const void *h, size_t nb) { return hipMemcpy(d, h, nb, hipMemcpyHostToDevice); };
  auto D2H = [&](void *h, const void *d, size_t nb) { return hipMemcpy(h, d, nb, hipMemcpyDeviceToHost); };
  do {
    if ((e = H2D(dP, P, B * N * N * 8)) != hipSuccess) break;
    if ((e = H2D(dq, q, B * N * 8)) != hipSuccess) break;
    if ((e = H2D(dA, A, B * M * N * 8)) != hipSuccess) break;
    if ((e = H2D(dl, l, B * M * 8)) != hipSuccess) break;
    if ((e = H2D(du, u, B * M * 8)) != hipSuccess) break;
    if (warm_x && ((e = H2D(dwx, warm_x, B * N * 8)) != hipSuccess || (e = H2D(dwy, warm_y, B * M * 8)) != hipSuccess)) break;
    if (TR) {  // unused rows keep ITER = -1
      std::vector<double> init(TR, 0.0);
      for (size_t r = 0; r < TR; r += 5) init[r] = -1.0;
      if ((e = H2D(dtr, init.data(), TR * 8)) != hipSuccess) break;
    }
    if (PH && (e = hipMemset(dph, 0, PH * 8)) != hipSuccess) break;
    st = sfb_qp_dense_solve_batch_phases(prm, batch, n, m, dP, dq, dA, dl, du, dwx, dwy, dx, dy, dobj, dit, dcode, TR ? dtr : nullptr, trace_rows,
                                         PH ? dph : nullptr, nullptr);
    if (st != SFB_OK) break;
    if ((e = hipDeviceSynchronize()) != hipSuccess) break;
    if ((e = D2H(x, dx, B * N * 8)) != hipSuccess) break;
    if ((e = D2H(y, dy, B * M * 8)) != hipSuccess) break;
    if (obj && (e = D2H(obj, dobj, B * 8)) != hipSuccess) break;
    if (iter && (e = D2H(iter, dit, B * 4)) != hipSuccess) break;
    if ((e = D2H(code, dcode, B * 4)) != hipSuccess) break;
    if (TR && (e = D2H(trace, dtr, TR * 8)) != hipSuccess) break;
    if (PH && (e = D2H(phase_us, dph, B * 6 * 8)) != hipSuccess) break;
  } while (false);
  (void)hipFree(mem);
  if (e != hipSuccess) return hip_fail(e, "sfb_qp_dense_solve_batch_host_trace");
  return st;
}

}  // extern "C"
