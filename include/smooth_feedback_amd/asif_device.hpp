// ASI safety filter for a swarm with the ASSEMBLY on the GPU as well (HIP only: include from a translation unit compiled
// by hipcc).  The host version (asif.hpp, ASIFSwarm) integrates every agent's backup trajectory and its sensitivity on
// the CPU -- asif_func.hpp:145-179, 250 Euler steps of a 6 x 6 matrix ODE per vehicle of examples/mpc_asif_vehicle.cpp
// -- which takes 40 times longer than the batched QP solve.  Here one GPU thread does that for one agent, running the
// very same function (asif_fill, asif.hpp), writes the agent's QP into the batch arrays of sfb_qp_dense_solve_batch, and
// the solve follows on the same stream; states go up (sizeof(G) per agent), filtered inputs come down.
//
// What it asks of the caller: the dynamics f(x, u), the barrier h(agent, t, x) and the backup controller bu(agent, t, x)
// must be functors callable in device code (`__host__ __device__` members; lie.hpp's groups already are), optionally
// with analytic `jacobian` members as in asif.hpp.  Results: the QPs equal the host assembly's up to the last bits of
// sin / cos / atan2 (two maths libraries); the solve of the assembled QPs is the same kernel, bit-identical to the oracle.
#pragma once
#ifndef __HIPCC__
#error "asif_device.hpp needs hipcc"
#endif
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "asif.hpp"

namespace smooth_feedback_amd {

namespace detail {
template<class G, class U, class Dyn, class H, class BU>
__global__ void __launch_bounds__(64) asif_assemble_kernel(const int64_t B, ASIFProblemView<G, U> proto, const G * __restrict__ g,
                                                           const U * __restrict__ u_des, const Dyn f, const H h, const BU bu,
                                                           const int N, const int M, double * __restrict__ P,
                                                           double * __restrict__ q, double * __restrict__ A,
                                                           double * __restrict__ l, double * __restrict__ u)
{
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  proto.x0    = g[b];
  proto.u_des = u_des[b];
  double *Pb = P + b * N * N, *qb = q + b * N, *Ab = A + b * (int64_t)M * N, *lb = l + b * M, *ub = u + b * M;
  for (int e = 0; e < N * N; ++e) Pb[e] = 0.0;  // asif_to_qp_allocate (asif_func.hpp:78-99)
  for (int e = 0; e < N; ++e) qb[e] = 0.0;
  for (int e = 0; e < M * N; ++e) Ab[e] = 0.0;
  for (int e = 0; e < M; ++e) { lb[e] = 0.0; ub[e] = 0.0; }
  asif_fill<G, U>(Pb, qb, Ab, lb, ub, proto, f, AgentFn<H, G>{h, (std::size_t)b}, AgentFn<BU, G>{bu, (std::size_t)b});
}

// asif.hpp:99: only an Optimal solution becomes the next warm start
__global__ void __launch_bounds__(256) asif_store_kernel(const int64_t B, const int n, const int m, const double * __restrict__ x,
                                                         const double * __restrict__ y, const int32_t * __restrict__ code,
                                                         double * __restrict__ wx, double * __restrict__ wy)
{
  const int64_t b = blockIdx.x;
  if (b >= B || code[b] != 0) return;
  for (int e = threadIdx.x; e < n; e += 256) wx[b * n + e] = x[b * n + e];
  for (int e = threadIdx.x; e < m; e += 256) wy[b * m + e] = y[b * m + e];
}

inline void hip_check(hipError_t e, const char * what)
{
  if (e != hipSuccess) throw std::runtime_error(std::string("asif_device: ") + what + ": " + hipGetErrorString(e));
}
}  // namespace detail

/// ASIFSwarm (asif.hpp) with device-side assembly.  h(agent, t, x) -> Vec<nh>, bu(agent, t, x) -> U.
template<class G, class U, class Dyn, class H, class BU>
class ASIFSwarmDevice {
public:
  ASIFSwarmDevice(Dyn f, H h, BU bu, std::size_t agents, ASIFilterParams<U> prm = {})
      : f_(f), h_(h), bu_(bu), B_((int64_t)agents), prm_(std::move(prm))
  {
    n_ = U::Dof + 1;
    m_ = int(prm_.asif.K * prm_.nh + (std::size_t)prm_.ulim.rows + 1);
    const size_t B = (size_t)B_, qpd = (size_t)n_ * n_ + n_ + (size_t)m_ * n_ + 2 * (size_t)m_, sol = 2 * ((size_t)n_ + m_);
    const size_t ul = (size_t)prm_.ulim.rows * (U::Dof + 2);
    detail::hip_check(hipMalloc(reinterpret_cast<void **>(&mem_), (B * (qpd + sol) + ul) * sizeof(double) + B * (sizeof(G) + sizeof(U) + 8)),
                      "hipMalloc");
    double * d = mem_;
    P_ = d; d += B * n_ * n_;  q_ = d; d += B * n_;  A_ = d; d += B * m_ * n_;  l_ = d; d += B * m_;  u_ = d; d += B * m_;
    x_ = d; d += B * n_;  y_ = d; d += B * m_;  wx_ = d; d += B * n_;  wy_ = d; d += B * m_;
    ulA_ = d; d += (size_t)prm_.ulim.rows * U::Dof;  ull_ = d; d += prm_.ulim.rows;  ulu_ = d; d += prm_.ulim.rows;
    g_     = reinterpret_cast<G *>(d);
    udes_  = reinterpret_cast<U *>(g_ + B);
    iter_  = reinterpret_cast<uint32_t *>(udes_ + B);
    code_  = reinterpret_cast<int32_t *>(iter_ + B);
    if (prm_.ulim.rows > 0) {
      detail::hip_check(hipMemcpy(ulA_, prm_.ulim.A.data(), prm_.ulim.A.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
      detail::hip_check(hipMemcpy(ull_, prm_.ulim.l.data(), prm_.ulim.l.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
      detail::hip_check(hipMemcpy(ulu_, prm_.ulim.u.data(), prm_.ulim.u.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
    }
    const sfb_qp_params c = prm_.qp.to_c();
    int64_t wsb = 0;
    sfb_check(sfb_qp_dense_workspace_bytes(&c, B_, n_, m_, &wsb));
    sfb_check(sfb_workspace_create(wsb, &ws_));
    hx_.resize(B * n_);
    hcode_.assign(B, 6);
    hiter_.assign(B, 0);
  }
  ASIFSwarmDevice(const ASIFSwarmDevice &)             = delete;
  ASIFSwarmDevice & operator=(const ASIFSwarmDevice &) = delete;
  ~ASIFSwarmDevice()
  {
    sfb_workspace_destroy(ws_);
    if (mem_) (void)hipFree(mem_);
  }

  /// one tick: the filtered inputs of all agents
  std::vector<U> operator()(const std::vector<G> & g, const std::vector<U> & u_des)
  {
    if ((int64_t)g.size() != B_ || (int64_t)u_des.size() != B_) throw std::invalid_argument("ASIFSwarmDevice: one state and input per agent");
    detail::hip_check(hipMemcpy(g_, g.data(), (size_t)B_ * sizeof(G), hipMemcpyHostToDevice), "hipMemcpy(states)");
    detail::hip_check(hipMemcpy(udes_, u_des.data(), (size_t)B_ * sizeof(U), hipMemcpyHostToDevice), "hipMemcpy(inputs)");
    ASIFProblemView<G, U> proto{prm_.T, G::Identity(), U::Identity(), prm_.u_weight, prm_.ulim.rows, ulA_, ull_, ulu_, prm_.ulim.c,
                                int(prm_.asif.K), prm_.asif.alpha, prm_.asif.dt, prm_.asif.relax_cost};
    hipLaunchKernelGGL((detail::asif_assemble_kernel<G, U, Dyn, H, BU>), dim3((unsigned)((B_ + 63) / 64)), dim3(64), 0, nullptr, B_, proto,
                       g_, udes_, f_, h_, bu_, n_, m_, P_, q_, A_, l_, u_);
    detail::hip_check(hipGetLastError(), "asif_assemble_kernel");
    const sfb_qp_params c = prm_.qp.to_c();
    sfb_check(sfb_qp_dense_solve_batch_ws(&c, B_, n_, m_, P_, q_, A_, l_, u_, have_warm_ ? wx_ : nullptr, have_warm_ ? wy_ : nullptr,
                                          x_, y_, nullptr, iter_, code_, ws_, nullptr));
    if (!have_warm_) detail::hip_check(hipMemsetAsync(wx_, 0, (size_t)B_ * (n_ + m_) * 8, nullptr), "hipMemsetAsync");  // wx, wy adjacent
    hipLaunchKernelGGL(detail::asif_store_kernel, dim3((unsigned)B_), dim3(256), 0, nullptr, B_, n_, m_, x_, y_, code_, wx_, wy_);
    detail::hip_check(hipGetLastError(), "asif_store_kernel");
    detail::hip_check(hipMemcpy(hx_.data(), x_, (size_t)B_ * n_ * 8, hipMemcpyDeviceToHost), "hipMemcpy(x)");
    detail::hip_check(hipMemcpy(hcode_.data(), code_, (size_t)B_ * 4, hipMemcpyDeviceToHost), "hipMemcpy(code)");
    detail::hip_check(hipMemcpy(hiter_.data(), iter_, (size_t)B_ * 4, hipMemcpyDeviceToHost), "hipMemcpy(iter)");
    have_warm_ = true;  // agents that never were Optimal keep the zero start (== no warm start)
    std::vector<U> out((size_t)B_);
    for (int64_t b = 0; b < B_; ++b) {
      typename U::Tangent du{};
      for (int i = 0; i < U::Dof; ++i) du[i] = hx_[(size_t)b * n_ + i];
      out[(size_t)b] = rplus(u_des[(size_t)b], du);  // asif.hpp:101
    }
    return out;
  }

  const std::vector<int32_t> & codes() const { return hcode_; }
  const std::vector<uint32_t> & iterations() const { return hiter_; }
  int n() const { return n_; }
  int m() const { return m_; }
  /// the QPs of the last call (batch-major, column-major matrices) and their primal / dual solutions
  void copy_problem(double * P, double * q, double * A, double * l, double * u, double * x, double * y) const
  {
    const size_t B = (size_t)B_;
    auto dl = [&](double * dst, const double * src, size_t cnt) { detail::hip_check(hipMemcpy(dst, src, cnt * 8, hipMemcpyDeviceToHost), "hipMemcpy"); };
    dl(P, P_, B * n_ * n_); dl(q, q_, B * n_); dl(A, A_, B * m_ * n_); dl(l, l_, B * m_); dl(u, u_, B * m_); dl(x, x_, B * n_); dl(y, y_, B * m_);
  }
  /// the warm start the NEXT call will use (zeros before the first call)
  void copy_warm_start(double * wx, double * wy) const
  {
    if (!have_warm_) {
      std::fill(wx, wx + (size_t)B_ * n_, 0.0);
      std::fill(wy, wy + (size_t)B_ * m_, 0.0);
      return;
    }
    detail::hip_check(hipMemcpy(wx, wx_, (size_t)B_ * n_ * 8, hipMemcpyDeviceToHost), "hipMemcpy");
    detail::hip_check(hipMemcpy(wy, wy_, (size_t)B_ * m_ * 8, hipMemcpyDeviceToHost), "hipMemcpy");
  }

private:
  Dyn f_;
  H h_;
  BU bu_;
  int64_t B_;
  ASIFilterParams<U> prm_;
  int n_ = 0, m_ = 0;
  double *mem_ = nullptr, *P_ = nullptr, *q_ = nullptr, *A_ = nullptr, *l_ = nullptr, *u_ = nullptr, *x_ = nullptr, *y_ = nullptr,
         *wx_ = nullptr, *wy_ = nullptr, *ulA_ = nullptr, *ull_ = nullptr, *ulu_ = nullptr;
  G * g_        = nullptr;
  U * udes_     = nullptr;
  uint32_t * iter_ = nullptr;
  int32_t * code_  = nullptr;
  sfb_workspace * ws_ = nullptr;
  std::vector<double> hx_;
  std::vector<int32_t> hcode_;
  std::vector<uint32_t> hiter_;
  bool have_warm_ = false;
};

}  // namespace smooth_feedback_amd
