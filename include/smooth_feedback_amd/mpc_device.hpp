// MPC swarm whose LINEARISATION runs on the GPU as well (HIP only: include from a translation unit compiled by hipcc).
// MPCSwarmDevice (mpc.hpp) already keeps assembly, solve and warm starts on the device; what is left on the host there
// is MPC::fill_record -- the desired trajectory, the dynamics and the running constraint with their Jacobians at every
// collocation node (ocp_to_qp.hpp:250-257, :300-313, :345-357) -- and the upload of the records.  For a model whose
// functors are device-callable, one GPU thread per agent and node writes its part of the record straight into the
// swarm's device buffer (sfb_mpc_swarm_device_records) and the tick continues from there
// (sfb_mpc_swarm_step_resident): only the agents' times and states go up.
//
// Model: a struct with `__host__ __device__` members
//   X xdes(double t), Vec<Nx> dxdes(double t), U udes(double t)   desired trajectory (MPC::set_xdes / set_udes)
//   F f, CR cr                                                    dynamics f(x, u) and running constraint cr(x, u),
//                                                                 optionally with jacobian(x, u, dx, du) members
// -- the same functions the host MPC object of the swarm was built from.  Records equal the host's fill_record up to
// what sin / cos of the two maths libraries differ; everything downstream is the same kernels.
#pragma once
#ifndef __HIPCC__
#error "mpc_device.hpp needs hipcc"
#endif
#include <hip/hip_runtime.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mpc.hpp"

namespace smooth_feedback_amd {

namespace detail {

/// where the pieces of a record go (sfb.h: [ f | dxdes | dfdx | dfdu | c | dcdx | dcdu | e | J ], Jacobians packed by pos_*)
template<int Nx, int Nu, int Ncr>
struct RecordMapDev {
  int N;
  double tf;
  int n_fx, n_fu, n_cx, n_cu, n_J;
  int o_f, o_dx, o_dfdx, o_dfdu, o_c, o_dcdx, o_dcdu, o_e, o_J;
  int64_t rec_doubles;
  int16_t pos_fx[Nx * Nx], pos_fu[Nx * Nu], pos_cx[(Ncr > 0 ? Ncr : 1) * Nx], pos_cu[(Ncr > 0 ? Ncr : 1) * Nu], pos_J[Nx * Nx];  // -1: not carried
};

template<class X, class U, int Ncr, class Model>
__global__ void __launch_bounds__(64) mpc_linearise_kernel(const int64_t B, const RecordMapDev<X::Dof, U::Dof, Ncr> map, const Model model,
                                                           const double * __restrict__ tau, const double * __restrict__ t,
                                                           const X * __restrict__ xs, double * __restrict__ records,
                                                           int * __restrict__ misfit)
{
  constexpr int Nx = X::Dof, Nu = U::Dof;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per     = map.N + 1;
  const int64_t b   = gid / per;
  const int node    = (int)(gid - b * per);
  if (b >= B) return;
  double * rec = records + b * map.rec_doubles;
  bool bad     = false;
  auto put = [&](double * dst, const int16_t * pos, int kept, int block, int e, double v) {
    const int p = pos[e];
    if (p >= 0) dst[(int64_t)block * kept + p] = v;
    else bad = bad || !(v == 0.0);
  };
  if (node < map.N) {
    const double ti = t[b] + map.tf * tau[node];
    const X xl      = model.xdes(ti);
    const auto dxl  = model.dxdes(ti);
    const U ul      = model.udes(ti);
    Vec<Nx> fv;
    Mat<Nx, Nx> dfdx;
    Mat<Nx, Nu> dfdu;
    xu_jacobian<Nx>(model.f, xl, ul, fv, dfdx, dfdu);
    // (every loop over matrix entries unrolled: with run-time indices the matrices live in scratch memory -- 1.4 KB per lane at Nx = 12)
#pragma unroll
    for (int d = 0; d < Nx; ++d) {
      rec[map.o_f + node * Nx + d]  = fv[d];
      rec[map.o_dx + node * Nx + d] = dxl[d];
#pragma unroll
      for (int c = 0; c < Nx; ++c) put(rec + map.o_dfdx, map.pos_fx, map.n_fx, node, d * Nx + c, dfdx(d, c));
#pragma unroll
      for (int c = 0; c < Nu; ++c) put(rec + map.o_dfdu, map.pos_fu, map.n_fu, node, d * Nu + c, dfdu(d, c));
    }
    if constexpr (Ncr > 0) {
      Vec<Ncr> cv;
      Mat<Ncr, Nx> dcdx;
      Mat<Ncr, Nu> dcdu;
      xu_jacobian<Ncr>(model.cr, xl, ul, cv, dcdx, dcdu);
#pragma unroll
      for (int d = 0; d < Ncr; ++d) {
        rec[map.o_c + node * Ncr + d] = cv[d];
#pragma unroll
        for (int c = 0; c < Nx; ++c) put(rec + map.o_dcdx, map.pos_cx, map.n_cx, node, d * Nx + c, dcdx(d, c));
#pragma unroll
        for (int c = 0; c < Nu; ++c) put(rec + map.o_dcdu, map.pos_cu, map.n_cu, node, d * Nu + c, dcdu(d, c));
      }
    }
  } else {  // the initial-state constraint: e = xdes(t) (-) x, J = d^r exp^-1(e)   (MPCCE, mpc.hpp:288-301)
    const auto e = rminus(model.xdes(t[b]), xs[b]);
    const auto J = X::dr_expinv(e);
#pragma unroll
    for (int d = 0; d < Nx; ++d) {
      rec[map.o_e + d] = e[d];
#pragma unroll
      for (int c = 0; c < Nx; ++c) put(rec + map.o_J, map.pos_J, map.n_J, 0, d * Nx + c, J(d, c));
    }
  }
  if (bad) atomicOr(misfit, 1);
}

inline void mpc_hip_check(hipError_t e, const char * what)
{
  if (e != hipSuccess) throw std::runtime_error(std::string("mpc_device: ") + what + ": " + hipGetErrorString(e));
}
}  // namespace detail

/// MPCSwarmDevice with the linearisation on the GPU.  `proto` is the host MPC object of the same model (structure probing,
/// symbolic analysis, mesh, parameters), `model` its device-callable twin.
template<class MPCT, class Model>
class MPCSwarmDeviceLin {
public:
  using X = decltype(std::declval<const Model &>().xdes(0.0));
  using U = decltype(std::declval<const Model &>().udes(0.0));
  static constexpr int Nx = MPCT::Nx, Nu = MPCT::Nu, Ncr = MPCT::Ncr;
  using Map = detail::RecordMapDev<Nx, Nu, Ncr>;

  MPCSwarmDeviceLin(MPCT & proto, Model model, int64_t agents, double t_probe = 0.0, bool probe_empty = false)
      : mpc_(proto), model_(model), B_(agents)
  {
    if (!mpc_.solver().analyzed()) {
      std::vector<uint8_t> keep;
      mpc_.probe_default(t_probe, keep);
      mpc_.analyze_solver(&keep);
    }
    if (!probe_empty) pack_ = mpc_.probe_record_default(t_probe);  // probe_empty: tests of the misfit fallback
    packed_ = mpc_.params().prune_explicit_zeros && pack_.doubles(mpc_.N()) < MPCT::record_doubles(mpc_.N());
    layout_ = mpc_.device_layout(packed_ ? &pack_.keep : nullptr);
    const auto & qp = mpc_.qp();
    static_assert(std::is_same_v<typename MPCT::TimeT, double>, "the device-side linearisation takes time as double seconds");
    sfb_check(sfb_mpc_swarm_create(mpc_.solver().plan(), &layout_->c, qp.P_val.data(), qp.q.data(), B_, &swarm_));
    mpc_.solver().pin_plan();  // the swarm holds the raw plan pointer
    sfb_check(sfb_mpc_swarm_device_records(swarm_, &drec_, nullptr));
    const int Nn = mpc_.N();
    std::vector<double> tau((size_t)Nn + 1);
    for (int i = 0; i <= Nn; ++i) tau[(size_t)i] = mpc_.mesh().node(i);
    detail::mpc_hip_check(hipMalloc(reinterpret_cast<void **>(&dmem_), ((size_t)Nn + 1 + (size_t)B_) * 8 + (size_t)B_ * sizeof(X) + 16), "hipMalloc");
    dtau_    = reinterpret_cast<double *>(dmem_);
    dt_      = dtau_ + Nn + 1;
    dx_      = reinterpret_cast<X *>(dt_ + B_);
    dmisfit_ = reinterpret_cast<int *>(dx_ + B_);
    detail::mpc_hip_check(hipMemcpy(dtau_, tau.data(), tau.size() * 8, hipMemcpyHostToDevice), "hipMemcpy(tau)");
    build_map();
    du0_.resize((size_t)B_ * Nu);
    iter_.resize((size_t)B_);
    code_.resize((size_t)B_);
  }
  MPCSwarmDeviceLin(const MPCSwarmDeviceLin &)             = delete;
  MPCSwarmDeviceLin & operator=(const MPCSwarmDeviceLin &) = delete;
  ~MPCSwarmDeviceLin()
  {
    sfb_mpc_swarm_destroy(swarm_);
    mpc_.solver().unpin_plan();
    if (dmem_) (void)hipFree(dmem_);
  }

  void reset_warmstart() { sfb_check(sfb_mpc_swarm_reset_warmstart(swarm_)); }

  /// one control tick for all agents
  void step(const std::vector<double> & t, const std::vector<X> & xs, std::vector<U> & us, std::vector<QPSolutionStatus> & codes)
  {
    if ((int64_t)t.size() != B_ || (int64_t)xs.size() != B_) throw std::invalid_argument("MPCSwarmDeviceLin: one time and state per agent");
    us.resize((size_t)B_);
    codes.resize((size_t)B_);
    step(t.data(), xs.data(), us.data(), codes.data());
  }
  /// the same on arrays of size() entries (a shard of a larger swarm: multi_device.hpp)
  void step(const double * t, const X * xs, U * us, QPSolutionStatus * codes)
  {
    detail::mpc_hip_check(hipMemcpy(dt_, t, (size_t)B_ * 8, hipMemcpyHostToDevice), "hipMemcpy(t)");
    detail::mpc_hip_check(hipMemcpy(dx_, xs, (size_t)B_ * sizeof(X), hipMemcpyHostToDevice), "hipMemcpy(x)");
    linearise();
    int misfit = 0;
    detail::mpc_hip_check(hipMemcpy(&misfit, dmisfit_, sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy(flag)");
    if (misfit && packed_) {
      // a linearisation has a non-zero where the probe saw none: unpacked records from now on (same results; warm
      // starts and solver memory of the swarm are untouched), and this tick's records once more
      packed_ = false;
      sfb_check(sfb_mpc_swarm_set_jac_keep(swarm_, nullptr, nullptr));
      build_map();
      linearise();
    }
    const sfb_qp_params c = mpc_.solver().params().to_c();
    sfb_check(sfb_mpc_swarm_step_resident(swarm_, &c, mpc_.params().warmstart ? 1 : 0, du0_.data(), iter_.data(), code_.data(), nullptr,
                                          nullptr));
    for (int64_t b = 0; b < B_; ++b) {
      us[(size_t)b]    = mpc_.input_from_du0(t[(size_t)b], &du0_[(size_t)b * Nu]);
      codes[(size_t)b] = static_cast<QPSolutionStatus>(code_[(size_t)b]);
    }
  }
  int64_t size() const { return B_; }
  const std::vector<uint32_t> & iterations() const { return iter_; }
  bool packed_records() const { return packed_; }
  int64_t record_doubles() const { return map_.rec_doubles; }
  /// the records of the last tick (device -> host), [agents][record_doubles()]
  void copy_records(double * out) const
  {
    detail::mpc_hip_check(hipMemcpy(out, drec_, (size_t)B_ * (size_t)map_.rec_doubles * 8, hipMemcpyDeviceToHost), "hipMemcpy(records)");
  }

private:
  void build_map()
  {
    const int Nn = mpc_.N();
    Map & m = map_;
    m.N  = Nn;
    m.tf = mpc_.params().tf;
    const uint8_t * k = pack_.keep.data();
    auto fill = [&](int16_t * pos, int len) {
      int cnt = 0;
      for (int e = 0; e < len; ++e) pos[e] = (!packed_ || k[e]) ? (int16_t)cnt++ : (int16_t)-1;
      k += len;
      return cnt;
    };
    m.n_fx = fill(m.pos_fx, Nx * Nx); m.n_fu = fill(m.pos_fu, Nx * Nu); m.n_cx = fill(m.pos_cx, Ncr * Nx);
    m.n_cu = fill(m.pos_cu, Ncr * Nu); m.n_J = fill(m.pos_J, Nx * Nx);
    int64_t o = 0;
    m.o_f = (int)o; o += (int64_t)Nn * Nx;
    m.o_dx = (int)o; o += (int64_t)Nn * Nx;
    m.o_dfdx = (int)o; o += (int64_t)Nn * m.n_fx;
    m.o_dfdu = (int)o; o += (int64_t)Nn * m.n_fu;
    m.o_c = (int)o; o += (int64_t)Nn * Ncr;
    m.o_dcdx = (int)o; o += (int64_t)Nn * m.n_cx;
    m.o_dcdu = (int)o; o += (int64_t)Nn * m.n_cu;
    m.o_e = (int)o; o += Nx;
    m.o_J = (int)o; o += m.n_J;
    m.rec_doubles = o;
    int64_t rd = 0;
    sfb_check(sfb_mpc_swarm_device_records(swarm_, &drec_, &rd));
    if (rd != o) throw std::logic_error("mpc_device: record layout differs from the swarm's");
  }
  void linearise()
  {
    detail::mpc_hip_check(hipMemsetAsync(dmisfit_, 0, sizeof(int), nullptr), "hipMemsetAsync");
    const int64_t threads = B_ * (map_.N + 1);
    hipLaunchKernelGGL((detail::mpc_linearise_kernel<X, U, Ncr, Model>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, nullptr, B_, map_,
                       model_, dtau_, dt_, dx_, drec_, dmisfit_);
    detail::mpc_hip_check(hipGetLastError(), "mpc_linearise_kernel");
  }

  MPCT & mpc_;
  Model model_;
  int64_t B_;
  typename MPCT::RecordPacking pack_;
  bool packed_ = false;
  std::unique_ptr<typename MPCT::DeviceLayout> layout_;
  sfb_mpc_swarm * swarm_ = nullptr;
  Map map_{};
  char * dmem_   = nullptr;
  double * dtau_ = nullptr, *dt_ = nullptr, *drec_ = nullptr;
  X * dx_        = nullptr;
  int * dmisfit_ = nullptr;
  std::vector<double> du0_;
  std::vector<uint32_t> iter_;
  std::vector<int32_t> code_;
};

}  // namespace smooth_feedback_amd
