// Device-resident swarms of ONE process sharded over several GPUs.
//
// The reference has no parallelism at all (benchmarks/bench_types.hpp:93 is a sequential loop over the problems; one
// MPC::operator() per agent, mpc.hpp:458-519; one EKF object per filter, ekf.hpp:43-147).  A swarm is a batch of
// independent agents, so it shards trivially: contiguous ranges of agents, one resident swarm (MPCSwarmDeviceLin /
// EKFSwarmDevice: own device memory, own plan upload, own workspace) and one host thread per device.  Per tick only the
// agents' states go up and the inputs / codes / iteration counts (24 B per agent) come down; nothing is exchanged
// between devices -- host memory is the gathering point, like in the *_host_multi entry points of sfb.h.
// Device list: sfb_get_devices() (sfb_set_devices; default: every visible device) or the constructor's argument.  An
// ordinal may appear more than once: its shards then run one after the other on that device's thread (the tests on
// 1-GPU boxes; results are those of the single-device swarm bit for bit, whatever the list).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdint>
#include <exception>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../sfb.h"
#include "ekf_device.hpp"
#include "mpc_device.hpp"

namespace smooth_feedback_amd {

/// contiguous shards of a batch over a device list
class DeviceShards {
public:
  struct Shard { int device; int64_t first, count; };

  /// devices empty: the list of the process (sfb_get_devices).  Shards that would be empty are left out.
  explicit DeviceShards(int64_t batch, std::vector<int> devices = {})
  {
    if (batch < 1) throw std::invalid_argument("DeviceShards: empty batch");
    if (devices.empty()) {
      int cnt = 0;
      if (sfb_get_devices(nullptr, 0, &cnt) != SFB_OK || cnt < 1) throw std::runtime_error("DeviceShards: no HIP device");
      devices.resize((size_t)cnt);
      (void)sfb_get_devices(devices.data(), cnt, &cnt);
    }
    const int64_t G = (int64_t)devices.size();
    for (int64_t g = 0; g < G; ++g) {  // the cut of sfb's *_host_multi entry points: shard g = [g B / G, (g + 1) B / G)
      const int64_t a = g * batch / G, b = (g + 1) * batch / G;
      if (b > a) shards_.push_back({devices[(size_t)g], a, b - a});
    }
  }
  const std::vector<Shard> & shards() const { return shards_; }
  size_t size() const { return shards_.size(); }
  /// test hook for 1-GPU boxes: a host thread per SHARD, also for shards of one device (they then use it concurrently)
  void thread_per_shard(bool on) const { thread_per_shard_ = on; }

  /// f(shard index, shard) for every shard, with the shard's device current: one host thread per DISTINCT device, the
  /// shards of a device in order on its thread.  The first exception of any thread is rethrown here.
  template<class F>
  void for_each(F && f) const
  {
    // work lists: the shards of one device (or, thread_per_shard, every shard on its own)
    std::vector<int> distinct;
    std::vector<std::vector<size_t>> todo;
    for (size_t i = 0; i < shards_.size(); ++i) {
      size_t di = distinct.size();
      if (!thread_per_shard_)
        for (size_t d = 0; d < distinct.size(); ++d)
          if (distinct[d] == shards_[i].device) di = d;
      if (di == distinct.size()) { distinct.push_back(shards_[i].device); todo.emplace_back(); }
      todo[di].push_back(i);
    }
    std::vector<std::exception_ptr> err(distinct.size());
    auto work = [&](size_t di) {
      try {
        if (hipSetDevice(distinct[di]) != hipSuccess) throw std::runtime_error("DeviceShards: hipSetDevice(" + std::to_string(distinct[di]) + ")");
        for (size_t i : todo[di]) f(i, shards_[i]);
      } catch (...) {
        err[di] = std::current_exception();
      }
    };
    if (distinct.size() == 1) {  // nothing to overlap: the caller's thread, its current device restored afterwards
      int prev = -1;
      (void)hipGetDevice(&prev);
      work(0);
      if (prev >= 0) (void)hipSetDevice(prev);
    } else {
      std::vector<std::thread> th;
      try {
        for (size_t di = 0; di < distinct.size(); ++di) th.emplace_back(work, di);
      } catch (...) {  // (thread creation failed: finish what was started, then report)
        for (auto & t : th) t.join();
        throw;
      }
      for (auto & t : th) t.join();
    }
    for (auto & e : err)
      if (e) std::rethrow_exception(e);
  }

private:
  std::vector<Shard> shards_;
  mutable bool thread_per_shard_ = false;
};

/// MPCSwarmDeviceLin sharded over the devices of the process: same constructor arguments (+ the device list), same step().
template<class MPCT, class Model>
class MPCSwarmMultiDeviceLin {
public:
  using One = MPCSwarmDeviceLin<MPCT, Model>;
  using X   = typename One::X;
  using U   = typename One::U;

  MPCSwarmMultiDeviceLin(MPCT & proto, Model model, int64_t agents, std::vector<int> devices = {}, double t_probe = 0.0)
      : cut_(agents, std::move(devices)), B_(agents)
  {
    // the shards share the host MPC object (structure probe, symbolic plan): created one after the other, each with its
    // device current (the plan's index arrays are uploaded per device on first use)
    int prev = -1;
    (void)hipGetDevice(&prev);
    part_.resize(cut_.size());
    try {
      for (size_t i = 0; i < cut_.size(); ++i) {
        if (hipSetDevice(cut_.shards()[i].device) != hipSuccess) throw std::runtime_error("MPCSwarmMultiDeviceLin: hipSetDevice");
        part_[i] = std::make_unique<One>(proto, model, cut_.shards()[i].count, t_probe);
      }
    } catch (...) {
      destroy();
      if (prev >= 0) (void)hipSetDevice(prev);
      throw;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    iter_.resize((size_t)B_);
  }
  MPCSwarmMultiDeviceLin(const MPCSwarmMultiDeviceLin &)             = delete;
  MPCSwarmMultiDeviceLin & operator=(const MPCSwarmMultiDeviceLin &) = delete;
  ~MPCSwarmMultiDeviceLin() { destroy(); }

  int64_t size() const { return B_; }
  const DeviceShards & shards() const { return cut_; }

  void reset_warmstart()
  {
    cut_.for_each([&](size_t i, const DeviceShards::Shard &) { part_[i]->reset_warmstart(); });
  }

  /// one control tick for all agents (MPCSwarmDeviceLin::step on every shard at once)
  void step(const std::vector<double> & t, const std::vector<X> & xs, std::vector<U> & us, std::vector<QPSolutionStatus> & codes)
  {
    if ((int64_t)t.size() != B_ || (int64_t)xs.size() != B_) throw std::invalid_argument("MPCSwarmMultiDeviceLin: one time and state per agent");
    us.resize((size_t)B_);
    codes.resize((size_t)B_);
    cut_.for_each([&](size_t i, const DeviceShards::Shard & s) {
      part_[i]->step(t.data() + s.first, xs.data() + s.first, us.data() + s.first, codes.data() + s.first);
      std::copy(part_[i]->iterations().begin(), part_[i]->iterations().end(), iter_.begin() + s.first);
    });
  }
  const std::vector<uint32_t> & iterations() const { return iter_; }

private:
  void destroy()
  {
    // a swarm's memory is freed with its device current
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < part_.size(); ++i)
      if (part_[i]) {
        (void)hipSetDevice(cut_.shards()[i].device);
        part_[i].reset();
      }
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceShards cut_;
  int64_t B_;
  std::vector<std::unique_ptr<One>> part_;
  std::vector<uint32_t> iter_;
};

/// EKFSwarmDevice sharded over the devices of the process: same interface on whole-swarm vectors.
template<class G, class Dyn, class Meas, int Ny, EKFStepper Stp = EKFStepper::Euler>
class EKFSwarmMultiDevice {
public:
  using One  = EKFSwarmDevice<G, Dyn, Meas, Ny, Stp>;
  using CovT = typename One::CovT;
  static constexpr int N = G::Dof;

  EKFSwarmMultiDevice(Dyn f, Meas h, int64_t filters, std::vector<int> devices = {}) : cut_(filters, std::move(devices)), B_(filters)
  {
    part_.resize(cut_.size());
    try {
      cut_.for_each([&](size_t i, const DeviceShards::Shard & s) { part_[i] = std::make_unique<One>(f, h, s.count); });
    } catch (...) {  // the shards built so far own memory of THEIR devices: released with those devices current
      try {
        cut_.for_each([&](size_t i, const DeviceShards::Shard &) { part_[i].reset(); });
      } catch (...) {
      }
      throw;
    }
  }
  EKFSwarmMultiDevice(const EKFSwarmMultiDevice &)             = delete;
  EKFSwarmMultiDevice & operator=(const EKFSwarmMultiDevice &) = delete;
  ~EKFSwarmMultiDevice()
  {
    try {
      cut_.for_each([&](size_t i, const DeviceShards::Shard &) { part_[i].reset(); });
    } catch (...) {
    }
  }

  int64_t size() const { return B_; }
  const DeviceShards & shards() const { return cut_; }
  void one_launch(bool on)
  {
    for (auto & p : part_) p->one_launch(on);
  }

  void reset(const std::vector<G> & g, const std::vector<CovT> & P)
  {
    if ((int64_t)g.size() != B_ || (int64_t)P.size() != B_) throw std::invalid_argument("EKFSwarmMultiDevice: one state and covariance per filter");
    cut_.for_each([&](size_t i, const DeviceShards::Shard & s) { part_[i]->reset(g.data() + s.first, P.data() + s.first); });
  }
  std::vector<G> estimates() const
  {
    std::vector<G> out((size_t)B_);
    cut_.for_each([&](size_t i, const DeviceShards::Shard & s) { part_[i]->estimates(out.data() + s.first); });
    return out;
  }
  std::vector<CovT> covariances() const
  {
    std::vector<CovT> out((size_t)B_);
    cut_.for_each([&](size_t i, const DeviceShards::Shard & s) { part_[i]->covariances(out.data() + s.first); });
    return out;
  }
  std::vector<int32_t> update_info() const
  {
    std::vector<int32_t> out((size_t)B_);
    cut_.for_each([&](size_t i, const DeviceShards::Shard & s) { part_[i]->update_info(out.data() + s.first); });
    return out;
  }

  void predict(const CovT & Q, double tau, std::optional<double> dt = {})
  {
    cut_.for_each([&](size_t i, const DeviceShards::Shard &) { part_[i]->predict(Q, tau, dt); });
  }
  void update(const std::vector<Vec<Ny>> & y, const Mat<Ny, Ny> & R)
  {
    if ((int64_t)y.size() != B_) throw std::invalid_argument("EKFSwarmMultiDevice: one measurement per filter");
    cut_.for_each([&](size_t i, const DeviceShards::Shard & s) {
      part_[i]->upload_measurements(y.data() + s.first);
      part_[i]->update_resident(R);
    });
  }
  /// predict(Q, tau) with one substep followed by update(y, R), per shard in one launch (EKFSwarmDevice::step)
  void step(const CovT & Q, double tau, const std::vector<Vec<Ny>> & y, const Mat<Ny, Ny> & R)
  {
    if ((int64_t)y.size() != B_) throw std::invalid_argument("EKFSwarmMultiDevice: one measurement per filter");
    cut_.for_each([&](size_t i, const DeviceShards::Shard & s) {
      part_[i]->upload_measurements(y.data() + s.first);
      part_[i]->step_resident(Q, tau, R);
      detail::ekf_hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");  // (asynchronous launch: the tick is over -- and its faults are reported -- when every shard's is)
    });
  }

private:
  DeviceShards cut_;
  int64_t B_;
  std::vector<std::unique_ptr<One>> part_;
};

}  // namespace smooth_feedback_amd
