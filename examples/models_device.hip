// Example / test harness of the device-side fronts (compiled by hipcc into libsfb_models_dev.so): the vehicle safety
// filter of examples/mpc_asif_vehicle.cpp for a swarm, assembled AND solved on the GPU (ASIFSwarmDevice), and the vehicle
// MPC swarm linearised on the GPU (MPCSwarmDeviceLin).
#include <cstdint>
#include <cstdio>
#include <vector>

#include <chrono>
#include <random>

#include <smooth_feedback_amd/asif_device.hpp>
#include <smooth_feedback_amd/ekf_device.hpp>
#include <smooth_feedback_amd/mpc_device.hpp>
#include <smooth_feedback_amd/multi_device.hpp>

#include "vehicle_model.h"

using namespace smooth_feedback_amd;
using sfbx::U2;
using sfbx::X6;

namespace {
// the swarm of sfbx_mpc_swarm_step (models.cpp): agent b at time 0.025 (b mod 400), its state off the desired one
template<class X>
X perturbed(const X & x0, uint64_t seed)
{
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> d(-0.5, 0.5);
  typename X::Tangent xi{};
  for (auto & v : xi) v = d(rng);
  return rplus(x0, xi);
}

// `ticks` closed-loop ticks of the swarm of sfbx_mpc_swarm_step through any swarm type with MPCSwarmDeviceLin's step()
template<class Model, class Swarm>
void devlin_loop(Swarm & swarm, const Model & mdl, int64_t batch, uint64_t seed, int ticks, double * u0, int32_t * codes, uint32_t * iters,
                 double * seconds)
{
  using X = decltype(std::declval<const Model &>().xdes(0.0));
  std::vector<double> t((size_t)batch);
  std::vector<X> xs((size_t)batch);
  for (int64_t b = 0; b < batch; ++b) {
    t[b]  = 0.025 * double(b % 400);
    xs[b] = perturbed(mdl.xdes(t[b]), seed + (uint64_t)b);
  }
  std::vector<sfbx::U2> us;
  std::vector<QPSolutionStatus> cs;
  for (int k = 0; k < ticks; ++k) {
    if (k > 0)
      for (int64_t b = 0; b < batch; ++b) {  // crude closed loop, as in models.cpp
        auto f = mdl.f(xs[b], us[b]);
        for (auto & v : f) v *= 0.025;
        xs[b] = rplus(xs[b], f);
        t[b] += 0.025;
      }
    const auto t0 = std::chrono::steady_clock::now();
    swarm.step(t, xs, us, cs);
    if (seconds) seconds[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  for (int64_t b = 0; b < batch; ++b) {
    u0[2 * b] = us[b].v[0]; u0[2 * b + 1] = us[b].v[1];
    codes[b] = (int32_t)cs[b];
    iters[b] = swarm.iterations()[b];
  }
}

template<class MPCT, class Model>
int devlin_step(int K, double tf, int64_t batch, uint64_t seed, int ticks, int probe_empty, double * u0, int32_t * codes, uint32_t * iters,
                double * records, int64_t * record_doubles, int32_t * packed, double * seconds)
{
  const Model mdl{};
  auto mpc = sfbx::make_vehicle_mpc<MPCT, Model>(K, tf);
  MPCSwarmDeviceLin<MPCT, Model> swarm(mpc, mdl, batch, 0.0, probe_empty != 0);
  devlin_loop(swarm, mdl, batch, seed, ticks, u0, codes, iters, seconds);
  if (record_doubles) *record_doubles = swarm.record_doubles();
  if (packed) *packed = swarm.packed_records() ? 1 : 0;
  if (records) swarm.copy_records(records);
  return 0;
}

template<class MPCT, class Model>
int devlin_step_multi(int K, double tf, int64_t batch, uint64_t seed, int ticks, const int * devices, int ndev, int thread_per_shard, double * u0,
                      int32_t * codes, uint32_t * iters, double * seconds)
{
  const Model mdl{};
  auto mpc = sfbx::make_vehicle_mpc<MPCT, Model>(K, tf);
  MPCSwarmMultiDeviceLin<MPCT, Model> swarm(mpc, mdl, batch, std::vector<int>(devices, devices + ndev));
  swarm.shards().thread_per_shard(thread_per_shard != 0);
  devlin_loop(swarm, mdl, batch, seed, ticks, u0, codes, iters, seconds);
  return 0;
}

// the same rounds through EKFSwarmMultiDevice (fused: 0 = predict + update, 1 = step())
template<EKFStepper Stp>
int ekf_swarm_multi(int64_t batch, int steps, int fused, double tau, double dt, const int * devices, int ndev, int thread_per_shard,
                    const double * states, const double * P0, const double * y, double * states_out, double * P_out, int32_t * info)
{
  EKFSwarmMultiDevice<X6, sfbx::VehicleEkfDyn, sfbx::VehicleEkfMeas, 3, Stp> swarm(sfbx::VehicleEkfDyn{}, sfbx::VehicleEkfMeas{}, batch,
                                                                                  std::vector<int>(devices, devices + ndev));
  swarm.shards().thread_per_shard(thread_per_shard != 0);
  std::vector<X6> g((size_t)batch);
  std::vector<Mat<6, 6>> P((size_t)batch);
  for (int64_t b = 0; b < batch; ++b) {
    g[b] = sfbx::vehicle_state(states + 7 * b);
    std::copy(P0 + 36 * b, P0 + 36 * (b + 1), P[b].a.begin());
  }
  swarm.reset(g, P);
  const auto Q = sfbx::vehicle_ekf_Q();
  const auto R = sfbx::vehicle_ekf_R();
  std::vector<Vec<3>> ys((size_t)batch);
  for (int k = 0; k < steps; ++k) {
    for (int64_t b = 0; b < batch; ++b) ys[b] = {y[((size_t)k * batch + b) * 3], y[((size_t)k * batch + b) * 3 + 1], y[((size_t)k * batch + b) * 3 + 2]};
    if (fused) {
      swarm.step(Q, tau, ys, R);
    } else {
      swarm.predict(Q, tau, dt > 0 ? std::optional<double>(dt) : std::nullopt);
      swarm.update(ys, R);
    }
  }
  g = swarm.estimates();
  P = swarm.covariances();
  const auto inf = swarm.update_info();
  for (int64_t b = 0; b < batch; ++b) {
    sfbx::vehicle_state_out(g[b], states_out + 7 * b);
    std::copy(P[b].a.begin(), P[b].a.end(), P_out + 36 * b);
    info[b] = inf[b];
  }
  return 0;
}

template<EKFStepper Stp>
int ekf_swarm(int64_t batch, int steps, int fused, double tau, double dt, const double * states, const double * P0, const double * y,
              double * states_out, double * P_out, int32_t * info, double * seconds)
{
  EKFSwarmDevice<X6, sfbx::VehicleEkfDyn, sfbx::VehicleEkfMeas, 3, Stp> swarm(sfbx::VehicleEkfDyn{}, sfbx::VehicleEkfMeas{}, batch);
  std::vector<X6> g((size_t)batch);
  std::vector<Mat<6, 6>> P((size_t)batch);
  for (int64_t b = 0; b < batch; ++b) {
    g[b] = sfbx::vehicle_state(states + 7 * b);
    std::copy(P0 + 36 * b, P0 + 36 * (b + 1), P[b].a.begin());
  }
  swarm.reset(g, P);
  swarm.one_launch(fused != 2);  // fused: 1 = step() in one launch, 2 = step() as separate launches, 3 = 1 with resident measurements
  double * dy = nullptr;
  if (fused == 3) {
    if (hipMalloc(reinterpret_cast<void **>(&dy), (size_t)steps * batch * 24) != hipSuccess) return -3;
    (void)hipMemcpy(dy, y, (size_t)steps * batch * 24, hipMemcpyHostToDevice);
  }
  const auto Q = sfbx::vehicle_ekf_Q();
  const auto R = sfbx::vehicle_ekf_R();
  std::vector<Vec<3>> ys((size_t)batch);
  for (int k = 0; k < steps; ++k) {
    for (int64_t b = 0; b < batch; ++b) ys[b] = {y[((size_t)k * batch + b) * 3], y[((size_t)k * batch + b) * 3 + 1], y[((size_t)k * batch + b) * 3 + 2]};
    const auto t0 = std::chrono::steady_clock::now();
    if (fused == 3) {  // measurements already on the device: copied device-to-device into the swarm's buffer
      (void)hipMemcpy(swarm.device_measurements(), dy + (size_t)k * batch * 3, (size_t)batch * 24, hipMemcpyDeviceToDevice);
      swarm.step_resident(Q, tau, R);
    } else if (fused) {
      swarm.step(Q, tau, ys, R);
    } else {
      swarm.predict(Q, tau, dt > 0 ? std::optional<double>(dt) : std::nullopt);
      swarm.update(ys, R);
    }
    (void)hipDeviceSynchronize();
    if (seconds) seconds[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (dy) (void)hipFree(dy);
  g = swarm.estimates();
  P = swarm.covariances();
  const auto inf = swarm.update_info();
  for (int64_t b = 0; b < batch; ++b) {
    sfbx::vehicle_state_out(g[b], states_out + 7 * b);
    std::copy(P[b].a.begin(), P[b].a.end(), P_out + 36 * b);
    info[b] = inf[b];
  }
  return 0;
}
}  // namespace

extern "C" {

/* A swarm of vehicle EKFs on the GPU (EKFSwarmDevice; model: vehicle_model.h VehicleEkfDyn / VehicleEkfMeas): `steps` times
 * predict(Q, tau, dt) + update(y[step], R) -- or the fused step() (fused != 0, Euler, one substep) -- from states [batch][7]
 * and covariances P0 [batch][36]; y [steps][batch][3].  dt <= 0: the default single substep.  Out: states, covariances, the
 * last update's info, seconds[steps].  sfbx_ekf_swarm_host (models.h) does the same with one host EKF<> object per filter. */
int sfbx_ekf_swarm_device(int64_t batch, int steps, int rk4, int fused, double tau, double dt, const double * states, const double * P0,
                          const double * y, double * states_out, double * P_out, int32_t * info, double * seconds)
{
  try {
    return rk4 ? ekf_swarm<EKFStepper::RK4>(batch, steps, fused, tau, dt, states, P0, y, states_out, P_out, info, seconds)
               : ekf_swarm<EKFStepper::Euler>(batch, steps, fused, tau, dt, states, P0, y, states_out, P_out, info, seconds);
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sfbx_ekf_swarm_device: %s\n", e.what());
    return -2;
  }
}

/* sfbx_ekf_swarm_device through EKFSwarmMultiDevice (multi_device.hpp): the filters sharded over `devices` (an ordinal may repeat),
 * one resident swarm per shard.  thread_per_shard != 0: a host thread per shard even on one device (test hook). */
int sfbx_ekf_swarm_device_multi(int64_t batch, int steps, int rk4, int fused, double tau, double dt, const int * devices, int ndev,
                                int thread_per_shard, const double * states, const double * P0, const double * y, double * states_out,
                                double * P_out, int32_t * info)
{
  try {
    return rk4 ? ekf_swarm_multi<EKFStepper::RK4>(batch, steps, fused, tau, dt, devices, ndev, thread_per_shard, states, P0, y, states_out, P_out, info)
               : ekf_swarm_multi<EKFStepper::Euler>(batch, steps, fused, tau, dt, devices, ndev, thread_per_shard, states, P0, y, states_out, P_out,
                                                    info);
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sfbx_ekf_swarm_device_multi: %s\n", e.what());
    return -2;
  }
}

/* sfbx_mpc_swarm_devlin_step through MPCSwarmMultiDeviceLin (multi_device.hpp): the agents sharded over `devices`. */
int sfbx_mpc_swarm_devlin_step_multi(int variant, int K, double tf, int64_t batch, uint64_t seed, int ticks, const int * devices, int ndev,
                                     int thread_per_shard, double * u0, int32_t * codes, uint32_t * iters, double * seconds)
{
  try {
    if (variant == 6)
      return devlin_step_multi<sfbx::MPC6, sfbx::VehicleModel6>(K, tf, batch, seed, ticks, devices, ndev, thread_per_shard, u0, codes, iters, seconds);
    if (variant == 12)
      return devlin_step_multi<sfbx::MPC12, sfbx::VehicleModel12>(K, tf, batch, seed, ticks, devices, ndev, thread_per_shard, u0, codes, iters, seconds);
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sfbx_mpc_swarm_devlin_step_multi: %s\n", e.what());
    return -2;
  }
  return -1;
}

/* examples/mpc_asif_vehicle.cpp:151-175 for a swarm, both controllers on the GPU: per 25 ms tick the MPC input of every vehicle
 * (MPCSwarmDeviceLin, K_mpc), filtered by the ASI filter (ASIFSwarmDevice, K_asif), then one runge_kutta4 step of the
 * closed loop.  Vehicle 0 starts at the identity like the example, the others off it by xi ~ U(-0.3, 0.3)^6 (mt19937_64(seed + b)).
 * Out per tick: positions xy [ticks][batch][2], u_mpc / u_asif [ticks][batch][2], the number of non-Optimal MPC / ASIF
 * solves, the smallest barrier value h(x) over the swarm, seconds [ticks][2] (MPC, ASIF). */
int sfbx_vehicle_swarm_sim(int64_t batch, int K_mpc, int K_asif, int ticks, uint64_t seed, double * xy, double * u_mpc, double * u_asif,
                           int32_t * mpc_bad, int32_t * asif_bad, double * hmin, double * seconds)
{
  try {
    const sfbx::VehicleModel6 mdl{};
    auto mpc = sfbx::make_vehicle_mpc<sfbx::MPC6, sfbx::VehicleModel6>(K_mpc, 5.0);
    MPCSwarmDeviceLin<sfbx::MPC6, sfbx::VehicleModel6> ctrl(mpc, mdl, batch);
    ASIFSwarmDevice<X6, U2, sfbx::VehicleDyn6, sfbx::VehicleH, sfbx::VehicleBU> filt(sfbx::VehicleDyn6{}, sfbx::VehicleH{}, sfbx::VehicleBU{},
                                                                                    (size_t)batch, sfbx::vehicle_asif_params(K_asif));
    std::vector<X6> x((size_t)batch);
    for (int64_t b = 1; b < batch; ++b) {
      std::mt19937_64 rng(seed + (uint64_t)b);
      std::uniform_real_distribution<double> d(-0.3, 0.3);
      X6::Tangent xi{};
      for (auto & v : xi) v = d(rng);
      x[b] = rplus(X6::Identity(), xi);
    }
    std::vector<double> t((size_t)batch, 0.0);
    std::vector<U2> um, ua;
    std::vector<QPSolutionStatus> cs;
    const double dt = 0.025;
    for (int k = 0; k < ticks; ++k) {
      auto t0 = std::chrono::steady_clock::now();
      ctrl.step(t, x, um, cs);
      auto t1 = std::chrono::steady_clock::now();
      ua = filt(x, um);
      auto t2 = std::chrono::steady_clock::now();
      seconds[2 * k]     = std::chrono::duration<double>(t1 - t0).count();
      seconds[2 * k + 1] = std::chrono::duration<double>(t2 - t1).count();
      mpc_bad[k] = asif_bad[k] = 0;
      hmin[k]    = 1e300;
      for (int64_t b = 0; b < batch; ++b) {
        mpc_bad[k] += cs[b] != QPSolutionStatus::Optimal;
        asif_bad[k] += filt.codes()[b] != 0;
        double * o = xy + ((size_t)k * batch + b) * 2;
        o[0] = x[b].part<0>().x; o[1] = x[b].part<0>().y;
        double * pm = u_mpc + ((size_t)k * batch + b) * 2, *pa = u_asif + ((size_t)k * batch + b) * 2;
        pm[0] = um[b].v[0]; pm[1] = um[b].v[1]; pa[0] = ua[b].v[0]; pa[1] = ua[b].v[1];
        hmin[k] = std::min(hmin[k], sfbx::VehicleH{}(0.0, x[b])[0]);
        const U2 u = ua[b];
        auto f     = [&](double, const X6 & g) { return sfbx::VehicleDyn6{}(g, u); };
        x[b]       = detail::ekf_rk4_state(f, 0.0, dt, x[b], f(0.0, x[b]));  // runge_kutta4 on the group (:146-147)
        t[b] += dt;
      }
    }
    return 0;
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sfbx_vehicle_swarm_sim: %s\n", e.what());
    return -2;
  }
}

/* sfbx_mpc_swarm_device_step (models.h) with the linearisation on the GPU as well (MPCSwarmDeviceLin): same agents, same
 * closed loop.  Also out: the records of the LAST tick as the device wrote them ([batch][*record_doubles], room for the
 * unpacked size; NULL to skip), whether they are packed, seconds[ticks].  probe_empty != 0: start from an empty packing
 * (every non-zero Jacobian entry is then a misfit -> the unpacked fallback runs). */
int sfbx_mpc_swarm_devlin_step(int variant, int K, double tf, int64_t batch, uint64_t seed, int ticks, int probe_empty, double * u0,
                               int32_t * codes, uint32_t * iters, double * records, int64_t * record_doubles, int32_t * packed,
                               double * seconds)
{
  try {
    if (variant == 6)
      return devlin_step<sfbx::MPC6, sfbx::VehicleModel6>(K, tf, batch, seed, ticks, probe_empty, u0, codes, iters, records, record_doubles,
                                                          packed, seconds);
    if (variant == 12)
      return devlin_step<sfbx::MPC12, sfbx::VehicleModel12>(K, tf, batch, seed, ticks, probe_empty, u0, codes, iters, records,
                                                            record_doubles, packed, seconds);
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sfbx_mpc_swarm_devlin_step: %s\n", e.what());
    return -2;
  }
  return -1;
}

/* Like sfbx_asif_swarm_step (models.h) from given states [batch][7] = (x, y, cos, sin, v0, v1, v2) and desired inputs
 * [batch][2]: `ticks` consecutive filter calls, the vehicles moved 25 ms along their filtered inputs in between.  Out for
 * the LAST tick: u [batch][2], codes, iters, the QPs (n = 3, m = K + 3), their primal / dual and the warm start used;
 * seconds[ticks]: wall time of every filter call. */
int sfbx_asif_swarm_device_step(int64_t batch, int K, int ticks, const double * states, const double * udes, double * u_out,
                                int32_t * codes, uint32_t * iters, double * P, double * q, double * A, double * l, double * u,
                                double * x, double * y, double * wx, double * wy, double * seconds)
{
  try {
    ASIFSwarmDevice<X6, U2, sfbx::VehicleDyn6, sfbx::VehicleH, sfbx::VehicleBU> swarm(sfbx::VehicleDyn6{}, sfbx::VehicleH{}, sfbx::VehicleBU{},
                                                                                     (size_t)batch, sfbx::vehicle_asif_params(K));
    std::vector<X6> g((size_t)batch);
    std::vector<U2> ud((size_t)batch);
    for (int64_t b = 0; b < batch; ++b) {
      const double * s = states + 7 * b;
      g[b].part<0>()   = SE2{s[0], s[1], s[2], s[3]};
      g[b].part<1>().v = {s[4], s[5], s[6]};
      ud[b].v          = {udes[2 * b], udes[2 * b + 1]};
    }
    std::vector<U2> out;
    for (int tick = 0; tick < ticks; ++tick) {
      if (tick > 0) {
        for (int64_t b = 0; b < batch; ++b) {
          auto dx = sfbx::VehicleDyn6{}(g[b], out[b]);
          for (auto & v : dx) v *= 0.025;
          g[b] = rplus(g[b], dx);
        }
      }
      if (tick == ticks - 1) swarm.copy_warm_start(wx, wy);
      const auto t0 = std::chrono::steady_clock::now();
      out = swarm(g, ud);
      if (seconds) seconds[tick] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    for (int64_t b = 0; b < batch; ++b) { u_out[2 * b] = out[b].v[0]; u_out[2 * b + 1] = out[b].v[1]; }
    std::copy(swarm.codes().begin(), swarm.codes().end(), codes);
    std::copy(swarm.iterations().begin(), swarm.iterations().end(), iters);
    swarm.copy_problem(P, q, A, l, u, x, y);
    return 0;
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sfbx_asif_swarm_device_step: %s\n", e.what());
    return 1;
  }
}

}  // extern "C"
