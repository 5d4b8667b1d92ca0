// Example / test harness of the device-side fronts (compiled by hipcc into libsfb_models_dev.so): the vehicle safety
// filter of examples/mpc_asif_vehicle.cpp for a swarm, assembled AND solved on the GPU (ASIFSwarmDevice).
#include <cstdint>
#include <cstdio>
#include <vector>

#include <smooth_feedback_amd/asif_device.hpp>

#include "vehicle_model.h"

using namespace smooth_feedback_amd;
using sfbx::U2;
using sfbx::X6;

extern "C" {

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
