// Debug knobs of libsfb.so: launch shapes and engine choices for tests, A/B measurements and diagnostics -- none of them
// changes a result (the parity tests run the variants against each other).  They are set through ONE entry point,
// sfb_debug_set(name, value) (include/sfb.h); the library reads NO environment variable, so a stale environment cannot
// steer a production launch.  Every set is announced on stderr.
//
//   sparse kernel, launch shape   SFB_SP_GRID        cap on the resident workgroups (tests: a tiny grid forces time slicing)
//                                 SFB_SP_SLICE       iterations per time slice (0: one workgroup per item)
//                                 SFB_SP_PAUSE       iteration of the first launch's pause (launch in predicted order)
//                                 SFB_SP_PREDICT     0: a single time-sliced launch instead of the launch in predicted order
//                                 SFB_SP_LAT         0: the standard form of the kernel for the loop launch
//                                 SFB_SP_FORCE_LAT   1: the LAT form of the kernel for a whole launch
//                                 SFB_SP_POLISHERS   0: no polishers next to the loop launch (polish and report of everything in the finish launch)
//                                 SFB_SP_LAT_WAVES   cap on the waves of the LAT loop launch (default: what the chip holds, three per CU)
//                                 SFB_SP_PHASED      1: setup / ADMM loop / polish + report as three launches (profiling)
//                                 SFB_SP_LEAN_WAVES  busy waves above which the sweeps use masked non-temporal loads
//   dense kernels                 SFB_MID_GRID, SFB_MID_SLICE   32 < n + m <= 128: resident waves / checks per slice
//                                 SFB_QP4_MAX_WAVES  n + m <= 32: cap on the persistent grid
//                                 SFB_QP_DENSE_BIG   0: sizes beyond 128 through the sparse kernel
//   EKF                           SFB_EKF_PERSISTENT 0: the fused step as one tile per wave instead of persistent waves with the next tile's
//                                                    covariances requested straight into LDS
//   plan                          SFB_PLAN_UNITS     0: the supernodal engine of the numeric factorisation for every plan
//                                 SFB_PLAN_DEBUG     1: print segments, units and sweep schedules of a plan
//   MPC swarm                     SFB_MPC_TIMING     1: synchronise after every stage of a tick and print its wall time
#pragma once

namespace sfb {

// value of a knob set through sfb_debug_set, or nullptr (knobs.cpp)
const char *knob(const char *name);
// 0 = ok, 1 = unknown name; value == nullptr clears the knob
int knob_set(const char *name, const char *value);

}  // namespace sfb
