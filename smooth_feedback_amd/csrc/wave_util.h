// Wavefront-level helpers for gfx950 (wave64).  One problem per wavefront: lanes exchange data with
// v_readlane (uniform source lane -> SGPR broadcast), DPP row rotations and 64-bit ballots; LDS is
// only used for data that has to be indexed dynamically.
#pragma once
#include <hip/hip_runtime.h>

namespace sfb {

constexpr int kWave = 64;

// Broadcast lane `src` (wave-uniform) of a double to every lane.
__device__ __forceinline__ double lane_bcast(double v, int src)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo     = __builtin_amdgcn_readlane(lo, src);
  hi     = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}

template<int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// max over all 64 lanes, result uniform.  NaNs are ignored (fmax), like the reference's running
// `std::max(r, |v|)` that starts from 0.  Inactive problem lanes must pass a neutral value.
__device__ __forceinline__ double wave_max(double v)
{
  v = fmax(v, dpp_mov<0x128>(v));  // row_ror:8
  v = fmax(v, dpp_mov<0x124>(v));  // row_ror:4
  v = fmax(v, dpp_mov<0x122>(v));  // row_ror:2
  v = fmax(v, dpp_mov<0x121>(v));  // row_ror:1  -> every lane of a 16-lane row holds the row max
  const double a = lane_bcast(v, 0), b = lane_bcast(v, 16), c = lane_bcast(v, 32), d = lane_bcast(v, 48);
  return fmax(fmax(a, b), fmax(c, d));
}

__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p ? 1 : 0); }

__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// All LDS traffic of a problem comes from its own single-wave workgroup; this orders it for the
// compiler (and is a cheap s_barrier for a 64-thread block).
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

}  // namespace sfb
