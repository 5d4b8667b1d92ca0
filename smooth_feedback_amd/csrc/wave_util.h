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

// sum over all 64 lanes in a fixed tree order, result uniform (for quantities whose summation order is free)
__device__ __forceinline__ double wave_sum(double v)
{
  v = v + dpp_mov<0x128>(v);  // row_ror:8
  v = v + dpp_mov<0x124>(v);  // row_ror:4
  v = v + dpp_mov<0x122>(v);  // row_ror:2
  v = v + dpp_mov<0x121>(v);  // row_ror:1  -> every lane of a 16-lane row holds the row sum
  const double a = lane_bcast(v, 0), b = lane_bcast(v, 16), c = lane_bcast(v, 32), d = lane_bcast(v, 48);
  return (a + b) + (c + d);
}

// ---- 16-lane-row primitives (DPP on DP-ALU ops supports row_newbcast only, gfx90a+) ----
// t += bcast_row(t, J) * negl on the rows selected by ROWMASK (other rows keep t).  The pivot of
// an elimination step never leaves the VALU: no v_readlane -> SGPR -> VALU round trip.
// "VALU writes VGPR -> DPP reads it" needs 2 wait states and the assembler/compiler does not look
// inside the asm, so the self-dependent form carries its own `s_nop 1` (18 cycles per dependent
// step; without it the chain runs at 10 cycles/step but k == 32 problems come out wrong on
// gfx950 -- measured, tests/test_qp_dense_gpu.py::test_other_sizes[16-16]).
template<int J, int ROWMASK>
__device__ __forceinline__ void fmac_rowbcast_self(double &t, const double negl)
{
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:%3 bank_mask:0xf"
               : "+v"(t)
               : "v"(negl), "n"(J), "n"(ROWMASK));
}
// t += bcast_row(x, J) * negl on the rows selected by ROWMASK.  Although x is not written by the
// chain, dropping the wait states here also breaks k == 32 problems (the accumulator written by
// the previous DPP op is read early), so every hand-written DPP op carries `s_nop 1`.
template<int J, int ROWMASK>
__device__ __forceinline__ void fmac_rowbcast(double &t, const double x, const double negl)
{
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:%4 bank_mask:0xf"
               : "+v"(t)
               : "v"(x), "v"(negl), "n"(J), "n"(ROWMASK));
}

// The hazard recogniser does not look inside inline asm: after a hand-written DPP fmac the next
// cross-lane reader of `t` emitted by the compiler (v_permlane*_swap, v_readlane) needs its two
// wait states spelled out.  Threading `t` through the asm pins the order.
__device__ __forceinline__ void cross_lane_fence(double &t) { asm volatile("s_nop 1" : "+v"(t)); }

// ... and after a compiler-emitted v_permlane*_swap the first hand-written DPP reader of its result
// needs the swap's own wait states (not visible to the compiler either); 4 are spelled here.
__device__ __forceinline__ void swap_settle(double &a, double &b) { asm volatile("s_nop 3" : "+v"(a), "+v"(b)); }

// rows of 16 lanes r0..r3:  even_dup = [r0, r0, r2, r2],  odd_dup = [r1, r1, r3, r3]
__device__ __forceinline__ void row_swap16(const double t, double &even_dup, double &odd_dup)
{
  const unsigned lo = (unsigned)__double2loint(t), hi = (unsigned)__double2hiint(t);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  even_dup = __hiloint2double((int)b[0], (int)a[0]);
  odd_dup  = __hiloint2double((int)b[1], (int)a[1]);
  swap_settle(even_dup, odd_dup);
}
// halves of 32 lanes:  low_dup = [lo, lo],  high_dup = [hi, hi]
__device__ __forceinline__ void half_swap32(const double t, double &low_dup, double &high_dup)
{
  const unsigned lo = (unsigned)__double2loint(t), hi = (unsigned)__double2hiint(t);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  low_dup  = __hiloint2double((int)b[0], (int)a[0]);
  high_dup = __hiloint2double((int)b[1], (int)a[1]);
  swap_settle(low_dup, high_dup);
}

// QPSolverParams::max_time (qp_solver.hpp:504-507): the reference compares a steady clock with the time its solve
// started, at every stopping check that leaves the status open.  Device equivalent: the constant 100 MHz wall clock.
__device__ __forceinline__ bool max_time_exceeded(const long long max_time_ns, const unsigned long long t0_ticks)
{
  return max_time_ns >= 0 && (long long)((wall_clock64() - t0_ticks) * 10ull) > max_time_ns;
}

__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p ? 1 : 0); }

__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// All LDS traffic of a problem comes from its own single-wave workgroup; this orders it for the
// compiler (and is a cheap s_barrier for a 64-thread block).
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// LDS-only ordering for a single-wave workgroup: the hardware executes one wave's LDS operations in
// order, so phases that communicate across lanes through LDS only need the COMPILER to keep them in
// program order.  No s_barrier and, unlike wave_sync(), no s_waitcnt vmcnt(0) that would drain the
// wave's outstanding global stores.
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

}  // namespace sfb
