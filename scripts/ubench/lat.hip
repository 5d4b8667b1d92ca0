// Micro-benchmark: latency of the dependent "broadcast pivot + fma" step used by the triangular
// sweeps of the dense QP kernel, in several forms.  One wave, s_memtime around unrolled chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__device__ __forceinline__ double bcast(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src); hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}

#ifndef NOPSTR
#define NOPSTR "s_nop 1\n\t"
#endif
template<int J> __device__ __forceinline__ void fmac_dpp(double &t, double negL) {
  asm volatile(NOPSTR "v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
               : "+v"(t) : "v"(negL), "n"(J));
}

template<int MODE> __global__ void k(const double* in, double* out, long long* cyc, int reps) {
  const int lane = threadIdx.x;
  double L[16];
  for (int j = 0; j < 16; ++j) L[j] = in[64 + lane * 16 + j];
  double t = in[lane];
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    if constexpr (MODE == 0) {        // plain dependent fma chain
#pragma unroll
      for (int j = 0; j < 16; ++j) t = fma(L[j], t, t);
    } else if constexpr (MODE == 1) { // readlane broadcast + fma
#pragma unroll
      for (int j = 0; j < 16; ++j) { double tj = bcast(t, j); t = fma(L[j], tj, t); }
    } else if constexpr (MODE == 2) { // fused DPP row_newbcast fmac
      fmac_dpp<0>(t, L[0]); fmac_dpp<1>(t, L[1]); fmac_dpp<2>(t, L[2]); fmac_dpp<3>(t, L[3]);
      fmac_dpp<4>(t, L[4]); fmac_dpp<5>(t, L[5]); fmac_dpp<6>(t, L[6]); fmac_dpp<7>(t, L[7]);
      fmac_dpp<8>(t, L[8]); fmac_dpp<9>(t, L[9]); fmac_dpp<10>(t, L[10]); fmac_dpp<11>(t, L[11]);
      fmac_dpp<12>(t, L[12]); fmac_dpp<13>(t, L[13]); fmac_dpp<14>(t, L[14]); fmac_dpp<15>(t, L[15]);
    } else if constexpr (MODE == 3) { // ds_bpermute broadcast + fma
#pragma unroll
      for (int j = 0; j < 16; ++j) { double tj = __shfl(t, j); t = fma(L[j], tj, t); }
    } else if constexpr (MODE == 4) { // f32 readlane + fma for comparison
      float tf = (float)t;
#pragma unroll
      for (int j = 0; j < 16; ++j) { float tj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tf), j)); tf = fmaf((float)L[j], tj, tf); }
      t = tf;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = t;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int reps = 1000;
  std::vector<double> h(64 + 64 * 16);
  srand(1);
  for (auto &v : h) v = (rand() / (double)RAND_MAX - 0.5) * 0.1;
  double *din, *dout; long long *dc;
  hipMalloc(&din, h.size() * 8); hipMalloc(&dout, 64 * 8 * 8192); hipMalloc(&dc, 8 * 8192);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  // CPU reference for the DPP form (MODE 2 == MODE 1 semantics within a 16-lane row) -- 1 rep
  auto run = [&](auto kern, const char *name, int blocks) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, din, dout, dc, reps);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, din, dout, dc, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    double o[64]; hipMemcpy(o, dout, 64 * 8, hipMemcpyDeviceToHost);
    printf("%-28s blocks=%5d  %7.2f ticks/step (s_memtime, 100MHz?)  wall %8.3f ms -> %7.2f ns/step/wave  out0=%g out17=%g\n",
           name, blocks, (double)c / (reps * 16.0), ms, ms * 1e6 / (reps * 16.0), o[0], o[17]);
  };
  for (int blocks : {1, 4096, 8192}) {
    run(k<0>, "fma chain", blocks);
    run(k<1>, "readlane+fma", blocks);
    run(k<2>, "fmac_dpp newbcast", blocks);
    run(k<3>, "shfl(bpermute)+fma", blocks);
    run(k<4>, "f32 readlane+fma", blocks);
  }
  // correctness of MODE 2 vs MODE 1 for rows (reps=1)
  {
    double o1[64], o2[64];
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, din, dout, dc, 1); hipMemcpy(o1, dout, 512, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, din, dout, dc, 1); hipMemcpy(o2, dout, 512, hipMemcpyDeviceToHost);
    int same = 0; for (int i = 0; i < 16; ++i) same += (o1[i] == o2[i]);
    printf("row0 lanes bit-identical readlane vs dpp: %d/16\n", same);
  }
  return 0;
}
