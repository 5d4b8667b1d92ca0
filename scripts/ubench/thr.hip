// Micro-benchmark: THROUGHPUT (SIMD cycles per step at a given occupancy) of the dependent DPP
// row_newbcast fmac chain of the dense QP sweeps, against the plain FP64 fma rate.  Occupancy is set
// with the dynamic LDS size (160 KB per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template<int J, int NOP> __device__ __forceinline__ void fmac_dpp(double &t, double negL)
{
  if constexpr (NOP == 2)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(negL), "n"(J));
  else if constexpr (NOP == 1)
    asm volatile("s_nop 0\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(negL), "n"(J));
  else
    asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(negL), "n"(J));
}

template<int J, int NOP, int CH> struct Chain {
  static __device__ __forceinline__ void run(double (&t)[CH], const double (&L)[16])
  {
    if constexpr (J < 16) {
#pragma unroll
      for (int c = 0; c < CH; ++c) fmac_dpp<J, NOP>(t[c], L[J]);
      Chain<J + 1, NOP, CH>::run(t, L);
    }
  }
};

// MODE 0: plain dependent v_fma_f64 chain; MODE 1: dpp chains (CH interleaved chains, NOP wait-state variant)
template<int MODE, int NOP, int CH> __global__ void __launch_bounds__(64) k(const double *in, double *out, int reps)
{
  extern __shared__ double sm[];
  const int lane = threadIdx.x;
  double L[16];
  for (int j = 0; j < 16; ++j) L[j] = in[64 + lane * 16 + j];
  double t[CH];
  for (int c = 0; c < CH; ++c) t[c] = in[lane] + c;
  for (int r = 0; r < reps; ++r) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int c = 0; c < CH; ++c) t[c] = fma(L[j], t[c], t[c]);
    } else {
      Chain<0, NOP, CH>::run(t, L);
    }
  }
  double s = 0;
  for (int c = 0; c < CH; ++c) s += t[c];
  if (s == 123.456) sm[lane] = s;
  out[(size_t)blockIdx.x * 64 + lane] = s;
}

int main()
{
  const int reps = 2000;
  std::vector<double> h(64 + 64 * 16);
  srand(1);
  for (auto &v : h) v = (rand() / (double)RAND_MAX - 0.5) * 0.1;
  double *din, *dout;
  const int blocks = 256 * 32 * 4;
  hipMalloc(&din, h.size() * 8);
  hipMalloc(&dout, (size_t)blocks * 64 * 8);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  auto run = [&](auto kern, const char *name, int ch, int wavesPerSimd) {
    const size_t lds = 160 * 1024 / (4 * wavesPerSimd) - 256;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, din, dout, reps);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, din, dout, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double steps_per_simd = (double)blocks * reps * 16.0 * ch / 1024.0;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %6.2f SIMD-cycles/step @2.4GHz\n", name, wavesPerSimd, ms,
           ms * 1e-3 * 2.4e9 / steps_per_simd);
  };
  for (int w : {1, 2, 4, 8}) {
    run(k<0, 0, 1>, "fma chain", 1, w);
    run(k<0, 0, 4>, "fma 4 chains", 4, w);
    run(k<1, 2, 1>, "dpp s_nop1 1 chain", 1, w);
    run(k<1, 1, 1>, "dpp s_nop0 1 chain (timing only)", 1, w);
    run(k<1, 0, 1>, "dpp no nop 1 chain (timing only)", 1, w);
    run(k<1, 1, 2>, "dpp s_nop0 2 chains", 2, w);
    run(k<1, 0, 2>, "dpp no nop 2 chains", 2, w);
    run(k<1, 0, 3>, "dpp no nop 3 chains", 3, w);
    run(k<1, 0, 4>, "dpp no nop 4 chains", 4, w);
  }
  return 0;
}
