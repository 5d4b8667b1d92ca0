// Micro-benchmark: issue cost of instruction mixes for a LONE wave per SIMD (and 2, 4 waves): what a sweep step of the
// block engine (sweep_rows.h) costs -- SALU exec writes, DPP FP64 fmacs (dependent chain + independent riders), LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template<int MODE> __global__ void __launch_bounds__(64) k(const double *in, double *out, int reps)
{
  extern __shared__ double sm[];
  const int lane = threadIdx.x;
  double tp = in[lane], t0 = in[lane] + 1, t1 = in[lane] + 2, l0 = in[64 + lane], l1 = in[128 + lane], l2 = in[192 + lane];
  unsigned long long m = 0xFFFEFFFEFFFEFFFEull;
  asm volatile("" : "+s"(m));
  for (int e = lane; e < 1024; e += 64) sm[e] = in[e % 256];
  __syncthreads();
  const unsigned addr = (unsigned)(lane * 8);
  double a = 0, b = 0;
  for (int r = 0; r < reps; ++r) {
    if constexpr (MODE == 0) {  // 16 x [s_mov exec; chain fmac; s_mov exec]
      asm volatile(REP16("s_mov_b64 exec, %1\n\tv_fmac_f64_dpp %0, %0, -%2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1\n\t")
                   : "+v"(tp) : "s"(m), "v"(l0));
    } else if constexpr (MODE == 1) {  // 16 x [s_mov; chain; s_mov; rider; rider]
      asm volatile(REP16("s_mov_b64 exec, %3\n\tv_fmac_f64_dpp %0, %0, -%4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1\n\t"
                         "v_fmac_f64_dpp %1, %0, -%5 row_newbcast:3 row_mask:0xe bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %0, -%6 row_newbcast:3 row_mask:0x3 bank_mask:0xf\n\t")
                   : "+v"(tp), "+v"(t0), "+v"(t1) : "s"(m), "v"(l0), "v"(l1), "v"(l2));
    } else if constexpr (MODE == 2) {  // 16 x [chain; rider; rider] no exec (timing only)
      asm volatile(REP16("v_fmac_f64_dpp %0, %0, -%3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, -%4 row_newbcast:3 row_mask:0xe bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %0, -%5 row_newbcast:3 row_mask:0x3 bank_mask:0xf\n\t")
                   : "+v"(tp), "+v"(t0), "+v"(t1) : "v"(l0), "v"(l1), "v"(l2));
    } else if constexpr (MODE == 3) {  // 48 independent-ish riders (two accumulators alternate)
      asm volatile(REP16("v_fmac_f64_dpp %1, %0, -%3 row_newbcast:3 row_mask:0xe bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %0, -%4 row_newbcast:3 row_mask:0x3 bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, -%5 row_newbcast:3 row_mask:0xe bank_mask:0xf\n\t")
                   : "+v"(tp), "+v"(t0), "+v"(t1) : "v"(l0), "v"(l1), "v"(l2));
    } else if constexpr (MODE == 4) {  // 48 SALU exec writes
      asm volatile(REP16("s_mov_b64 exec, %0\n\ts_mov_b64 exec, -1\n\ts_mov_b64 exec, -1\n\t") : : "s"(m));
    } else if constexpr (MODE == 5) {  // 48 ds_read_b64 (3 per group, one wait per 48)
      asm volatile(REP16("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512\n\tds_read_b64 %0, %2 offset:1024\n\t") "s_waitcnt lgkmcnt(0)"
                   : "=v"(a), "=v"(b) : "v"(addr));
    } else if constexpr (MODE == 6) {  // 16 x [s_mov; chain; s_mov; rider; rider; ds_read; ds_read] one wait at the end
      asm volatile(REP16("s_mov_b64 exec, %3\n\tv_fmac_f64_dpp %0, %0, -%4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1\n\t"
                         "v_fmac_f64_dpp %1, %0, -%5 row_newbcast:3 row_mask:0xe bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %0, -%6 row_newbcast:3 row_mask:0x3 bank_mask:0xf\n\t"
                         "ds_read_b64 %7, %9\n\tds_read_b64 %8, %9 offset:512\n\t") "s_waitcnt lgkmcnt(0)"
                   : "+v"(tp), "+v"(t0), "+v"(t1) : "s"(m), "v"(l0), "v"(l1), "v"(l2), "v"(a), "v"(b), "v"(addr));
    } else if constexpr (MODE == 7) {  // 16 x [s_mov narrow; chain] then 32 riders (riders batched after the chain)
      asm volatile(REP16("s_mov_b64 exec, %3\n\tv_fmac_f64_dpp %0, %0, -%4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
                   "s_mov_b64 exec, -1\n\t"
                   REP16("v_fmac_f64_dpp %1, %0, -%5 row_newbcast:3 row_mask:0xe bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %0, -%6 row_newbcast:3 row_mask:0x3 bank_mask:0xf\n\t")
                   : "+v"(tp), "+v"(t0), "+v"(t1) : "s"(m), "v"(l0), "v"(l1), "v"(l2));
    } else if constexpr (MODE == 8) {  // 48 plain v_fma_f64, 3 accumulators
      asm volatile(REP16("v_fma_f64 %0, %3, %4, %0\n\tv_fma_f64 %1, %3, %4, %1\n\tv_fma_f64 %2, %3, %4, %2\n\t")
                   : "+v"(tp), "+v"(t0), "+v"(t1) : "v"(l0), "v"(l1));
    } else if constexpr (MODE == 9) {  // 48 s_nop 0
      asm volatile(REP16("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"));
    } else if constexpr (MODE == 10) {  // 48 v_mov_b32
      unsigned w = addr;
      asm volatile(REP16("v_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\t") : "+v"(w) : "v"(addr));
      a += w;
    } else if constexpr (MODE == 11) {  // 16 x [s_mov; chain; s_mov; ds_read] : chain with one LDS read riding
      asm volatile(REP16("s_mov_b64 exec, %1\n\tv_fmac_f64_dpp %0, %0, -%2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1\n\t"
                         "ds_read_b64 %3, %4\n\t") "s_waitcnt lgkmcnt(0)"
                   : "+v"(tp) : "s"(m), "v"(l0), "v"(a), "v"(addr));
    }
  }
  double s = tp + t0 + t1 + a + b;
  out[(size_t)blockIdx.x * 64 + lane] = s;
}

int main()
{
  const int reps = 500;
  std::vector<double> h(1024);
  srand(1);
  for (auto &v : h) v = (rand() / (double)RAND_MAX - 0.5) * 0.1;
  double *din, *dout;
  const int blocks = 256 * 8;
  hipMalloc(&din, h.size() * 8);
  hipMalloc(&dout, (size_t)blocks * 4 * 64 * 8);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  auto run = [&](auto kern, const char *name, int instrs, int w) {
    const size_t lds = 160 * 1024 / (4 * w) - 256;
    const int nb = 256 * 4 * w;  // exactly one resident set
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64), lds, 0, din, dout, reps);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64), lds, 0, din, dout, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-62s waves/SIMD=%d %8.3f ms  %6.2f cycles per instruction per wave, %6.2f SIMD-cycles per instruction\n", name, w, ms,
           ms * 1e-3 * 2.4e9 / ((double)reps * instrs), ms * 1e-3 * 2.4e9 / ((double)reps * instrs * w));
  };
  for (int w : {1, 2, 4}) {
    run(k<0>, "16 x [s_mov exec; chain fmac; s_mov exec]", 48, w);
    run(k<1>, "16 x [s_mov; chain; s_mov; rider; rider]", 80, w);
    run(k<2>, "16 x [chain; rider; rider] (no exec)", 48, w);
    run(k<3>, "48 riders (2 accumulators)", 48, w);
    run(k<4>, "48 s_mov exec", 48, w);
    run(k<5>, "48 ds_read_b64 + 1 wait", 49, w);
    run(k<6>, "16 x [s_mov; chain; s_mov; rider; rider; ds_read; ds_read]", 113, w);
    run(k<7>, "16 x [s_mov; chain] + 32 riders", 65, w);
    run(k<8>, "48 v_fma_f64 (3 accumulators)", 48, w);
    run(k<9>, "48 s_nop 0", 48, w);
    run(k<10>, "48 v_mov_b32", 48, w);
    run(k<11>, "16 x [s_mov; chain; s_mov; ds_read]", 65, w);
  }
  return 0;
}
