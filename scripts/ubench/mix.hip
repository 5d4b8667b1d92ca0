// What the memory system of this box delivers for the EKF kernel's traffic MIX (1 104 B read + 336 B written per item =
// 77 % reads) with no arithmetic in the way: every wave streams 64 items' worth of contiguous input
// (138 doubles per item, 16-byte non-temporal loads, all requests of a tile in flight together) and writes 42 doubles per
// item (16-byte non-temporal stores); 4 waves per CU x several tiles in flight like ekf_kernel (one wave per workgroup),
// and a variant with 8 waves per CU.  Prints TB/s (read + write) for both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double vd2 __attribute__((ext_vector_type(2)));
template<int RD2, int WR2>  // 16-byte units per lane read / written per tile of 64 items
__global__ void __launch_bounds__(64) mix_kernel(const vd2 *__restrict__ in, vd2 *__restrict__ out, const long tiles)
{
  const int lane = threadIdx.x;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const vd2 *src = in + t * (long)RD2 * 64;
    vd2 v[RD2];
#pragma unroll
    for (int c = 0; c < RD2; ++c) v[c] = __builtin_nontemporal_load(&src[c * 64 + lane]);
    vd2 acc[WR2];
#pragma unroll
    for (int c = 0; c < WR2; ++c) acc[c] = vd2{0.0, 0.0};
#pragma unroll
    for (int c = 0; c < RD2; ++c) acc[c % WR2] += v[c];
    vd2 *dst = out + t * (long)WR2 * 64;
#pragma unroll
    for (int c = 0; c < WR2; ++c) __builtin_nontemporal_store(acc[c], &dst[c * 64 + lane]);
  }
}
int main()
{
  constexpr int RD2 = 69, WR2 = 21;  // 138 / 42 doubles per item
  const long items = 1l << 20, tiles = items / 64;
  vd2 *in, *out;
  hipMalloc(&in, tiles * RD2 * 64 * sizeof(vd2));
  hipMalloc(&out, tiles * WR2 * 64 * sizeof(vd2));
  hipMemset(in, 0, tiles * RD2 * 64 * sizeof(vd2));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)tiles * 64 * (RD2 + WR2) * 16;
  for (int grid : {(int)tiles, 256 * 4, 256 * 8, 256 * 16}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((mix_kernel<RD2, WR2>), dim3(grid), dim3(64), 0, 0, in, out, tiles);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((mix_kernel<RD2, WR2>), dim3(grid), dim3(64), 0, 0, in, out, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("mix 77/23 (1 104 B read + 336 B written per item, %ld items), grid %7d waves: %.3f ms -> %.2f TB/s = %.3f of 8 TB/s\n", items, grid, ms,
           bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
  }
  return 0;
}
