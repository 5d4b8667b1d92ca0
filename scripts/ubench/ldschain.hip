// Micro-benchmark: dependent LDS read-modify-write chain of the sparse sweeps:
//   t[r_s] = fma(c, t[p_s], t[r_s]),  with p_{s+1} = r_s for lane 0 (true dependency through LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template<int UNROLL> __global__ void k(const int* idx, double* out, long long* cyc, int steps, int mode) {
  __shared__ double t[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) t[i] = 1.0 + i * 1e-3;
  __syncthreads();
  const int* my = idx + lane;
  long long t0 = __builtin_readcyclecounter();
  if (mode == 0) {       // indices from global memory each step (L2-resident), no prefetch distance
    for (int s = 0; s < steps; ++s) {
      const unsigned pk = (unsigned)my[s * 64];
      const int r = pk & 0xFFFF, p = pk >> 16;
      t[r] = fma(-1e-3, t[p], t[r]);
    }
  } else {               // indices preloaded in registers (UNROLL steps), pure LDS chain
    for (int s0 = 0; s0 < steps; s0 += UNROLL) {
      unsigned pk[UNROLL];
#pragma unroll
      for (int d = 0; d < UNROLL; ++d) pk[d] = (unsigned)my[(s0 + d) * 64];
#pragma unroll
      for (int d = 0; d < UNROLL; ++d) { const int r = pk[d] & 0xFFFF, p = pk[d] >> 16; t[r] = fma(-1e-3, t[p], t[r]); }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  out[blockIdx.x * 64 + lane] = t[lane];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int steps = 4096;
  std::vector<int> h(steps * 64);
  // lane l at step s: target r = (s*7 + l) % 1024 + 1024*(l&0)  ; pivot = previous step's target of lane 0 (dependency)
  int prev = 0;
  for (int s = 0; s < steps; ++s) {
    for (int l = 0; l < 64; ++l) { int r = (s * 67 + l) % 2048; h[s * 64 + l] = r | (prev << 16); }
    prev = (s * 67 + 0) % 2048;
  }
  int* d; double* o; long long* c;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 8 * 4096); hipMalloc(&c, 8 * 4096);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) for (int blocks : {1, 256 * 8}) {
    hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(64), 0, 0, d, o, c, steps, mode); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(64), 0, 0, d, o, c, steps, mode); hipDeviceSynchronize();
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("mode %d blocks %5d : %.1f cycles/step\n", mode, blocks, (double)cy / steps);
  }
  return 0;
}
