// Experiment: 16-MFMA blocks (the conv kernel's item) separated by F filler instructions (scalar / vector / LDS),
// 4 waves per SIMD on the whole chip: how much of the MFMA pipe survives the bookkeeping between the blocks?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int seed) {
  __shared__ int lds[4096];
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  int s = seed, v = threadIdx.x;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 || MODE == 3) {            // ~30 scalar ops
#pragma unroll
      for (int u = 0; u < 15; ++u) { s = s * 1103515245 + 12345; s ^= (s >> 7); }
      asm volatile("" : "+s"(s));
    }
    if (MODE == 2 || MODE == 3) {            // ~12 vector ops + an LDS read with its wait (the neighbour-row lookup)
#pragma unroll
      for (int u = 0; u < 6; ++u) { v = v * 5 + 1; v ^= (v >> 3); }
      v += lds[(v & 1023)];
      x += (float)(v & 1) * 1e-9f;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (a0[0] + a1[0] == 12345.678f || s == 0x7fffffff) out[threadIdx.x] = a0[1] + a1[2] + v;
}
template <int MODE> void run(float* out, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000, wps = 4, grid = 256 * wps;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 1);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  printf("%-28s %.3f ms, %.2f ns per MFMA per SIMD (floor 13.3-13.5)\n", name, best, best * 1e6 / ((double)iters * 16 * wps));
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  run<0>(out, "MFMA only");
  run<1>(out, "+30 scalar per block");
  run<2>(out, "+12 vector + LDS per block");
  run<3>(out, "+both");
  return 0;
}
