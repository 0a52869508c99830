// Experiment: sustained rate of v_mfma_f32_16x16x4_f32 (the exact-fp32 MFMA the sparse conv uses) on the whole chip.
// Every wave issues back-to-back MFMAs on two alternating accumulators (the conv kernel's order); 1, 2 or 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    }
  }
  if (a0[0] + a1[0] == 12345.678f) out[threadIdx.x] = a0[1] + a1[2];
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;                       // 64000 MFMAs per wave
  for (int wps : {1, 2, 4}) {                   // waves per SIMD: blocks of 256 threads (4 waves, one per SIMD) x wps per CU
    const int grid = 256 * wps;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double mfma_per_simd = (double)iters * 16 * wps;
    const double ns_per_mfma = best * 1e6 / mfma_per_simd;
    const double tflops = (double)grid * 4 * iters * 16 * 2048 / (best * 1e-3) / 1e12;
    printf("waves/SIMD %d: %.3f ms, %.2f ns per MFMA per SIMD (32 cycles @ 2.4 GHz = 13.33 ns, @ 2.1 GHz = 15.24 ns), %.1f TFLOP/s\n",
           wps, best, ns_per_mfma, tflops);
  }
  return 0;
}
