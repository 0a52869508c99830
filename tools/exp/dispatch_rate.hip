// Experiment: how fast does the chip start workgroups?  Trivial kernel (one store per workgroup), grids of 3200 / 12800 /
// 51200 workgroups of 64 / 256 threads, with 0 / 8 / 33 KB of dynamic LDS and ~100 VGPRs worth of launch bounds.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int spin) {
  extern __shared__ float lds[];
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { lds[0] = x; out[blockIdx.x & 1023] = lds[0]; }
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int spin : {0, 2000}) for (int threads : {64, 256}) for (int lds : {0, 8192, 33792}) for (int grid : {3200, 12800, 51200}) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, out, spin);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, out, spin);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("spin %4d threads %3d lds %5d grid %5d: %7.1f us  -> %.1f WG/us (%.0f ns per WG per XCD)\n", spin, threads, lds, grid, best * 1e3,
           grid / (best * 1e3), best * 1e6 / (grid / 8.0));
  }
  return 0;
}
