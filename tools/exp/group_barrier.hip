// Micro-benchmark for the resident tail kernel: what does a barrier among the workgroups that share one
// sample (or the whole grid) cost on MI355X?  Each workgroup runs `iters` rounds of: a little LDS work, arrive (device-scope
// atomic add on the group's counter), spin until everybody of the round arrived.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/group_barrier tools/exp/group_barrier.hip && /tmp/group_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counters, int group_size, int iters, int payload, float* sink) {
  const int group = blockIdx.x / group_size;
  unsigned* ctr = counters + group * 32;      // one 128-byte line per group
  float acc = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    for (int p = 0; p < payload; ++p) acc = acc * 1.0001f + 0.5f;
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(it + 1) * (unsigned)group_size;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (acc == -1.f) sink[0] = acc;
}

int main() {
  unsigned* ctr; float* sink;
  CK(hipMalloc(&ctr, 4096 * 128)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 200;
  struct Cfg { int groups, gsize; };
  const Cfg cfgs[] = {{16, 8}, {16, 16}, {64, 8}, {1, 128}, {1, 256}, {1, 512}, {16, 4}, {16, 1}};
  for (const Cfg& c : cfgs) {
    for (int payload : {0, 2000}) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(ctr, 0, 4096 * 128));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(barrier_kernel, dim3(c.groups * c.gsize), dim3(256), 0, 0, ctr, c.gsize, iters, payload, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("groups %3d x %3d workgroups, payload %4d: %8.2f us per launch, %6.3f us per round\n", c.groups, c.gsize, payload, best * 1e3, best * 1e3 / iters);
    }
  }
  // four instances in flight on four streams (what the bench does): 16 groups x 8 each
  hipStream_t st[4];
  for (auto& s : st) CK(hipStreamCreate(&s));
  CK(hipMemset(ctr, 0, 4096 * 128));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(barrier_kernel, dim3(128), dim3(256), 0, st[i], ctr + i * 1024 * 8, 8, iters, 2000, sink);
  for (auto& s : st) CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("4 instances x (16 groups x 8) on 4 streams, payload 2000: %8.2f us total, %6.3f us per round\n", ms * 1e3, ms * 1e3 / iters);
  return 0;
}
