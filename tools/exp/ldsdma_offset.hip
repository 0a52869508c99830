// Probe: does the instruction offset of `buffer_load_dwordx4 ... offen offset:N lds` move the LDS destination as well as
// the global source?  (tail.hip issues the three 1 KB pieces of a W item with one M0 and offsets 0 / 1024 / 2048.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* W, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 64) reinterpret_cast<uint32_t*>(smem)[i] = 0xDEADBEEFu;
  __syncthreads();
  const uint64_t wb = (uint64_t)(uintptr_t)W;
  u32x4 rs;
  rs[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wb);
  rs[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wb >> 32)) & 0xFFFFu;
  rs[2] = 1 << 20;
  rs[3] = 0x00020000u;
  const uint32_t dst = (uint32_t)(uintptr_t)(lds_char*)(smem + 4096);
  const int so = __builtin_amdgcn_readfirstlane(0);
  const int vo = lane * 16;
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
               "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
               "buffer_load_dwordx4 %1, %2, %4 offen offset:1024 lds\n\t"
               "buffer_load_dwordx4 %1, %2, %4 offen offset:2048 lds\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(vo), "s"(rs), "s"(dst), "s"(so) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = reinterpret_cast<uint32_t*>(smem)[i];
}
int main() {
  std::vector<uint32_t> h(4096), o(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  uint32_t *d, *od;
  hipMalloc(&d, 16384); hipMalloc(&od, 16384);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d, od);
  hipMemcpy(o.data(), od, 16384, hipMemcpyDeviceToHost);
  // expected if the offset moves both sides: LDS words [1024 + i] = i for i < 768
  int ok = 1;
  for (int i = 0; i < 768; ++i) ok &= (o[1024 + i] == (uint32_t)i);
  printf("offset moves LDS destination and global source together: %s\n", ok ? "yes" : "NO");
  if (!ok) for (int p = 0; p < 3; ++p) printf("piece %d: lds[%d]=%u lds[%d]=%u\n", p, 1024 + 256 * p, o[1024 + 256 * p], 1024, o[1024]);
  return ok ? 0 : 1;
}
