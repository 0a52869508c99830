// Experiment (not part of the library): gather 16 feature rows of 128 B with LDS-DMA (buffer_load_dwordx4 ... lds) in
// full-line pieces (8 lanes per row, 8 rows per instruction), XOR swizzle on the SOURCE chunk, read back as MFMA operand
// fragments (lane = (row, g): chunks g and 4+g).  Checks: data lands where expected, out-of-bounds rows (-1) give zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, int nbytes, const int* idx, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* my = smem + wave * 4096;
  for (int i = lane; i < 1024; i += 64) reinterpret_cast<float*>(my)[i] = -777.f;      // poison: stale data would show
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, nbytes, 0x00020000);
  int row = idx[(blockIdx.x * 4 + wave) * 16 + (lane >> 3)];
  int row2 = idx[(blockIdx.x * 4 + wave) * 16 + 8 + (lane >> 3)];
  uint32_t off = (uint32_t)row * 128u + (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
  uint32_t off2 = (uint32_t)row2 * 128u + (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my), 16, (int)off, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + 1024), 16, (int)off2, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int l15 = lane & 15, g = lane >> 4;
  f32x4 a0 = *reinterpret_cast<f32x4*>(my + l15 * 128 + ((g ^ (l15 & 7)) * 16));
  f32x4 a1 = *reinterpret_cast<f32x4*>(my + l15 * 128 + (((4 + g) ^ (l15 & 7)) * 16));
  f32x4* o = reinterpret_cast<f32x4*>(out + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 8);
  o[0] = a0; o[1] = a1;
}
int main() {
  const int R = 1000, NB = 8;
  std::vector<float> in(R * 32);
  for (int i = 0; i < R * 32; ++i) in[i] = (float)i;
  std::vector<int> idx(NB * 4 * 16);
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = (i % 5 == 3) ? -1 : (int)((i * 37) % R);
  float *din, *dout; int* didx;
  (void)hipMalloc(&din, in.size() * 4); (void)hipMalloc(&dout, NB * 4 * 64 * 8 * 4); (void)hipMalloc(&didx, idx.size() * 4);
  (void)hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(NB), dim3(256), 16384, 0, din, R * 32 * 4, didx, dout);
  std::vector<float> out(NB * 4 * 64 * 8);
  (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < NB * 4; ++w)
    for (int lane = 0; lane < 64; ++lane) {
      int l15 = lane & 15, g = lane >> 4;
      int row = idx[w * 16 + l15];
      for (int tt = 0; tt < 2; ++tt)
        for (int u = 0; u < 4; ++u) {
          float want = row < 0 ? 0.f : in[row * 32 + 16 * tt + 4 * g + u];
          float got = out[((size_t)w * 64 + lane) * 8 + tt * 4 + u];
          if (want != got) { if (bad < 10) printf("w %d lane %d tt %d u %d row %d want %f got %f\n", w, lane, tt, u, row, want, got); ++bad; }
        }
    }
  printf("bad = %d of %zu\n", bad, out.size());
  return bad != 0;
}
