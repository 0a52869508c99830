/* ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-alone driver of oracle/egonn_cpu.c for a -fsanitize=address,undefined build
 * (sanitizer run of the host restatement; tests/test_oracle.py::test_c_oracle_under_sanitizers).
 * Input file (little endian): int64 n_points, int32 mode, float step[3], int32 n_k, int32 n_tensors, then per tensor
 * int64 count + floats, then n_points*3 floats.  Prints the level counts, the number of keypoints and simple checksums. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int egonn_cpu_compute_embedding_q(const float* points, int64_t n_points, int mode, const float* step,
                                  const float* const* weights, int n_weights, int n_k, float* out_global,
                                  int32_t* level_counts, int32_t* out_kp_coords, float* out_kp, float* out_desc,
                                  float* out_sigma, int n_threads);

static void need(size_t got, size_t want) {
  if (got != want) { fprintf(stderr, "short read\n"); exit(2); }
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int64_t n; int32_t mode, n_k, nt; float step[3];
  need(fread(&n, 8, 1, f), 1); need(fread(&mode, 4, 1, f), 1); need(fread(step, 4, 3, f), 3);
  need(fread(&n_k, 4, 1, f), 1); need(fread(&nt, 4, 1, f), 1);
  float** w = (float**)calloc((size_t)nt, sizeof(float*));
  for (int i = 0; i < nt; ++i) {
    int64_t c; need(fread(&c, 8, 1, f), 1);
    w[i] = (float*)malloc(sizeof(float) * (size_t)(c > 0 ? c : 1));
    need(fread(w[i], 4, (size_t)c, f), (size_t)c);
  }
  float* pts = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? 3 * n : 1));
  need(fread(pts, 4, (size_t)(3 * n), f), (size_t)(3 * n));
  fclose(f);
  float* g = (float*)calloc(256, 4);
  int32_t cnt[8] = {0};
  int32_t* kc = (int32_t*)calloc((size_t)n_k * 3, 4);
  float* kp = (float*)calloc((size_t)n_k * 3, 4);
  float* de = (float*)calloc((size_t)n_k * 128, 4);
  float* sg = (float*)calloc((size_t)n_k, 4);
  const int m = egonn_cpu_compute_embedding_q(pts, n, mode, step, (const float* const*)w, nt, n_k, g, cnt, kc, kp, de, sg, 2);
  double sgl = 0, sde = 0;
  for (int i = 0; i < 256; ++i) sgl += g[i];
  for (int i = 0; i < (m > 0 ? m : 0) * 128; ++i) sde += de[i];
  printf("m %d counts %d %d %d %d %d %d %d %d sum_global %.6e sum_desc %.6e\n", m, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5],
         cnt[6], cnt[7], sgl, sde);
  for (int i = 0; i < nt; ++i) free(w[i]);
  free(w); free(pts); free(g); free(kc); free(kp); free(de); free(sg);
  return m < 0 ? 1 : 0;
}
