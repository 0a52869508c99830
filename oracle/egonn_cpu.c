/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or called from the product path (egonn_amd/).
 *
 * C / OpenMP restatement of the reference's EgoNN descriptor extraction for ONE scan, Cartesian quantiser (the
 * configuration of BASELINE.json configs[1]) or polar quantiser (the reference's shipped configuration,
 * models/egonn.txt:3-5): quantise -> MinkGL.forward (eval mode) -> top-n_k keypoints.
 * It is the CPU baseline bench.py times next to the GPU path ("the build's own CPU oracle
 * (C++/OpenMP restatement, same graph, same clouds, fp32)") and a second, independently written checker: it is
 * validated against the numpy oracle (oracle/egonn_ref.py) and the reference-graph fixtures in tests/test_oracle.py.
 *
 * Follows (reference file:line):
 *   datasets/quantization.py:79-85   floor(pc / q) -> unique voxels                       (quantize)
 *   models/minkgl.py:136-153         MinkTrunk.forward: conv k5 + 7 x (conv k2s2, ECABasicBlock)
 *   layers/eca_block.py:21-36,56-73  ECALayer / ECABasicBlock.forward
 *   models/minkgl.py:46-60           MinkHead.forward (1x1 lateral, transposed conv onto cached coordinates)
 *   models/minkgl.py:175-225         KeypointRegressor / SigmaRegressor / DescriptorDecoder
 *   layers/pooling.py:82-86          GeM
 *   datasets/quantization.py:93-103  keypoint_position
 *   eval/evaluate.py:352-361         the n_k lowest-sigma keypoints, ascending (ties: Z-order key of the super-voxel)
 * MinkowskiEngine semantics per SURVEY.md Appendix A (kernel index x-fastest, even kernels non-centred, floor parents,
 * transposed conv reuses the strided map).  PARITY STATUS: as oracle/me_ops.py — sparse-conv primitive semantics
 * "parity unpinned" (MinkowskiEngine absent), graph pinned through the fixtures.
 *
 * Weights arrive as an array of pointers in the traversal order documented in oracle/egonn_cpu.py (WEIGHT_ORDER).
 * Parallelism: OpenMP over output rows of every layer.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NLEV 8
#define BIAS (1 << 20)

typedef struct {
  int n;
  int32_t* c;      /* [n][3] coordinates (multiples of 2^level) */
  uint64_t* key;   /* [n] sorted packed keys */
} Level;

static uint64_t pack(int32_t x, int32_t y, int32_t z) {
  return ((uint64_t)(uint32_t)(z + BIAS) << 42) | ((uint64_t)(uint32_t)(y + BIAS) << 21) | (uint64_t)(uint32_t)(x + BIAS);
}
static int cmp_u64(const void* a, const void* b) {
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
static int find(const Level* L, int32_t x, int32_t y, int32_t z) {
  const uint64_t k = pack(x, y, z);
  int lo = 0, hi = L->n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (L->key[mid] == k) return mid;
    if (L->key[mid] < k) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}
static int32_t floor_div_mul(int32_t c, int s) {   /* floor(c / s) * s */
  int32_t q = c / s;
  if ((c % s) && (c < 0)) --q;
  return q * s;
}
static void level_from_keys(Level* L, uint64_t* keys, int n) {
  qsort(keys, n, sizeof(uint64_t), cmp_u64);
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (i == 0 || keys[i] != keys[i - 1]) keys[m++] = keys[i];
  L->n = m;
  L->key = keys;
  L->c = (int32_t*)malloc((size_t)(m > 0 ? m : 1) * 3 * sizeof(int32_t));
  for (int i = 0; i < m; ++i) {
    L->c[3 * i + 0] = (int32_t)(keys[i] & 0x1FFFFF) - BIAS;
    L->c[3 * i + 1] = (int32_t)((keys[i] >> 21) & 0x1FFFFF) - BIAS;
    L->c[3 * i + 2] = (int32_t)((keys[i] >> 42) & 0x1FFFFF) - BIAS;
  }
}

/* out[o] = sum_k in[nbr[o][k]] @ W[k]   (W: [K][cin][cout]) */
static void conv_map(const float* in, int cin, const int* nbr, int K, const float* W, int cout, float* out, int n_out) {
#pragma omp parallel for schedule(static)
  for (int o = 0; o < n_out; ++o) {
    float* dst = out + (size_t)o * cout;
    for (int c = 0; c < cout; ++c) dst[c] = 0.f;
    for (int k = 0; k < K; ++k) {
      const int j = nbr[(size_t)o * K + k];
      if (j < 0) continue;
      const float* src = in + (size_t)j * cin;
      const float* w = W + (size_t)k * cin * cout;
      for (int ci = 0; ci < cin; ++ci) {
        const float a = src[ci];
        const float* wr = w + (size_t)ci * cout;
        for (int c = 0; c < cout; ++c) dst[c] += a * wr[c];
      }
    }
  }
}
/* rows @ W (+ bias): w_out_in = 0: W[cin][cout]; 1: W[cout][cin] (nn.Linear) */
static void dense(const float* in, int n, int cin, const float* W, int w_out_in, const float* bias, int cout, float* out) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < n; ++r) {
    const float* src = in + (size_t)r * cin;
    float* dst = out + (size_t)r * cout;
    for (int c = 0; c < cout; ++c) dst[c] = bias ? bias[c] : 0.f;
    if (!w_out_in) {
      for (int ci = 0; ci < cin; ++ci) {
        const float a = src[ci];
        const float* wr = W + (size_t)ci * cout;
        for (int c = 0; c < cout; ++c) dst[c] += a * wr[c];
      }
    } else {
      for (int c = 0; c < cout; ++c) {
        const float* wr = W + (size_t)c * cin;
        float s = 0.f;
        for (int ci = 0; ci < cin; ++ci) s += src[ci] * wr[ci];
        dst[c] += s;
      }
    }
  }
}
/* eval-mode BatchNorm1d (eps 1e-5) + optional ReLU, in place.  bn = {weight, bias, running_mean, running_var} */
static void bn_act(float* x, int n, int c, const float* const* bn, int relu) {
  float* sc = (float*)malloc(sizeof(float) * 2 * c);
  float* sh = sc + c;
  for (int i = 0; i < c; ++i) {
    sc[i] = bn[0][i] / sqrtf(bn[3][i] + 1e-5f);
    sh[i] = bn[1][i] - bn[2][i] * sc[i];
  }
#pragma omp parallel for schedule(static)
  for (int r = 0; r < n; ++r)
    for (int i = 0; i < c; ++i) {
      float v = x[(size_t)r * c + i] * sc[i] + sh[i];
      x[(size_t)r * c + i] = (relu && v < 0.f) ? 0.f : v;
    }
  free(sc);
}

typedef struct {
  const float* const* w;   /* weight pointers in WEIGHT_ORDER */
  int i;
} Weights;
static const float* nextw(Weights* W) { return W->w[W->i++]; }

static const int PLANES[7] = {32, 64, 64, 128, 128, 128, 128};

/* MinkHead.forward: 1x1 on the top level, then per level transposed conv (k2s2 onto the cached finer coordinates) +
 * 1x1 lateral.  feats[l]: trunk output of level l.  Returns malloc'ed (n[lo], ch). */
static float* head(Weights* W, const Level* lv, int* const* nbr8, float* const* feats, int lo, int hi, int ch) {
  float* y = (float*)malloc(sizeof(float) * (size_t)(lv[hi].n > 0 ? lv[hi].n : 1) * ch);
  dense(feats[hi], lv[hi].n, PLANES[hi - 1], nextw(W), 0, NULL, ch, y);            /* conv1x1[hi] */
  for (int l = hi - 1; l >= lo; --l) {
    const float* tk = nextw(W);                                                       /* tconv[l+1]: [8][ch][ch] */
    float* up = (float*)malloc(sizeof(float) * (size_t)(lv[l].n > 0 ? lv[l].n : 1) * ch);
    const int nl = lv[l].n;
    const Level* P = &lv[l + 1];
    const int s = 1 << l;
#pragma omp parallel for schedule(static)
    for (int f = 0; f < nl; ++f) {                     /* out[f] = in[parent(f)] @ W[slot(f)]   */
      const int32_t* c = lv[l].c + 3 * f;
      const int32_t px = floor_div_mul(c[0], 2 * s), py = floor_div_mul(c[1], 2 * s), pz = floor_div_mul(c[2], 2 * s);
      const int p = find(P, px, py, pz);
      const int slot = (c[0] - px) / s + 2 * ((c[1] - py) / s) + 4 * ((c[2] - pz) / s);
      const float* src = y + (size_t)p * ch;
      const float* w = tk + (size_t)slot * ch * ch;
      float* dst = up + (size_t)f * ch;
      for (int co = 0; co < ch; ++co) dst[co] = 0.f;
      for (int ci = 0; ci < ch; ++ci) {
        const float a = src[ci];
        for (int co = 0; co < ch; ++co) dst[co] += a * w[(size_t)ci * ch + co];
      }
    }
    free(y);
    y = up;
    const float* lat = nextw(W);                                                      /* conv1x1[l] */
    float* t = (float*)malloc(sizeof(float) * (size_t)(nl > 0 ? nl : 1) * ch);
    dense(feats[l], nl, PLANES[l - 1], lat, 0, NULL, ch, t);
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < (int64_t)nl * ch; ++e) y[e] += t[e];
    free(t);
  }
  (void)nbr8;
  return y;
}

/* Linear -> ReLU -> Linear */
static float* mlp(Weights* W, const float* x, int n, int cin, int mid, int cout) {
  const float *w0 = nextw(W), *b0 = nextw(W), *w1 = nextw(W), *b1 = nextw(W);
  float* h = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * mid);
  float* o = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * cout);
  dense(x, n, cin, w0, 1, b0, mid, h);
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < (int64_t)n * mid; ++e) h[e] = h[e] > 0.f ? h[e] : 0.f;
  dense(h, n, mid, w1, 1, b1, cout, o);
  free(h);
  return o;
}

static uint64_t morton3(int32_t x, int32_t y, int32_t z) {     /* tie-break key: x in bit 0 of every triple */
  uint64_t k = 0;
  const uint32_t ux = (uint32_t)(x + (1 << 15)), uy = (uint32_t)(y + (1 << 15)), uz = (uint32_t)(z + (1 << 15));
  for (int b = 0; b < 16; ++b)
    k |= ((uint64_t)((ux >> b) & 1) << (3 * b)) | ((uint64_t)((uy >> b) & 1) << (3 * b + 1)) |
         ((uint64_t)((uz >> b) & 1) << (3 * b + 2));
  return k;
}
typedef struct { float s; uint64_t t; int i; } SigKey;
static int cmp_sig(const void* a, const void* b) {
  const SigKey *x = (const SigKey*)a, *y = (const SigKey*)b;
  if (x->s != y->s) return x->s < y->s ? -1 : 1;
  return x->t < y->t ? -1 : (x->t > y->t ? 1 : 0);
}

/* One scan.  Outputs (caller allocated): out_global[256]; out_n3; for the first min(n3, n_k) selected keypoints in
 * ascending-sigma order: sel_coords[n_k][3] (super-voxel coordinate), sel_kp[n_k][3], sel_desc[n_k][128],
 * sel_sigma[n_k]; level_counts[8].  Returns the number of selected keypoints, < 0 on error. */
/* mode 0: CartesianQuantizer(step[0]) (datasets/quantization.py:79-103); mode 1: PolarQuantizer(step[0..2]) = sector in
 * degrees, ring and z in metres (datasets/quantization.py:29-72; the shipped configuration, models/egonn.txt:3-5). */
int egonn_cpu_compute_embedding_q(const float* points, int64_t n_points, int mode, const float* step,
                                  const float* const* weights, int n_weights, int n_k, float* out_global,
                                  int32_t* level_counts, int32_t* sel_coords, float* sel_kp, float* sel_desc,
                                  float* sel_sigma, int n_threads) {
  if (n_points <= 0 || n_weights < 1 || (mode != 0 && mode != 1)) return -1;
  const float quant_step = step[0];
  const float qs[3] = {step[0], mode ? step[1] : step[0], mode ? step[2] : step[0]};
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);     /* per calling thread: several scans may run side by side */
#endif
  Weights W = {weights, 0};
  Level lv[NLEV];
  /* ---- quantise: floor(p / q) (fp32 division, as torch.floor(pc / q)), unique voxels */
  uint64_t* k0 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n_points);
  for (int64_t i = 0; i < n_points; ++i) {
    float a = points[3 * i], b = points[3 * i + 1];
    const float c = points[3 * i + 2];
    if (mode) {   /* theta = 180 + atan2(y, x) * 180 / pi (degrees, evaluated left to right in fp32); dist = sqrt(x^2 + y^2) */
      const float px = a, py = b;
      a = 180.0f + (atan2f(py, px) * 180.0f) / 3.14159265358979323846f;
      b = sqrtf(px * px + py * py);
    }
    const int32_t x = (int32_t)floorf(a / qs[0]), y = (int32_t)floorf(b / qs[1]), z = (int32_t)floorf(c / qs[2]);
    k0[i] = pack(x, y, z);
  }
  level_from_keys(&lv[0], k0, (int)n_points);
  for (int l = 1; l < NLEV; ++l) {            /* strided maps: floor(c / 2^l) * 2^l, de-duplicated */
    const int s = 1 << l;
    uint64_t* kl = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(lv[l - 1].n > 0 ? lv[l - 1].n : 1));
    for (int i = 0; i < lv[l - 1].n; ++i) {
      const int32_t* c = lv[l - 1].c + 3 * i;
      kl[i] = pack(floor_div_mul(c[0], s), floor_div_mul(c[1], s), floor_div_mul(c[2], s));
    }
    level_from_keys(&lv[l], kl, lv[l - 1].n);
  }
  for (int l = 0; l < NLEV; ++l) level_counts[l] = lv[l].n;

  /* ---- conv0: k=5, 1 -> 32, all-ones features: out[o][c] = sum over occupied offsets k of W0[k][0][c] */
  const int n0 = lv[0].n;
  float* x = (float*)malloc(sizeof(float) * (size_t)n0 * 32);
  {
    const float* w0 = nextw(&W);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < n0; ++o) {
      float* dst = x + (size_t)o * 32;
      for (int c = 0; c < 32; ++c) dst[c] = 0.f;
      const int32_t* c0 = lv[0].c + 3 * o;
      for (int k = 0; k < 125; ++k) {
        const int dx = k % 5 - 2, dy = (k / 5) % 5 - 2, dz = k / 25 - 2;
        if (find(&lv[0], c0[0] + dx, c0[1] + dy, c0[2] + dz) < 0) continue;
        for (int c = 0; c < 32; ++c) dst[c] += w0[k * 32 + c];
      }
    }
    const float* bn[4] = {nextw(&W), nextw(&W), nextw(&W), nextw(&W)};
    bn_act(x, n0, 32, bn, 1);
  }
  float* feats[NLEV] = {0};
  int cin = 32;
  for (int l = 1; l < NLEV; ++l) {
    const int n = lv[l].n, s = 1 << l, cout = PLANES[l - 1];
    /* conv k=2 s=2 cin -> cin: every level-(l-1) voxel feeds its floor-parent through slot (c - parent) / 2^(l-1) */
    int* nbr8 = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1) * 8);
    for (int i = 0; i < n * 8; ++i) nbr8[i] = -1;
    for (int f = 0; f < lv[l - 1].n; ++f) {
      const int32_t* c = lv[l - 1].c + 3 * f;
      const int32_t px = floor_div_mul(c[0], s), py = floor_div_mul(c[1], s), pz = floor_div_mul(c[2], s);
      const int p = find(&lv[l], px, py, pz);
      const int h = s / 2;
      nbr8[(size_t)p * 8 + (c[0] - px) / h + 2 * ((c[1] - py) / h) + 4 * ((c[2] - pz) / h)] = f;
    }
    float* y = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * cin);
    conv_map(x, cin, nbr8, 8, nextw(&W), cin, y, n);
    {
      const float* bn[4] = {nextw(&W), nextw(&W), nextw(&W), nextw(&W)};
      bn_act(y, n, cin, bn, 1);
    }
    free(x);
    free(nbr8);
    /* ECABasicBlock */
    int* nbr27 = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1) * 27);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < n; ++o) {
      const int32_t* c = lv[l].c + 3 * o;
      for (int k = 0; k < 27; ++k)
        nbr27[(size_t)o * 27 + k] = find(&lv[l], c[0] + (k % 3 - 1) * s, c[1] + ((k / 3) % 3 - 1) * s, c[2] + (k / 9 - 1) * s);
    }
    float* t1 = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * cout);
    float* t2 = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * cout);
    conv_map(y, cin, nbr27, 27, nextw(&W), cout, t1, n);                                   /* conv1 */
    {
      const float* bn[4] = {nextw(&W), nextw(&W), nextw(&W), nextw(&W)};
      bn_act(t1, n, cout, bn, 1);
    }
    conv_map(t1, cout, nbr27, 27, nextw(&W), cout, t2, n);                                 /* conv2 */
    {
      const float* bn[4] = {nextw(&W), nextw(&W), nextw(&W), nextw(&W)};
      bn_act(t2, n, cout, bn, 0);
    }
    free(nbr27);
    float* res = y;
    if (cin != cout) {                                                                       /* downsample 1x1 + BN */
      res = t1;   /* reuse */
      dense(y, n, cin, nextw(&W), 0, NULL, cout, res);
      const float* bn[4] = {nextw(&W), nextw(&W), nextw(&W), nextw(&W)};
      bn_act(res, n, cout, bn, 0);
    }
    {                                                                                        /* ECA gate */
      const float* ew = nextw(&W);
      const int ks = (cout == 128) ? 5 : 3, pad = (ks - 1) / 2;
      float mean[128], gate[128];
      for (int c = 0; c < cout; ++c) {
        double sacc = 0.0;
        for (int r = 0; r < n; ++r) sacc += t2[(size_t)r * cout + c];
        mean[c] = n > 0 ? (float)(sacc / n) : 0.f;
      }
      for (int c = 0; c < cout; ++c) {
        float v = 0.f;
        for (int j = 0; j < ks; ++j) {
          const int q = c + j - pad;
          if (q >= 0 && q < cout) v += ew[j] * mean[q];
        }
        gate[c] = 1.f / (1.f + expf(-v));
      }
#pragma omp parallel for schedule(static)
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < cout; ++c) {
          const float v = t2[(size_t)r * cout + c] * gate[c] + res[(size_t)r * cout + c];
          t2[(size_t)r * cout + c] = v > 0.f ? v : 0.f;
        }
    }
    free(y);
    free(t1);
    x = t2;
    feats[l] = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * cout);
    memcpy(feats[l], x, sizeof(float) * (size_t)n * cout);
    cin = cout;
  }
  free(x);

  /* ---- global branch: head(5..7) -> MLP 128-192-256 -> GeM */
  {
    float* g = head(&W, lv, NULL, feats, 5, 7, 128);
    float* d = mlp(&W, g, lv[5].n, 128, 192, 256);
    const float p = nextw(&W)[0];
    const int n = lv[5].n;
    for (int c = 0; c < 256; ++c) {
      double sacc = 0.0;
      for (int r = 0; r < n; ++r) sacc += powf(fmaxf(d[(size_t)r * 256 + c], 1e-6f), p);
      out_global[c] = powf((float)(sacc / (n > 0 ? n : 1)), 1.f / p);
    }
    free(g);
    free(d);
  }
  /* ---- local branch: head(3..4) -> descriptors (L2 norm), keypoints (tanh + keypoint_position), sigma (softplus) */
  int n_sel = 0;
  {
    const int n3 = lv[3].n;
    float* h = head(&W, lv, NULL, feats, 3, 4, 64);
    float* desc = mlp(&W, h, n3, 64, 96, 128);
    float* kp = mlp(&W, h, n3, 64, 32, 3);
    float* sg = mlp(&W, h, n3, 64, 32, 1);
    SigKey* order = (SigKey*)malloc(sizeof(SigKey) * (size_t)(n3 > 0 ? n3 : 1));
    for (int r = 0; r < n3; ++r) {
      const float v = sg[r];
      sg[r] = v > 20.f ? v : log1pf(expf(v));
      order[r].s = sg[r];
      order[r].t = morton3(lv[3].c[3 * r], lv[3].c[3 * r + 1], lv[3].c[3 * r + 2]);
      order[r].i = r;
    }
    qsort(order, n3, sizeof(SigKey), cmp_sig);
    n_sel = n3 < n_k ? n3 : n_k;
    for (int q = 0; q < n_sel; ++q) {
      const int r = order[q].i;
      float nrm = 0.f;
      for (int c = 0; c < 128; ++c) nrm += desc[(size_t)r * 128 + c] * desc[(size_t)r * 128 + c];
      nrm = fmaxf(sqrtf(nrm), 1e-12f);
      for (int c = 0; c < 128; ++c) sel_desc[(size_t)q * 128 + c] = desc[(size_t)r * 128 + c] / nrm;
      float pos[3];
      for (int a = 0; a < 3; ++a) {
        const int32_t cc = lv[3].c[3 * r + a];
        sel_coords[3 * q + a] = cc;
        /* (C + 0.5) q + tanh(offset) * (stride q) / 2,  stride = 8 */
        pos[a] = ((float)cc + 0.5f) * qs[a] + tanhf(kp[(size_t)r * 3 + a]) * (8.f * qs[a]) / 2.f;
      }
      if (mode) {   /* PolarQuantizer.to_cartesian: theta = pi (deg - 180) / 180 */
        const float theta = 3.14159265358979323846f * (pos[0] - 180.0f) / 180.0f, rr = pos[1];
        pos[0] = cosf(theta) * rr;
        pos[1] = sinf(theta) * rr;
      }
      for (int a = 0; a < 3; ++a) sel_kp[3 * q + a] = pos[a];
      sel_sigma[q] = sg[r];
    }
    free(order);
    free(h);
    free(desc);
    free(kp);
    free(sg);
  }
  for (int l = 0; l < NLEV; ++l) {
    free(lv[l].c);
    free(lv[l].key);
    if (feats[l]) free(feats[l]);
  }
  return (W.i == n_weights) ? n_sel : -2;      /* -2: the weight list was not consumed exactly */
}

int egonn_cpu_compute_embedding(const float* points, int64_t n_points, float quant_step, const float* const* weights,
                                int n_weights, int n_k, float* out_global, int32_t* level_counts, int32_t* sel_coords,
                                float* sel_kp, float* sel_desc, float* sel_sigma, int n_threads) {
  const float step[3] = {quant_step, quant_step, quant_step};
  return egonn_cpu_compute_embedding_q(points, n_points, 0, step, weights, n_weights, n_k, out_global, level_counts,
                                       sel_coords, sel_kp, sel_desc, sel_sigma, n_threads);
}

int egonn_cpu_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
