/*
 * spconv_oracle.c -- CPU restatement of the sparse 3-D convolution used by the reference's
 * BEVFusionSparseEncoder.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as bevfusion_oracle.c).
 *
 * PARITY UNPINNED: the arithmetic lives in the third-party `spconv` 2.x / `cumm` wheels
 * (reference: `pip install spconv-cu120`, projects/BEVFusion/README.md:13; version gate
 * mmdet3d/models/layers/spconv/__init__.py:9), which are not in /root/reference and not
 * installed.  The reference's tests at this boundary assert shapes only
 * (tests/test_models/test_layers/test_spconv/test_spconv_module.py:15-48).  This file restates
 * the published traveller59 semantics as they are visible from the reference's call sites:
 *   - output spatial size (in + 2p - d(k-1) - 1)//s + 1         projects/SparseConvolution/sparse_conv.py:88-90
 *     (reproduces the chain the reference records: 1440->720->360->180, 41->21->11->5->2,
 *      projects/BEVFusion/bevfusion/sparse_encoder.py:132,148)
 *   - SubM: out indices == in indices                            projects/SparseConvolution/sparse_functional.py:142-143
 *   - pair table pair_fwd[KV, N_out] int32, -1 = no input          projects/SparseConvolution/sparse_functional.py:57-61,139-162
 *   - weight layout (out, k0, k1, k2, in)                          mmdet3d/models/layers/spconv/overwrite_spconv/write_spconv2.py:50-51
 *   - out[n] = sum_k W[:,k,:] . in[pair_fwd[k][n]]  (cross-correlation, as torch.nn.Conv3d)
 * The numeric check that is independent of this file is torch.nn.functional.conv3d on the
 * densified tensor (tests/test_spconv.py).
 * Output row order of a strided SparseConv3d is implementation-defined in spconv (hash order);
 * the canonical order used here and by the HIP path is ascending linear index ((b*X+x)*Y+y)*Z+z.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ---- tiny open-addressing map int64 -> int32 ------------------------------------------- */
typedef struct { int64_t *keys; int32_t *vals; uint64_t mask; } imap_t;
static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static int imap_init(imap_t *m, size_t n) {
  uint64_t cap = 16;
  while (cap < 2 * n + 1) cap <<= 1;
  m->keys = (int64_t *)malloc(cap * sizeof(int64_t));
  m->vals = (int32_t *)malloc(cap * sizeof(int32_t));
  if (!m->keys || !m->vals) return -1;
  for (uint64_t i = 0; i < cap; ++i) m->keys[i] = -1;
  m->mask = cap - 1;
  return 0;
}
static void imap_free(imap_t *m) { free(m->keys); free(m->vals); }
/* insert if absent; returns stored value */
static int32_t imap_put(imap_t *m, int64_t k, int32_t v) {
  uint64_t s = mix64((uint64_t)k) & m->mask;
  while (m->keys[s] != -1 && m->keys[s] != k) s = (s + 1) & m->mask;
  if (m->keys[s] == -1) { m->keys[s] = k; m->vals[s] = v; }
  return m->vals[s];
}
static int32_t imap_get(const imap_t *m, int64_t k) {
  uint64_t s = mix64((uint64_t)k) & m->mask;
  while (m->keys[s] != -1 && m->keys[s] != k) s = (s + 1) & m->mask;
  return m->keys[s] == -1 ? -1 : m->vals[s];
}

static inline int64_t lin(int b, int x, int y, int z, const int *shape) {
  return (((int64_t)b * shape[0] + x) * shape[1] + y) * shape[2] + z;
}

ORACLE_API void oracle_conv_out_shape(const int *in_shape, const int *ksize, const int *stride,
                                      const int *padding, const int *dilation, int *out_shape) {
  for (int i = 0; i < 3; ++i)
    out_shape[i] = (in_shape[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) / stride[i] + 1;
}

/* SubM rulebook.  pair_fwd[k*N + n] = row j with coord(j) = coord(n) - pad + k*dil, or -1.
 * pad = dil*(ksize//2) (submanifold convs are centred).  returns number of valid pairs. */
ORACLE_API int64_t oracle_rulebook_subm(const int32_t *indices, int N, const int *shape,
                                        const int *ksize, const int *dilation, int32_t *pair_fwd) {
  imap_t m;
  if (imap_init(&m, (size_t)N)) return -1;
  for (int i = 0; i < N; ++i) {
    const int32_t *c = indices + (size_t)i * 4;
    imap_put(&m, lin(c[0], c[1], c[2], c[3], shape), i);
  }
  int KV = ksize[0] * ksize[1] * ksize[2];
  int64_t pairs = 0;
  for (int n = 0; n < N; ++n) {
    const int32_t *c = indices + (size_t)n * 4;
    int k = 0;
    for (int i = 0; i < ksize[0]; ++i)
      for (int j = 0; j < ksize[1]; ++j)
        for (int l = 0; l < ksize[2]; ++l, ++k) {
          int x = c[1] + (i - ksize[0] / 2) * dilation[0];
          int y = c[2] + (j - ksize[1] / 2) * dilation[1];
          int z = c[3] + (l - ksize[2] / 2) * dilation[2];
          int32_t r = -1;
          if (x >= 0 && x < shape[0] && y >= 0 && y < shape[1] && z >= 0 && z < shape[2])
            r = imap_get(&m, lin(c[0], x, y, z, shape));
          pair_fwd[(size_t)k * N + n] = r;
          pairs += (r >= 0);
        }
  }
  imap_free(&m);
  (void)KV;
  return pairs;
}

static int cmp_i64(const void *a, const void *b) {
  int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return x < y ? -1 : (x > y);
}

/* Regular (strided) SparseConv3d rulebook.
 * Output site o is active iff some input p and offset k satisfy o*s - pad + k*dil == p.
 * out_indices: canonical ascending linear order.  Buffers sized for N*KV rows worst case
 * (caller passes max_out).  pair_fwd[k*max_out + o] (row stride max_out) = input row or -1;
 * pair_bwd[k*N + i] = output row reached from input i through offset k, or -1.
 * returns N_out (or -1 on overflow / alloc failure). */
ORACLE_API int oracle_rulebook_sparse(const int32_t *indices, int N, const int *in_shape,
                                      const int *ksize, const int *stride, const int *padding,
                                      const int *dilation, int max_out, int32_t *out_indices,
                                      int32_t *pair_fwd, int32_t *pair_bwd, int64_t *n_pairs) {
  int out_shape[3];
  oracle_conv_out_shape(in_shape, ksize, stride, padding, dilation, out_shape);
  int KV = ksize[0] * ksize[1] * ksize[2];
  int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)N * KV + 8);
  if (!cand) return -1;
  size_t nc = 0;
  for (int n = 0; n < N; ++n) {
    const int32_t *c = indices + (size_t)n * 4;
    for (int i = 0; i < ksize[0]; ++i)
      for (int j = 0; j < ksize[1]; ++j)
        for (int l = 0; l < ksize[2]; ++l) {
          int ox = c[1] + padding[0] - i * dilation[0];
          int oy = c[2] + padding[1] - j * dilation[1];
          int oz = c[3] + padding[2] - l * dilation[2];
          if (ox < 0 || oy < 0 || oz < 0) continue;
          if (ox % stride[0] || oy % stride[1] || oz % stride[2]) continue;
          ox /= stride[0]; oy /= stride[1]; oz /= stride[2];
          if (ox >= out_shape[0] || oy >= out_shape[1] || oz >= out_shape[2]) continue;
          cand[nc++] = lin(c[0], ox, oy, oz, out_shape);
        }
  }
  qsort(cand, nc, sizeof(int64_t), cmp_i64);
  size_t n_out = 0;
  for (size_t i = 0; i < nc; ++i)
    if (i == 0 || cand[i] != cand[i - 1]) cand[n_out++] = cand[i];
  if ((int64_t)n_out > max_out) { free(cand); return -1; }
  imap_t m;
  if (imap_init(&m, n_out)) { free(cand); return -1; }
  for (size_t o = 0; o < n_out; ++o) {
    int64_t key = cand[o];
    imap_put(&m, key, (int32_t)o);
    int z = (int)(key % out_shape[2]); key /= out_shape[2];
    int y = (int)(key % out_shape[1]); key /= out_shape[1];
    int x = (int)(key % out_shape[0]); key /= out_shape[0];
    out_indices[o * 4 + 0] = (int32_t)key;
    out_indices[o * 4 + 1] = x; out_indices[o * 4 + 2] = y; out_indices[o * 4 + 3] = z;
  }
  for (size_t i = 0; i < (size_t)KV * max_out; ++i) pair_fwd[i] = -1;
  for (size_t i = 0; i < (size_t)KV * N; ++i) pair_bwd[i] = -1;
  int64_t pairs = 0;
  for (int n = 0; n < N; ++n) {
    const int32_t *c = indices + (size_t)n * 4;
    int k = 0;
    for (int i = 0; i < ksize[0]; ++i)
      for (int j = 0; j < ksize[1]; ++j)
        for (int l = 0; l < ksize[2]; ++l, ++k) {
          int ox = c[1] + padding[0] - i * dilation[0];
          int oy = c[2] + padding[1] - j * dilation[1];
          int oz = c[3] + padding[2] - l * dilation[2];
          if (ox < 0 || oy < 0 || oz < 0) continue;
          if (ox % stride[0] || oy % stride[1] || oz % stride[2]) continue;
          ox /= stride[0]; oy /= stride[1]; oz /= stride[2];
          if (ox >= out_shape[0] || oy >= out_shape[1] || oz >= out_shape[2]) continue;
          int32_t o = imap_get(&m, lin(c[0], ox, oy, oz, out_shape));
          pair_fwd[(size_t)k * max_out + o] = n;
          pair_bwd[(size_t)k * N + n] = o;
          ++pairs;
        }
  }
  imap_free(&m);
  free(cand);
  if (n_pairs) *n_pairs = pairs;
  return (int)n_out;
}

/* Forward: out[n][co] = sum_k sum_ci W[co][k][ci] * in[pair_fwd[k*ld + n]][ci]; fp64 accumulate
 * (the oracle gives the reference VALUE; fp32 kernels are compared within tolerance). */
ORACLE_API void oracle_spconv_fwd(const float *feat_in, const float *W, const int32_t *pair_fwd,
                                  int ld, int N_out, int KV, int Cin, int Cout, float *out) {
  double *acc = (double *)malloc(sizeof(double) * (size_t)Cout);
  for (int n = 0; n < N_out; ++n) {
    for (int co = 0; co < Cout; ++co) acc[co] = 0.0;
    for (int k = 0; k < KV; ++k) {
      int32_t r = pair_fwd[(size_t)k * ld + n];
      if (r < 0) continue;
      const float *x = feat_in + (size_t)r * Cin;
      for (int co = 0; co < Cout; ++co) {
        const float *w = W + ((size_t)co * KV + k) * Cin;
        double s = 0.0;
        for (int ci = 0; ci < Cin; ++ci) s += (double)w[ci] * (double)x[ci];
        acc[co] += s;
      }
    }
    for (int co = 0; co < Cout; ++co) out[(size_t)n * Cout + co] = (float)acc[co];
  }
  free(acc);
}

/* Backward: d_in[j][ci] = sum over (k,n) with pair_fwd[k][n]==j of sum_co W[co][k][ci]*d_out[n][co]
 *           d_W[co][k][ci] = sum_n d_out[n][co] * in[pair_fwd[k][n]][ci]          (fp64 accumulate) */
ORACLE_API void oracle_spconv_bwd(const float *feat_in, const float *W, const float *d_out,
                                  const int32_t *pair_fwd, int ld, int N_in, int N_out, int KV,
                                  int Cin, int Cout, float *d_in, float *d_W) {
  double *din = (double *)calloc((size_t)N_in * Cin, sizeof(double));
  double *dw = (double *)calloc((size_t)Cout * KV * Cin, sizeof(double));
  for (int k = 0; k < KV; ++k)
    for (int n = 0; n < N_out; ++n) {
      int32_t r = pair_fwd[(size_t)k * ld + n];
      if (r < 0) continue;
      const float *x = feat_in + (size_t)r * Cin;
      const float *g = d_out + (size_t)n * Cout;
      for (int co = 0; co < Cout; ++co) {
        const float *w = W + ((size_t)co * KV + k) * Cin;
        double *dwr = dw + ((size_t)co * KV + k) * Cin;
        double gg = (double)g[co];
        for (int ci = 0; ci < Cin; ++ci) {
          din[(size_t)r * Cin + ci] += (double)w[ci] * gg;
          dwr[ci] += gg * (double)x[ci];
        }
      }
    }
  for (size_t i = 0; i < (size_t)N_in * Cin; ++i) d_in[i] = (float)din[i];
  for (size_t i = 0; i < (size_t)Cout * KV * Cin; ++i) d_W[i] = (float)dw[i];
  free(din);
  free(dw);
}

/* SparseConvTensor.dense() + the BEVFusion permute: BF/sparse_encoder.py:147-151
 *   dense [B, C, X, Y, Z] -> permute(0,1,4,2,3) -> view [B, C*Z, X, Y]; channel index = c*Z + z */
ORACLE_API void oracle_sparse_to_bev(const float *feats, const int32_t *indices, int N, int C,
                                     int B, int X, int Y, int Z, float *out) {
  memset(out, 0, sizeof(float) * (size_t)B * C * Z * X * Y);
  for (int n = 0; n < N; ++n) {
    const int32_t *c = indices + (size_t)n * 4;
    for (int ch = 0; ch < C; ++ch)
      out[((((size_t)c[0] * C + ch) * Z + c[3]) * X + c[1]) * Y + c[2]] = feats[(size_t)n * C + ch];
  }
}
