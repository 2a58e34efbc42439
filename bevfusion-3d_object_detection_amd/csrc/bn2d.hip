// bn2d.hip -- training-mode BatchNorm2d (+ residual add) (+ ReLU) on channels-last activations, fwd + bwd (gfx950).
//
// Every dense conv of the model (ResNet-50 bottlenecks, LSS-FPN, dtransform / depthnet / downsample of the view
// transform BF/depth_lss.py:581-620, ConvFuser BF/bevfusion_head.py:25-38, SECOND / SECONDFPN
// mmdet3d/models/backbones/second.py:27-95, the heat-map head :104-126) is followed by BatchNorm2d and ReLU, and the
// bottleneck adds its identity before the last ReLU.  As separate library kernels that is 5 passes over the
// activation forward and 8 backward (BN statistics, BN apply, [add], ReLU; ReLU backward, BN reduce, BN dx), the
// BatchNorm ones split into three launches each for mid-sized tensors.  An NHWC tensor is a row-major [M = N*H*W, C]
// matrix, so the layer is a column reduction + an elementwise pass:
//   forward  (3 passes): column sum / sum of squares per row slab (fp32, 16-B vector loads) -> fp64 combine -> per-channel
//                        (a, b) = (gamma*invstd, beta - mean*a) -> y = relu(a*x + b [+ residual])
//   backward (5 passes): g = dy * [y > 0] with the mask RECOMPUTED from x (a*x + b > 0; y is re-read only when a residual was
//                        added); dbeta = sum g, dgamma = invstd * sum g*(x - mean) -> dx = c1*g + c2*x + c3; d_residual = g
// A thread owns one 16-byte channel vector (8 bf16 / 4 fp32) and strides over rows, so its per-channel coefficients
// stay in registers; reductions are two-stage in a fixed order (deterministic, no atomics).
#include <algorithm>

#include "common.h"

namespace bfhip {
namespace {

typedef unsigned short bf16_t;

template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void load(const float *p, float *o) {
    float4 v = *(const float4 *)p;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float *p, const float *o) { *(float4 *)p = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct Vec<bf16_t> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void load(const bf16_t *p, float *o) {
    uint4 v = *(const uint4 *)p;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ unsigned rne(float f) {  // fp32 -> bf16, round to nearest even (NaN kept quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  }
  static __device__ __forceinline__ void store(bf16_t *p, const float *o) {
    uint4 v;
    v.x = rne(o[0]) | (rne(o[1]) << 16);
    v.y = rne(o[2]) | (rne(o[3]) << 16);
    v.z = rne(o[4]) | (rne(o[5]) << 16);
    v.w = rne(o[6]) | (rne(o[7]) << 16);
    *(uint4 *)p = v;
  }
};

// A block covers Lb vector columns (Lb*V channels, column tile blockIdx.y) x R row lanes; blockIdx.x is the row slab.
struct Map { int Lb, R, rpb; };  // vector columns per block, row lanes per block, rows per slab

// block-level combine of per-thread (s0[V], s1[V]) over the R row lanes -> partial[blk][2][C]
template <int V>
__device__ __forceinline__ void combine_rows(const float *s0, const float *s1, int Lb, int R, int C,
                                             float *__restrict__ partial, float *sm) {
  const int t = threadIdx.x;
  float *m0 = sm, *m1 = sm + 256 * V;
#pragma unroll
  for (int j = 0; j < V; ++j) { m0[t * V + j] = s0[j]; m1[t * V + j] = s1[j]; }
  __syncthreads();
  // local channel cl = cvl*V + j lives at thread (rl*Lb + cvl): element index (rl*Lb*V + cl)
  const int c0 = blockIdx.y * Lb * V;
  for (int cl = t; cl < Lb * V && c0 + cl < C; cl += 256) {
    float a = 0.f, b = 0.f;
    for (int rl = 0; rl < R; ++rl) { a += m0[rl * Lb * V + cl]; b += m1[rl * Lb * V + cl]; }
    partial[((size_t)blockIdx.x * 2 + 0) * C + c0 + cl] = a;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c0 + cl] = b;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn2d_stats_kernel(const T *__restrict__ x, long long M, int C, Map mp,
                                                         float *__restrict__ partial) {
  constexpr int V = Vec<T>::V;
  __shared__ float sm[2 * 256 * V];
  const int t = threadIdx.x, cv = blockIdx.y * mp.Lb + t % mp.Lb, rl = t / mp.Lb;
  const bool live = rl < mp.R && cv * V < C;
  float s0[V], s1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  if (live) {
    const long long r0 = (long long)blockIdx.x * mp.rpb, r1 = r0 + mp.rpb < M ? r0 + mp.rpb : M;
    const T *p = x + (size_t)cv * V;
    long long r = r0 + rl;
    for (; r + 3LL * mp.R < r1; r += 4LL * mp.R) {
      float v[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u) Vec<T>::load(p + (size_t)(r + (long long)u * mp.R) * C, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < V; ++j) { s0[j] += v[u][j]; s1[j] += v[u][j] * v[u][j]; }
    }
    for (; r < r1; r += mp.R) {
      float v[V];
      Vec<T>::load(p + (size_t)r * C, v);
#pragma unroll
      for (int j = 0; j < V; ++j) { s0[j] += v[j]; s1[j] += v[j] * v[j]; }
    }
  }
  combine_rows<V>(s0, s1, mp.Lb, mp.R, C, partial, sm);
}

// 8 channels per block x 32 slab lanes; every lane issues its (<= 32) loads in independent groups of 4, then a fixed-order
// fp64 combine through LDS.  The serial chain per launch is ~8 L2 round trips whatever the number of slabs.
__device__ __forceinline__ void reduce_partials8(const float *__restrict__ partial, int nblk, int C, int c, bool ok,
                                                 double &s, double &s2) {
  __shared__ double sm[2][256];
  const int kl = threadIdx.x >> 3;  // 0..31
  double a = 0.0, b = 0.0;
  if (ok) {
    int k = kl;
    for (; k + 96 < nblk; k += 128) {
      float v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v0[u] = partial[((size_t)(k + 32 * u) * 2 + 0) * C + c];
        v1[u] = partial[((size_t)(k + 32 * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a += (double)v0[u]; b += (double)v1[u]; }
    }
    for (; k < nblk; k += 32) {
      a += (double)partial[((size_t)k * 2 + 0) * C + c];
      b += (double)partial[((size_t)k * 2 + 1) * C + c];
    }
  }
  sm[0][threadIdx.x] = a;
  sm[1][threadIdx.x] = b;
  __syncthreads();
  s = 0.0; s2 = 0.0;
  if (threadIdx.x < 8)
    for (int k = 0; k < 32; ++k) { s += sm[0][k * 8 + threadIdx.x]; s2 += sm[1][k * 8 + threadIdx.x]; }
}

// stats[0..C) mean, [C..2C) invstd, [2C..3C) a = gamma*invstd, [3C..4C) b = beta - mean*a
// m_dev (optional): number of ACTIVE rows, on the device; the rows beyond it are exact zeros (feature matrices of the sparse
// encoder in static capacity mode), so only the divisor changes
__global__ __launch_bounds__(256) void bn2d_finalize_kernel(const float *__restrict__ partial, int nblk, long long M,
                                                            const int *__restrict__ m_dev, int C, float eps, float momentum,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            float *__restrict__ stats,
                                                            float *__restrict__ running_mean,
                                                            float *__restrict__ running_var) {
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  double s, s2;
  reduce_partials8(partial, nblk, C, c, c < C, s, s2);
  if (threadIdx.x >= 8 || c >= C) return;
  if (m_dev) { long long mv = *m_dev; M = mv < 1 ? 1 : (mv < M ? mv : M); }
  double mean = s / (double)M;
  double var = s2 / (double)M - mean * mean;
  if (var < 0.0) var = 0.0;
  float invstd = (float)(1.0 / sqrt(var + (double)eps));
  float a = gamma[c] * invstd;
  stats[c] = (float)mean;
  stats[C + c] = invstd;
  stats[2 * C + c] = a;
  stats[3 * C + c] = beta[c] - (float)mean * a;
  if (running_mean) {
    double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const T *__restrict__ x, const T *__restrict__ res,
                                                         const float *__restrict__ stats, long long M, int C, Map mp,
                                                         T *__restrict__ y) {
  constexpr int V = Vec<T>::V;
  const int t = threadIdx.x, cv = blockIdx.y * mp.Lb + t % mp.Lb, rl = t / mp.Lb;
  const bool live = rl < mp.R && cv * V < C;
  if (!live) return;
  float a[V], b[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a[j] = stats[2 * C + cv * V + j]; b[j] = stats[3 * C + cv * V + j]; }
  const long long r0 = (long long)blockIdx.x * mp.rpb, r1 = r0 + mp.rpb < M ? r0 + mp.rpb : M;
  const size_t col = (size_t)cv * V;
  long long r = r0 + rl;
  for (; r + mp.R < r1; r += 2LL * mp.R) {
    float v[2][V], q[2][V];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      Vec<T>::load(x + (size_t)(r + (long long)u * mp.R) * C + col, v[u]);
      if (RES) Vec<T>::load(res + (size_t)(r + (long long)u * mp.R) * C + col, q[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float o = v[u][j] * a[j] + b[j];
        if (RES) o += q[u][j];
        v[u][j] = (RELU && !(o > 0.f)) ? 0.f : o;
      }
      Vec<T>::store(y + (size_t)(r + (long long)u * mp.R) * C + col, v[u]);
    }
  }
  for (; r < r1; r += mp.R) {
    float v[V], q[V];
    Vec<T>::load(x + (size_t)r * C + col, v);
    if (RES) Vec<T>::load(res + (size_t)r * C + col, q);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float o = v[j] * a[j] + b[j];
      if (RES) o += q[j];
      v[j] = (RELU && !(o > 0.f)) ? 0.f : o;
    }
    Vec<T>::store(y + (size_t)r * C + col, v);
  }
}

// MASK: 0 = no ReLU, 1 = ReLU mask recomputed from x (a*x + b > 0), 2 = ReLU mask from the saved output y
template <typename T, int MASK>
__device__ __forceinline__ void masked_grad(float *g, const float *xv, const float *yv, const float *a, const float *b) {
  constexpr int V = Vec<T>::V;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    if (MASK == 1 && !(xv[j] * a[j] + b[j] > 0.f)) g[j] = 0.f;
    if (MASK == 2 && !(yv[j] > 0.f)) g[j] = 0.f;
  }
}

// partial[blk][0][c] = sum g, partial[blk][1][c] = sum g * (x - mean)
template <typename T, int MASK>
__global__ __launch_bounds__(256) void bn2d_bwd_reduce_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                              const T *__restrict__ y,
                                                              const float *__restrict__ stats, long long M, int C,
                                                              Map mp, float *__restrict__ partial) {
  constexpr int V = Vec<T>::V;
  __shared__ float sm[2 * 256 * V];
  const int t = threadIdx.x, cv = blockIdx.y * mp.Lb + t % mp.Lb, rl = t / mp.Lb;
  const bool live = rl < mp.R && cv * V < C;
  float s0[V], s1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  if (live) {
    float mean[V], a[V], b[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      mean[j] = stats[cv * V + j];
      a[j] = stats[2 * C + cv * V + j];
      b[j] = stats[3 * C + cv * V + j];
    }
    const long long r0 = (long long)blockIdx.x * mp.rpb, r1 = r0 + mp.rpb < M ? r0 + mp.rpb : M;
    const size_t col = (size_t)cv * V;
    long long r = r0 + rl;
    for (; r + mp.R < r1; r += 2LL * mp.R) {
      float g[2][V], xv[2][V], yv[2][V];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        size_t off = (size_t)(r + (long long)u * mp.R) * C + col;
        Vec<T>::load(dy + off, g[u]);
        Vec<T>::load(x + off, xv[u]);
        if (MASK == 2) Vec<T>::load(y + off, yv[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        masked_grad<T, MASK>(g[u], xv[u], yv[u], a, b);
#pragma unroll
        for (int j = 0; j < V; ++j) { s0[j] += g[u][j]; s1[j] += g[u][j] * (xv[u][j] - mean[j]); }
      }
    }
    for (; r < r1; r += mp.R) {
      float g[V], xv[V], yv[V];
      size_t off = (size_t)r * C + col;
      Vec<T>::load(dy + off, g);
      Vec<T>::load(x + off, xv);
      if (MASK == 2) Vec<T>::load(y + off, yv);
      masked_grad<T, MASK>(g, xv, yv, a, b);
#pragma unroll
      for (int j = 0; j < V; ++j) { s0[j] += g[j]; s1[j] += g[j] * (xv[j] - mean[j]); }
    }
  }
  combine_rows<V>(s0, s1, mp.Lb, mp.R, C, partial, sm);
}

// dgb[0..C) dgamma, [C..2C) dbeta; coef[0..C) c1, [C..2C) c2, [2C..3C) c3 with dx = c1*g + c2*x + c3
__global__ __launch_bounds__(256) void bn2d_bwd_finalize_kernel(const float *__restrict__ partial, int nblk,
                                                                long long M, const int *__restrict__ m_dev, int C,
                                                                const float *__restrict__ stats,
                                                                float *__restrict__ dgb, float *__restrict__ coef) {
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  double s, s2;
  reduce_partials8(partial, nblk, C, c, c < C, s, s2);
  if (threadIdx.x >= 8 || c >= C) return;
  if (m_dev) { long long mv = *m_dev; M = mv < 1 ? 1 : (mv < M ? mv : M); }
  const float mean = stats[c], invstd = stats[C + c], a = stats[2 * C + c];
  const float dbeta = (float)s, dgamma = (float)(s2 * (double)invstd);
  dgb[c] = dgamma;
  dgb[C + c] = dbeta;
  const float invM = (float)(1.0 / (double)M);
  const float c2 = -a * invstd * dgamma * invM;
  coef[c] = a;
  coef[C + c] = c2;
  coef[2 * C + c] = -a * dbeta * invM - c2 * mean;
}

template <typename T, int MASK, bool DRES>
__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                             const T *__restrict__ y,
                                                             const float *__restrict__ stats,
                                                             const float *__restrict__ coef, long long M, int C,
                                                             Map mp, T *__restrict__ dx, T *__restrict__ dres) {
  constexpr int V = Vec<T>::V;
  const int t = threadIdx.x, cv = blockIdx.y * mp.Lb + t % mp.Lb, rl = t / mp.Lb;
  const bool live = rl < mp.R && cv * V < C;
  if (!live) return;
  float a[V], b[V], c1[V], c2[V], c3[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    a[j] = stats[2 * C + cv * V + j];
    b[j] = stats[3 * C + cv * V + j];
    c1[j] = coef[cv * V + j];
    c2[j] = coef[C + cv * V + j];
    c3[j] = coef[2 * C + cv * V + j];
  }
  const long long r0 = (long long)blockIdx.x * mp.rpb, r1 = r0 + mp.rpb < M ? r0 + mp.rpb : M;
  const size_t col = (size_t)cv * V;
  long long r = r0 + rl;
  for (; r + mp.R < r1; r += 2LL * mp.R) {
    float g[2][V], xv[2][V], yv[2][V];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      size_t off = (size_t)(r + (long long)u * mp.R) * C + col;
      Vec<T>::load(dy + off, g[u]);
      Vec<T>::load(x + off, xv[u]);
      if (MASK == 2) Vec<T>::load(y + off, yv[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      size_t off = (size_t)(r + (long long)u * mp.R) * C + col;
      masked_grad<T, MASK>(g[u], xv[u], yv[u], a, b);
      if (DRES) Vec<T>::store(dres + off, g[u]);
#pragma unroll
      for (int j = 0; j < V; ++j) xv[u][j] = c1[j] * g[u][j] + c2[j] * xv[u][j] + c3[j];
      Vec<T>::store(dx + off, xv[u]);
    }
  }
  for (; r < r1; r += mp.R) {
    float g[V], xv[V], yv[V];
    size_t off = (size_t)r * C + col;
    Vec<T>::load(dy + off, g);
    Vec<T>::load(x + off, xv);
    if (MASK == 2) Vec<T>::load(y + off, yv);
    masked_grad<T, MASK>(g, xv, yv, a, b);
    if (DRES) Vec<T>::store(dres + off, g);
#pragma unroll
    for (int j = 0; j < V; ++j) xv[j] = c1[j] * g[j] + c2[j] * xv[j] + c3[j];
    Vec<T>::store(dx + off, xv);
  }
}

inline int elem_size(int dtype) { return dtype == 1 ? 2 : 4; }

// C must be a multiple of the 16-byte vector
inline bool shape_ok(long long M, int C, int dtype) {
  const int V = 16 / elem_size(dtype);
  return M > 0 && C >= V && C % V == 0 && C <= 65536;
}

// grid = (row slabs S, column tiles CT).  One block iteration moves 256 x 16 B = 4 KB whatever the shape; the slab count
// aims at ~2048 blocks (8 per CU) with >= 4 iterations each, at most 1024 slabs, and partial sums <= 1/16 of the tensor.
inline Map make_map(long long M, int C, int dtype, dim3 *grid) {
  const int es = elem_size(dtype), V = 16 / es, L = C / V;
  Map mp;
  mp.Lb = L < 32 ? L : 32;
  mp.R = 256 / mp.Lb;
  const int CT = (L + mp.Lb - 1) / mp.Lb;
  long long S = 1024;
  S = std::min(S, (M + 4LL * mp.R - 1) / (4LL * mp.R));
  S = std::min(S, (long long)std::max(1, 2048 / CT));
  S = std::min(S, std::max(1LL, M * es / 128));
  S = std::max(S, 1LL);
  long long rpb = ((M + S - 1) / S + mp.R - 1) / mp.R * mp.R;
  S = (M + rpb - 1) / rpb;
  mp.rpb = (int)rpb;
  *grid = dim3((unsigned)S, (unsigned)CT);
  return mp;
}

template <typename T>
int run_fwd(const void *x, const void *res, const float *stats, long long M, int C, Map mp, dim3 grid, int relu, void *y,
            hipStream_t s) {
#define BFHIP_BN2D_APPLY(RES, RELU)                                                                             \
  hipLaunchKernelGGL((bn2d_apply_kernel<T, RES, RELU>), grid, dim3(256), 0, s, (const T *)x, (const T *)res, \
                     stats, M, C, mp, (T *)y)
  if (res) { if (relu) BFHIP_BN2D_APPLY(true, true); else BFHIP_BN2D_APPLY(true, false); }
  else { if (relu) BFHIP_BN2D_APPLY(false, true); else BFHIP_BN2D_APPLY(false, false); }
#undef BFHIP_BN2D_APPLY
  return 0;
}

template <typename T, int MASK>
void run_bwd(const void *dy, const void *x, const void *y, const float *stats, const float *gamma, long long M, int C,
             Map mp, dim3 grid, float *partial, float *coef, float *dgb, void *dx, void *dres, const int *m_dev, hipStream_t s) {
  const int nblk = (int)grid.x;
  hipLaunchKernelGGL((bn2d_bwd_reduce_kernel<T, MASK>), grid, dim3(256), 0, s, (const T *)dy, (const T *)x,
                     (const T *)y, stats, M, C, mp, partial);
  hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3(ceil_div(C, 8)), dim3(256), 0, s, partial, nblk, M, m_dev, C, stats, dgb,
                     coef);
  if (dres)
    hipLaunchKernelGGL((bn2d_bwd_apply_kernel<T, MASK, true>), grid, dim3(256), 0, s, (const T *)dy, (const T *)x,
                       (const T *)y, stats, coef, M, C, mp, (T *)dx, (T *)dres);
  else
    hipLaunchKernelGGL((bn2d_bwd_apply_kernel<T, MASK, false>), grid, dim3(256), 0, s, (const T *)dy,
                       (const T *)x, (const T *)y, stats, coef, M, C, mp, (T *)dx, (T *)nullptr);
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_bn2d_supported(long long M, int C, int dtype) {
  return (dtype == 0 || dtype == 1) && shape_ok(M, C, dtype) ? 1 : 0;
}

BFHIP_EXPORT size_t bfhip_bn2d_workspace_bytes(long long M, int C, int dtype) {
  if (!bfhip_bn2d_supported(M, C, dtype)) return 0;
  dim3 grid;
  make_map(M, C, dtype, &grid);
  return align_up((size_t)grid.x * 2 * C * sizeof(float), 256) + align_up((size_t)3 * C * sizeof(float), 256);
}

BFHIP_EXPORT int bfhip_bn2d_fwd(const void *x, const void *residual, const float *gamma, const float *beta,
                                long long M, int C, int dtype, float eps, float momentum, int relu,
                                float *running_mean, float *running_var, float *stats, void *y, const int32_t *m_dev,
                                void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(bfhip_bn2d_supported(M, C, dtype), "bn2d_fwd: unsupported shape M=%lld C=%d dtype=%d", M, C, dtype);
  BFHIP_REQUIRE(x && gamma && beta && stats && y, "bn2d_fwd: null pointer");
  BFHIP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)residual % 16) == 0,
                "bn2d_fwd: tensors must be 16-byte aligned");
  if (!workspace || workspace_bytes < bfhip_bn2d_workspace_bytes(M, C, dtype)) { set_error("bn2d_fwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  dim3 grid;
  Map mp = make_map(M, C, dtype, &grid);
  const int nblk = (int)grid.x;
  float *partial = (float *)workspace;
  if (dtype == 1)
    hipLaunchKernelGGL(bn2d_stats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t *)x, M, C, mp, partial);
  else
    hipLaunchKernelGGL(bn2d_stats_kernel<float>, grid, dim3(256), 0, s, (const float *)x, M, C, mp, partial);
  hipLaunchKernelGGL(bn2d_finalize_kernel, dim3(ceil_div(C, 8)), dim3(256), 0, s, partial, nblk, M, m_dev, C, eps, momentum,
                     gamma, beta, stats, running_mean, running_var);
  if (dtype == 1) run_fwd<bf16_t>(x, residual, stats, M, C, mp, grid, relu, y, s);
  else run_fwd<float>(x, residual, stats, M, C, mp, grid, relu, y, s);
  return check_launch("bn2d_fwd");
}

BFHIP_EXPORT int bfhip_bn2d_fwd_partials(const void *x, const void *residual, const float *gamma, const float *beta,
                                         long long M, int C, int dtype, float eps, float momentum, int relu,
                                         float *running_mean, float *running_var, float *stats, void *y,
                                         const float *partial, int nblk, const int32_t *m_dev, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(bfhip_bn2d_supported(M, C, dtype), "bn2d_fwd_partials: unsupported shape M=%lld C=%d dtype=%d", M, C, dtype);
  BFHIP_REQUIRE(x && gamma && beta && stats && y && partial && nblk > 0, "bn2d_fwd_partials: null pointer");
  BFHIP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)residual % 16) == 0,
                "bn2d_fwd_partials: tensors must be 16-byte aligned");
  dim3 grid;
  Map mp = make_map(M, C, dtype, &grid);
  hipLaunchKernelGGL(bn2d_finalize_kernel, dim3(ceil_div(C, 8)), dim3(256), 0, s, partial, nblk, M, m_dev, C, eps, momentum,
                     gamma, beta, stats, running_mean, running_var);
  if (dtype == 1) run_fwd<bf16_t>(x, residual, stats, M, C, mp, grid, relu, y, s);
  else run_fwd<float>(x, residual, stats, M, C, mp, grid, relu, y, s);
  return check_launch("bn2d_fwd_partials");
}

BFHIP_EXPORT int bfhip_bn2d_bwd(const void *dy, const void *x, const void *y, const float *stats, const float *gamma,
                                long long M, int C, int dtype, int relu, void *dx, void *dres, float *dgb,
                                const int32_t *m_dev, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(bfhip_bn2d_supported(M, C, dtype), "bn2d_bwd: unsupported shape M=%lld C=%d dtype=%d", M, C, dtype);
  BFHIP_REQUIRE(dy && x && stats && gamma && dx && dgb, "bn2d_bwd: null pointer");
  BFHIP_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
                    ((uintptr_t)y % 16) == 0 && ((uintptr_t)dres % 16) == 0, "bn2d_bwd: tensors must be 16-byte aligned");
  if (!workspace || workspace_bytes < bfhip_bn2d_workspace_bytes(M, C, dtype)) { set_error("bn2d_bwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  dim3 grid;
  Map mp = make_map(M, C, dtype, &grid);
  const int nblk = (int)grid.x;
  float *partial = (float *)workspace;
  float *coef = (float *)((char *)workspace + align_up((size_t)nblk * 2 * C * sizeof(float), 256));
  // ReLU mask: from the saved output when one is given (residual layers), otherwise recomputed from x
  const int mask = !relu ? 0 : (y ? 2 : 1);
  if (dtype == 1) {
    if (mask == 0) run_bwd<bf16_t, 0>(dy, x, y, stats, gamma, M, C, mp, grid, partial, coef, dgb, dx, dres, m_dev, s);
    else if (mask == 1) run_bwd<bf16_t, 1>(dy, x, y, stats, gamma, M, C, mp, grid, partial, coef, dgb, dx, dres, m_dev, s);
    else run_bwd<bf16_t, 2>(dy, x, y, stats, gamma, M, C, mp, grid, partial, coef, dgb, dx, dres, m_dev, s);
  } else {
    if (mask == 0) run_bwd<float, 0>(dy, x, y, stats, gamma, M, C, mp, grid, partial, coef, dgb, dx, dres, m_dev, s);
    else if (mask == 1) run_bwd<float, 1>(dy, x, y, stats, gamma, M, C, mp, grid, partial, coef, dgb, dx, dres, m_dev, s);
    else run_bwd<float, 2>(dy, x, y, stats, gamma, M, C, mp, grid, partial, coef, dgb, dx, dres, m_dev, s);
  }
  return check_launch("bn2d_bwd");
}
