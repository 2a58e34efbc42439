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
#include <map>
#include <mutex>

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

// Device-coherent accesses for the data that blocks hand to one another inside a launch (finish_in_last_block): agent-scope
// relaxed atomics go to the coherence point themselves (sc1 loads / write-through stores), so no cache-wide release / acquire
// is needed.  (__threadfence() here = an L2 write-back + invalidate per block: measured +16 ms on the 36 ms training step.)
template <typename T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all of this thread's stores have been acknowledged (s_waitcnt only)
__device__ __forceinline__ void stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }

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
    st_agent(&partial[((size_t)blockIdx.x * 2 + 0) * C + c0 + cl], a);
    st_agent(&partial[((size_t)blockIdx.x * 2 + 1) * C + c0 + cl], b);
  }
}

// ---- finalize inside the reduction launch -------------------------------------------------------------------------------
// The per-channel finalisation (mean / invstd / affine coefficients, or dgamma / dbeta / dx coefficients) used to be a third
// launch between the reduction and the elementwise pass: ~5 us of kernel plus a launch gap, 208 times per training step.
// It now runs in the LAST-ARRIVING block of the reduction launch, as a two-level tree so that no block ever reads more than
// 32 partial rows: the slabs of a column tile form groups of 32; the last block of a group to arrive adds the group's partial
// rows (fp64, slab order) into gp[group]; the last group to finish adds the <= 32 group rows (group order) and finalises.
// Sums are therefore in a fixed order whatever the arrival order (deterministic).  Arrival counters live in a small
// zero-initialised, self-resetting buffer owned by the library, one per stream (kernels of one stream run in order).
constexpr int kGroup = 32;
struct Tree {
  int *cnt;    // [column tile][1 + ng] arrival counters (all zero between launches)
  double *gp;  // [ng][2][C] group sums
  int nblk, ng;
};

template <typename Fin>
__device__ __forceinline__ void finish_in_last_block(const float *__restrict__ partial, Tree tr, int C, int c0, int ncl,
                                                     Fin fin) {
  __shared__ int s_flag;
  if (!tr.cnt) return;  // finalisation as a separate launch (BFHIP_BN2D_FOLD=0, A/B)
  const int t = threadIdx.x;
  const int g = blockIdx.x / kGroup;
  const int k0 = g * kGroup;
  const int gsize = tr.nblk - k0 < kGroup ? tr.nblk - k0 : kGroup;
  int *cnt = tr.cnt + (size_t)blockIdx.y * (1 + tr.ng);
  stores_done();  // this block's partial row has reached the coherence point before its arrival is counted
  __syncthreads();
  if (t == 0) s_flag = __hip_atomic_fetch_add(&cnt[1 + g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1;
  __syncthreads();
  if (!s_flag) return;
  const int c = c0 + t;
  const bool own = t < ncl && c < C;
  if (own) {
    double a = 0.0, b = 0.0;
    int k = 0;
    for (; k + 8 <= gsize; k += 8) {
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v0[u] = ld_agent(&partial[((size_t)(k0 + k + u) * 2 + 0) * C + c]);
        v1[u] = ld_agent(&partial[((size_t)(k0 + k + u) * 2 + 1) * C + c]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += (double)v0[u]; b += (double)v1[u]; }
    }
    for (; k < gsize; ++k) {
      a += (double)ld_agent(&partial[((size_t)(k0 + k) * 2 + 0) * C + c]);
      b += (double)ld_agent(&partial[((size_t)(k0 + k) * 2 + 1) * C + c]);
    }
    st_agent(&tr.gp[((size_t)g * 2 + 0) * C + c], a);
    st_agent(&tr.gp[((size_t)g * 2 + 1) * C + c], b);
  }
  stores_done();
  __syncthreads();
  if (t == 0) {
    st_agent(&cnt[1 + g], 0);  // every block of the group has arrived: leave the counter as it was found
    s_flag = __hip_atomic_fetch_add(&cnt[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tr.ng - 1;
  }
  __syncthreads();
  if (!s_flag) return;
  if (t == 0) st_agent(&cnt[0], 0);
  if (own) {
    double s = 0.0, s2 = 0.0;
    for (int g2 = 0; g2 < tr.ng; ++g2) {
      s += ld_agent(&tr.gp[((size_t)g2 * 2 + 0) * C + c]);
      s2 += ld_agent(&tr.gp[((size_t)g2 * 2 + 1) * C + c]);
    }
    fin(c, s, s2);
  }
}

// stats[0..C) mean, [C..2C) invstd, [2C..3C) a = gamma*invstd, [3C..4C) b = beta - mean*a
// m_dev (optional): number of ACTIVE rows, on the device; the rows beyond it are exact zeros (feature matrices of the sparse
// encoder in static capacity mode), so only the divisor changes
struct FwdFin {
  long long M;
  const int *m_dev;
  int C;
  float eps, momentum;
  const float *gamma, *beta;
  float *stats, *running_mean, *running_var;
  __device__ __forceinline__ void operator()(int c, double s, double s2) const {
    long long Mv = M;
    if (m_dev) { long long mv = *m_dev; Mv = mv < 1 ? 1 : (mv < M ? mv : M); }
    double mean = s / (double)Mv;
    double var = s2 / (double)Mv - mean * mean;
    if (var < 0.0) var = 0.0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float a = gamma[c] * invstd;
    stats[c] = (float)mean;
    stats[C + c] = invstd;
    stats[2 * C + c] = a;
    stats[3 * C + c] = beta[c] - (float)mean * a;
    if (running_mean) {
      double unbiased = Mv > 1 ? var * (double)Mv / (double)(Mv - 1) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
};

// dgb[0..C) dgamma, [C..2C) dbeta; coef[0..C) c1, [C..2C) c2, [2C..3C) c3 with dx = c1*g + c2*x + c3
struct BwdFin {
  long long M;
  const int *m_dev;
  int C;
  const float *stats;
  float *dgb, *coef;
  __device__ __forceinline__ void operator()(int c, double s, double s2) const {
    long long Mv = M;
    if (m_dev) { long long mv = *m_dev; Mv = mv < 1 ? 1 : (mv < M ? mv : M); }
    const float mean = stats[c], invstd = stats[C + c], a = stats[2 * C + c];
    const float dbeta = (float)s, dgamma = (float)(s2 * (double)invstd);
    dgb[c] = dgamma;
    dgb[C + c] = dbeta;
    const float invM = (float)(1.0 / (double)Mv);
    const float c2 = -a * invstd * dgamma * invM;
    coef[c] = a;
    coef[C + c] = c2;
    coef[2 * C + c] = -a * dbeta * invM - c2 * mean;
  }
};

// out[c] = column sum (the bias gradient of a convolution / linear layer: sum of dy over all pixels / rows)
struct SumFin {
  int C;
  float *out;
  __device__ __forceinline__ void operator()(int c, double s, double) const { out[c] = (float)s; }
};

template <typename T>
__global__ __launch_bounds__(256) void bn2d_stats_kernel(const T *__restrict__ x, long long M, int C, Map mp,
                                                         float *__restrict__ partial, Tree tr, FwdFin fin) {
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
  finish_in_last_block(partial, tr, C, blockIdx.y * mp.Lb * V, mp.Lb * V, fin);
}

// 8 channels per block x 64 slab lanes (512 threads); a lane issues ALL of its loads (<= 16 for up to 1024 slabs) before the
// first add, then a fixed-order fp64 combine through LDS in two levels (8 x 8 lanes, then 8).  The kernel is pure latency --
// a few KB of partial sums per channel -- and runs 208 times per training step: with round 2's groups of 4 loads its serial
// chain was ~8 L2 round trips (6.4 us per launch); now it is one round trip, two barriers and the per-channel arithmetic.
constexpr int kFinLanes = 64, kFinThreads = 8 * kFinLanes;
__device__ __forceinline__ void reduce_partials8(const float *__restrict__ partial, int nblk, int C, int c, bool ok,
                                                 double &s, double &s2) {
  __shared__ double sm[2][kFinThreads];
  const int t = threadIdx.x, kl = t >> 3;  // slab lane 0..63
  double a = 0.0, b = 0.0;
  if (ok) {
    for (int k0 = kl; k0 < nblk; k0 += 16 * kFinLanes) {
      float v0[16], v1[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        // unconditional loads from a clamped row (a predicated load is a branch the compiler will not issue loads across)
        const int k = k0 + u * kFinLanes, kc = k < nblk ? k : nblk - 1;
        v0[u] = partial[((size_t)kc * 2 + 0) * C + c];
        v1[u] = partial[((size_t)kc * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (k0 + u * kFinLanes < nblk) { a += (double)v0[u]; b += (double)v1[u]; }
    }
  }
  sm[0][t] = a;
  sm[1][t] = b;
  __syncthreads();
  double x = 0.0, y = 0.0;
  if (t < 64) {  // channel t & 7, lanes 8 * (t >> 3) ... + 7
    const int ch = t & 7, g = t >> 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x += sm[0][(g * 8 + j) * 8 + ch]; y += sm[1][(g * 8 + j) * 8 + ch]; }
  }
  __syncthreads();
  if (t < 64) { sm[0][t] = x; sm[1][t] = y; }
  __syncthreads();
  s = 0.0; s2 = 0.0;
  if (t < 8)
#pragma unroll
    for (int g = 0; g < 8; ++g) { s += sm[0][g * 8 + t]; s2 += sm[1][g * 8 + t]; }
}

// finalisation as its own launch: for partial sums that come from somewhere else (the conv epilogue, bfhip_bn2d_fwd_partials)
template <typename Fin>
__global__ __launch_bounds__(kFinThreads) void bn2d_finalize_kernel(const float *__restrict__ partial, int nblk, Fin fin) {
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  double s, s2;
  reduce_partials8(partial, nblk, fin.C, c, c < fin.C, s, s2);
  if (threadIdx.x >= 8 || c >= fin.C) return;
  fin(c, s, s2);
}

template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const T *__restrict__ x, const T *__restrict__ res,
                                                         const float *__restrict__ stats, long long M, int C, Map mp,
                                                         T *__restrict__ y, unsigned char *__restrict__ relu_mask) {
  // relu_mask (optional, bf16 + residual + ReLU): one BIT per element, [M][C / 8] bytes -- what the backward needs of y (y > 0);
  // reading it instead of y saves two passes over the tensor per residual layer (bfhip_bn2d_bwd_mask)
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
      unsigned bits = 0;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float o = v[u][j] * a[j] + b[j];
        if (RES) o += q[u][j];
        if (o > 0.f) bits |= 1u << j;
        v[u][j] = (RELU && !(o > 0.f)) ? 0.f : o;
      }
      Vec<T>::store(y + (size_t)(r + (long long)u * mp.R) * C + col, v[u]);
      if (V == 8 && relu_mask) relu_mask[(size_t)(r + (long long)u * mp.R) * (C >> 3) + cv] = (unsigned char)bits;
    }
  }
  for (; r < r1; r += mp.R) {
    float v[V], q[V];
    Vec<T>::load(x + (size_t)r * C + col, v);
    if (RES) Vec<T>::load(res + (size_t)r * C + col, q);
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float o = v[j] * a[j] + b[j];
      if (RES) o += q[j];
      if (o > 0.f) bits |= 1u << j;
      v[j] = (RELU && !(o > 0.f)) ? 0.f : o;
    }
    Vec<T>::store(y + (size_t)r * C + col, v);
    if (V == 8 && relu_mask) relu_mask[(size_t)r * (C >> 3) + cv] = (unsigned char)bits;
  }
}

// MASK: 0 = no ReLU, 1 = ReLU mask recomputed from x (a*x + b > 0), 2 = ReLU mask from the saved output y, 3 = from the bit mask
// the forward stored (yv[0] carries the byte of this 8-channel vector)
template <typename T, int MASK>
__device__ __forceinline__ void masked_grad(float *g, const float *xv, const float *yv, const float *a, const float *b) {
  constexpr int V = Vec<T>::V;
  const unsigned bits = MASK == 3 ? __float_as_uint(yv[0]) : 0u;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    if (MASK == 1 && !(xv[j] * a[j] + b[j] > 0.f)) g[j] = 0.f;
    if (MASK == 2 && !(yv[j] > 0.f)) g[j] = 0.f;
    if (MASK == 3 && !((bits >> j) & 1u)) g[j] = 0.f;
  }
}

// partial[blk][0][c] = sum g, partial[blk][1][c] = sum g * (x - mean)
template <typename T, int MASK>
__global__ __launch_bounds__(256) void bn2d_bwd_reduce_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                              const T *__restrict__ y,
                                                              const float *__restrict__ stats, long long M, int C,
                                                              Map mp, float *__restrict__ partial, Tree tr, BwdFin fin) {
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
        if (MASK == 3) yv[u][0] = __uint_as_float(((const unsigned char *)y)[(size_t)(r + (long long)u * mp.R) * (C >> 3) + cv]);
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
      if (MASK == 3) yv[0] = __uint_as_float(((const unsigned char *)y)[(size_t)r * (C >> 3) + cv]);
      masked_grad<T, MASK>(g, xv, yv, a, b);
#pragma unroll
      for (int j = 0; j < V; ++j) { s0[j] += g[j]; s1[j] += g[j] * (xv[j] - mean[j]); }
    }
  }
  combine_rows<V>(s0, s1, mp.Lb, mp.R, C, partial, sm);
  finish_in_last_block(partial, tr, C, blockIdx.y * mp.Lb * V, mp.Lb * V, fin);
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
      if (MASK == 3) yv[u][0] = __uint_as_float(((const unsigned char *)y)[(size_t)(r + (long long)u * mp.R) * (C >> 3) + cv]);
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
    if (MASK == 3) yv[0] = __uint_as_float(((const unsigned char *)y)[(size_t)r * (C >> 3) + cv]);
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
            hipStream_t s, unsigned char *relu_mask = nullptr) {
#define BFHIP_BN2D_APPLY(RES, RELU)                                                                             \
  hipLaunchKernelGGL((bn2d_apply_kernel<T, RES, RELU>), grid, dim3(256), 0, s, (const T *)x, (const T *)res, \
                     stats, M, C, mp, (T *)y, (RES && RELU) ? relu_mask : (unsigned char *)nullptr)
  if (res) { if (relu) BFHIP_BN2D_APPLY(true, true); else BFHIP_BN2D_APPLY(true, false); }
  else { if (relu) BFHIP_BN2D_APPLY(false, true); else BFHIP_BN2D_APPLY(false, false); }
#undef BFHIP_BN2D_APPLY
  return 0;
}

template <typename T, int MASK>
void run_bwd(const void *dy, const void *x, const void *y, const float *stats, long long M, int C, Map mp, dim3 grid,
             float *partial, float *coef, Tree tr, BwdFin fin, void *dx, void *dres, hipStream_t s) {
  hipLaunchKernelGGL((bn2d_bwd_reduce_kernel<T, MASK>), grid, dim3(256), 0, s, (const T *)dy, (const T *)x,
                     (const T *)y, stats, M, C, mp, partial, tr, fin);
  if (!tr.cnt) hipLaunchKernelGGL(bn2d_finalize_kernel<BwdFin>, dim3(ceil_div(C, 8)), dim3(kFinThreads), 0, s, partial, tr.nblk, fin);
  if (dres)
    hipLaunchKernelGGL((bn2d_bwd_apply_kernel<T, MASK, true>), grid, dim3(256), 0, s, (const T *)dy, (const T *)x,
                       (const T *)y, stats, coef, M, C, mp, (T *)dx, (T *)dres);
  else
    hipLaunchKernelGGL((bn2d_bwd_apply_kernel<T, MASK, false>), grid, dim3(256), 0, s, (const T *)dy,
                       (const T *)x, (const T *)y, stats, coef, M, C, mp, (T *)dx, (T *)nullptr);
}

// Arrival counters of finish_in_last_block: one zero-initialised slab per (device, stream), handed out from a pool that is
// allocated on the first call on a device (a warm-up step, never inside a stream capture), so that a stream first seen during
// a capture still gets its slab without an allocation.  Kernels of one stream run in order and leave their slab zeroed.
constexpr int kSlabInts = 16384;  // >= CT * (1 + ng) = 256 * 33
constexpr int kSlabs = 32;

inline bool fold_enabled() {
  static const bool on = [] { const char *e = getenv("BFHIP_BN2D_FOLD"); return e && e[0] == '1'; }();
  return on;
}

int *arrival_counters(hipStream_t stream) {
  static std::mutex mu;
  static std::map<int, int *> pool;                         // device -> kSlabs slabs
  static std::map<std::pair<int, hipStream_t>, int> slot;   // (device, stream) -> slab index
  static std::map<int, int> used;                           // device -> slabs handed out
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  auto it = pool.find(dev);
  if (it == pool.end()) {
    int *p = nullptr;
    const size_t bytes = (size_t)kSlabs * kSlabInts * sizeof(int);
    if (hipMalloc((void **)&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) {
      set_error("bn2d: cannot allocate the arrival counters (first call on a device must be outside a stream capture)");
      return nullptr;
    }
    it = pool.emplace(dev, p).first;
  }
  auto key = std::make_pair(dev, stream);
  auto st = slot.find(key);
  if (st == slot.end()) {
    // more streams than slabs: share the last slab round-robin is NOT safe, so refuse
    if (used[dev] >= kSlabs) { set_error("bn2d: more than %d streams per device in use", kSlabs); return nullptr; }
    st = slot.emplace(key, used[dev]++).first;
  }
  return it->second + (size_t)st->second * kSlabInts;
}

struct Plan {
  Map mp;
  dim3 grid;
  int nblk, ng;
  float *partial, *coef;
  double *gp;
};

inline size_t plan_bytes(long long M, int C, int dtype, Plan *pl, void *workspace) {
  pl->mp = make_map(M, C, dtype, &pl->grid);
  pl->nblk = (int)pl->grid.x;
  pl->ng = (pl->nblk + kGroup - 1) / kGroup;
  size_t off = 0;
  pl->partial = (float *)((char *)workspace + off); off += align_up((size_t)pl->nblk * 2 * C * sizeof(float), 256);
  pl->coef = (float *)((char *)workspace + off); off += align_up((size_t)3 * C * sizeof(float), 256);
  pl->gp = (double *)((char *)workspace + off); off += align_up((size_t)pl->ng * 2 * C * sizeof(double), 256);
  return off;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_bn2d_supported(long long M, int C, int dtype) {
  return (dtype == 0 || dtype == 1) && shape_ok(M, C, dtype) ? 1 : 0;
}

BFHIP_EXPORT size_t bfhip_bn2d_workspace_bytes(long long M, int C, int dtype) {
  if (!bfhip_bn2d_supported(M, C, dtype)) return 0;
  Plan pl;
  return plan_bytes(M, C, dtype, &pl, nullptr);
}

// out f32[C] = sum over the M rows of x [M][C] (f32 | bf16, dense): the statistics pass of the fused BatchNorm with a sum-only
// finalisation (fixed-order, fp64 combine).  Bias gradients are this: torch's strided reduction streams at ~1.1 TB/s here.
BFHIP_EXPORT int bfhip_colsum(const void *x, long long M, int C, int dtype, float *out, void *workspace, size_t workspace_bytes,
                              void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(bfhip_bn2d_supported(M, C, dtype), "colsum: unsupported shape M=%lld C=%d dtype=%d", M, C, dtype);
  BFHIP_REQUIRE(x && out && ((uintptr_t)x % 16) == 0, "colsum: null or misaligned pointer");
  if (!workspace || workspace_bytes < bfhip_bn2d_workspace_bytes(M, C, dtype)) { set_error("colsum: workspace too small"); return BFHIP_E_WORKSPACE; }
  Plan pl;
  plan_bytes(M, C, dtype, &pl, workspace);
  Tree tr{nullptr, pl.gp, pl.nblk, pl.ng};
  FwdFin none{};
  if (dtype == 1)
    hipLaunchKernelGGL(bn2d_stats_kernel<bf16_t>, pl.grid, dim3(256), 0, s, (const bf16_t *)x, M, C, pl.mp, pl.partial, tr, none);
  else
    hipLaunchKernelGGL(bn2d_stats_kernel<float>, pl.grid, dim3(256), 0, s, (const float *)x, M, C, pl.mp, pl.partial, tr, none);
  hipLaunchKernelGGL(bn2d_finalize_kernel<SumFin>, dim3(ceil_div(C, 8)), dim3(kFinThreads), 0, s, pl.partial, pl.nblk, SumFin{C, out});
  return check_launch("colsum");
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
  Plan pl;
  plan_bytes(M, C, dtype, &pl, workspace);
  int *cnt = fold_enabled() ? arrival_counters(s) : nullptr;
  if (!cnt && fold_enabled()) return BFHIP_E_LAUNCH;
  BFHIP_REQUIRE((long long)pl.grid.y * (1 + pl.ng) <= kSlabInts, "bn2d_fwd: too many column tiles");
  Tree tr{cnt, pl.gp, pl.nblk, pl.ng};
  FwdFin fin{M, m_dev, C, eps, momentum, gamma, beta, stats, running_mean, running_var};
  ProfScope ps;
  prof_begin(BFHIP_OP_BN2D_FWD, s, &ps);
  if (dtype == 1)
    hipLaunchKernelGGL(bn2d_stats_kernel<bf16_t>, pl.grid, dim3(256), 0, s, (const bf16_t *)x, M, C, pl.mp, pl.partial, tr, fin);
  else
    hipLaunchKernelGGL(bn2d_stats_kernel<float>, pl.grid, dim3(256), 0, s, (const float *)x, M, C, pl.mp, pl.partial, tr, fin);
  if (!cnt) hipLaunchKernelGGL(bn2d_finalize_kernel<FwdFin>, dim3(ceil_div(C, 8)), dim3(kFinThreads), 0, s, pl.partial, pl.nblk, fin);
  if (dtype == 1) run_fwd<bf16_t>(x, residual, stats, M, C, pl.mp, pl.grid, relu, y, s);
  else run_fwd<float>(x, residual, stats, M, C, pl.mp, pl.grid, relu, y, s);
  prof_end(&ps);
  return check_launch("bn2d_fwd");
}

static int bn2d_fwd_partials_impl(const void *x, const void *residual, const float *gamma, const float *beta, long long M, int C,
                                  int dtype, float eps, float momentum, int relu, float *running_mean, float *running_var,
                                  float *stats, void *y, const float *partial, int nblk, const int32_t *m_dev,
                                  unsigned char *relu_mask, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(!relu_mask || (dtype == 1 && relu && residual), "bn2d_fwd_partials: a ReLU bit mask needs bf16 + residual + ReLU");
  BFHIP_REQUIRE(bfhip_bn2d_supported(M, C, dtype), "bn2d_fwd_partials: unsupported shape M=%lld C=%d dtype=%d", M, C, dtype);
  BFHIP_REQUIRE(x && gamma && beta && stats && y && partial && nblk > 0, "bn2d_fwd_partials: null pointer");
  BFHIP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)residual % 16) == 0,
                "bn2d_fwd_partials: tensors must be 16-byte aligned");
  dim3 grid;
  Map mp = make_map(M, C, dtype, &grid);
  FwdFin fin{M, m_dev, C, eps, momentum, gamma, beta, stats, running_mean, running_var};
  ProfScope ps;
  prof_begin(BFHIP_OP_BN2D_FWD, s, &ps);
  hipLaunchKernelGGL(bn2d_finalize_kernel<FwdFin>, dim3(ceil_div(C, 8)), dim3(kFinThreads), 0, s, partial, nblk, fin);
  if (dtype == 1) run_fwd<bf16_t>(x, residual, stats, M, C, mp, grid, relu, y, s, relu_mask);
  else run_fwd<float>(x, residual, stats, M, C, mp, grid, relu, y, s);
  prof_end(&ps);
  return check_launch("bn2d_fwd_partials");
}

BFHIP_EXPORT int bfhip_bn2d_fwd_partials(const void *x, const void *residual, const float *gamma, const float *beta,
                                         long long M, int C, int dtype, float eps, float momentum, int relu,
                                         float *running_mean, float *running_var, float *stats, void *y,
                                         const float *partial, int nblk, const int32_t *m_dev, void *stream_) {
  return bn2d_fwd_partials_impl(x, residual, gamma, beta, M, C, dtype, eps, momentum, relu, running_mean, running_var, stats, y,
                                partial, nblk, m_dev, nullptr, stream_);
}

// ... and the ReLU decision of every element as one bit (relu_mask u8[M][C / 8]; bf16, residual + ReLU only): the backward
// (bfhip_bn2d_bwd_mask) reads the bits instead of the saved output
BFHIP_EXPORT int bfhip_bn2d_fwd_partials_mask(const void *x, const void *residual, const float *gamma, const float *beta,
                                              long long M, int C, int dtype, float eps, float momentum, int relu,
                                              float *running_mean, float *running_var, float *stats, void *y,
                                              const float *partial, int nblk, const int32_t *m_dev, unsigned char *relu_mask,
                                              void *stream_) {
  BFHIP_REQUIRE(relu_mask, "bn2d_fwd_partials_mask: null mask");
  return bn2d_fwd_partials_impl(x, residual, gamma, beta, M, C, dtype, eps, momentum, relu, running_mean, running_var, stats, y,
                                partial, nblk, m_dev, relu_mask, stream_);
}

static int bn2d_bwd_impl(const void *dy, const void *x, const void *y, const unsigned char *relu_mask, const float *stats,
                         const float *gamma, long long M, int C, int dtype, int relu, void *dx, void *dres, float *dgb,
                         const int32_t *m_dev, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(!relu_mask || (dtype == 1 && relu), "bn2d_bwd: a ReLU bit mask needs bf16 + ReLU");
  if (relu_mask) y = relu_mask;  // the kernels' MASK == 3 mode reads the bits through the y pointer
  BFHIP_REQUIRE(bfhip_bn2d_supported(M, C, dtype), "bn2d_bwd: unsupported shape M=%lld C=%d dtype=%d", M, C, dtype);
  BFHIP_REQUIRE(dy && x && stats && gamma && dx && dgb, "bn2d_bwd: null pointer");
  BFHIP_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
                    ((uintptr_t)y % 16) == 0 && ((uintptr_t)dres % 16) == 0, "bn2d_bwd: tensors must be 16-byte aligned");
  if (!workspace || workspace_bytes < bfhip_bn2d_workspace_bytes(M, C, dtype)) { set_error("bn2d_bwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  Plan pl;
  plan_bytes(M, C, dtype, &pl, workspace);
  int *cnt = fold_enabled() ? arrival_counters(s) : nullptr;
  if (!cnt && fold_enabled()) return BFHIP_E_LAUNCH;
  BFHIP_REQUIRE((long long)pl.grid.y * (1 + pl.ng) <= kSlabInts, "bn2d_bwd: too many column tiles");
  Tree tr{cnt, pl.gp, pl.nblk, pl.ng};
  BwdFin fin{M, m_dev, C, stats, dgb, pl.coef};
  // ReLU mask: from the saved output when one is given (residual layers), otherwise recomputed from x
  const int mask = !relu ? 0 : (relu_mask ? 3 : (y ? 2 : 1));
  ProfScope ps;
  prof_begin(BFHIP_OP_BN2D_BWD, s, &ps);
#define BFHIP_BN2D_BWD(T, MASK) run_bwd<T, MASK>(dy, x, y, stats, M, C, pl.mp, pl.grid, pl.partial, pl.coef, tr, fin, dx, dres, s)
  if (dtype == 1) {
    if (mask == 0) BFHIP_BN2D_BWD(bf16_t, 0);
    else if (mask == 1) BFHIP_BN2D_BWD(bf16_t, 1);
    else if (mask == 3) BFHIP_BN2D_BWD(bf16_t, 3);
    else BFHIP_BN2D_BWD(bf16_t, 2);
  } else {
    if (mask == 0) BFHIP_BN2D_BWD(float, 0);
    else if (mask == 1) BFHIP_BN2D_BWD(float, 1);
    else BFHIP_BN2D_BWD(float, 2);
  }
#undef BFHIP_BN2D_BWD
  prof_end(&ps);
  return check_launch("bn2d_bwd");
}

BFHIP_EXPORT int bfhip_bn2d_bwd(const void *dy, const void *x, const void *y, const float *stats, const float *gamma,
                                long long M, int C, int dtype, int relu, void *dx, void *dres, float *dgb,
                                const int32_t *m_dev, void *workspace, size_t workspace_bytes, void *stream_) {
  return bn2d_bwd_impl(dy, x, y, nullptr, stats, gamma, M, C, dtype, relu, dx, dres, dgb, m_dev, workspace, workspace_bytes, stream_);
}

// the backward of a bf16 residual + ReLU layer from the forward's bit mask (bfhip_bn2d_fwd_partials_mask) instead of the saved
// output: 1/16 of the bytes, in both of its passes
BFHIP_EXPORT int bfhip_bn2d_bwd_mask(const void *dy, const void *x, const unsigned char *relu_mask, const float *stats,
                                     const float *gamma, long long M, int C, int dtype, void *dx, void *dres, float *dgb,
                                     const int32_t *m_dev, void *workspace, size_t workspace_bytes, void *stream_) {
  BFHIP_REQUIRE(relu_mask, "bn2d_bwd_mask: null mask");
  return bn2d_bwd_impl(dy, x, nullptr, relu_mask, stats, gamma, M, C, dtype, 1, dx, dres, dgb, m_dev, workspace, workspace_bytes,
                       stream_);
}
