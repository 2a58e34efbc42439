// bn1d.hip -- BatchNorm1d (+ residual add) (+ ReLU) on sparse-tensor feature matrices f32[N, C], fwd + bwd.
//
// In the reference every sparse conv is followed by BatchNorm1d(eps 1e-3, momentum 0.01) and ReLU, and the
// SparseBasicBlock adds the identity before its last ReLU (mmdet3d/models/layers/sparse_block.py:135-154,
// make_sparse_convmodule :157-224); torch runs that as 4-6 generic kernels per layer that are latency-bound at
// these sizes (N = 26 k - 136 k rows, C = 16 - 128: ~48 us for a 17 MB column reduction).  Here:
//   forward : column sum / sum of squares per 64-row slab (fp32 partials, fp64 combine) -> mean, biased var,
//             running-stat update -> y = relu(gamma * (x - mean) * invstd + beta [+ residual])
//   backward: g = dy * (y > 0);  dbeta = sum g, dgamma = sum g * xhat  (slab partials, fp64 combine)
//             dx = gamma * invstd * (g - dbeta/N - xhat * dgamma/N);  d_residual = g
// Two-stage reductions in a fixed order: deterministic, no atomics.
#include "common.h"

namespace bfhip {
namespace {

constexpr int kSlab = 64;  // rows per block (many small slabs: the reductions are latency-bound, not byte-bound)

// partial[blk][0][c] = sum_x, partial[blk][1][c] = sum_x2 over the slab.  256 threads: (256 / C) row lanes x C channels
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ x, int N, int C,
                                                       float *__restrict__ partial) {
  __shared__ float sm[2][256];
  const int c = threadIdx.x % C, rl = threadIdx.x / C, nrl = 256 / C;
  const int r0 = blockIdx.x * kSlab, r1 = min(N, r0 + kSlab);
  float s = 0.f, s2 = 0.f;
  if (rl < nrl)
    for (int r = r0 + rl; r < r1; r += nrl) {
      float v = x[(size_t)r * C + c];
      s += v;
      s2 += v * v;
    }
  sm[0][threadIdx.x] = s;
  sm[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < C) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < nrl; ++k) { a += sm[0][k * C + threadIdx.x]; b += sm[1][k * C + threadIdx.x]; }
    partial[((size_t)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
    partial[((size_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

// one block per channel: 256 threads sum the slab partials in a fixed order (thread t takes slabs t, t+256, ...;
// then a fixed LDS tree), fp64.  returns (sum0, sum1) in every thread.
__device__ __forceinline__ void reduce_partials(const float *__restrict__ partial, int nblk, int C, int c,
                                                double &s, double &s2) {
  __shared__ double sm[2][256];
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 256) {
    a += (double)partial[((size_t)k * 2 + 0) * C + c];
    b += (double)partial[((size_t)k * 2 + 1) * C + c];
  }
  sm[0][threadIdx.x] = a;
  sm[1][threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sm[0][threadIdx.x] += sm[0][threadIdx.x + o];
      sm[1][threadIdx.x] += sm[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  s = sm[0][0];
  s2 = sm[1][0];
}

// stats[0][c] = mean, stats[1][c] = invstd; running stats updated like torch (unbiased var in running_var)
// n_dev (optional): the number of ACTIVE rows lives on the device; rows beyond it are exact zeros (spconv.py, static
// capacity mode), so only the divisor changes
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *__restrict__ partial, int nblk, int N,
                                                          const int *__restrict__ n_dev, int C, float eps, float momentum,
                                                          float *__restrict__ stats,
                                                          float *__restrict__ running_mean,
                                                          float *__restrict__ running_var) {
  const int c = blockIdx.x;
  double s, s2;
  reduce_partials(partial, nblk, C, c, s, s2);
  if (threadIdx.x != 0) return;
  if (n_dev) { int nv = *n_dev; N = nv < 1 ? 1 : (nv < N ? nv : N); }
  double mean = s / N;
  double var = s2 / N - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float4 *__restrict__ x,
                                                       const float4 *__restrict__ residual,
                                                       const float *__restrict__ stats,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, long long total4,
                                                       int C, int relu, float4 *__restrict__ y) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total4) return;
  const int c = (int)((t * 4) % C);
  float4 v = x[t];
  float o[4] = {v.x, v.y, v.z, v.w};
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  if (residual) { float4 q = residual[t]; r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w; }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a = gamma[c + j] * stats[C + c + j];
    float val = (o[j] - stats[c + j]) * a + beta[c + j] + r[j];
    o[j] = (relu && val < 0.f) ? 0.f : val;
  }
  y[t] = make_float4(o[0], o[1], o[2], o[3]);
}

// partial[blk][0][c] = sum g, partial[blk][1][c] = sum g * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ stats, int N, int C,
                                                            int relu, float *__restrict__ partial) {
  __shared__ float sm[2][256];
  const int c = threadIdx.x % C, rl = threadIdx.x / C, nrl = 256 / C;
  const int r0 = blockIdx.x * kSlab, r1 = min(N, r0 + kSlab);
  float s = 0.f, s2 = 0.f;
  if (rl < nrl) {
    const float mean = stats[c], invstd = stats[C + c];
    for (int r = r0 + rl; r < r1; r += nrl) {
      size_t i = (size_t)r * C + c;
      float g = dy[i];
      if (relu && !(y[i] > 0.f)) g = 0.f;
      s += g;
      s2 += g * ((x[i] - mean) * invstd);
    }
  }
  sm[0][threadIdx.x] = s;
  sm[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < C) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < nrl; ++k) { a += sm[0][k * C + threadIdx.x]; b += sm[1][k * C + threadIdx.x]; }
    partial[((size_t)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
    partial[((size_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

// dgb[0][c] = dgamma, dgb[1][c] = dbeta
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int nblk, int C,
                                                              float *__restrict__ dgb) {
  const int c = blockIdx.x;
  double s, s2;
  reduce_partials(partial, nblk, C, c, s, s2);
  if (threadIdx.x == 0) {
    dgb[c] = (float)s2;
    dgb[C + c] = (float)s;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ y,
                                                           const float4 *__restrict__ x,
                                                           const float *__restrict__ stats,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ dgb, long long total4, int C,
                                                           int N, const int *__restrict__ n_dev, int relu,
                                                           float4 *__restrict__ dx, float4 *__restrict__ dres) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total4) return;
  if (n_dev) { int nv = *n_dev; N = nv < 1 ? 1 : (nv < N ? nv : N); }
  const int c = (int)((t * 4) % C);
  float4 gv = dy[t], yv = y[t], xv = x[t];
  float g[4] = {gv.x, gv.y, gv.z, gv.w};
  const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
  const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
  float o[4];
  const float invN = 1.0f / (float)N;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (relu && !(yy[j] > 0.f)) g[j] = 0.f;
    const float invstd = stats[C + c + j];
    const float xhat = (xx[j] - stats[c + j]) * invstd;
    o[j] = gamma[c + j] * invstd * (g[j] - dgb[C + c + j] * invN - xhat * dgb[c + j] * invN);
  }
  dx[t] = make_float4(o[0], o[1], o[2], o[3]);
  if (dres) dres[t] = make_float4(g[0], g[1], g[2], g[3]);
}

inline bool shape_ok(int N, int C) { return N > 0 && C >= 4 && C <= 256 && (C % 4 == 0) && (256 % C == 0); }

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT size_t bfhip_bn1d_workspace_bytes(int N, int C) {
  size_t nblk = (size_t)ceil_div(N > 0 ? N : 1, kSlab);
  return align_up(nblk * 2 * (size_t)C * sizeof(float), 256) + 256;
}

// Training-mode forward.  stats f32[2*C] receives (mean, invstd) for the backward; running_mean/var may be NULL.
BFHIP_EXPORT int bfhip_bn1d_fwd(const float *x, const float *residual, const float *gamma, const float *beta, int N,
                                int C, float eps, float momentum, int relu, float *running_mean,
                                float *running_var, float *stats, float *y, const int32_t *n_rows_dev, void *workspace,
                                size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(shape_ok(N, C), "bn1d_fwd: needs N > 0 and C in {4,8,16,32,64,128,256} (N=%d C=%d)", N, C);
  BFHIP_REQUIRE(x && gamma && beta && stats && y, "bn1d_fwd: null pointer");
  BFHIP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (!residual || ((uintptr_t)residual % 16) == 0),
                "bn1d_fwd: tensors must be 16-byte aligned");
  if (workspace_bytes < bfhip_bn1d_workspace_bytes(N, C) || !workspace) { set_error("bn1d_fwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  float *partial = (float *)workspace;
  int nblk = ceil_div(N, kSlab);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(256), 0, stream, x, N, C, partial);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, stream, partial, nblk, N, n_rows_dev, C, eps, momentum,
                     stats, running_mean, running_var);
  long long total4 = (long long)N * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ceil_div(total4, 256)), dim3(256), 0, stream, (const float4 *)x,
                     (const float4 *)residual, stats, gamma, beta, total4, C, relu, (float4 *)y);
  return check_launch("bn1d_fwd");
}

// Backward.  dgb f32[2*C] receives (dgamma, dbeta); dres (optional) the gradient of the residual input.
BFHIP_EXPORT int bfhip_bn1d_bwd(const float *dy, const float *y, const float *x, const float *stats,
                                const float *gamma, int N, int C, int relu, float *dx, float *dres, float *dgb,
                                const int32_t *n_rows_dev, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(shape_ok(N, C), "bn1d_bwd: needs N > 0 and C in {4,8,16,32,64,128,256} (N=%d C=%d)", N, C);
  BFHIP_REQUIRE(dy && y && x && stats && gamma && dx && dgb, "bn1d_bwd: null pointer");
  if (workspace_bytes < bfhip_bn1d_workspace_bytes(N, C) || !workspace) { set_error("bn1d_bwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  float *partial = (float *)workspace;
  int nblk = ceil_div(N, kSlab);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, stream, dy, y, x, stats, N, C, relu, partial);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, partial, nblk, C, dgb);
  long long total4 = (long long)N * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ceil_div(total4, 256)), dim3(256), 0, stream, (const float4 *)dy,
                     (const float4 *)y, (const float4 *)x, stats, gamma, dgb, total4, C, N, n_rows_dev, relu, (float4 *)dx,
                     (float4 *)dres);
  return check_launch("bn1d_bwd");
}
