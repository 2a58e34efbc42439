// bev_pool.hip -- sorted-interval sum pooling of frustum features into the BEV grid (gfx950).
//
// Replaces the reference kernels BF/ops/bev_pool/src/bev_pool_cuda.cu:20-42 (fwd) and :61-84
// (bwd).  The reference runs one thread per (interval, channel) with a serial loop over the
// interval; here one lane owns FOUR channels (one 16-B float4) of one interval, so a C=80 row is
// covered by 20 adjacent lanes and a wave64 carries three intervals (60 lanes, 4 idle).  Every
// lane keeps up to 16 independent 16-B loads in flight and adds them in row order, so the
// per-(interval, channel) fp32 sum is formed in exactly the reference's order (bit-identical
// result) while the loads are pipelined.  HBM-bound: x is streamed once (n*c*4 bytes), out is
// written once after a memset.
#include "common.h"

namespace bfhip {
namespace {

constexpr int kUnroll = 16;

__device__ __forceinline__ size_t cell_offset(int4 g, int d, int h, int w) {
  // geom row layout (x, y, z, b) -> out[b][z][x][y]   (bev_pool_cuda.cu:34-36)
  return (((size_t)g.w * d + g.z) * h + g.x) * w + g.y;
}

// cq = c/4 float4 per row; groups = 64/cq intervals per wave.
__global__ __launch_bounds__(256) void bev_pool_fwd_v4(
    const float4 *__restrict__ x, const int4 *__restrict__ geom, const int *__restrict__ starts,
    const int *__restrict__ lengths, float4 *__restrict__ out, int m, int cq, int groups, int d,
    int h, int w, const int *__restrict__ m_dev) {
  if (m_dev) m = min(m, *m_dev);
  const int lane = threadIdx.x & (kWave - 1);
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int g = lane / cq;
  const int q = lane - g * cq;
  const long long k = wave * groups + g;
  if (g >= groups || k >= m) return;
  const int s = starts[k];
  const int len = lengths[k];
  const int4 gm = geom[s];
  const float4 *px = x + (size_t)s * cq + q;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int i = 0;
  // full chunks: kUnroll independent 16-B loads in flight, added in row order
  for (; i + kUnroll <= len; i += kUnroll) {
    float4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = px[(size_t)(i + u) * cq];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
    }
  }
  // remainder (< kUnroll rows) in chunks of 4 with the last chunk clamped: at most 3 redundant loads per
  // interval (the mean interval is ~19 rows, so a single clamped kUnroll-chunk would double the load count)
  for (; i < len; i += 4) {
    float4 v[4];
    const int rem = len - i;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = px[(size_t)(i + (u < rem ? u : rem - 1)) * cq];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u < rem) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  out[cell_offset(gm, d, h, w) * cq + q] = acc;
}

// generic channel count: one thread per (interval, channel), the reference's own mapping.
__global__ __launch_bounds__(256) void bev_pool_fwd_scalar(
    const float *__restrict__ x, const int4 *__restrict__ geom, const int *__restrict__ starts,
    const int *__restrict__ lengths, float *__restrict__ out, int m, int c, int d, int h, int w,
    const int *__restrict__ m_dev) {
  if (m_dev) m = min(m, *m_dev);
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long k = idx / c;
  const int ch = (int)(idx - k * c);
  if (k >= m) return;
  const int s = starts[k];
  const int len = lengths[k];
  const int4 gm = geom[s];
  const float *px = x + (size_t)s * c + ch;
  float psum = 0.f;
  for (int i = 0; i < len; ++i) psum += px[(size_t)i * c];
  out[cell_offset(gm, d, h, w) * c + ch] = psum;
}

__global__ __launch_bounds__(256) void bev_pool_bwd_v4(
    const float4 *__restrict__ out_grad, const int4 *__restrict__ geom,
    const int *__restrict__ starts, const int *__restrict__ lengths, float4 *__restrict__ x_grad,
    int m, int cq, int groups, int d, int h, int w, const int *__restrict__ m_dev) {
  if (m_dev) m = min(m, *m_dev);
  const int lane = threadIdx.x & (kWave - 1);
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int g = lane / cq;
  const int q = lane - g * cq;
  const long long k = wave * groups + g;
  if (g >= groups || k >= m) return;
  const int s = starts[k];
  const int len = lengths[k];
  const int4 gm = geom[s];
  const float4 gval = out_grad[cell_offset(gm, d, h, w) * cq + q];
  float4 *px = x_grad + (size_t)s * cq + q;
  for (int i = 0; i < len; ++i) px[(size_t)i * cq] = gval;
}

__global__ __launch_bounds__(256) void bev_pool_bwd_scalar(
    const float *__restrict__ out_grad, const int4 *__restrict__ geom,
    const int *__restrict__ starts, const int *__restrict__ lengths, float *__restrict__ x_grad,
    int m, int c, int d, int h, int w, const int *__restrict__ m_dev) {
  if (m_dev) m = min(m, *m_dev);
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long k = idx / c;
  const int ch = (int)(idx - k * c);
  if (k >= m) return;
  const int s = starts[k];
  const int len = lengths[k];
  const int4 gm = geom[s];
  const float gval = out_grad[cell_offset(gm, d, h, w) * c + ch];
  float *px = x_grad + (size_t)s * c + ch;
  for (int i = 0; i < len; ++i) px[(size_t)i * c] = gval;
}

inline bool vec4_ok(int c, const void *a, const void *b) {
  return c % 4 == 0 && c / 4 <= kWave && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_bev_pool_fwd(const float *x, const int32_t *geom, const int32_t *starts,
                                    const int32_t *lengths, float *out, int n, int c, int m,
                                    int b, int d, int h, int w, const int32_t *m_dev,
                                    void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(n >= 0 && m >= 0 && c > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                "bev_pool_fwd: bad sizes n=%d c=%d m=%d b=%d d=%d h=%d w=%d", n, c, m, b, d, h, w);
  BFHIP_REQUIRE(out != nullptr, "bev_pool_fwd: out is null");
  BFHIP_REQUIRE(((uintptr_t)geom % 16) == 0, "bev_pool_fwd: geom must be 16-byte aligned");
  size_t out_bytes = (size_t)b * d * h * w * c * sizeof(float);
  if (hipMemsetAsync(out, 0, out_bytes, stream) != hipSuccess) return check_launch("bev_pool_fwd memset");
  if (m == 0 || n == 0) return BFHIP_OK;
  BFHIP_REQUIRE(x && geom && starts && lengths, "bev_pool_fwd: null input");
  ProfScope ps;
  prof_begin(BFHIP_OP_BEV_POOL_FWD, stream, &ps);
  if (vec4_ok(c, x, out)) {
    int cq = c / 4, groups = kWave / cq;
    long long waves = ((long long)m + groups - 1) / groups;
    int blocks = ceil_div(waves * kWave, 256);
    hipLaunchKernelGGL(bev_pool_fwd_v4, dim3(blocks), dim3(256), 0, stream, (const float4 *)x,
                       (const int4 *)geom, starts, lengths, (float4 *)out, m, cq, groups, d, h, w,
                       m_dev);
  } else {
    int blocks = ceil_div((long long)m * c, 256);
    hipLaunchKernelGGL(bev_pool_fwd_scalar, dim3(blocks), dim3(256), 0, stream, x,
                       (const int4 *)geom, starts, lengths, out, m, c, d, h, w, m_dev);
  }
  prof_end(&ps);
  return check_launch("bev_pool_fwd");
}

BFHIP_EXPORT int bfhip_bev_pool_bwd(const float *out_grad, const int32_t *geom,
                                    const int32_t *starts, const int32_t *lengths, float *x_grad,
                                    int n, int c, int m, int b, int d, int h, int w,
                                    int intervals_cover_all_rows, const int32_t *m_dev,
                                    void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(n >= 0 && m >= 0 && c > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                "bev_pool_bwd: bad sizes n=%d c=%d m=%d b=%d d=%d h=%d w=%d", n, c, m, b, d, h, w);
  if (n == 0) return BFHIP_OK;
  BFHIP_REQUIRE(x_grad != nullptr, "bev_pool_bwd: x_grad is null");
  BFHIP_REQUIRE(((uintptr_t)geom % 16) == 0, "bev_pool_bwd: geom must be 16-byte aligned");
  if (!intervals_cover_all_rows || m == 0 || m_dev) {
    if (hipMemsetAsync(x_grad, 0, (size_t)n * c * sizeof(float), stream) != hipSuccess)
      return check_launch("bev_pool_bwd memset");
  }
  if (m == 0) return BFHIP_OK;
  BFHIP_REQUIRE(out_grad && geom && starts && lengths, "bev_pool_bwd: null input");
  ProfScope ps;
  prof_begin(BFHIP_OP_BEV_POOL_BWD, stream, &ps);
  if (vec4_ok(c, out_grad, x_grad)) {
    int cq = c / 4, groups = kWave / cq;
    long long waves = ((long long)m + groups - 1) / groups;
    int blocks = ceil_div(waves * kWave, 256);
    hipLaunchKernelGGL(bev_pool_bwd_v4, dim3(blocks), dim3(256), 0, stream,
                       (const float4 *)out_grad, (const int4 *)geom, starts, lengths,
                       (float4 *)x_grad, m, cq, groups, d, h, w, m_dev);
  } else {
    int blocks = ceil_div((long long)m * c, 256);
    hipLaunchKernelGGL(bev_pool_bwd_scalar, dim3(blocks), dim3(256), 0, stream, out_grad,
                       (const int4 *)geom, starts, lengths, x_grad, m, c, d, h, w, m_dev);
  }
  prof_end(&ps);
  return check_launch("bev_pool_bwd");
}
