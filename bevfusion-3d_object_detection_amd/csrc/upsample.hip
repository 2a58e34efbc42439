// upsample.hip -- bilinear 2x upsampling of channels-last feature maps, forward and backward (gfx950).
// The LSS-FPN upsamples the coarser backbone level before every lateral conv (BF/bevfusion_necks.py:76-88,
// F.interpolate(mode='bilinear', align_corners=False)).  The library backward scatters with atomics (0.33 ms per call in
// fp32, 1.2 ms in bf16 for [24, 2048, 16, 44]); for an exact 2x factor every input pixel receives from at most 3 x 3 output
// pixels with fixed weights, so the backward is a GATHER: no atomics, deterministic, one pass.
// Index rule = torch's (align_corners=False): src = max(0, (o + 0.5) / 2 - 0.5), i0 = floor(src), i1 = min(i0 + 1, n - 1).
#include "common.h"

namespace bfhip {
namespace {

typedef unsigned short bf16_t;

template <typename T> struct V;
template <> struct V<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void ld(const float *p, float *o) { float4 v = *(const float4 *)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  static __device__ __forceinline__ void st(float *p, const float *o) { *(float4 *)p = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct V<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void ld(const bf16_t *p, float *o) {
    uint4 v = *(const uint4 *)p;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ unsigned r(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  }
  static __device__ __forceinline__ void st(bf16_t *p, const float *o) {
    uint4 v;
    v.x = r(o[0]) | (r(o[1]) << 16); v.y = r(o[2]) | (r(o[3]) << 16);
    v.z = r(o[4]) | (r(o[5]) << 16); v.w = r(o[6]) | (r(o[7]) << 16);
    *(uint4 *)p = v;
  }
};

__device__ __forceinline__ void src_of(int o, int n, int &i0, int &i1, float &w0, float &w1) {
  float s = fmaxf(0.f, ((float)o + 0.5f) * 0.5f - 0.5f);
  i0 = (int)s;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  w1 = s - (float)i0;
  w0 = 1.f - w1;
}

// in [B, H, W, C] -> out [B, 2H, 2W, C]; one thread per (output pixel, channel vector)
template <typename T>
__global__ __launch_bounds__(256) void up2x_fwd_kernel(const T *__restrict__ in, int B, int H, int W, int C,
                                                       T *__restrict__ out) {
  constexpr int N = V<T>::N;
  const int cv = C / N;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * 2 * H * 2 * W * cv) return;
  int c = (int)(t % cv) * N;
  long long p = t / cv;
  int ox = (int)(p % (2 * W)); p /= 2 * W;
  int oy = (int)(p % (2 * H));
  int b = (int)(p / (2 * H));
  int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
  src_of(oy, H, y0, y1, wy0, wy1);
  src_of(ox, W, x0, x1, wx0, wx1);
  const T *base = in + (size_t)b * H * W * C + c;
  float a[N], q[N], acc[N];
  V<T>::ld(base + ((size_t)y0 * W + x0) * C, a);
  V<T>::ld(base + ((size_t)y0 * W + x1) * C, q);
#pragma unroll
  for (int j = 0; j < N; ++j) acc[j] = wy0 * (wx0 * a[j] + wx1 * q[j]);
  V<T>::ld(base + ((size_t)y1 * W + x0) * C, a);
  V<T>::ld(base + ((size_t)y1 * W + x1) * C, q);
#pragma unroll
  for (int j = 0; j < N; ++j) acc[j] += wy1 * (wx0 * a[j] + wx1 * q[j]);
  V<T>::st(out + (((size_t)b * 2 * H + oy) * 2 * W + ox) * C + c, acc);
}

// gin [B, H, W, C] <- gout [B, 2H, 2W, C]; one thread per (input pixel, channel vector) gathers its <= 5 x 5 candidates
template <typename T>
__global__ __launch_bounds__(256) void up2x_bwd_kernel(const T *__restrict__ gout, int B, int H, int W, int C,
                                                       T *__restrict__ gin) {
  constexpr int N = V<T>::N;
  const int cv = C / N;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * H * W * cv) return;
  int c = (int)(t % cv) * N;
  long long p = t / cv;
  int ix = (int)(p % W); p /= W;
  int iy = (int)(p % H);
  int b = (int)(p / H);
  float acc[N];
#pragma unroll
  for (int j = 0; j < N; ++j) acc[j] = 0.f;
  const T *base = gout + (size_t)b * 4 * H * W * C + c;
  for (int oy = max(0, 2 * iy - 2); oy <= min(2 * H - 1, 2 * iy + 2); ++oy) {
    int y0, y1; float wy0, wy1;
    src_of(oy, H, y0, y1, wy0, wy1);
    float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
    if (wy == 0.f) continue;
    for (int ox = max(0, 2 * ix - 2); ox <= min(2 * W - 1, 2 * ix + 2); ++ox) {
      int x0, x1; float wx0, wx1;
      src_of(ox, W, x0, x1, wx0, wx1);
      float w = wy * ((x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f));
      if (w == 0.f) continue;
      float g[N];
      V<T>::ld(base + ((size_t)oy * 2 * W + ox) * C, g);
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += w * g[j];
    }
  }
  V<T>::st(gin + (((size_t)b * H + iy) * W + ix) * C + c, acc);
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

// dir 0: forward (src = in [B,H,W,C], dst = out [B,2H,2W,C]); dir 1: backward (src = grad_out [B,2H,2W,C], dst = grad_in)
BFHIP_EXPORT int bfhip_upsample2x_nhwc(const void *src, void *dst, int B, int H, int W, int C, int dtype, int dir,
                                       void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  const int N = dtype == 1 ? 8 : 4;
  BFHIP_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && (dtype == 0 || dtype == 1) && (dir == 0 || dir == 1),
                "upsample2x: bad arguments");
  BFHIP_REQUIRE(C % N == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                "upsample2x: C must be a multiple of the 16-byte vector and the tensors 16-byte aligned (C=%d)", C);
  const long long n = (long long)B * H * W * (C / N) * (dir == 0 ? 4 : 1);
  dim3 grid(ceil_div(n, 256));
  if (dtype == 1) {
    if (dir == 0) hipLaunchKernelGGL(up2x_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t *)src, B, H, W, C, (bf16_t *)dst);
    else hipLaunchKernelGGL(up2x_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t *)src, B, H, W, C, (bf16_t *)dst);
  } else {
    if (dir == 0) hipLaunchKernelGGL(up2x_fwd_kernel<float>, grid, dim3(256), 0, s, (const float *)src, B, H, W, C, (float *)dst);
    else hipLaunchKernelGGL(up2x_bwd_kernel<float>, grid, dim3(256), 0, s, (const float *)src, B, H, W, C, (float *)dst);
  }
  return check_launch("upsample2x");
}
