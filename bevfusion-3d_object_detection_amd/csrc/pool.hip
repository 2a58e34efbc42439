// pool.hip -- 3x3 stride-2 pad-1 max pooling of a channels-last bf16 map, forward (with argmax) and backward (gfx950).
//
// The ResNet-50 stem pools its [24, 64, 128, 352] activation once per step (img_backbone: mmdet.ResNet, an external dependency
// of the reference; `nn.MaxPool2d(kernel_size=3, stride=2, padding=1)` as in torchvision).  The library's channels-last
// kernels take 0.10 ms forward and 0.25 ms backward for it (the backward walks every output window per input element); here
// a thread owns one 16-byte channel vector of one pixel:
//   forward : 9 vector loads, running max in torch's scan order (kh, kw ascending; a later element replaces the maximum only
//             if it is greater or NaN -- so ties keep the FIRST maximum and NaNs propagate, like at::max_pool2d), the winning
//             tap (0..8) of every channel stored as one byte
//   backward: a GATHER -- an input pixel lies in at most 2 x 2 windows; for each, 8 tap bytes + 8 gradients are loaded and the
//             gradients whose tap points back at this pixel are added (fp32, windows in ascending (oh, ow) order); no atomics,
//             deterministic, every byte of dx written exactly once (no zero-fill pass).
#include "common.h"

namespace bfhip {
namespace {

typedef unsigned short bf16_t;

__device__ __forceinline__ void ld8(const bf16_t *p, float *o) {
  const uint4 v = *(const uint4 *)p;
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

__device__ __forceinline__ unsigned rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// x [N, H, W, C] -> y [N, OH, OW, C], tap [N, OH, OW, C] (u8); OH = (H - 1) / 2 + 1
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t *__restrict__ x, int N, int H, int W, int C, int OH, int OW,
                                                          bf16_t *__restrict__ y, unsigned char *__restrict__ tap) {
  const int cv = C >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * OH * OW * cv) return;
  const int c = (int)(t % cv) << 3;
  long long p = t / cv;
  const int ow = (int)(p % OW); p /= OW;
  const int oh = (int)(p % OH);
  const int n = (int)(p / OH);
  const bf16_t *base = x + (size_t)n * H * W * C + c;
  // the maximum is kept as the 16-bit pattern of the winning element (max of bf16 values IS one of them): no re-rounding
  float m[8];
  unsigned short mb[8];
  unsigned char k[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; mb[j] = 0xff80u; k[j] = 0; }
  bool first = true;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oh * 2 - 1 + kh;
    if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ow * 2 - 1 + kw;
      if ((unsigned)iw >= (unsigned)W) continue;
      const uint4 raw = *(const uint4 *)(base + ((size_t)ih * W + iw) * C);
      const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned short b = (unsigned short)((j & 1) ? (w4[j >> 1] >> 16) : (w4[j >> 1] & 0xffffu));
        const float v = __uint_as_float((unsigned)b << 16);
        // torch: the first in-range element initialises the maximum; later ones replace it if greater or NaN
        if (first || v > m[j] || v != v) { m[j] = v; mb[j] = b; k[j] = (unsigned char)(kh * 3 + kw); }
      }
      first = false;
    }
  }
  uint4 o;
  o.x = mb[0] | ((unsigned)mb[1] << 16); o.y = mb[2] | ((unsigned)mb[3] << 16);
  o.z = mb[4] | ((unsigned)mb[5] << 16); o.w = mb[6] | ((unsigned)mb[7] << 16);
  const size_t off = (((size_t)n * OH + oh) * OW + ow) * C + c;
  *(uint4 *)(y + off) = o;
  uint2 kk;
  kk.x = k[0] | (k[1] << 8) | (k[2] << 16) | ((unsigned)k[3] << 24);
  kk.y = k[4] | (k[5] << 8) | (k[6] << 16) | ((unsigned)k[7] << 24);
  *(uint2 *)(tap + off) = kk;
}

// dx [N, H, W, C] <- dy [N, OH, OW, C], tap
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t *__restrict__ dy, const unsigned char *__restrict__ tap,
                                                          int N, int H, int W, int C, int OH, int OW, bf16_t *__restrict__ dx) {
  const int cv = C >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * H * W * cv) return;
  const int c = (int)(t % cv) << 3;
  long long p = t / cv;
  const int w = (int)(p % W); p /= W;
  const int h = (int)(p % H);
  const int n = (int)(p / H);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  // windows that contain row h: 2*oh - 1 <= h <= 2*oh + 1  <=>  oh in [h / 2, (h + 1) / 2]
  const int oh0 = h >> 1, oh1 = (h + 1) >> 1, ow0 = w >> 1, ow1 = (w + 1) >> 1;
  for (int oh = oh0; oh <= oh1; ++oh) {
    if (oh >= OH) break;
    const int kh = h - (oh * 2 - 1);
    for (int ow = ow0; ow <= ow1; ++ow) {
      if (ow >= OW) break;
      const unsigned kk = (unsigned)(kh * 3 + (w - (ow * 2 - 1)));
      const size_t off = (((size_t)n * OH + oh) * OW + ow) * C + c;
      const uint2 tk = *(const uint2 *)(tap + off);
      float g[8];
      ld8(dy + off, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned tj = ((j < 4 ? tk.x : tk.y) >> (8 * (j & 3))) & 0xffu;
        if (tj == kk) acc[j] += g[j];
      }
    }
  }
  uint4 o;
  o.x = rne(acc[0]) | (rne(acc[1]) << 16); o.y = rne(acc[2]) | (rne(acc[3]) << 16);
  o.z = rne(acc[4]) | (rne(acc[5]) << 16); o.w = rne(acc[6]) | (rne(acc[7]) << 16);
  *(uint4 *)(dx + (((size_t)n * H + h) * W + w) * C + c) = o;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_maxpool3x3s2_fwd(const void *x, int N, int H, int W, int C, void *y, unsigned char *tap, void *stream_) {
  BFHIP_REQUIRE(x && y && tap, "maxpool3x3s2_fwd: null pointer");
  BFHIP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool3x3s2_fwd: bad shape N=%d H=%d W=%d C=%d (C %% 8 == 0)", N, H, W, C);
  BFHIP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)tap % 8) == 0, "maxpool3x3s2_fwd: misaligned tensor");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 8);
  BFHIP_REQUIRE(total < (1LL << 31) * 256, "maxpool3x3s2_fwd: tensor too large");
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)x, N, H,
                     W, C, OH, OW, (bf16_t *)y, tap);
  return check_launch("maxpool3x3s2_fwd");
}

BFHIP_EXPORT int bfhip_maxpool3x3s2_bwd(const void *dy, const unsigned char *tap, int N, int H, int W, int C, void *dx, void *stream_) {
  BFHIP_REQUIRE(dy && dx && tap, "maxpool3x3s2_bwd: null pointer");
  BFHIP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool3x3s2_bwd: bad shape N=%d H=%d W=%d C=%d (C %% 8 == 0)", N, H, W, C);
  BFHIP_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)tap % 8) == 0, "maxpool3x3s2_bwd: misaligned tensor");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * H * W * (C / 8);
  BFHIP_REQUIRE(total < (1LL << 31) * 256, "maxpool3x3s2_bwd: tensor too large");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)dy, tap, N,
                     H, W, C, OH, OW, (bf16_t *)dx);
  return check_launch("maxpool3x3s2_bwd");
}
