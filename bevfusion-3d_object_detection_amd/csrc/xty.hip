// xty.hip -- out[M, N] = X^T Y for tall-skinny row-major operands X [K, M], Y [K, N], K >> M, N   (gfx950).
//
// This is the weight gradient of every "linear layer over positions" of the TransFusion head: the K / V projections of
// the cross attention and the position-embedding MLP act on all 180 x 180 BEV cells of the batch (K = 129 600 rows of 128
// channels, BF/transformer.py:10-23,60-105), so dW = dY^T X is a GEMM with a 128 x 128 output and a 129 600-long
// reduction.  Library GEMMs tile the output and walk K serially (0.42 ms each here); this kernel splits K over the chip:
//   wave = (K slice s, 64 x 64 output block): per 4-row K step one 4-channel vector of X and one of Y per lane feed
//   16 v_mfma_f32_16x16x4_f32 (output element (4a + c, 4a' + d) from components c, d of lanes a, a' -- no operand
//   transposition needed for row-major inputs); bf16 inputs are widened on load, products and sums are fp32.
//   partial[s][M][N] is then summed over s in a fixed order (deterministic).
#include "common.h"

namespace bfhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ f32x4 load4(const T *p, int valid);
template <> __device__ __forceinline__ f32x4 load4<float>(const float *p, int valid) {
  if (valid >= 4) return *(const f32x4 *)p;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < valid; ++i) v[i] = p[i];
  return v;
}
template <> __device__ __forceinline__ f32x4 load4<unsigned short>(const unsigned short *p, int valid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid >= 4) {
    uint2 u = *(const uint2 *)p;
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    return v;
  }
  for (int i = 0; i < valid; ++i) v[i] = __uint_as_float((unsigned)p[i] << 16);
  return v;
}

// VEC: rows are 4-element aligned (M % 4 == 0 and N % 4 == 0 and 8/16-byte aligned bases) -> unguarded vector loads
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void xty_kernel(const T *__restrict__ X, const T *__restrict__ Y, long long K, int M,
                                                  int N, int S, int GI, int GJ, float *__restrict__ partial) {
  const int lane = threadIdx.x & 63;
  long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= (long long)S * GI * GJ) return;
  const int gj = (int)(wave % GJ); wave /= GJ;
  const int gi = (int)(wave % GI);
  const int s = (int)(wave / GI);
  const int la = lane & 15, lq = lane >> 4;
  const long long rows_per = (((K + S - 1) / S) + 15) & ~15LL;
  const long long r0 = (long long)s * rows_per, r1 = r0 + rows_per < K ? r0 + rows_per : K;
  const int ci = gi * 64 + la * 4, cj = gj * 64 + la * 4;
  const int vi = VEC ? (ci < M ? 4 : 0) : (M - ci > 4 ? 4 : (M - ci > 0 ? M - ci : 0));
  const int vj = VEC ? (cj < N ? 4 : 0) : (N - cj > 4 ? 4 : (N - cj > 0 ? N - cj : 0));
  f32x4 acc[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  for (long long n0 = r0; n0 < r1; n0 += 4 * U) {
    f32x4 av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = n0 + 4 * u + lq;
      const bool ok = r < r1;
      av[u] = (ok && vi) ? load4<T>(X + (size_t)r * M + ci, vi) : (f32x4){0.f, 0.f, 0.f, 0.f};
      bv[u] = (ok && vj) ? load4<T>(Y + (size_t)r * N + cj, vj) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d)
          acc[c][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][c], bv[u][d], acc[c][d], 0, 0, 0);
  }
  // D layout: row = (lane>>4)*4 + i -> a (m = 4a + c), col = lane&15 -> a' (n = 4a' + d)
  float *dst = partial + (size_t)s * M * N;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = gi * 64 + (lq * 4 + i) * 4 + c;
      if (m >= M) continue;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int n = gj * 64 + la * 4 + d;
        if (n < N) dst[(size_t)m * N + n] = acc[c][d][i];
      }
    }
}

__global__ __launch_bounds__(256) void xty_reduce_kernel(const float *__restrict__ partial, int S, long long MN,
                                                         float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= MN) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 3 < S; s += 4) {  // four independent chains; combined in a fixed order
    a0 += partial[(size_t)s * MN + t];
    a1 += partial[(size_t)(s + 1) * MN + t];
    a2 += partial[(size_t)(s + 2) * MN + t];
    a3 += partial[(size_t)(s + 3) * MN + t];
  }
  for (; s < S; ++s) a0 += partial[(size_t)s * MN + t];
  out[t] = (a0 + a1) + (a2 + a3);
}

inline int pick_splits(long long K, int GI, int GJ) {
  long long blocks = (long long)GI * GJ;
  long long S = (2048 + blocks - 1) / blocks;       // ~2048 waves = 2 per SIMD
  long long maxS = (K + 63) / 64;                   // at least 64 rows per slice
  if (S > maxS) S = maxS;
  if (S > 512) S = 512;
  if (S < 1) S = 1;
  return (int)S;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT size_t bfhip_xty_workspace_bytes(long long K, int M, int N) {
  if (K <= 0 || M <= 0 || N <= 0) return 0;
  const int GI = (M + 63) / 64, GJ = (N + 63) / 64;
  return align_up((size_t)pick_splits(K, GI, GJ) * M * N * sizeof(float), 256);
}

BFHIP_EXPORT int bfhip_xty(const void *X, const void *Y, long long K, int M, int N, int dtype, float *out,
                           void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(X && Y && out && K > 0 && M > 0 && N > 0 && (dtype == 0 || dtype == 1), "xty: bad arguments");
  BFHIP_REQUIRE((long long)M * N < (1LL << 31), "xty: output too large");
  if (!workspace || workspace_bytes < bfhip_xty_workspace_bytes(K, M, N)) { set_error("xty: workspace too small"); return BFHIP_E_WORKSPACE; }
  const int GI = (M + 63) / 64, GJ = (N + 63) / 64;
  const int S = pick_splits(K, GI, GJ);
  float *partial = (float *)workspace;
  const long long waves = (long long)S * GI * GJ;
  dim3 grid(ceil_div(waves * 64, 256));
  const size_t es = dtype == 1 ? 2 : 4;
  const bool vec = (M % 4 == 0) && (N % 4 == 0) && ((uintptr_t)X % (4 * es) == 0) && ((uintptr_t)Y % (4 * es) == 0);
  if (dtype == 1) {
    if (vec) hipLaunchKernelGGL((xty_kernel<unsigned short, true>), grid, dim3(256), 0, s, (const unsigned short *)X, (const unsigned short *)Y, K, M, N, S, GI, GJ, partial);
    else hipLaunchKernelGGL((xty_kernel<unsigned short, false>), grid, dim3(256), 0, s, (const unsigned short *)X, (const unsigned short *)Y, K, M, N, S, GI, GJ, partial);
  } else {
    if (vec) hipLaunchKernelGGL((xty_kernel<float, true>), grid, dim3(256), 0, s, (const float *)X, (const float *)Y, K, M, N, S, GI, GJ, partial);
    else hipLaunchKernelGGL((xty_kernel<float, false>), grid, dim3(256), 0, s, (const float *)X, (const float *)Y, K, M, N, S, GI, GJ, partial);
  }
  const long long MN = (long long)M * N;
  hipLaunchKernelGGL(xty_reduce_kernel, dim3(ceil_div(MN, 256)), dim3(256), 0, s, partial, S, MN, out);
  return check_launch("xty");
}
