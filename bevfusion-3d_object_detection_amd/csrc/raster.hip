// raster.hip -- sparse LiDAR depth images and the GT depth histogram of the camera branch (gfx950).
//
// Replaces the per-sample torch loop of BaseDepthTransform.forward (BF/depth_lss.py:372-449: five small
// matmuls, clamp, divide, boolean mask, nonzero, scatter_ with duplicate indices) and the dense
// scatter_add_ histogram of get_cam_feats (:636-686, 6*256*704 atomics per sample, almost all into bin 0
// which is then zeroed) with:
//   raster_project_kernel   one thread per (camera, point): project, test, 64-bit atomicMax of
//                           (point index + 1) << 32 | depth bits  -> the LAST point wins on a duplicate
//                           pixel, deterministically (torch leaves it unspecified, :410-417)
//   raster_resolve_kernel   depth image from the winners; in the same pass the depth bin of every hit
//                           pixel is added to the (feature cell, bin) histogram (only hit pixels can land
//                           in a bin > 0; bin 0 is zeroed by the reference anyway)
//   hist_normalise_kernel   distr = counts / (sum + 1e-8)
#include "common.h"

namespace bfhip {
namespace {

__device__ __forceinline__ void m3(const float *__restrict__ m, float p0, float p1, float p2, float &o0,
                                   float &o1, float &o2) {
  o0 = __fadd_rn(__fadd_rn(__fmul_rn(m[0], p0), __fmul_rn(m[1], p1)), __fmul_rn(m[2], p2));
  o1 = __fadd_rn(__fadd_rn(__fmul_rn(m[4], p0), __fmul_rn(m[5], p1)), __fmul_rn(m[6], p2));
  o2 = __fadd_rn(__fadd_rn(__fmul_rn(m[8], p0), __fmul_rn(m[9], p1)), __fmul_rn(m[10], p2));
}

// inv_rot is a packed 3x3 (9 floats); l2i / img_aug are row-major 4x4 per camera
__global__ __launch_bounds__(256) void raster_project_kernel(
    const float *__restrict__ points, int n, int f, const float *__restrict__ inv_rot,
    const float *__restrict__ aug_trans, const float *__restrict__ l2i,
    const float *__restrict__ img_aug, int ncam, int iH, int iW,
    unsigned long long *__restrict__ winner) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)ncam * n) return;
  int c = (int)(t / n), i = (int)(t - (long long)c * n);
  const float *pt = points + (size_t)i * f;
  float p0 = __fsub_rn(pt[0], aug_trans[0]), p1 = __fsub_rn(pt[1], aug_trans[1]), p2 = __fsub_rn(pt[2], aug_trans[2]);
  float q0 = __fadd_rn(__fadd_rn(__fmul_rn(inv_rot[0], p0), __fmul_rn(inv_rot[1], p1)), __fmul_rn(inv_rot[2], p2));
  float q1 = __fadd_rn(__fadd_rn(__fmul_rn(inv_rot[3], p0), __fmul_rn(inv_rot[4], p1)), __fmul_rn(inv_rot[5], p2));
  float q2 = __fadd_rn(__fadd_rn(__fmul_rn(inv_rot[6], p0), __fmul_rn(inv_rot[7], p1)), __fmul_rn(inv_rot[8], p2));
  const float *L = l2i + c * 16, *A = img_aug + c * 16;
  m3(L, q0, q1, q2, p0, p1, p2);
  p0 = __fadd_rn(p0, L[3]); p1 = __fadd_rn(p1, L[7]); p2 = __fadd_rn(p2, L[11]);
  const float dist = p2;
  float z = dist < 1e-5f ? 1e-5f : (dist > 1e5f ? 1e5f : dist);
  if (!(dist == dist)) z = dist;
  p0 = __fdiv_rn(p0, z); p1 = __fdiv_rn(p1, z);
  m3(A, p0, p1, z, q0, q1, q2);
  const float col = __fadd_rn(q0, A[3]), row = __fadd_rn(q1, A[7]);
  if (!(row < (float)iH && row >= 0.f && col < (float)iW && col >= 0.f)) return;
  unsigned long long v = ((unsigned long long)(unsigned)(i + 1) << 32) | (unsigned long long)__float_as_uint(dist);
  atomicMax(&winner[((size_t)c * iH + (int)row) * iW + (int)col], v);
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const unsigned long long *__restrict__ winner,
                                                             long long npix, int iH, int iW, int fH, int fW,
                                                             int D, float lo, float cmax, float half,
                                                             float step, float *__restrict__ depth,
                                                             float *__restrict__ counts) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npix) return;
  unsigned long long v = winner[t];
  float d = (v >> 32) ? __uint_as_float((unsigned)(v & 0xffffffffull)) : 0.f;
  depth[t] = d;
  if (counts && (v >> 32)) {
    float cl = d < lo ? lo : (d > cmax ? cmax : d);
    int bin = (int)__fdiv_rn(__fsub_rn(__fadd_rn(cl, half), lo), step);
    if (bin > 0 && bin < D) {  // bin 0 is zeroed by the reference (:670)
      int col = (int)(t % iW);
      long long r = t / iW;
      int row = (int)(r % iH);
      long long cam = r / iH;
      size_t cell = ((size_t)cam * fH + row / (iH / fH)) * fW + col / (iW / fW);
      atomicAdd(&counts[cell * D + bin], 1.0f);  // small exact integers: order-independent
    }
  }
}

// histogram of an EXISTING depth image (API parity with gt_depth_distribution on arbitrary input)
__global__ __launch_bounds__(256) void hist_from_depth_kernel(const float *__restrict__ depth, long long npix,
                                                              int iH, int iW, int fH, int fW, int D, float lo,
                                                              float cmax, float half, float step,
                                                              float *__restrict__ counts) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npix) return;
  float d = depth[t];
  float cl = d < lo ? lo : (d > cmax ? cmax : d);
  int bin = (int)__fdiv_rn(__fsub_rn(__fadd_rn(cl, half), lo), step);
  if (bin > 0 && bin < D) {
    int col = (int)(t % iW);
    long long r = t / iW;
    int row = (int)(r % iH);
    long long cam = r / iH;
    size_t cell = ((size_t)cam * fH + row / (iH / fH)) * fW + col / (iW / fW);
    atomicAdd(&counts[cell * D + bin], 1.0f);
  }
}

// one wave per (camera, feature cell): lanes stride over the D bins (coalesced); counts are small integers, so the
// order of the sum does not matter (exact in fp32)
__global__ __launch_bounds__(256) void hist_normalise_kernel(const float *__restrict__ counts, long long ncell,
                                                             int D, float *__restrict__ distr) {
  const long long cell = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (cell >= ncell) return;
  const float *c = counts + cell * D;
  float s = 0.f;
  for (int b = lane; b < D; b += 64) s += c[b];
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
  const float den = s + 1e-8f;
  for (int b = lane; b < D; b += 64) distr[cell * D + b] = c[b] / den;
}

// ---- gradient of the first dtransform layer, Conv2d(1, 8, 1) on the one-channel depth image (BF/depth_lss.py:592-594):
// y[m][c] = b[c] + d[m] * w[c], so dw[c] = sum_m dy[m][c] * d[m] and db[c] = sum_m dy[m][c] over the BN * 256 * 704 pixels.
// torch forms dy * d as a tensor and reduces two [4.3 M, 8] tensors along their long side (2 x 160 us at 0.43 TB/s: eight-wide
// rows defeat its reduction's vectorisation); here one pass over dy (16 B per row) and d, fp32 accumulation, block partials
// summed in a fixed order by the second kernel.
constexpr int kLiftBlocks = 2048;

__global__ __launch_bounds__(256) void depth_lift_bwd_kernel(const uint4 *__restrict__ dy, const unsigned short *__restrict__ d,
                                                             long long M, float *__restrict__ partial) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    const uint4 v = dy[m];
    const float dv = __uint_as_float((unsigned)d[m] << 16);
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = __uint_as_float(u[k] << 16), hi = __uint_as_float(u[k] & 0xffff0000u);
      acc[2 * k] += lo;
      acc[2 * k + 1] += hi;
      acc[8 + 2 * k] += lo * dv;
      acc[8 + 2 * k + 1] += hi * dv;
    }
  }
  __shared__ float red[4][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float a = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = a;
  }
  __syncthreads();
  if (threadIdx.x < 16) partial[(size_t)blockIdx.x * 16 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[0..8) = db, out[8..16) = dw: 16 threads per output, each a fixed slice of the block partials, then a fixed tree
__global__ __launch_bounds__(256) void depth_lift_bwd_final_kernel(const float *__restrict__ partial, int nblk, float *__restrict__ out) {
  const int o = threadIdx.x >> 4, part = threadIdx.x & 15;
  double a = 0.0;
  for (int b = part; b < nblk; b += 16) a += (double)partial[(size_t)b * 16 + o];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) a += __shfl_down(a, off, 16);
  if (part == 0) out[o] = (float)a;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT size_t bfhip_rasterise_depth_workspace_bytes(int ncam, int iH, int iW) {
  return align_up((size_t)(ncam > 0 ? ncam : 1) * iH * iW * sizeof(unsigned long long), 256) + 256;
}

// One sample: points f32[n,f] -> depth f32[ncam,iH,iW].  If counts != NULL (f32[ncam,fH,fW,D], NOT cleared
// here) the GT-depth histogram of the hit pixels is accumulated into it in the same pass.
BFHIP_EXPORT int bfhip_rasterise_depth(const float *points, int n, int f, const float *inv_rot,
                                       const float *aug_trans, const float *lidar2image, const float *img_aug,
                                       int ncam, int iH, int iW, float *depth, float *counts, int fH, int fW, int D,
                                       const float *dbound_host, void *workspace, size_t workspace_bytes,
                                       void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(n >= 0 && f >= 3 && ncam > 0 && iH > 0 && iW > 0, "rasterise_depth: bad sizes");
  BFHIP_REQUIRE(depth && inv_rot && aug_trans && lidar2image && img_aug, "rasterise_depth: null pointer");
  BFHIP_REQUIRE(!counts || (fH > 0 && fW > 0 && D > 0 && iH % fH == 0 && iW % fW == 0 && dbound_host),
                "rasterise_depth: histogram needs fH | iH, fW | iW and dbound");
  if (workspace_bytes < bfhip_rasterise_depth_workspace_bytes(ncam, iH, iW) || !workspace) { set_error("rasterise_depth: workspace too small"); return BFHIP_E_WORKSPACE; }
  unsigned long long *winner = (unsigned long long *)workspace;
  long long npix = (long long)ncam * iH * iW;
  ProfScope ps;
  prof_begin(BFHIP_OP_RASTER, stream, &ps);
  if (hipMemsetAsync(winner, 0, (size_t)npix * sizeof(unsigned long long), stream) != hipSuccess) return check_launch("rasterise_depth memset");
  if (n > 0) {
    BFHIP_REQUIRE(points, "rasterise_depth: points is null");
    hipLaunchKernelGGL(raster_project_kernel, dim3(ceil_div((long long)ncam * n, 256)), dim3(256), 0, stream, points, n, f,
                       inv_rot, aug_trans, lidar2image, img_aug, ncam, iH, iW, winner);
  }
  float lo = 0.f, cmax = 0.f, half = 0.f, step = 1.f;
  if (counts) {
    lo = dbound_host[0]; step = dbound_host[2];
    half = (float)(0.5 * (double)step);
    cmax = (float)((double)dbound_host[1] - 0.5 * (double)step);
  }
  hipLaunchKernelGGL(raster_resolve_kernel, dim3(ceil_div(npix, 256)), dim3(256), 0, stream, winner, npix, iH, iW, fH, fW, D,
                     lo, cmax, half, step, depth, counts);
  prof_end(&ps);
  return check_launch("rasterise_depth");
}

BFHIP_EXPORT size_t bfhip_depth_lift_bwd_workspace_bytes(void) { return (size_t)kLiftBlocks * 16 * sizeof(float); }

// dy bf16 [M][8] (dense), d bf16 [M] -> out f32[16] = {db[8], dw[8]} of y[m][c] = b[c] + d[m] * w[c]
BFHIP_EXPORT int bfhip_depth_lift_bwd(const void *dy, const void *d, long long M, float *out, void *workspace, size_t workspace_bytes,
                                      void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(dy && d && out && M > 0, "depth_lift_bwd: bad arguments");
  BFHIP_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)d % 2) == 0, "depth_lift_bwd: misaligned tensor");
  if (!workspace || workspace_bytes < bfhip_depth_lift_bwd_workspace_bytes()) { set_error("depth_lift_bwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  const int nblk = (int)(M < (long long)kLiftBlocks * 256 ? ceil_div(M, 256) : kLiftBlocks);
  hipLaunchKernelGGL(depth_lift_bwd_kernel, dim3(nblk), dim3(256), 0, stream, (const uint4 *)dy, (const unsigned short *)d, M,
                     (float *)workspace);
  hipLaunchKernelGGL(depth_lift_bwd_final_kernel, dim3(1), dim3(256), 0, stream, (const float *)workspace, nblk, out);
  return check_launch("depth_lift_bwd");
}

// counts (optional input, f32[BN,fH,fW,D]) -> distr; when depth != NULL the counts are first rebuilt from it.
BFHIP_EXPORT int bfhip_depth_histogram(const float *depth, int BN, int iH, int iW, int fH, int fW, int D,
                                       const float *dbound_host, float *counts, float *distr, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(BN > 0 && iH > 0 && iW > 0 && fH > 0 && fW > 0 && D > 0 && iH % fH == 0 && iW % fW == 0, "depth_histogram: bad sizes");
  BFHIP_REQUIRE(counts && distr && dbound_host, "depth_histogram: null pointer");
  long long ncell = (long long)BN * fH * fW;
  if (depth) {
    float lo = dbound_host[0], step = dbound_host[2];
    float half = (float)(0.5 * (double)step), cmax = (float)((double)dbound_host[1] - 0.5 * (double)step);
    if (hipMemsetAsync(counts, 0, (size_t)ncell * D * sizeof(float), stream) != hipSuccess) return check_launch("depth_histogram memset");
    long long npix = (long long)BN * iH * iW;
    hipLaunchKernelGGL(hist_from_depth_kernel, dim3(ceil_div(npix, 256)), dim3(256), 0, stream, depth, npix, iH, iW, fH, fW, D,
                       lo, cmax, half, step, counts);
  }
  hipLaunchKernelGGL(hist_normalise_kernel, dim3(ceil_div(ncell * 64, 256)), dim3(256), 0, stream, counts, ncell, D, distr);
  return check_launch("depth_histogram");
}
