// lift_splat.hip -- camera frustum -> BEV on gfx950: geometry/rank/sort plan + fused lift-splat.
//
// Reference path (BF/depth_lss.py):
//   get_geometry :68-112        frustum pixel-depth grid -> lidar xyz   [B,N,D,fH,fW,3] materialised
//   bev_pool_aux :118-176       cells (trunc), range mask, rank, argsort of ~1.8 M int64 per sample
//   get_cam_feats :723-725      x = depth (x) feat, materialised [B,N,D,fH,fW,C]  (638 MB / sample)
//   bev_pool :179-204           x[kept][indices] (two more 587 MB copies) + bev_pool op
// Here:
//   bfhip_bev_plan        one pass over the frustum computes geometry -> cell -> rank key (nothing
//                         materialised), a stable radix sort orders the kept points by rank, and the
//                         interval table is compacted on device.  No host sync; counts stay on device.
//   bfhip_lift_splat_fwd  cell-stationary: each BEV cell sums depth[p,d]*feat[p,:] over its interval.
//                         The [N',C] tensor never exists; feat rows are gathered from L2.
//   bfhip_lift_splat_bwd  pixel-stationary: each pixel walks its D depth bins, gathers the cell's
//                         out_grad row once and uses it for BOTH d_depth (dot with feat) and d_feat
//                         (axpy with depth).  No atomics, deterministic.
// Layouts (pixel-major so that a pixel's C features / D depths are contiguous):
//   depth f32[P, depth_pitch] (first D used), feat f32[P, feat_pitch] (first C used), P = B*N*fH*fW.
#include "common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace bfhip {
namespace {

constexpr int kScan = 1024;

struct PlanParams {
  int B, N, D, HW;         // frustum [B, N, D, HW]
  int nx0, nx1, nx2;       // BEV grid (x, y, z) = (H, W, Dz) of the op
  float ox, oy, oz;        // origin = bx - dx/2 (fp32, computed like torch does)
  float dx0, dx1, dx2;
  unsigned invalid_key;    // = nx0*nx1*nx2*B  (one past the largest rank)
};

__device__ __forceinline__ void mat3_vec(const float *__restrict__ m, float p0, float p1, float p2,
                                         float &o0, float &o1, float &o2) {
  // fixed association ((m0*p0 + m1*p1) + m2*p2), every op rounded to fp32 (no fma): identical to
  // oracle_frustum_geometry
  o0 = __fadd_rn(__fadd_rn(__fmul_rn(m[0], p0), __fmul_rn(m[1], p1)), __fmul_rn(m[2], p2));
  o1 = __fadd_rn(__fadd_rn(__fmul_rn(m[3], p0), __fmul_rn(m[4], p1)), __fmul_rn(m[5], p2));
  o2 = __fadd_rn(__fadd_rn(__fmul_rn(m[6], p0), __fmul_rn(m[7], p1)), __fmul_rn(m[8], p2));
}

__device__ __forceinline__ bool trunc_cell(float p, float o, float dx, int n, int &c) {
  // ((p - origin) / dx).long()  -- truncation toward zero (depth_lss.py:129), then 0 <= c < n (:141-148)
  float q = __fdiv_rn(__fsub_rn(p, o), dx);
  if (!(q > -2147483648.0f && q < 2147483648.0f)) return false;
  c = (int)q;
  return c >= 0 && c < n;
}

// one thread per frustum point i = ((b*N + n)*D + d)*HW + hw
__global__ __launch_bounds__(256) void plan_rank_kernel(
    const float *__restrict__ frustum,        // [D*HW, 3]
    const float *__restrict__ post_trans,     // [B*N, 3]
    const float *__restrict__ post_rots_inv,  // [B*N, 9]
    const float *__restrict__ combine,        // [B*N, 9]
    const float *__restrict__ c2l_trans,      // [B*N, 3]
    const float *__restrict__ extra_rots,     // [B, 9]
    const float *__restrict__ extra_trans,    // [B, 3]
    PlanParams P, long long nprime, unsigned *__restrict__ keys, unsigned *__restrict__ vals,
    int *__restrict__ cell_of_point, unsigned char *__restrict__ kept, float *__restrict__ geom_out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nprime) return;
  const int dhw = P.D * P.HW;
  const int cam = (int)(i / dhw);
  const int rem = (int)(i - (long long)cam * dhw);
  const int d = rem / P.HW;
  const int hw = rem - d * P.HW;
  const int b = cam / P.N;
  const float *fr = frustum + (size_t)rem * 3;
  const float *pt = post_trans + cam * 3;
  float p0 = __fsub_rn(fr[0], pt[0]), p1 = __fsub_rn(fr[1], pt[1]), p2 = __fsub_rn(fr[2], pt[2]);
  float q0, q1, q2;
  mat3_vec(post_rots_inv + cam * 9, p0, p1, p2, q0, q1, q2);
  p0 = __fmul_rn(q0, q2);
  p1 = __fmul_rn(q1, q2);
  p2 = q2;
  mat3_vec(combine + cam * 9, p0, p1, p2, q0, q1, q2);
  const float *ct = c2l_trans + cam * 3;
  q0 = __fadd_rn(q0, ct[0]); q1 = __fadd_rn(q1, ct[1]); q2 = __fadd_rn(q2, ct[2]);
  mat3_vec(extra_rots + b * 9, q0, q1, q2, p0, p1, p2);
  const float *et = extra_trans + b * 3;
  p0 = __fadd_rn(p0, et[0]); p1 = __fadd_rn(p1, et[1]); p2 = __fadd_rn(p2, et[2]);
  if (geom_out) {
    geom_out[i * 3 + 0] = p0; geom_out[i * 3 + 1] = p1; geom_out[i * 3 + 2] = p2;
  }
  int cx = -1, cy = -1, cz = -1;
  bool okx = trunc_cell(p0, P.ox, P.dx0, P.nx0, cx);
  bool oky = trunc_cell(p1, P.oy, P.dx1, P.nx1, cy);
  bool okz = trunc_cell(p2, P.oz, P.dx2, P.nx2, cz);
  bool ok = okx && oky && okz;
  // rank = x*(W*Dz*B) + y*(Dz*B) + z*B + b   (depth_lss.py:165-169; W = nx[1], Dz = nx[2])
  unsigned key = ok ? (unsigned)(((cx * P.nx1 + cy) * P.nx2 + cz) * P.B + b) : P.invalid_key;
  keys[i] = key;
  vals[i] = ((unsigned)(cam * P.HW + hw) << 8) | (unsigned)d;  // (pixel index, depth bin)
  if (kept) kept[i] = ok ? 1 : 0;
  // out layout [b][z][x][y]   (bev_pool_cuda.cu:34-36)
  if (cell_of_point) cell_of_point[i] = ok ? ((b * P.nx2 + cz) * P.nx0 + cx) * P.nx1 + cy : -1;
}

__device__ __forceinline__ int block_sum_1024(int v, int *sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  int r = 0;
  for (int k = 0; k < kScan / 64; ++k) r += sm[k];
  __syncthreads();
  return r;
}

// counts of interval starts and kept rows per block of sorted keys
__global__ __launch_bounds__(kScan) void plan_flag_count_kernel(const unsigned *__restrict__ keys,
                                                                long long n, unsigned invalid,
                                                                int *__restrict__ blk_starts,
                                                                int *__restrict__ blk_kept) {
  __shared__ int sm[kScan / 64];
  long long i = (long long)blockIdx.x * kScan + threadIdx.x;
  int valid = 0, flag = 0;
  if (i < n) {
    unsigned k = keys[i];
    valid = k != invalid;
    flag = valid && (i == 0 || keys[i - 1] != k);
  }
  int s = block_sum_1024(flag, sm);
  int v = block_sum_1024(valid, sm);
  if (threadIdx.x == 0) { blk_starts[blockIdx.x] = s; blk_kept[blockIdx.x] = v; }
}

// single-block exclusive scan of blk_starts, total of blk_kept -> counts[0] = n_kept, counts[1] = m
__global__ __launch_bounds__(kScan) void plan_scan_kernel(int *__restrict__ blk_starts,
                                                          const int *__restrict__ blk_kept, int nb,
                                                          int *__restrict__ counts) {
  __shared__ int sm[kScan];
  __shared__ int carry, kept_total;
  if (threadIdx.x == 0) { carry = 0; kept_total = 0; }
  __syncthreads();
  int kept_local = 0;
  for (int base = 0; base < nb; base += kScan) {
    int i = base + threadIdx.x;
    int v = i < nb ? blk_starts[i] : 0;
    kept_local += i < nb ? blk_kept[i] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < kScan; o <<= 1) {
      int t = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    int incl = sm[threadIdx.x];
    int c = carry;
    if (i < nb) blk_starts[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == kScan - 1) carry = c + incl;
    __syncthreads();
  }
  atomicAdd(&kept_total, kept_local);
  __syncthreads();
  if (threadIdx.x == 0) { counts[0] = kept_total; counts[1] = carry; }
}

// interval starts, per-interval cell offset, per-row geom (x,y,z,b), optional int64 ranks
__global__ __launch_bounds__(kScan) void plan_assign_kernel(
    const unsigned *__restrict__ keys, long long n, PlanParams P, const int *__restrict__ blk_offs,
    int *__restrict__ starts, int *__restrict__ cell_of_interval, int *__restrict__ geom,
    long long *__restrict__ ranks64) {
  __shared__ int wsum[kScan / 64];
  long long i = (long long)blockIdx.x * kScan + threadIdx.x;
  int valid = 0, flag = 0;
  unsigned k = P.invalid_key;
  if (i < n) {
    k = keys[i];
    valid = k != P.invalid_key;
    flag = valid && (i == 0 || keys[i - 1] != k);
  }
  int cx = 0, cy = 0, cz = 0, b = 0;
  if (valid) {
    unsigned t = k;
    b = t % P.B; t /= P.B;
    cz = t % P.nx2; t /= P.nx2;
    cy = t % P.nx1; cx = t / P.nx1;
    if (geom) {
      int4 g = make_int4(cx, cy, cz, b);
      ((int4 *)geom)[i] = g;
    }
    if (ranks64) ranks64[i] = (long long)k;
  }
  unsigned long long bal = __ballot(flag);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  if (flag) {
    int woff = 0;
    for (int j = 0; j < wv; ++j) woff += wsum[j];
    int idx = blk_offs[blockIdx.x] + woff + __popcll(bal & ((1ull << lane) - 1ull));
    starts[idx] = (int)i;
    cell_of_interval[idx] = ((b * P.nx2 + cz) * P.nx0 + cx) * P.nx1 + cy;
  }
}

__global__ __launch_bounds__(256) void plan_lengths_kernel(const int *__restrict__ starts,
                                                           const int *__restrict__ counts,
                                                           int *__restrict__ lengths, int mmax) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  int m = counts[1];
  if (k >= mmax) return;
  if (k >= m) { lengths[k] = 0; return; }
  int next = (k + 1 < m) ? starts[k + 1] : counts[0];
  lengths[k] = next - starts[k];
}

// Camera-major processing order of the intervals.  A BEV cell's members are frustum points of (almost always) ONE camera,
// and each feature row (320 B) is shared by the 118 depth points of its ray -- i.e. by cells strung out along the ray, far
// apart in rank order (x-major): walked in rank order the 21.6 MB row table of a batch misses the 4 MiB per-XCD L2 on nearly
// every gather (PMC: 1.24 GB fetched for 62 MB of algorithmic bytes per launch).  Keyed by (sample, camera) of the first
// member and sorted stably (rank order inside a camera), every XCD walks ~3 cameras whose rows (0.9 MB each) stay in ITS
// L2.  Only the order in which cells are produced changes; the sum inside a cell keeps rank order (bit-identical output).
__global__ __launch_bounds__(256) void plan_group_key_kernel(const unsigned *__restrict__ pd, const int *__restrict__ starts,
                                                             const int *__restrict__ counts, int HW, int mmax, unsigned ngroups,
                                                             unsigned *__restrict__ keys, unsigned *__restrict__ vals) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= mmax) return;
  vals[k] = (unsigned)k;
  keys[k] = k < counts[1] ? (pd[starts[k]] >> 8) / (unsigned)HW : ngroups;  // unused slots sort behind every camera
}

// ------------------------------------------------------------------------------ fused forward
constexpr int kU = 8;

__device__ __forceinline__ unsigned rne_bf16_bits(float f) {  // fp32 -> bf16, round to nearest even (NaN kept quiet)
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// cq lanes per interval, each lane owns 4 channels.  out[cell][c] = sum_i depth[pix_i, d_i] * feat[pix_i, c]
// OUT16: the fp32 sum is rounded once and stored as bf16 (what the consumer's cast would do: the view transform's bf16
// downsample convolution then takes the BEV map as it is)
template <bool OUT16>
__global__ __launch_bounds__(256) void lift_splat_fwd_kernel(
    const float *__restrict__ depth, int depth_pitch, const float *__restrict__ feat, int feat_pitch,
    const unsigned *__restrict__ pd, const int *__restrict__ starts, const int *__restrict__ lengths,
    const int *__restrict__ cell_of_interval, const int *__restrict__ counts, const int *__restrict__ order, int mmax,
    int cq, int groups, void *__restrict__ out_) {
  const int m = min(mmax, counts[1]);
  const int lane = threadIdx.x & (kWave - 1);
  // rank order (order == NULL): round-robin block->XCD placement on purpose -- the long intervals (cells next to the ego
  // vehicle) are contiguous in rank order and an XCD-chunked mapping puts them all on one XCD (0.548 ms chunked vs
  // 0.374 ms round-robin, batch-4 nuScenes frustum).  Camera-major order: XCD-chunked, each XCD owns whole cameras and
  // every camera has its share of long intervals.
  const long long blk = order ? xcd_chunked_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const long long wave = (blk * blockDim.x + threadIdx.x) >> 6;
  const int g = lane / cq;
  const int q = lane - g * cq;
  const long long kp = wave * groups + g;
  if (g >= groups || kp >= m) return;
  const long long k = order ? order[kp] : kp;
  const int s = starts[k];
  const int len = lengths[k];
  const unsigned *ppd = pd + s;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // Measured and dropped (batch-4 nuScenes frustum, same launch): index / row loads software-pipelined one and two groups
  // ahead of the adds (0.363 vs 0.374 ms), groups of 4 / 16 members (0.39 / 0.53 ms), intervals ordered by descending
  // length so that a wave's three intervals match (0.347 vs 0.350 ms).  The kernel sits at the per-CU rate of 320-byte row
  // gathers through the vector L1 (~8 TB/s of useful bytes chip-wide), not on latency, divergence or -- since the
  // camera-major order -- memory traffic.
  for (int i = 0; i < len; i += kU) {
    const int rem = len - i;
    unsigned e[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) e[u] = ppd[i + (u < rem ? u : rem - 1)];
    float dv[kU];
    float4 fv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const unsigned pix = e[u] >> 8, dd = e[u] & 255u;
      dv[u] = depth[(size_t)pix * depth_pitch + dd];
      fv[u] = *(const float4 *)(feat + (size_t)pix * feat_pitch + q * 4);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (u < rem) {
        // product rounded to fp32, then added (the materialised x tensor holds the rounded product)
        acc.x = __fadd_rn(acc.x, __fmul_rn(dv[u], fv[u].x));
        acc.y = __fadd_rn(acc.y, __fmul_rn(dv[u], fv[u].y));
        acc.z = __fadd_rn(acc.z, __fmul_rn(dv[u], fv[u].z));
        acc.w = __fadd_rn(acc.w, __fmul_rn(dv[u], fv[u].w));
      }
    }
  }
  const size_t o = (size_t)cell_of_interval[k] * cq + q;
  if (OUT16) {
    uint2 v;
    v.x = rne_bf16_bits(acc.x) | (rne_bf16_bits(acc.y) << 16);
    v.y = rne_bf16_bits(acc.z) | (rne_bf16_bits(acc.w) << 16);
    ((uint2 *)out_)[o] = v;
  } else {
    ((float4 *)out_)[o] = acc;
  }
}

// ------------------------------------------------------------------------------ fused backward
// cq lanes per PIXEL.  For d in [0, D): g = out_grad[cell(p,d)] (0 if not kept)
//   d_depth[p,d] = sum_c g[c]*feat[p,c] ;  d_feat[p,:] += depth[p,d]*g
// G16: out_grad arrives as bf16 (the gradient of a bf16 BEV map): widened on load, same arithmetic
template <bool G16>
__global__ __launch_bounds__(256) void lift_splat_bwd_kernel(
    const void *__restrict__ out_grad_, const float *__restrict__ depth, int depth_pitch,
    const float *__restrict__ feat, int feat_pitch, const int *__restrict__ cell_of_point, int P_,
    int D, int HW, int cq, int groups, float *__restrict__ d_depth, int d_depth_pitch,
    float *__restrict__ d_feat, int d_feat_pitch) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long wave = xcd_chunked_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int g = lane / cq;
  const int q = lane - g * cq;
  const long long p = wave * groups + g;
  const bool active = g < groups && p < P_;
  // no early return: the group reduction below uses cross-lane reads
  const long long pp = active ? p : 0;
  const int cam = (int)(pp / HW), hw = (int)(pp - (long long)cam * HW);
  const int *cop = cell_of_point + (size_t)cam * D * HW + hw;  // + d*HW
  const float4 f = active ? *(const float4 *)(feat + (size_t)pp * feat_pitch + q * 4) : make_float4(0, 0, 0, 0);
  const float *dep = depth + (size_t)pp * depth_pitch;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int gbase = g * cq;
  for (int d0 = 0; d0 < D; d0 += kU) {
    int cell[kU];
    float dv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int d = d0 + u;
      cell[u] = (active && d < D) ? cop[(size_t)d * HW] : -1;
      dv[u] = (active && d < D) ? dep[d] : 0.f;
    }
    float4 gv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u)
    {
      if (cell[u] < 0) {
        gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (G16) {
        const uint2 v = ((const uint2 *)out_grad_)[(size_t)cell[u] * cq + q];
        gv[u] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                            __uint_as_float(v.y & 0xffff0000u));
      } else {
        gv[u] = ((const float4 *)out_grad_)[(size_t)cell[u] * cq + q];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      // per-lane partial dot over its 4 channels (ascending), then a fixed-shape tree over the cq lanes
      float part = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(gv[u].x, f.x), __fmul_rn(gv[u].y, f.y)),
                                       __fmul_rn(gv[u].z, f.z)),
                             __fmul_rn(gv[u].w, f.w));
      // fold lanes >= 32 then >= 16 onto the first 16 lanes of the group, then xor-tree inside 16
      float hi = __shfl(part, gbase + q + 32);
      if (q < 32 && q + 32 < cq) part = __fadd_rn(part, hi);
      hi = __shfl(part, gbase + q + 16);
      if (q < 16 && q + 16 < (cq < 32 ? cq : 32)) part = __fadd_rn(part, hi);
      for (int o = 8; o > 0; o >>= 1) {
        float other = __shfl(part, gbase + (q ^ o));
        bool has = ((q ^ o) < cq) && ((q ^ o) < 16);
        if (q < 16 && has) part = __fadd_rn(part, other);
      }
      const int d = d0 + u;
      if (active && q == 0 && d < D) d_depth[(size_t)pp * d_depth_pitch + d] = part;
      acc.x = __fadd_rn(acc.x, __fmul_rn(dv[u], gv[u].x));
      acc.y = __fadd_rn(acc.y, __fmul_rn(dv[u], gv[u].y));
      acc.z = __fadd_rn(acc.z, __fmul_rn(dv[u], gv[u].z));
      acc.w = __fadd_rn(acc.w, __fmul_rn(dv[u], gv[u].w));
    }
  }
  if (active) *(float4 *)(d_feat + (size_t)pp * d_feat_pitch + q * 4) = acc;
}

// ------------------------------------------------------------------ bf16 feature rows (round 3)
// Under a bf16 depthnet the C features of a pixel ARE bf16 values (BF/depth_lss.py:467-468 widens them with x.float()); gathering
// them as stored -- 160-byte rows for C = 80 instead of 320 -- is bit-identical and halves the bytes of the gather this kernel
// is bound by.  A lane owns EIGHT channels (one 16-byte load per member), cq8 = C / 8 lanes per interval / pixel (10 for
// C = 80: six intervals per wave instead of three); products and sums are the same fp32 operations in the same order.
__device__ __forceinline__ void widen8(const uint4 v, float f[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 narrow8(const float f[8]) {
  uint4 v;
  v.x = rne_bf16_bits(f[0]) | (rne_bf16_bits(f[1]) << 16);
  v.y = rne_bf16_bits(f[2]) | (rne_bf16_bits(f[3]) << 16);
  v.z = rne_bf16_bits(f[4]) | (rne_bf16_bits(f[5]) << 16);
  v.w = rne_bf16_bits(f[6]) | (rne_bf16_bits(f[7]) << 16);
  return v;
}

template <bool OUT16>
__global__ __launch_bounds__(256) void lift_splat_fwd16_kernel(
    const float *__restrict__ depth, int depth_pitch, const unsigned short *__restrict__ feat, int feat_pitch,
    const unsigned *__restrict__ pd, const int *__restrict__ starts, const int *__restrict__ lengths,
    const int *__restrict__ cell_of_interval, const int *__restrict__ counts, const int *__restrict__ order, int mmax,
    int cq, int groups, void *__restrict__ out_) {
  const int m = min(mmax, counts[1]);
  const int lane = threadIdx.x & (kWave - 1);
  const long long blk = order ? xcd_chunked_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const long long wave = (blk * blockDim.x + threadIdx.x) >> 6;
  const int g = lane / cq;
  const int q = lane - g * cq;
  const long long kp = wave * groups + g;
  if (g >= groups || kp >= m) return;
  const long long k = order ? order[kp] : kp;
  const int s = starts[k];
  const int len = lengths[k];
  const unsigned *ppd = pd + s;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int i = 0; i < len; i += kU) {
    const int rem = len - i;
    unsigned e[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) e[u] = ppd[i + (u < rem ? u : rem - 1)];
    float dv[kU];
    uint4 fv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const unsigned pix = e[u] >> 8, dd = e[u] & 255u;
      dv[u] = depth[(size_t)pix * depth_pitch + dd];
      fv[u] = *(const uint4 *)(feat + (size_t)pix * feat_pitch + q * 8);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (u < rem) {
        float f[8];
        widen8(fv[u], f);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(dv[u], f[c]));
      }
    }
  }
  const size_t o = (size_t)cell_of_interval[k] * cq + q;  // in units of 8 channels
  if (OUT16) {
    ((uint4 *)out_)[o] = narrow8(acc);
  } else {
    ((float4 *)out_)[2 * o] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    ((float4 *)out_)[2 * o + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

// pixel-stationary backward with bf16 feature rows: cq lanes per pixel, 8 channels per lane; d_feat leaves as bf16 (the fp32
// sum rounded once: what the backward of the reference's x.float() does to it)
template <bool G16>
__global__ __launch_bounds__(256) void lift_splat_bwd16_kernel(
    const void *__restrict__ out_grad_, const float *__restrict__ depth, int depth_pitch,
    const unsigned short *__restrict__ feat, int feat_pitch, const int *__restrict__ cell_of_point, int P_,
    int D, int HW, int cq, int groups, float *__restrict__ d_depth, int d_depth_pitch,
    unsigned short *__restrict__ d_feat, int d_feat_pitch) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long wave = xcd_chunked_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int g = lane / cq;
  const int q = lane - g * cq;
  const long long p = wave * groups + g;
  const bool active = g < groups && p < P_;
  const long long pp = active ? p : 0;
  const int cam = (int)(pp / HW), hw = (int)(pp - (long long)cam * HW);
  const int *cop = cell_of_point + (size_t)cam * D * HW + hw;  // + d*HW
  float f[8];
  {
    const uint4 fr = active ? *(const uint4 *)(feat + (size_t)pp * feat_pitch + q * 8) : make_uint4(0, 0, 0, 0);
    widen8(fr, f);
  }
  const float *dep = depth + (size_t)pp * depth_pitch;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  const int gbase = g * cq;
  for (int d0 = 0; d0 < D; d0 += kU) {
    int cell[kU];
    float dv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int d = d0 + u;
      cell[u] = (active && d < D) ? cop[(size_t)d * HW] : -1;
      dv[u] = (active && d < D) ? dep[d] : 0.f;
    }
    uint4 g16[kU];
    float4 g32[kU][2];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (G16) {
        g16[u] = cell[u] < 0 ? make_uint4(0, 0, 0, 0) : ((const uint4 *)out_grad_)[(size_t)cell[u] * cq + q];
      } else {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        g32[u][0] = cell[u] < 0 ? z : ((const float4 *)out_grad_)[((size_t)cell[u] * cq + q) * 2];
        g32[u][1] = cell[u] < 0 ? z : ((const float4 *)out_grad_)[((size_t)cell[u] * cq + q) * 2 + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      float gv[8];
      if (G16) {
        widen8(g16[u], gv);
      } else {
        gv[0] = g32[u][0].x; gv[1] = g32[u][0].y; gv[2] = g32[u][0].z; gv[3] = g32[u][0].w;
        gv[4] = g32[u][1].x; gv[5] = g32[u][1].y; gv[6] = g32[u][1].z; gv[7] = g32[u][1].w;
      }
      // per-lane partial dot over its 8 channels (ascending), then a fixed-shape tree over the cq lanes
      float part = __fmul_rn(gv[0], f[0]);
#pragma unroll
      for (int c = 1; c < 8; ++c) part = __fadd_rn(part, __fmul_rn(gv[c], f[c]));
      float hi = __shfl(part, gbase + q + 16);
      if (q < 16 && q + 16 < cq) part = __fadd_rn(part, hi);
      for (int o = 8; o > 0; o >>= 1) {
        float other = __shfl(part, gbase + (q ^ o));
        bool has = ((q ^ o) < cq) && ((q ^ o) < 16);
        if (q < 16 && has) part = __fadd_rn(part, other);
      }
      const int d = d0 + u;
      if (active && q == 0 && d < D) d_depth[(size_t)pp * d_depth_pitch + d] = part;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(dv[u], gv[c]));
    }
  }
  if (active) *(uint4 *)(d_feat + (size_t)pp * d_feat_pitch + q * 8) = narrow8(acc);
}

inline int key_bits(unsigned max_key) {
  int b = 1;
  while (b < 32 && (max_key >> b) != 0) ++b;
  return b;
}

struct PlanWs {
  unsigned *keys_in, *vals_in, *keys_out;
  int *blk_starts, *blk_kept;
  void *sort_tmp;
  size_t sort_bytes;
};

inline size_t sort_temp_bytes(long long nprime, int bits) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<rocprim::default_config, unsigned *, unsigned *, unsigned *, unsigned *>(
      nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (size_t)nprime, 0, bits, 0);
  return bytes;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT size_t bfhip_bev_plan_workspace_bytes(long long nprime, long long ncells_times_b) {
  if (nprime <= 0) return 256;
  int bits = key_bits((unsigned)ncells_times_b);
  size_t nb = (size_t)ceil_div(nprime, kScan);
  size_t bytes = 3 * align_up((size_t)nprime * sizeof(unsigned), 256);
  bytes += 2 * align_up((nb + 1) * sizeof(int), 256);
  bytes += align_up(sort_temp_bytes(nprime, bits), 256);
  return bytes + 256;
}

BFHIP_EXPORT int bfhip_bev_plan(const float *frustum, const float *post_trans,
                                const float *post_rots_inv, const float *combine,
                                const float *c2l_trans, const float *extra_rots,
                                const float *extra_trans, int B, int N, int D, int HW,
                                const float *origin_host, const float *dx_host,
                                const int32_t *nx_host, uint32_t *sorted_pd, int32_t *starts,
                                int32_t *lengths, int32_t *cell_of_interval, int32_t *interval_order, int32_t *counts_dev,
                                int32_t *cell_of_point, int32_t *geom_sorted, int64_t *ranks_sorted,
                                uint8_t *kept, float *geom_xyz, int mmax, void *workspace,
                                size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(B > 0 && N > 0 && D > 0 && D <= 256 && HW > 0, "bev_plan: bad sizes B=%d N=%d D=%d HW=%d", B, N, D, HW);
  long long nprime = (long long)B * N * D * HW;
  long long npix = (long long)B * N * HW;
  BFHIP_REQUIRE(npix < (1ll << 24), "bev_plan: more than 2^24 pixels");
  long long ncb = (long long)nx_host[0] * nx_host[1] * nx_host[2] * B;
  BFHIP_REQUIRE(nx_host[0] > 0 && nx_host[1] > 0 && nx_host[2] > 0 && ncb < 0x7fffffffLL, "bev_plan: bad BEV grid");
  BFHIP_REQUIRE(nprime < 0x7fffffffLL, "bev_plan: frustum too large");
  BFHIP_REQUIRE(sorted_pd && starts && lengths && cell_of_interval && counts_dev, "bev_plan: null output");
  BFHIP_REQUIRE(mmax > 0, "bev_plan: mmax must be > 0");
  BFHIP_REQUIRE(geom_sorted == nullptr || ((uintptr_t)geom_sorted % 16) == 0, "bev_plan: geom_sorted must be 16-byte aligned");
  if (workspace_bytes < bfhip_bev_plan_workspace_bytes(nprime, ncb) || !workspace) {
    set_error("bev_plan: workspace too small (%zu < %zu)", workspace_bytes, bfhip_bev_plan_workspace_bytes(nprime, ncb));
    return BFHIP_E_WORKSPACE;
  }
  PlanParams P;
  P.B = B; P.N = N; P.D = D; P.HW = HW;
  P.nx0 = nx_host[0]; P.nx1 = nx_host[1]; P.nx2 = nx_host[2];
  P.ox = origin_host[0]; P.oy = origin_host[1]; P.oz = origin_host[2];
  P.dx0 = dx_host[0]; P.dx1 = dx_host[1]; P.dx2 = dx_host[2];
  P.invalid_key = (unsigned)ncb;
  int bits = key_bits(P.invalid_key);
  int nb = ceil_div(nprime, kScan);
  Workspace ws(workspace, workspace_bytes);
  unsigned *keys_in = ws.take<unsigned>(nprime), *vals_in = ws.take<unsigned>(nprime),
           *keys_out = ws.take<unsigned>(nprime);
  int *blk_starts = ws.take<int>(nb + 1), *blk_kept = ws.take<int>(nb + 1);
  size_t sort_bytes = sort_temp_bytes(nprime, bits);
  char *sort_tmp = ws.take<char>(sort_bytes);
  if (!ws.ok()) { set_error("bev_plan: workspace carve failed"); return BFHIP_E_WORKSPACE; }

  ProfScope ps;
  prof_begin(BFHIP_OP_BEV_AUX, stream, &ps);
  hipLaunchKernelGGL(plan_rank_kernel, dim3(ceil_div(nprime, 256)), dim3(256), 0, stream, frustum,
                     post_trans, post_rots_inv, combine, c2l_trans, extra_rots, extra_trans, P, nprime,
                     keys_in, vals_in, cell_of_point, kept, geom_xyz);
  hipError_t e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_in, keys_out, vals_in, sorted_pd,
                                           (size_t)nprime, 0, bits, stream);
  if (e != hipSuccess) { set_error("bev_plan: rocprim sort: %s", hipGetErrorString(e)); return BFHIP_E_LAUNCH; }
  hipLaunchKernelGGL(plan_flag_count_kernel, dim3(nb), dim3(kScan), 0, stream, keys_out, nprime,
                     P.invalid_key, blk_starts, blk_kept);
  hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(kScan), 0, stream, blk_starts, blk_kept, nb, counts_dev);
  hipLaunchKernelGGL(plan_assign_kernel, dim3(nb), dim3(kScan), 0, stream, keys_out, nprime, P, blk_starts,
                     starts, cell_of_interval, geom_sorted, (long long *)ranks_sorted);
  hipLaunchKernelGGL(plan_lengths_kernel, dim3(ceil_div(mmax, 256)), dim3(256), 0, stream, starts, counts_dev,
                     lengths, mmax);
  if (interval_order) {  // camera-major order of the intervals (the sort buffers are free again)
    const unsigned ngroups = (unsigned)(B * N);
    hipLaunchKernelGGL(plan_group_key_kernel, dim3(ceil_div(mmax, 256)), dim3(256), 0, stream, sorted_pd, starts, counts_dev,
                       HW, mmax, ngroups, keys_in, vals_in);
    e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_in, keys_out, vals_in, (unsigned *)interval_order, (size_t)mmax,
                                  0, key_bits(ngroups), stream);
    if (e != hipSuccess) { set_error("bev_plan: rocprim sort (order): %s", hipGetErrorString(e)); return BFHIP_E_LAUNCH; }
  }
  prof_end(&ps);
  return check_launch("bev_plan");
}

BFHIP_EXPORT int bfhip_lift_splat_fwd(const float *depth, int depth_pitch, const void *feat, int feat_bf16,
                                      int feat_pitch, const uint32_t *sorted_pd,
                                      const int32_t *starts, const int32_t *lengths,
                                      const int32_t *cell_of_interval, const int32_t *interval_order,
                                      const int32_t *counts_dev, int mmax, int C, long long out_cells, void *out,
                                      int out_bf16, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int per_lane = feat_bf16 ? 8 : 4;  // channels per lane = one 16-byte load
  BFHIP_REQUIRE(C > 0 && C % per_lane == 0 && C / per_lane <= kWave,
                "lift_splat_fwd: C must be a multiple of %d and <= %d (C=%d)", per_lane, per_lane * kWave, C);
  BFHIP_REQUIRE(feat_pitch % per_lane == 0 && ((uintptr_t)feat % 16) == 0 && ((uintptr_t)out % 16) == 0,
                "lift_splat_fwd: feat/out must be 16-byte aligned with a pitch that is a multiple of 16 bytes");
  BFHIP_REQUIRE(depth && feat && sorted_pd && starts && lengths && cell_of_interval && counts_dev && out,
                "lift_splat_fwd: null pointer");
  BFHIP_REQUIRE(mmax > 0 && out_cells > 0, "lift_splat_fwd: bad mmax/out_cells");
  if (hipMemsetAsync(out, 0, (size_t)out_cells * C * (out_bf16 ? 2 : 4), stream) != hipSuccess)
    return check_launch("lift_splat_fwd memset");
  int cq = C / per_lane, groups = kWave / cq;
  long long waves = ((long long)mmax + groups - 1) / groups;
  ProfScope ps;
  prof_begin(BFHIP_OP_LIFT_SPLAT_FWD, stream, &ps);
  if (feat_bf16 && out_bf16)
    hipLaunchKernelGGL(lift_splat_fwd16_kernel<true>, dim3(ceil_div(waves * kWave, 256)), dim3(256), 0, stream, depth,
                       depth_pitch, (const unsigned short *)feat, feat_pitch, sorted_pd, starts, lengths, cell_of_interval,
                       counts_dev, interval_order, mmax, cq, groups, out);
  else if (feat_bf16)
    hipLaunchKernelGGL(lift_splat_fwd16_kernel<false>, dim3(ceil_div(waves * kWave, 256)), dim3(256), 0, stream, depth,
                       depth_pitch, (const unsigned short *)feat, feat_pitch, sorted_pd, starts, lengths, cell_of_interval,
                       counts_dev, interval_order, mmax, cq, groups, out);
  else if (out_bf16)
    hipLaunchKernelGGL(lift_splat_fwd_kernel<true>, dim3(ceil_div(waves * kWave, 256)), dim3(256), 0, stream, depth,
                       depth_pitch, (const float *)feat, feat_pitch, sorted_pd, starts, lengths, cell_of_interval, counts_dev,
                       interval_order, mmax, cq, groups, out);
  else
    hipLaunchKernelGGL(lift_splat_fwd_kernel<false>, dim3(ceil_div(waves * kWave, 256)), dim3(256), 0, stream, depth,
                       depth_pitch, (const float *)feat, feat_pitch, sorted_pd, starts, lengths, cell_of_interval, counts_dev,
                       interval_order, mmax, cq, groups, out);
  prof_end(&ps);
  return check_launch("lift_splat_fwd");
}

BFHIP_EXPORT int bfhip_lift_splat_bwd(const void *out_grad, int grad_bf16, const float *depth, int depth_pitch,
                                      const void *feat, int feat_bf16, int feat_pitch,
                                      const int32_t *cell_of_point, int num_cams, int D, int HW,
                                      int C, float *d_depth, int d_depth_pitch, void *d_feat,
                                      int d_feat_pitch, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int per_lane = feat_bf16 ? 8 : 4;
  BFHIP_REQUIRE(C > 0 && C % per_lane == 0 && C / per_lane <= (feat_bf16 ? 32 : kWave),
                "lift_splat_bwd: C must be a multiple of %d and <= 256 (C=%d)", per_lane, C);
  BFHIP_REQUIRE(num_cams > 0 && D > 0 && HW > 0, "lift_splat_bwd: bad sizes");
  BFHIP_REQUIRE(feat_pitch % per_lane == 0 && d_feat_pitch % per_lane == 0 && ((uintptr_t)feat % 16) == 0 &&
                    ((uintptr_t)d_feat % 16) == 0 && ((uintptr_t)out_grad % ((grad_bf16 && !feat_bf16) ? 8 : 16)) == 0,
                "lift_splat_bwd: feat/d_feat/out_grad must be 16-byte aligned, pitches multiples of 16 bytes");
  BFHIP_REQUIRE(out_grad && depth && feat && cell_of_point && d_depth && d_feat, "lift_splat_bwd: null pointer");
  long long npix = (long long)num_cams * HW;
  int cq = C / per_lane, groups = kWave / cq;
  long long waves = (npix + groups - 1) / groups;
  ProfScope ps;
  prof_begin(BFHIP_OP_LIFT_SPLAT_BWD, stream, &ps);
  if (feat_bf16 && grad_bf16)
    hipLaunchKernelGGL(lift_splat_bwd16_kernel<true>, dim3(ceil_div(waves * kWave, 256)), dim3(256), 0, stream, out_grad, depth,
                       depth_pitch, (const unsigned short *)feat, feat_pitch, cell_of_point, (int)npix, D, HW, cq, groups, d_depth,
                       d_depth_pitch, (unsigned short *)d_feat, d_feat_pitch);
  else if (feat_bf16)
    hipLaunchKernelGGL(lift_splat_bwd16_kernel<false>, dim3(ceil_div(waves * kWave, 256)), dim3(256), 0, stream, out_grad, depth,
                       depth_pitch, (const unsigned short *)feat, feat_pitch, cell_of_point, (int)npix, D, HW, cq, groups, d_depth,
                       d_depth_pitch, (unsigned short *)d_feat, d_feat_pitch);
  else if (grad_bf16)
    hipLaunchKernelGGL(lift_splat_bwd_kernel<true>, dim3(ceil_div(waves * kWave, 256)), dim3(256), 0, stream, out_grad, depth,
                       depth_pitch, (const float *)feat, feat_pitch, cell_of_point, (int)npix, D, HW, cq, groups, d_depth,
                       d_depth_pitch, (float *)d_feat, d_feat_pitch);
  else
    hipLaunchKernelGGL(lift_splat_bwd_kernel<false>, dim3(ceil_div(waves * kWave, 256)), dim3(256), 0, stream, out_grad, depth,
                       depth_pitch, (const float *)feat, feat_pitch, cell_of_point, (int)npix, D, HW, cq, groups, d_depth,
                       d_depth_pitch, (float *)d_feat, d_feat_pitch);
  prof_end(&ps);
  return check_launch("lift_splat_bwd");
}
