// scatter.hip -- DynamicScatter: unique voxels (sorted) + max/mean/sum reduce of point features, fwd+bwd.
//
// Replaces BF/ops/voxel/src/scatter_points_cuda.cu: dynamic_point_to_voxel_forward_gpu :183-239
// (at::unique_dim sort + atomic CAS-max / atomicAdd reduce, order-dependent sums) and
// dynamic_point_to_voxel_backward_gpu :241-308.
//   1. key   : rows with any negative coordinate are invalid (:202); valid rows pack (c0,c1,c2) into a
//              63-bit key whose integer order is the lexicographic order unique_dim(sorted=True) uses
//   2. sort  : stable radix sort of (key, point index)  -> points of a voxel are contiguous, in point order
//   3. ids   : first-of-run flags -> prefix sum -> voxel id; voxel_coors, counts, point2voxel
//   4. reduce: one thread per (voxel, channel) walks the voxel's points IN POINT ORDER -> sums are
//              deterministic (the reference's atomics are not)
#include "common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace bfhip {
namespace {

constexpr int kScan = 1024;
constexpr unsigned long long kInvalid = ~0ull;

__global__ __launch_bounds__(256) void scatter_key_kernel(const int *__restrict__ coors, int N,
                                                          unsigned long long *__restrict__ keys,
                                                          unsigned *__restrict__ vals,
                                                          int *__restrict__ err) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c0 = coors[(size_t)i * 3], c1 = coors[(size_t)i * 3 + 1], c2 = coors[(size_t)i * 3 + 2];
  unsigned long long k = kInvalid;
  if (c0 >= 0 && c1 >= 0 && c2 >= 0) {
    if (c0 >= (1 << 21) || c1 >= (1 << 21) || c2 >= (1 << 21)) *err = 1;  // coordinate too large for the key
    k = ((unsigned long long)c0 << 42) | ((unsigned long long)(c1 & 0x1fffff) << 21) | (unsigned long long)(c2 & 0x1fffff);
  }
  keys[i] = k;
  vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(kScan) void scatter_flag_count_kernel(const unsigned long long *__restrict__ keys,
                                                                   int N, int *__restrict__ blk) {
  __shared__ int sm[kScan / 64];
  int i = blockIdx.x * kScan + threadIdx.x;
  int flag = 0;
  if (i < N) {
    unsigned long long k = keys[i];
    flag = k != kInvalid && (i == 0 || keys[i - 1] != k);
  }
  for (int o = 32; o > 0; o >>= 1) flag += __shfl_down(flag, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = flag;
  __syncthreads();
  if (threadIdx.x == 0) {
    int r = 0;
    for (int k = 0; k < kScan / 64; ++k) r += sm[k];
    blk[blockIdx.x] = r;
  }
}

__global__ __launch_bounds__(kScan) void scatter_scan_kernel(int *__restrict__ blk, int nb, int *__restrict__ total) {
  __shared__ int sm[kScan];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += kScan) {
    int i = base + threadIdx.x;
    int v = i < nb ? blk[i] : 0;
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
    if (i < nb) blk[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == kScan - 1) carry = c + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// voxel id of every sorted position; starts of each voxel; coors; point2voxel
__global__ __launch_bounds__(kScan) void scatter_assign_kernel(const unsigned long long *__restrict__ keys,
                                                               const unsigned *__restrict__ sorted_idx, int N,
                                                               const int *__restrict__ blk_offs,
                                                               int *__restrict__ voxel_start,
                                                               int *__restrict__ voxel_coors,
                                                               int *__restrict__ point2voxel) {
  __shared__ int wsum[kScan / 64];
  int i = blockIdx.x * kScan + threadIdx.x;
  unsigned long long k = kInvalid;
  int flag = 0;
  if (i < N) {
    k = keys[i];
    flag = k != kInvalid && (i == 0 || keys[i - 1] != k);
  }
  unsigned long long bal = __ballot(flag);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  int woff = 0;
  for (int j = 0; j < wv; ++j) woff += wsum[j];
  // inclusive count of starts up to and including i, minus 1 = voxel id of position i
  int vid = blk_offs[blockIdx.x] + woff + __popcll(bal & ((2ull << lane) - 1ull)) - 1;
  if (i < N) {
    if (k == kInvalid) {
      point2voxel[sorted_idx[i]] = -1;
    } else {
      point2voxel[sorted_idx[i]] = vid;
      if (flag) {
        voxel_start[vid] = i;
        voxel_coors[(size_t)vid * 3 + 0] = (int)(k >> 42);
        voxel_coors[(size_t)vid * 3 + 1] = (int)((k >> 21) & 0x1fffff);
        voxel_coors[(size_t)vid * 3 + 2] = (int)(k & 0x1fffff);
      }
    }
  }
}

// NOTE: a voxel run can straddle a block boundary: vid for a non-start position in a later block is
// (starts before it) - 1, which the inclusive count above provides because blk_offs is exclusive.

__global__ __launch_bounds__(256) void scatter_reduce_kernel(const float *__restrict__ feats, int C,
                                                             const unsigned *__restrict__ sorted_idx,
                                                             const int *__restrict__ voxel_start,
                                                             const int *__restrict__ counts2, int n_valid_end_unused,
                                                             int mcap, int reduce_type,
                                                             float *__restrict__ voxel_feats,
                                                             int *__restrict__ voxel_count) {
  (void)n_valid_end_unused;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long v = t / C;
  int c = (int)(t - v * C);
  int M = counts2[0];
  if (v >= M || v >= mcap) return;
  int s = voxel_start[v];
  int e = (v + 1 < M) ? voxel_start[v + 1] : counts2[1];  // counts2[1] = number of valid points
  float acc = (reduce_type == BFHIP_REDUCE_MAX) ? -INFINITY : 0.f;
  for (int i = s; i < e; ++i) {
    float f = feats[(size_t)sorted_idx[i] * C + c];
    acc = (reduce_type == BFHIP_REDUCE_MAX) ? fmaxf(acc, f) : acc + f;
  }
  if (reduce_type == BFHIP_REDUCE_MEAN) acc = acc / (float)(e - s);
  voxel_feats[(size_t)v * C + c] = acc;
  if (c == 0) voxel_count[v] = e - s;
}

__global__ __launch_bounds__(256) void count_valid_kernel(const unsigned long long *__restrict__ keys, int N,
                                                          int *__restrict__ n_valid) {
  // sorted keys: number of valid = first index whose key is invalid (binary search by one thread)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int lo = 0, hi = N;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (keys[mid] == kInvalid) hi = mid; else lo = mid + 1;
    }
    *n_valid = lo;
  }
}

__global__ __launch_bounds__(256) void scatter_bwd_add_kernel(float *__restrict__ grad_feats,
                                                              const float *__restrict__ grad_voxel,
                                                              const int *__restrict__ point2voxel,
                                                              const int *__restrict__ count, long long total,
                                                              int C, int mean) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long long i = t / C;
  int c = (int)(t - i * C);
  int v = point2voxel[i];
  float g = 0.f;
  if (v >= 0) {
    g = grad_voxel[(size_t)v * C + c];
    if (mean) g = g / (float)count[v];
  }
  grad_feats[t] = g;
}

__global__ __launch_bounds__(256) void scatter_bwd_argmax_kernel(const float *__restrict__ feats,
                                                                 const float *__restrict__ voxel_feats,
                                                                 const int *__restrict__ point2voxel,
                                                                 long long total, int C,
                                                                 int *__restrict__ reduce_from) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long long i = t / C;
  int c = (int)(t - i * C);
  int v = point2voxel[i];
  if (v < 0) return;
  if (feats[t] == voxel_feats[(size_t)v * C + c]) atomicMin(&reduce_from[(size_t)v * C + c], (int)i);
}

__global__ __launch_bounds__(256) void scatter_bwd_max_kernel(float *__restrict__ grad_feats,
                                                              const float *__restrict__ grad_voxel,
                                                              const int *__restrict__ reduce_from,
                                                              long long total_mc, int C, int N) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_mc) return;
  int c = (int)(t % C);
  int src = reduce_from[t];
  if (src >= 0 && src < N) grad_feats[(size_t)src * C + c] = grad_voxel[t];
}

inline size_t sort_bytes64(int N) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<rocprim::default_config, unsigned long long *, unsigned long long *, unsigned *,
                                  unsigned *>(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (size_t)N, 0, 64, 0);
  return bytes;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT size_t bfhip_dynamic_scatter_workspace_bytes(int N) {
  if (N <= 0) return 256;
  size_t nb = (size_t)ceil_div(N, kScan);
  return 2 * align_up((size_t)N * 8, 256) + 2 * align_up((size_t)N * 4, 256) + align_up((size_t)N * 4, 256) +
         align_up((nb + 1) * 4, 256) + align_up(sort_bytes64(N), 256) + 512;
}

// forward.  Outputs sized for N rows (M <= N); counts_dev[0] = M, counts_dev[1] = valid points,
// counts_dev[2] = 1 when a coordinate does not fit the 21-bit key (error).
BFHIP_EXPORT int bfhip_dynamic_scatter_fwd(const float *feats, const int32_t *coors, int N, int C, int reduce_type,
                                           float *voxel_feats, int32_t *voxel_coors, int32_t *point2voxel,
                                           int32_t *voxel_count, int32_t *counts_dev, void *workspace,
                                           size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(N >= 0 && C > 0, "dynamic_scatter_fwd: bad sizes");
  BFHIP_REQUIRE(reduce_type >= 0 && reduce_type <= 2, "dynamic_scatter_fwd: bad reduce type %d", reduce_type);
  BFHIP_REQUIRE(counts_dev, "dynamic_scatter_fwd: counts_dev is null");
  if (hipMemsetAsync(counts_dev, 0, 3 * sizeof(int), stream) != hipSuccess) return check_launch("dynamic_scatter memset");
  if (N == 0) return BFHIP_OK;
  BFHIP_REQUIRE(feats && coors && voxel_feats && voxel_coors && point2voxel && voxel_count, "dynamic_scatter_fwd: null pointer");
  if (workspace_bytes < bfhip_dynamic_scatter_workspace_bytes(N) || !workspace) { set_error("dynamic_scatter_fwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  Workspace ws(workspace, workspace_bytes);
  unsigned long long *keys_in = ws.take<unsigned long long>(N), *keys = ws.take<unsigned long long>(N);
  unsigned *vals_in = ws.take<unsigned>(N), *sorted_idx = ws.take<unsigned>(N);
  int *voxel_start = ws.take<int>(N);
  int nb = ceil_div(N, kScan);
  int *blk = ws.take<int>(nb + 1);
  size_t sb = sort_bytes64(N);
  char *tmp = ws.take<char>(sb);
  if (!ws.ok()) { set_error("dynamic_scatter_fwd: workspace carve failed"); return BFHIP_E_WORKSPACE; }
  ProfScope ps;
  prof_begin(BFHIP_OP_SCATTER_FWD, stream, &ps);
  hipLaunchKernelGGL(scatter_key_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, stream, coors, N, keys_in, vals_in, counts_dev + 2);
  hipError_t e = rocprim::radix_sort_pairs(tmp, sb, keys_in, keys, vals_in, sorted_idx, (size_t)N, 0, 64, stream);
  if (e != hipSuccess) { set_error("dynamic_scatter_fwd: sort: %s", hipGetErrorString(e)); return BFHIP_E_LAUNCH; }
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(64), 0, stream, keys, N, counts_dev + 1);
  hipLaunchKernelGGL(scatter_flag_count_kernel, dim3(nb), dim3(kScan), 0, stream, keys, N, blk);
  hipLaunchKernelGGL(scatter_scan_kernel, dim3(1), dim3(kScan), 0, stream, blk, nb, counts_dev);
  hipLaunchKernelGGL(scatter_assign_kernel, dim3(nb), dim3(kScan), 0, stream, keys, sorted_idx, N, blk, voxel_start,
                     voxel_coors, point2voxel);
  hipLaunchKernelGGL(scatter_reduce_kernel, dim3(ceil_div((long long)N * C, 256)), dim3(256), 0, stream, feats, C, sorted_idx,
                     voxel_start, counts_dev, 0, N, reduce_type, voxel_feats, voxel_count);
  prof_end(&ps);
  return check_launch("dynamic_scatter_fwd");
}

BFHIP_EXPORT size_t bfhip_dynamic_scatter_bwd_workspace_bytes(int M, int C) {
  return align_up((size_t)(M > 0 ? M : 1) * C * sizeof(int), 256) + 256;
}

BFHIP_EXPORT int bfhip_dynamic_scatter_bwd(float *grad_feats, const float *grad_voxel_feats, const float *feats,
                                           const float *voxel_feats, const int32_t *point2voxel,
                                           const int32_t *voxel_count, int N, int M, int C, int reduce_type,
                                           void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(N >= 0 && M >= 0 && C > 0, "dynamic_scatter_bwd: bad sizes");
  BFHIP_REQUIRE(reduce_type >= 0 && reduce_type <= 2, "dynamic_scatter_bwd: bad reduce type %d", reduce_type);
  if (N == 0) return BFHIP_OK;
  BFHIP_REQUIRE(grad_feats, "dynamic_scatter_bwd: grad_feats is null");
  if (M == 0) {
    if (hipMemsetAsync(grad_feats, 0, (size_t)N * C * sizeof(float), stream) != hipSuccess) return check_launch("dynamic_scatter_bwd memset");
    return BFHIP_OK;
  }
  BFHIP_REQUIRE(grad_voxel_feats && point2voxel && voxel_count, "dynamic_scatter_bwd: null pointer");
  long long total = (long long)N * C;
  ProfScope ps;
  prof_begin(BFHIP_OP_SCATTER_BWD, stream, &ps);
  if (reduce_type != BFHIP_REDUCE_MAX) {
    hipLaunchKernelGGL(scatter_bwd_add_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, grad_feats, grad_voxel_feats,
                       point2voxel, voxel_count, total, C, reduce_type == BFHIP_REDUCE_MEAN);
  } else {
    BFHIP_REQUIRE(feats && voxel_feats, "dynamic_scatter_bwd: null pointer");
    if (workspace_bytes < bfhip_dynamic_scatter_bwd_workspace_bytes(M, C) || !workspace) { set_error("dynamic_scatter_bwd: workspace too small"); return BFHIP_E_WORKSPACE; }
    int *reduce_from = (int *)workspace;
    hipMemsetAsync(grad_feats, 0, (size_t)N * C * sizeof(float), stream);
    (void)hipMemsetAsync(reduce_from, 0x7f, (size_t)M * C * sizeof(int), stream);
    hipLaunchKernelGGL(scatter_bwd_argmax_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, feats, voxel_feats,
                       point2voxel, total, C, reduce_from);
    long long mc = (long long)M * C;
    hipLaunchKernelGGL(scatter_bwd_max_kernel, dim3(ceil_div(mc, 256)), dim3(256), 0, stream, grad_feats, grad_voxel_feats,
                       reduce_from, mc, C, N);
  }
  prof_end(&ps);
  return check_launch("dynamic_scatter_bwd");
}

// Mean of hard voxels (BF/bevfusion.py:251-253): feats[v][f] = sum_p voxels[v][p][f] / num_points[v]
// sequential fp32 sum over the P slots (zero padded) like torch.sum(dim=1).
namespace bfhip { namespace {
__global__ __launch_bounds__(256) void voxel_mean_kernel(const float *__restrict__ voxels, const int *__restrict__ num,
                                                         long long total, int P, int F, float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long long v = t / F;
  int f = (int)(t - v * F);
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += voxels[((size_t)v * P + p) * F + f];
  out[t] = s / (float)num[v];
}

// BEVFusion.voxelize (BF/bevfusion.py:227-255) without its per-sample host reads: the B samples' hard-voxelization outputs
// (each MV rows, true counts on the device) -> ONE capacity-sized [cap, F] matrix of per-voxel means and [cap, 4] coordinates
// (b, x, y, z) whose ACTIVE rows are the prefix [0, min(sum of counts, cap)), in sample order like the reference's torch.cat;
// the remaining rows are zeros with batch index -1 (inactive).  n_total[0] = active rows, n_total[1] = sum of counts.
__global__ __launch_bounds__(256) void voxel_compact_mean_kernel(const float *__restrict__ voxels, const int *__restrict__ coors,
                                                                 const int *__restrict__ num, const int *__restrict__ counts,
                                                                 int B, int MV, int P, int F, int cap,
                                                                 float *__restrict__ feats, int4 *__restrict__ out_coords,
                                                                 int *__restrict__ n_total) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)cap * F) return;
  const int r = (int)(t / F), f = (int)(t - (long long)r * F);
  int off = 0, b = -1, local = 0;
  for (int k = 0; k < B; ++k) {
    int c = counts[k];
    c = c < 0 ? 0 : (c > MV ? MV : c);
    if (b < 0 && r < off + c) { b = k; local = r - off; }
    off += c;
  }
  if (t == 0) { n_total[0] = off < cap ? off : cap; n_total[1] = off; }
  if (b < 0) {
    feats[t] = 0.f;
    if (f == 0) out_coords[r] = make_int4(-1, -1, -1, -1);
    return;
  }
  const size_t v = (size_t)b * MV + local;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += voxels[(v * P + p) * F + f];
  feats[t] = s / (float)num[v];
  if (f == 0) out_coords[r] = make_int4(b, coors[v * 3 + 0], coors[v * 3 + 1], coors[v * 3 + 2]);
}
} }

BFHIP_EXPORT int bfhip_voxel_mean(const float *voxels, const int32_t *num_points, int M, int P, int F, float *out,
                                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(M >= 0 && P > 0 && F > 0, "voxel_mean: bad sizes");
  if (M == 0) return BFHIP_OK;
  BFHIP_REQUIRE(voxels && num_points && out, "voxel_mean: null pointer");
  long long total = (long long)M * F;
  hipLaunchKernelGGL(voxel_mean_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, voxels, num_points, total, P, F, out);
  return check_launch("voxel_mean");
}

BFHIP_EXPORT int bfhip_voxel_compact_mean(const float *voxels, const int32_t *coors, const int32_t *num_points,
                                          const int32_t *counts_dev, int B, int max_voxels, int P, int F, int cap,
                                          float *feats, int32_t *out_coords, int32_t *n_total_dev, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(B > 0 && B <= 64 && max_voxels > 0 && P > 0 && F > 0 && cap > 0, "voxel_compact_mean: bad sizes");
  BFHIP_REQUIRE(voxels && coors && num_points && counts_dev && feats && out_coords && n_total_dev, "voxel_compact_mean: null pointer");
  BFHIP_REQUIRE(((uintptr_t)out_coords % 16) == 0, "voxel_compact_mean: out_coords must be 16-byte aligned");
  hipLaunchKernelGGL(voxel_compact_mean_kernel, dim3(ceil_div((long long)cap * F, 256)), dim3(256), 0, stream, voxels, coors,
                     num_points, counts_dev, B, max_voxels, P, F, cap, feats, (int4 *)out_coords, n_total_dev);
  return check_launch("voxel_compact_mean");
}
