// voxelize.hip -- dynamic and deterministic hard voxelization for gfx950.
//
// Replaces BF/ops/voxel/src/voxelization_cuda.cu (dynamic_voxelize_kernel :24-61,
// point_to_voxelidx_kernel :105-147 [O(N^2)], determin_voxel_num :149-180 [<<<1,1>>>],
// assign_point_to_voxel :63-84, assign_voxel_coors :86-103, wrapper :231-373 with four
// cudaDeviceSynchronize and a D2H copy) with a sync-free pipeline that yields the identical
// first-come result of hard_voxelize_cpu (voxelization_cpu.cpp:46-101):
//
//   1. hash   : per point, IEEE fp32 (p - min) / voxel, floor -> linear cell key; open-addressing
//               hash insert (CAS on the key) and atomicMin of the point index -> first[slot]
//               = first point of that voxel; atomicAdd count[slot]
//   2. order  : flag = (first[slot[i]] == i); exclusive prefix sum of the flags over the point
//               index = voxel id in first-occurrence order (the serial order of the CPU loop);
//               ids >= max_voxels are dropped with all their points (voxelization_cpu.cpp:80)
//   3. rank   : each kept point is inserted into its voxel's sorted list of the max_points
//               smallest point indices by an atomicMin cascade (exact, order-independent)
//   4. scatter: voxels[v][r][:] = points[list[v][r]][:], num_points = min(count, max_points)
//
// Work is O(N * max_points) worst case instead of the reference's O(N^2).  The table is 8-16 B
// per point and lives in L2; the op is launch/latency bound (N = 40k moves < 6 MB).
#include "common.h"

namespace bfhip {
namespace {

constexpr int kInf = 0x7f7f7f7f;  // memset-able "+inf" for point indices
constexpr int kScanBlock = 1024;

struct VoxParams {
  float vx, vy, vz;
  float x0, y0, z0;
  int gx, gy, gz;
};

// voxelization_cpu.cpp:24-29 / voxelization_cuda.cu:37-41: c = floor((p - min) / voxel).
// fp32 subtract then IEEE divide (no reciprocal, no fma), range test on the floored float so
// that NaN / huge values fail exactly like the CPU's int conversion does.
__device__ __forceinline__ bool coord(float p, float lo, float vs, int grid, int &c) {
  float f = floorf(__fdiv_rn(__fsub_rn(p, lo), vs));
  if (!(f >= 0.0f && f < (float)grid)) return false;
  c = (int)f;
  return true;
}

__device__ __forceinline__ bool point_cell(const float *__restrict__ p, const VoxParams &P, int &cx,
                                           int &cy, int &cz) {
  return coord(p[0], P.x0, P.vx, P.gx, cx) && coord(p[1], P.y0, P.vy, P.gy, cy) &&
         coord(p[2], P.z0, P.vz, P.gz, cz);
}

__global__ __launch_bounds__(256) void dynamic_voxelize_kernel(const float *__restrict__ points,
                                                               int *__restrict__ coors, int n,
                                                               int f, VoxParams P) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  bool ok = point_cell(points + (size_t)i * f, P, cx, cy, cz);
  int *c = coors + (size_t)i * 3;
  c[0] = ok ? cx : -1;
  c[1] = ok ? cy : -1;
  c[2] = ok ? cz : -1;
}

__device__ __forceinline__ unsigned hash32(unsigned k) {
  k ^= k >> 16; k *= 0x85ebca6bu; k ^= k >> 13; k *= 0xc2b2ae35u; k ^= k >> 16;
  return k;
}

// ---- 0. table init: keys=-1, first=+inf, count=0, lists=+inf, voxel_num=0
__global__ __launch_bounds__(256) void vox_init_kernel(int *__restrict__ keys,
                                                       int *__restrict__ first,
                                                       int *__restrict__ count, int cap,
                                                       int *__restrict__ lists, long long nlist,
                                                       int *__restrict__ voxel_num) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  if (i == 0) *voxel_num = 0;
  for (long long j = i; j < cap; j += stride) {
    keys[j] = -1;
    first[j] = kInf;
    count[j] = 0;
  }
  for (long long j = i; j < nlist; j += stride) lists[j] = kInf;
}

// ---- 1. hash insert
__global__ __launch_bounds__(256) void vox_hash_kernel(const float *__restrict__ points, int n,
                                                       int f, VoxParams P, int *__restrict__ keys,
                                                       int *__restrict__ first,
                                                       int *__restrict__ count, unsigned mask,
                                                       int *__restrict__ slot_of) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  if (!point_cell(points + (size_t)i * f, P, cx, cy, cz)) {
    slot_of[i] = -1;
    return;
  }
  int key = (cx * P.gy + cy) * P.gz + cz;
  unsigned s = hash32((unsigned)key) & mask;
  for (unsigned probe = 0; probe <= mask; ++probe) {  // bounded: table is >= 2x the point count
    int old = atomicCAS(&keys[s], -1, key);
    if (old == -1 || old == key) break;
    s = (s + 1) & mask;
  }
  atomicMin(&first[s], i);
  atomicAdd(&count[s], 1);
  slot_of[i] = (int)s;
}

__device__ __forceinline__ int block_reduce_sum(int v, int *sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  int r = 0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sm[i];
    sm[0] = r;
  }
  __syncthreads();
  r = sm[0];
  __syncthreads();
  return r;
}

// ---- 2a. per-block count of first points
__global__ __launch_bounds__(kScanBlock) void vox_flag_count_kernel(
    const int *__restrict__ slot_of, const int *__restrict__ first, int n,
    int *__restrict__ block_sums) {
  __shared__ int sm[kScanBlock / 64];
  int i = blockIdx.x * kScanBlock + threadIdx.x;
  int flag = 0;
  if (i < n) {
    int s = slot_of[i];
    flag = (s >= 0 && first[s] == i) ? 1 : 0;
  }
  int tot = block_reduce_sum(flag, sm);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// ---- 2b. single-block exclusive scan of the block sums (nb <= a few thousand)
__global__ __launch_bounds__(kScanBlock) void vox_scan_blocks_kernel(int *__restrict__ block_sums,
                                                                     int nb, int max_voxels,
                                                                     int *__restrict__ voxel_num) {
  __shared__ int sm[kScanBlock];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += kScanBlock) {
    int i = base + threadIdx.x;
    int v = i < nb ? block_sums[i] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < kScanBlock; o <<= 1) {  // Hillis-Steele inclusive scan
      int t = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    int incl = sm[threadIdx.x];
    int c = carry;
    if (i < nb) block_sums[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == kScanBlock - 1) carry = c + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *voxel_num = (max_voxels >= 0 && carry > max_voxels) ? max_voxels : carry;
}

// ---- 2c. voxel ids in first-occurrence order; coors + clipped counts
__global__ __launch_bounds__(kScanBlock) void vox_assign_kernel(
    const int *__restrict__ slot_of, const int *__restrict__ first, const int *__restrict__ keys,
    const int *__restrict__ count, int n, const int *__restrict__ block_offs, int max_points,
    int max_voxels, VoxParams P, int *__restrict__ vid_of_slot, int *__restrict__ coors,
    int *__restrict__ num_points_per_voxel) {
  __shared__ int wsum[kScanBlock / 64];
  int i = blockIdx.x * kScanBlock + threadIdx.x;
  int s = -1, flag = 0;
  if (i < n) {
    s = slot_of[i];
    flag = (s >= 0 && first[s] == i) ? 1 : 0;
  }
  // block exclusive scan of flag: wave ballot + per-wave offsets
  unsigned long long bal = __ballot(flag);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < wv; ++k) woff += wsum[k];
  if (flag) {
    int v = block_offs[blockIdx.x] + woff + in_wave;
    bool keep = (max_voxels < 0) || (v < max_voxels);
    vid_of_slot[s] = keep ? v : -1;
    if (keep) {
      int key = keys[s];
      int cz = key % P.gz;
      int t = key / P.gz;
      int cy = t % P.gy;
      int cx = t / P.gy;
      coors[(size_t)v * 3 + 0] = cx;
      coors[(size_t)v * 3 + 1] = cy;
      coors[(size_t)v * 3 + 2] = cz;
      int cnt = count[s];
      num_points_per_voxel[v] = cnt < max_points ? cnt : max_points;
    }
  }
}

// ---- 3. sorted insertion of the point index into its voxel's list (max_points smallest).
// atomicMin cascade: slot r keeps the smaller of (resident, carried); the larger is carried on.
// At quiescence list[v] holds the max_points smallest indices ascending, whatever the
// interleaving (each value visits slots in order; the minimum of everything that reaches
// slot r stays there).
__global__ __launch_bounds__(256) void vox_rank_kernel(const int *__restrict__ slot_of,
                                                       const int *__restrict__ vid_of_slot,
                                                       const int *__restrict__ count, int n,
                                                       int max_points, int *__restrict__ lists) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = slot_of[i];
  if (s < 0) return;
  int v = vid_of_slot[s];
  if (v < 0) return;
  int *l = lists + (size_t)v * max_points;
  if (count[s] == 1) {  // sole point of its voxel
    l[0] = i;
    return;
  }
  int carried = i;
  for (int r = 0; r < max_points; ++r) {
    int old = atomicMin(&l[r], carried);
    if (old == kInf) break;          // slot was empty: carried value placed, nothing displaced
    if (old > carried) carried = old;  // we took the slot, push the former resident on
    // else: resident is smaller, keep carrying our value
  }
}

// ---- 4. gather point rows into voxels[v][r][:]
__global__ __launch_bounds__(256) void vox_scatter_kernel(const float *__restrict__ points, int f,
                                                          const int *__restrict__ lists,
                                                          const int *__restrict__ voxel_num,
                                                          int max_points,
                                                          float *__restrict__ voxels,
                                                          long long total) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long long vr = t / f;
  int k = (int)(t - vr * f);
  long long v = vr / max_points;
  if (v >= *voxel_num) return;
  int idx = lists[vr];
  if (idx == kInf) return;
  voxels[t] = points[(size_t)idx * f + k];
}

inline int make_params(const float *vs, const float *cr, VoxParams &P) {
  P.vx = vs[0]; P.vy = vs[1]; P.vz = vs[2];
  P.x0 = cr[0]; P.y0 = cr[1]; P.z0 = cr[2];
  // voxelization_cpu.cpp:121-124: grid = round((max - min) / voxel) in float arithmetic
  P.gx = (int)round((double)((cr[3] - cr[0]) / vs[0]));
  P.gy = (int)round((double)((cr[4] - cr[1]) / vs[1]));
  P.gz = (int)round((double)((cr[5] - cr[2]) / vs[2]));
  if (P.gx <= 0 || P.gy <= 0 || P.gz <= 0) return -1;
  if ((long long)P.gx * P.gy * P.gz >= 0x7fffffffLL) return -1;
  return 0;
}

inline unsigned table_cap(int n) {
  unsigned cap = 1024;
  while (cap < 2u * (unsigned)n) cap <<= 1;
  return cap;
}

inline long long list_rows(int n, int max_voxels) {
  return (max_voxels >= 0 && max_voxels < n) ? max_voxels : n;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_dynamic_voxelize(const float *points, int32_t *coors, int n, int f,
                                        const float *voxel_size_host,
                                        const float *coors_range_host, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(n >= 0 && f >= 3, "dynamic_voxelize: bad sizes n=%d f=%d", n, f);
  VoxParams P;
  BFHIP_REQUIRE(make_params(voxel_size_host, coors_range_host, P) == 0,
                "dynamic_voxelize: bad voxel_size/coors_range");
  if (n == 0) return BFHIP_OK;
  BFHIP_REQUIRE(points && coors, "dynamic_voxelize: null pointer");
  hipLaunchKernelGGL(dynamic_voxelize_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, points,
                     coors, n, f, P);
  return check_launch("dynamic_voxelize");
}

BFHIP_EXPORT size_t bfhip_hard_voxelize_workspace_bytes(int n, int max_points, int max_voxels) {
  if (n <= 0) return 256;
  size_t cap = table_cap(n);
  size_t nb = (size_t)ceil_div(n, kScanBlock);
  size_t bytes = 0;
  bytes += 4 * align_up(cap * sizeof(int), 256);                     // keys, first, count, vid
  bytes += align_up((size_t)n * sizeof(int), 256);                   // slot_of
  bytes += align_up((nb + 1) * sizeof(int), 256);                    // block sums
  bytes += align_up((size_t)list_rows(n, max_voxels) * (size_t)(max_points > 0 ? max_points : 1) * sizeof(int), 256);
  return bytes + 256;
}

BFHIP_EXPORT int bfhip_hard_voxelize(const float *points, int n, int f, float *voxels,
                                     int32_t *coors, int32_t *num_points_per_voxel,
                                     const float *voxel_size_host, const float *coors_range_host,
                                     int max_points, int max_voxels, void *workspace,
                                     size_t workspace_bytes, int32_t *voxel_num_dev,
                                     void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(n >= 0 && f >= 3, "hard_voxelize: bad sizes n=%d f=%d", n, f);
  BFHIP_REQUIRE(max_points > 0, "hard_voxelize: max_points must be > 0 (use dynamic_voxelize for -1)");
  BFHIP_REQUIRE(max_voxels >= -1, "hard_voxelize: bad max_voxels=%d", max_voxels);
  BFHIP_REQUIRE(voxel_num_dev != nullptr, "hard_voxelize: voxel_num_dev is null");
  VoxParams P;
  BFHIP_REQUIRE(make_params(voxel_size_host, coors_range_host, P) == 0,
                "hard_voxelize: bad voxel_size/coors_range (grid must be positive and < 2^31 cells)");
  if (n == 0 || max_voxels == 0) {
    if (hipMemsetAsync(voxel_num_dev, 0, sizeof(int), stream) != hipSuccess) return check_launch("hard_voxelize memset");
    return BFHIP_OK;
  }
  BFHIP_REQUIRE(points && voxels && coors && num_points_per_voxel, "hard_voxelize: null pointer");
  if (workspace_bytes < bfhip_hard_voxelize_workspace_bytes(n, max_points, max_voxels) || !workspace) {
    set_error("hard_voxelize: workspace too small (%zu < %zu)", workspace_bytes,
              bfhip_hard_voxelize_workspace_bytes(n, max_points, max_voxels));
    return BFHIP_E_WORKSPACE;
  }
  Workspace ws(workspace, workspace_bytes);
  unsigned cap = table_cap(n);
  int nb = ceil_div(n, kScanBlock);
  long long nlist = list_rows(n, max_voxels) * (long long)max_points;
  int *keys = ws.take<int>(cap), *first = ws.take<int>(cap), *count = ws.take<int>(cap),
      *vid = ws.take<int>(cap);
  int *slot_of = ws.take<int>(n);
  int *block_sums = ws.take<int>(nb + 1);
  int *lists = ws.take<int>(nlist);
  if (!ws.ok()) { set_error("hard_voxelize: workspace carve failed"); return BFHIP_E_WORKSPACE; }

  ProfScope ps;
  prof_begin(BFHIP_OP_HARD_VOXELIZE, stream, &ps);
  int init_blocks = ceil_div(cap > nlist ? cap : nlist, 256);
  if (init_blocks > 2048) init_blocks = 2048;
  hipLaunchKernelGGL(vox_init_kernel, dim3(init_blocks), dim3(256), 0, stream, keys, first, count,
                     (int)cap, lists, nlist, voxel_num_dev);
  hipLaunchKernelGGL(vox_hash_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, points, n, f,
                     P, keys, first, count, cap - 1, slot_of);
  hipLaunchKernelGGL(vox_flag_count_kernel, dim3(nb), dim3(kScanBlock), 0, stream, slot_of, first,
                     n, block_sums);
  hipLaunchKernelGGL(vox_scan_blocks_kernel, dim3(1), dim3(kScanBlock), 0, stream, block_sums, nb,
                     max_voxels, voxel_num_dev);
  hipLaunchKernelGGL(vox_assign_kernel, dim3(nb), dim3(kScanBlock), 0, stream, slot_of, first, keys,
                     count, n, block_sums, max_points, max_voxels, P, vid, coors,
                     num_points_per_voxel);
  hipLaunchKernelGGL(vox_rank_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, slot_of, vid,
                     count, n, max_points, lists);
  long long total = nlist * f;
  hipLaunchKernelGGL(vox_scatter_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, points,
                     f, lists, voxel_num_dev, max_points, voxels, total);
  prof_end(&ps);
  return check_launch("hard_voxelize");
}
