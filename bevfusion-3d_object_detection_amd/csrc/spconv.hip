// spconv.hip -- sparse 3-D convolution (SubMConv3d / SparseConv3d, traveller59 layout) for gfx950.
//
// Replaces what the reference delegates to the third-party spconv 2.x wheel
// (call sites: BF/sparse_encoder.py:133-147, mmdet3d/models/layers/sparse_block.py:190-217;
//  shim: projects/SparseConvolution/sparse_functional.py:118-137 get_indice_pairs_implicit_gemm,
//  :287-314 implicit_gemm).  Layout conventions kept: indices i32[N,4] = (b, x, y, z);
//  pair_fwd i32[KV, N_out] with -1 holes (sparse_functional.py:57-61); weights (out, k0, k1, k2, in)
//  (mmdet3d/models/layers/spconv/overwrite_spconv/write_spconv2.py:50-51); kernel offset index
//  k = (i*k1 + j)*k2 + l.
//
// Rulebook:  SubM   -> open-addressing hash (linear cell -> row) + one lookup per (row, offset)
//            strided-> bitmap over the output grid + prefix popcount = output rows in ascending
//                      linear order (canonical, deterministic; spconv's order is hash-dependent)
// Compute :  output-stationary implicit GEMM on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32):
//            one wave owns R*16 output rows x all C_out columns, loops over the KV offsets,
//            skips offsets none of its rows has, gathers A rows straight from L2 with one 16-B
//            load per lane (K permuted consistently in the pre-packed weights, so no shuffle),
//            no atomics -> deterministic.  The same kernel computes dgrad (weights transposed,
//            pair_bwd).  wgrad: wave per (offset, row split, tile group), K = rows, partial slabs
//            reduced in a fixed order.
#include "common.h"

#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>

namespace bfhip {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvGeom {
  int B;
  int in0, in1, in2;     // input spatial shape (X, Y, Z)
  int out0, out1, out2;  // output spatial shape
  int k0, k1, k2, s0, s1, s2, p0, p1, p2, d0, d1, d2;
  int KV;
};

__device__ __forceinline__ unsigned hash32(unsigned k) {
  k ^= k >> 16; k *= 0x85ebca6bu; k ^= k >> 13; k *= 0xc2b2ae35u; k ^= k >> 16;
  return k;
}

__global__ __launch_bounds__(256) void fill_pair_kernel(int2 *__restrict__ t, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) t[i] = make_int2(-1, 0x7f7f7f7f);
}

__global__ __launch_bounds__(256) void fill_i32_kernel(int *__restrict__ p, long long n, int v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// ------------------------------------------------------------------------------- SubM rulebook
// hash table of (key, row) pairs interleaved in one int2 (one cache line per probe)
__global__ __launch_bounds__(256) void subm_insert_kernel(const int4 *__restrict__ indices, int N,
                                                          ConvGeom G, int2 *__restrict__ table,
                                                          unsigned mask) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int4 c = indices[n];
  if (c.x < 0) return;  // inactive row of a capacity-sized tensor (spconv.py, static capacity mode)
  int key = ((c.x * G.in0 + c.y) * G.in1 + c.z) * G.in2 + c.w;
  unsigned s = hash32((unsigned)key) & mask;
  for (unsigned probe = 0; probe <= mask; ++probe) {
    int old = atomicCAS(&table[s].x, -1, key);
    if (old == -1 || old == key) break;
    s = (s + 1) & mask;
  }
  atomicMin(&table[s].y, n);  // duplicate coordinates (malformed input): lowest row wins, deterministically
}

// blockIdx.y = kernel offset k in the LOWER half (k <= KV/2).  A submanifold rulebook is symmetric:
// pair[k][n] = j  <=>  pair[KV-1-k][j] = n, so each found neighbour fills two entries and only half of the
// offsets are probed; the centre offset maps every row to itself.  pair_fwd must be pre-filled with -1.
__global__ __launch_bounds__(256) void subm_pairs_kernel(const int4 *__restrict__ indices, int N,
                                                         ConvGeom G, const int2 *__restrict__ table,
                                                         unsigned mask, int *__restrict__ pair_fwd,
                                                         int *__restrict__ n_pairs, int symmetric) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  int found = -1;
  if (n < N && indices[n].x >= 0) {
    int4 c = indices[n];
    int l = k % G.k2, j = (k / G.k2) % G.k1, i = k / (G.k2 * G.k1);
    int dx_ = (i - G.k0 / 2) * G.d0, dy_ = (j - G.k1 / 2) * G.d1, dz_ = (l - G.k2 / 2) * G.d2;
    if (dx_ == 0 && dy_ == 0 && dz_ == 0) {
      found = n;
    } else {
      int x = c.y + dx_, y = c.z + dy_, z = c.w + dz_;
      if (x >= 0 && x < G.in0 && y >= 0 && y < G.in1 && z >= 0 && z < G.in2) {
        int key = ((c.x * G.in0 + x) * G.in1 + y) * G.in2 + z;
        unsigned s = hash32((unsigned)key) & mask;
        for (unsigned probe = 0; probe <= mask; ++probe) {
          int2 e = table[s];
          if (e.x == key) { found = e.y; break; }
          if (e.x == -1) break;
          s = (s + 1) & mask;
        }
      }
    }
    if (found >= 0) {
      pair_fwd[(size_t)k * N + n] = found;
      if (symmetric && k != G.KV - 1 - k) pair_fwd[(size_t)(G.KV - 1 - k) * N + found] = n;
    }
  }
  // pair statistics: spread over 64 counters (one hot word saturates at ~88 atomics/us); summed by the host wrapper
  unsigned long long bal = __ballot(found >= 0);
  if (n_pairs && (threadIdx.x & 63) == 0 && bal) {
    int add = __popcll(bal) * ((symmetric && k != G.KV - 1 - k) ? 2 : 1);
    atomicAdd(&n_pairs[(blockIdx.x * 4 + (threadIdx.x >> 6) + k) & 63], add);
  }
}

// Row-block form of the SubM rulebook: a workgroup is 64 rows x (k0*k1) offset columns, each thread probes the k2 offsets of
// its (i, j) column for its row.  Every pair_fwd entry is written exactly once (holes included: no pre-fill of the table),
// the k-th plane's 64 entries of a wave are one coalesced store, and the row's offset mask is assembled in LDS, so the row
// mask, the identity permutation and the (region, mask) sort key leave in the same launch (what row_mask_kernel did in a
// second pass over the table).  No atomics on global memory: the pair count is the popcount of the masks.
template <int K2>
__global__ __launch_bounds__(1024) void subm_pairs_rows_kernel(const int4 *__restrict__ indices, int N, ConvGeom G,
                                                               const int2 *__restrict__ table, unsigned tmask,
                                                               int *__restrict__ pair_fwd, unsigned *__restrict__ row_mask,
                                                               unsigned *__restrict__ iota, unsigned *__restrict__ keys,
                                                               int regions, int *__restrict__ n_pairs) {
  __shared__ unsigned smask[64];
  const int lane = threadIdx.x, ty = threadIdx.y;
  const int n = blockIdx.x * 64 + lane;
  if (ty == 0) smask[lane] = 0u;
  __syncthreads();
  const int k2 = K2 ? K2 : G.k2;
  const int i = ty / G.k1, j = ty - i * G.k1;
  int4 c = n < N ? indices[n] : make_int4(-1, 0, 0, 0);
  const int x = c.y + (i - G.k0 / 2) * G.d0, y = c.z + (j - G.k1 / 2) * G.d1;
  const bool xy_ok = c.x >= 0 && x >= 0 && x < G.in0 && y >= 0 && y < G.in1;
  const bool centre_col = x == c.y && y == c.z;
  const int base = ((c.x * G.in0 + x) * G.in1 + y) * G.in2;
  unsigned bits = 0u;
  for (int l = 0; l < k2; ++l) {
    const int k = ty * k2 + l;
    const int z = c.w + (l - k2 / 2) * G.d2;
    int found = -1;
    if (xy_ok && z >= 0 && z < G.in2) {
      if (centre_col && z == c.w) {
        found = n;
      } else {
        const int key = base + z;
        unsigned s = hash32((unsigned)key) & tmask;
        for (unsigned probe = 0; probe <= tmask; ++probe) {
          int2 e = table[s];
          if (e.x == key) { found = e.y; break; }
          if (e.x == -1) break;
          s = (s + 1) & tmask;
        }
      }
    }
    if (n < N) pair_fwd[(size_t)k * N + n] = found;
    bits |= (found >= 0 ? 1u : 0u) << (k & 31);
  }
  if (bits) atomicOr(&smask[lane], bits);
  __syncthreads();
  if (ty == 0) {
    const unsigned m = smask[lane];
    if (n < N) {
      if (row_mask) row_mask[n] = m;
      if (iota) iota[n] = (unsigned)n;
      if (keys) keys[n] = (regions > 1 ? (unsigned)(((long long)n * regions) / N) << G.KV : 0u) | m;
    }
    if (n_pairs) {
      int cnt = __popc(m);
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
      if (lane == 0 && cnt) atomicAdd(&n_pairs[blockIdx.x & 63], cnt);
    }
  }
}

// ---------------------------------------------------------------------------- strided rulebook
__device__ __forceinline__ bool out_coord(const ConvGeom &G, int4 c, int k, int &ox, int &oy, int &oz) {
  int l = k % G.k2, j = (k / G.k2) % G.k1, i = k / (G.k2 * G.k1);
  ox = c.y + G.p0 - i * G.d0;
  oy = c.z + G.p1 - j * G.d1;
  oz = c.w + G.p2 - l * G.d2;
  if (ox < 0 || oy < 0 || oz < 0) return false;
  if (ox % G.s0 || oy % G.s1 || oz % G.s2) return false;
  ox /= G.s0; oy /= G.s1; oz /= G.s2;
  return ox < G.out0 && oy < G.out1 && oz < G.out2;
}

// n_dev (optional): the true number of input rows lives on the device (N is then the host-side bound of the launch)
__global__ __launch_bounds__(256) void sparse_mark_kernel(const int4 *__restrict__ indices, int N,
                                                          const int *__restrict__ n_dev, ConvGeom G,
                                                          unsigned *__restrict__ bitmap) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (n >= N || (n_dev && n >= *n_dev)) return;
  int ox, oy, oz;
  int4 c = indices[n];
  if (c.x < 0 || !out_coord(G, c, k, ox, oy, oz)) return;
  long long cell = (((long long)c.x * G.out0 + ox) * G.out1 + oy) * G.out2 + oz;
  atomicOr(&bitmap[cell >> 5], 1u << (cell & 31));
}

constexpr int kScan = 1024;

__global__ __launch_bounds__(kScan) void words_count_kernel(const unsigned *__restrict__ bitmap,
                                                            long long nwords,
                                                            int *__restrict__ blk) {
  __shared__ int sm[kScan / 64];
  long long i = (long long)blockIdx.x * kScan + threadIdx.x;
  int v = i < nwords ? __popc(bitmap[i]) : 0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    int r = 0;
    for (int k = 0; k < kScan / 64; ++k) r += sm[k];
    blk[blockIdx.x] = r;
  }
}

__global__ __launch_bounds__(kScan) void blocks_scan_kernel(int *__restrict__ blk, int nb,
                                                            int *__restrict__ total) {
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

// exclusive prefix of the per-word popcounts
__global__ __launch_bounds__(kScan) void words_prefix_kernel(const unsigned *__restrict__ bitmap,
                                                             long long nwords,
                                                             const int *__restrict__ blk_offs,
                                                             int *__restrict__ word_prefix) {
  __shared__ int sm[kScan];
  long long i = (long long)blockIdx.x * kScan + threadIdx.x;
  int v = i < nwords ? __popc(bitmap[i]) : 0;
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < kScan; o <<= 1) {
    int t = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  if (i < nwords) word_prefix[i] = blk_offs[blockIdx.x] + sm[threadIdx.x] - v;
}

__global__ __launch_bounds__(256) void sparse_out_indices_kernel(const unsigned *__restrict__ bitmap,
                                                                 const int *__restrict__ word_prefix,
                                                                 long long nwords, ConvGeom G,
                                                                 int max_out,
                                                                 int4 *__restrict__ out_indices) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  unsigned bits = bitmap[w];
  int o = word_prefix[w];
  while (bits) {
    int bpos = __ffs(bits) - 1;
    bits &= bits - 1;
    long long cell = (w << 5) + bpos;
    int z = (int)(cell % G.out2); cell /= G.out2;
    int y = (int)(cell % G.out1); cell /= G.out1;
    int x = (int)(cell % G.out0); cell /= G.out0;
    if (o < max_out) out_indices[o] = make_int4((int)cell, x, y, z);
    ++o;
  }
}

__global__ __launch_bounds__(256) void sparse_pairs_kernel(const int4 *__restrict__ indices, int N,
                                                           ConvGeom G,
                                                           const unsigned *__restrict__ bitmap,
                                                           const int *__restrict__ word_prefix,
                                                           int ld_out, int *__restrict__ pair_fwd,
                                                           int *__restrict__ pair_bwd,
                                                           int *__restrict__ n_pairs) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  const long long t = (long long)k * N + n;
  bool ok = false;
  if (n < N) {
    int ox, oy, oz;
    int4 c = indices[n];
    ok = c.x >= 0 && out_coord(G, c, k, ox, oy, oz);
    int o = -1;
    if (ok) {
      long long cell = (((long long)c.x * G.out0 + ox) * G.out1 + oy) * G.out2 + oz;
      unsigned wbits = bitmap[cell >> 5];
      o = word_prefix[cell >> 5] + __popc(wbits & ((1u << (cell & 31)) - 1u));
      // ld_out may be a capacity (static capacity mode): an output row beyond it does not exist in any buffer of this
      // layer, so the pair is dropped on BOTH sides (an unclamped pair_bwd entry would make the data gradient gather
      // grad_out rows past the buffer)
      // atomicMax over the -1 pre-fill instead of a plain store: with duplicate input coordinates (malformed input; the
      // reference's own encoder test feeds them) several rows claim one slot, and the HIGHEST row wins deterministically
      if (o < ld_out) atomicMax(&pair_fwd[(size_t)k * ld_out + o], n);
      else { o = -1; ok = false; }
    }
    pair_bwd[t] = o;
  }
  unsigned long long bal = __ballot(ok);
  if (n_pairs && (threadIdx.x & 63) == 0 && bal) atomicAdd(&n_pairs[(blockIdx.x * 4 + (threadIdx.x >> 6) + k) & 63], __popcll(bal));
}

// Row-block form of sparse_pairs_kernel (same tables): workgroup = 64 input rows x (k0*k1) offset columns, a thread walks the
// k2 offsets of its column.  pair_bwd planes are written coalesced, the input rows' offset masks (the backward table's row
// masks) are assembled in LDS and leave with the identity permutation and the sort key; no global atomics.
template <int K2>
__global__ __launch_bounds__(1024) void sparse_pairs_rows_kernel(const int4 *__restrict__ indices, int N, ConvGeom G,
                                                                 const unsigned *__restrict__ bitmap,
                                                                 const int *__restrict__ word_prefix, int ld_out,
                                                                 int *__restrict__ pair_fwd, int *__restrict__ pair_bwd,
                                                                 unsigned *__restrict__ row_mask, unsigned *__restrict__ iota,
                                                                 unsigned *__restrict__ keys, int regions,
                                                                 int *__restrict__ n_pairs) {
  __shared__ unsigned smask[64];
  const int lane = threadIdx.x, ty = threadIdx.y;
  const int n = blockIdx.x * 64 + lane;
  if (ty == 0) smask[lane] = 0u;
  __syncthreads();
  const int k2 = K2 ? K2 : G.k2;
  const int i = ty / G.k1, j = ty - i * G.k1;
  int4 c = n < N ? indices[n] : make_int4(-1, 0, 0, 0);
  int ox = c.y + G.p0 - i * G.d0, oy = c.z + G.p1 - j * G.d1;
  bool xy_ok = c.x >= 0 && ox >= 0 && oy >= 0 && (ox % G.s0) == 0 && (oy % G.s1) == 0;
  ox /= G.s0; oy /= G.s1;
  xy_ok = xy_ok && ox < G.out0 && oy < G.out1;
  const long long base = (((long long)c.x * G.out0 + ox) * G.out1 + oy) * G.out2;
  unsigned bits = 0u;
  for (int l = 0; l < k2; ++l) {
    const int k = ty * k2 + l;
    int oz = c.w + G.p2 - l * G.d2;
    int o = -1;
    if (xy_ok && oz >= 0 && (oz % G.s2) == 0 && oz / G.s2 < G.out2) {
      const long long cell = base + oz / G.s2;
      const unsigned wbits = bitmap[cell >> 5];
      o = word_prefix[cell >> 5] + __popc(wbits & ((1u << (cell & 31)) - 1u));
      if (o < ld_out) {  // beyond a static capacity: the pair is dropped on both sides (see sparse_pairs_kernel)
        atomicMax(&pair_fwd[(size_t)k * ld_out + o], n);  // duplicates: highest row wins (see sparse_pairs_kernel)
        bits |= 1u << (k & 31);
      } else {
        o = -1;
      }
    }
    if (n < N) pair_bwd[(size_t)k * N + n] = o;
  }
  if (bits) atomicOr(&smask[lane], bits);
  __syncthreads();
  if (ty == 0) {
    const unsigned m = smask[lane];
    if (n < N) {
      if (row_mask) row_mask[n] = m;
      if (iota) iota[n] = (unsigned)n;
      if (keys) keys[n] = (regions > 1 ? (unsigned)(((long long)n * regions) / N) << G.KV : 0u) | m;
    }
    if (n_pairs) {
      int cnt = __popc(m);
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
      if (lane == 0 && cnt) atomicAdd(&n_pairs[blockIdx.x & 63], cnt);
    }
  }
}

// -------------------------------------------------------------------------------- row masks
// mask[n] bit k = pair[k][n] >= 0.  Rows are then sorted by mask so that the 16 rows of an MFMA tile
// use (nearly) the same kernel offsets and whole offsets can be skipped per tile (the idea of spconv's
// mask_argsort, projects/SparseConvolution/sparse_functional.py:139-162).
__global__ __launch_bounds__(256) void row_mask_kernel(const int *__restrict__ pairs, int ld, int KV,
                                                       int n_rows, unsigned *__restrict__ mask,
                                                       unsigned *__restrict__ iota, unsigned *__restrict__ keys,
                                                       int regions) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_rows) return;
  unsigned m = 0u;
  for (int k = 0; k < KV; ++k) m |= (pairs[(size_t)k * ld + n] >= 0 ? 1u : 0u) << k;
  mask[n] = m;
  iota[n] = (unsigned)n;
  // sort key: region of the row (rows are in voxel order, so a region is a slab of space) above the mask
  if (keys) keys[n] = (unsigned)(((long long)n * regions) / n_rows) << KV | m;
}

// -------------------------------------------------------------------------------- weight packing
// Wp[((k*CC + cc)*NT + nt)*64 + lane][j] = M_k[cc*16 + 4*(lane>>4) + j][nt*16 + (lane&15)]
//   forward : M_k[ci][co] = W[co][k][ci]                      (K = C_in,  N = C_out)
//   dgrad   : M_k[co][ci] = W[co][flip ? KV-1-k : k][ci]      (K = C_out, N = C_in)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ W, int Cout,
                                                           int KV, int Cin, int transpose, int flip,
                                                           int CC, int NT, float *__restrict__ Wp) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)KV * CC * NT * 64 * 4;
  if (t >= total) return;
  int j = (int)(t & 3);
  int lane = (int)((t >> 2) & 63);
  long long r = t >> 8;
  int nt = (int)(r % NT); r /= NT;
  int cc = (int)(r % CC);
  int k = (int)(r / CC);
  int kk = cc * 16 + 4 * (lane >> 4) + j;  // K index
  int nn = nt * 16 + (lane & 15);          // N index
  float v = 0.f;
  if (!transpose) {
    if (kk < Cin && nn < Cout) v = W[((size_t)nn * KV + k) * Cin + kk];
  } else {
    int ks = flip ? KV - 1 - k : k;
    if (kk < Cout && nn < Cin) v = W[((size_t)kk * KV + ks) * Cin + nn];
  }
  Wp[t] = v;
}

// -------------------------------------------------------------------------------- forward / dgrad
// Workgroup version: 4 waves share every weight tile through LDS (double buffered, one barrier per
// (offset, 16-channel chunk) step); the workgroup skips offsets none of its 4*R*16 rows uses.
// Weight traffic from L2 drops 4x versus one weight fetch per wave; A rows are still gathered per wave.
template <int NT, int R>
__global__ __launch_bounds__(256) void spconv_gemm_lds_kernel(const float *__restrict__ in, int Kdim,
                                                              const f32x4 *__restrict__ Wp,
                                                              const int *__restrict__ pairs, int ld,
                                                              int KV, int n_rows, int Ndim,
                                                              const int *__restrict__ perm,
                                                              const unsigned *__restrict__ row_mask,
                                                              float *__restrict__ out) {
  __shared__ f32x4 sB[2][NT * 64];
  __shared__ unsigned s_mask;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  // XCD-chunked block order (common.h): each XCD walks one contiguous eighth of the (region-major sorted) rows
  const long long row_base = (xcd_chunked_block(blockIdx.x, gridDim.x) * 4 + wv) * (R * 16);
  const int lr = lane & 15, lq = lane >> 4;
  const int CC = Kdim >> 4;
  if (tid == 0) s_mask = 0u;
  __syncthreads();
  // rows of this wave: position in mask-sorted order -> actual row
  int my_row[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    long long sp = row_base + r * 16 + lr;
    my_row[r] = sp < n_rows ? (perm ? perm[sp] : (int)sp) : -1;
  }
  // which offsets does this wave / workgroup need
  unsigned wmask = 0u;
  if (row_mask) {
    unsigned mm = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) mm |= my_row[r] >= 0 ? row_mask[my_row[r]] : 0u;
    for (int o = 32; o > 0; o >>= 1) mm |= __shfl_xor(mm, o);
    wmask = mm;
  } else {
    for (int k = 0; k < KV; ++k) {
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) any |= (my_row[r] >= 0) && (pairs[(size_t)k * ld + my_row[r]] >= 0);
      if (__any(any)) wmask |= 1u << k;
    }
  }
  if (lane == 0 && wmask) atomicOr(&s_mask, wmask);
  __syncthreads();
  unsigned gmask = s_mask;

  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int LOADS = (NT * 64 + 255) / 256;  // float4 per thread per weight tile
  f32x4 pre[LOADS];
  auto prefetch = [&](int k, int cc) {
    const f32x4 *wp = Wp + ((size_t)(k * CC + cc) * NT) * 64;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      int e = tid + i * 256;
      if (e < NT * 64) pre[i] = wp[e];
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      int e = tid + i * 256;
      if (e < NT * 64) sB[buf][e] = pre[i];
    }
  };
  auto next_set = [&](int kk) {  // next offset the workgroup needs after kk (KV if none)
    unsigned rest = gmask & ~((2u << kk) - 1u);
    return rest ? __ffs(rest) - 1 : KV;
  };
  auto load_idx = [&](int kk, int *dst) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      dst[r] = (kk < KV && my_row[r] >= 0 && ((wmask >> kk) & 1u)) ? pairs[(size_t)kk * ld + my_row[r]] : -1;
  };
  auto gather = [&](const int *ix, int c, f32x4 *dst) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      dst[r] = ix[r] >= 0 ? *(const f32x4 *)(in + (size_t)ix[r] * Kdim + c * 16 + lq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // software pipeline: weight tile i+1 (global->regs->LDS), A rows of step i+1 and the pair indices of
  // the NEXT offset are all in flight while the MFMAs of step i run
  int k = gmask ? __ffs(gmask) - 1 : KV;
  int cc = 0, buf = 0;
  int idx[R], idx_nxt[R];
  f32x4 a[R], a_nxt[R];
  if (k < KV) {
    prefetch(k, 0);
    load_idx(k, idx);
    load_idx(next_set(k), idx_nxt);
    gather(idx, 0, a);
  }
  while (k < KV) {
    stash(buf);
    __syncthreads();
    int nk = k, ncc = cc + 1;
    if (ncc == CC) { ncc = 0; nk = next_set(k); }
    if (nk < KV) {
      prefetch(nk, ncc);
      gather(ncc == 0 ? idx_nxt : idx, ncc, a_nxt);
    }
    if ((wmask >> k) & 1u) {
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = sB[buf][nt * 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][j], b[nt][j], acc[r][nt], 0, 0, 0);
    }
    if (ncc == 0 && nk < KV) {  // moving on to offset nk: rotate the index registers, fetch the one after
#pragma unroll
      for (int r = 0; r < R; ++r) idx[r] = idx_nxt[r];
      load_idx(next_set(nk), idx_nxt);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = a_nxt[r];
    k = nk;
    cc = ncc;
    buf ^= 1;
  }
  // C/D layout: col = lane&15, row = (lane>>4)*4 + i ; the actual output row lives in lane (lq*4+i) of my_row
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = __shfl(my_row[r], lq * 4 + i);
      if (row >= 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          int col = nt * 16 + lr;
          if (col < Ndim) out[(size_t)row * Ndim + col] = acc[r][nt][i];
        }
      }
    }
}

// ---- bf16-input variant (fp32 features in HBM, converted on load; fp32 accumulate): for the bf16 configs.
// v_mfma_f32_16x16x32_bf16: lane l supplies A[row = l&15][k = 8*(l>>4) + j], B[k = 8*(l>>4) + j][col = l&15], j < 8,
// so one step covers 32 input channels (two float4 gathers per lane and row tile) with ONE MFMA per tile.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Wp16[((k*CC32 + c32)*NT + nt)*64 + lane][j] = bf16(M_k[c32*32 + 8*(lane>>4) + j][nt*16 + (lane&15)])
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float *__restrict__ W, int Cout,
                                                                int KV, int Cin, int transpose, int flip,
                                                                int CC32, int NT, __bf16 *__restrict__ Wp) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)KV * CC32 * NT * 64 * 8;
  if (t >= total) return;
  int j = (int)(t & 7);
  int lane = (int)((t >> 3) & 63);
  long long r = t >> 9;
  int nt = (int)(r % NT); r /= NT;
  int c32 = (int)(r % CC32);
  int k = (int)(r / CC32);
  int kk = c32 * 32 + 8 * (lane >> 4) + j;
  int nn = nt * 16 + (lane & 15);
  float v = 0.f;
  if (!transpose) {
    if (kk < Cin && nn < Cout) v = W[((size_t)nn * KV + k) * Cin + kk];
  } else {
    int ks = flip ? KV - 1 - k : k;
    if (kk < Cout && nn < Cin) v = W[((size_t)kk * KV + ks) * Cin + nn];
  }
  Wp[t] = (__bf16)v;
}

template <int NT, int R, bool IO16>
__global__ __launch_bounds__(256) void spconv_gemm_bf16_kernel(const void *__restrict__ in_, int Kdim,
                                                               const bf16x8 *__restrict__ Wp,
                                                               const int *__restrict__ pairs, int ld,
                                                               int KV, int n_rows, int Ndim,
                                                               const int *__restrict__ perm,
                                                               const unsigned *__restrict__ row_mask,
                                                               void *__restrict__ out_) {
  // IO16: features stored in bf16 (gathered rows are the MFMA operand as they are: one 16-byte load per lane and step,
  // half the gather traffic of fp32 storage) and the output rounded to bf16 once; otherwise fp32 in / fp32 out.
  const float *in = (const float *)in_;
  const __bf16 *in16 = (const __bf16 *)in_;
  float *out = (float *)out_;
  __bf16 *out16 = (__bf16 *)out_;
  __shared__ bf16x8 sB[2][NT * 64];
  __shared__ unsigned s_mask;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  // XCD-chunked block order (common.h): each XCD walks one contiguous eighth of the (region-major sorted) rows
  const long long row_base = (xcd_chunked_block(blockIdx.x, gridDim.x) * 4 + wv) * (R * 16);
  const int lr = lane & 15, lq = lane >> 4;
  const int CC = (Kdim + 31) >> 5;
  if (tid == 0) s_mask = 0u;
  __syncthreads();
  int my_row[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    long long sp = row_base + r * 16 + lr;
    my_row[r] = sp < n_rows ? (perm ? perm[sp] : (int)sp) : -1;
  }
  unsigned wmask = 0u;
  if (row_mask) {
    unsigned mm = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) mm |= my_row[r] >= 0 ? row_mask[my_row[r]] : 0u;
    for (int o = 32; o > 0; o >>= 1) mm |= __shfl_xor(mm, o);
    wmask = mm;
  } else {
    for (int k = 0; k < KV; ++k) {
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) any |= (my_row[r] >= 0) && (pairs[(size_t)k * ld + my_row[r]] >= 0);
      if (__any(any)) wmask |= 1u << k;
    }
  }
  if (lane == 0 && wmask) atomicOr(&s_mask, wmask);
  __syncthreads();
  unsigned gmask = s_mask;

  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int LOADS = (NT * 64 + 255) / 256;
  bf16x8 pre[LOADS];
  auto prefetch = [&](int k, int cc) {
    const bf16x8 *wp = Wp + ((size_t)(k * CC + cc) * NT) * 64;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      int e = tid + i * 256;
      if (e < NT * 64) pre[i] = wp[e];
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      int e = tid + i * 256;
      if (e < NT * 64) sB[buf][e] = pre[i];
    }
  };
  auto next_set = [&](int kk) {
    unsigned rest = gmask & ~((2u << kk) - 1u);
    return rest ? __ffs(rest) - 1 : KV;
  };
  auto load_idx = [&](int kk, int *dst) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      dst[r] = (kk < KV && my_row[r] >= 0 && ((wmask >> kk) & 1u)) ? pairs[(size_t)kk * ld + my_row[r]] : -1;
  };
  auto gather = [&](const int *ix, int c, f32x4 *lo, f32x4 *hi) {
    const int ch = c * 32 + lq * 8;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (ix[r] >= 0 && ch < Kdim) {
        if (IO16) {
          lo[r] = *(const f32x4 *)(in16 + (size_t)ix[r] * Kdim + ch);  // 8 bf16 carried in one 16-byte register group
          hi[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
          const float *src = in + (size_t)ix[r] * Kdim + ch;
          lo[r] = *(const f32x4 *)src;
          hi[r] = *(const f32x4 *)(src + 4);
        }
      } else {
        lo[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        hi[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  int k = gmask ? __ffs(gmask) - 1 : KV;
  int cc = 0, buf = 0;
  int idx[R], idx_nxt[R];
  f32x4 alo[R], ahi[R], nlo[R], nhi[R];
  if (k < KV) {
    prefetch(k, 0);
    load_idx(k, idx);
    load_idx(next_set(k), idx_nxt);
    gather(idx, 0, alo, ahi);
  }
  while (k < KV) {
    stash(buf);
    __syncthreads();
    int nk = k, ncc = cc + 1;
    if (ncc == CC) { ncc = 0; nk = next_set(k); }
    if (nk < KV) {
      prefetch(nk, ncc);
      gather(ncc == 0 ? idx_nxt : idx, ncc, nlo, nhi);
    }
    if ((wmask >> k) & 1u) {
      bf16x8 a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (IO16) {
          a[r] = __builtin_bit_cast(bf16x8, alo[r]);
        } else {
          a[r][0] = (__bf16)alo[r][0]; a[r][1] = (__bf16)alo[r][1]; a[r][2] = (__bf16)alo[r][2]; a[r][3] = (__bf16)alo[r][3];
          a[r][4] = (__bf16)ahi[r][0]; a[r][5] = (__bf16)ahi[r][1]; a[r][6] = (__bf16)ahi[r][2]; a[r][7] = (__bf16)ahi[r][3];
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 b = sB[buf][nt * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r)
          acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b, acc[r][nt], 0, 0, 0);
      }
    }
    if (ncc == 0 && nk < KV) {
#pragma unroll
      for (int r = 0; r < R; ++r) idx[r] = idx_nxt[r];
      load_idx(next_set(nk), idx_nxt);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { alo[r] = nlo[r]; ahi[r] = nhi[r]; }
    k = nk;
    cc = ncc;
    buf ^= 1;
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = __shfl(my_row[r], lq * 4 + i);
      if (row >= 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          int col = nt * 16 + lr;
          if (col < Ndim) {
            if (IO16) out16[(size_t)row * Ndim + col] = (__bf16)acc[r][nt][i];
            else out[(size_t)row * Ndim + col] = acc[r][nt][i];
          }
        }
      }
    }
}

// generic fallback (any channel counts): one thread per (row, out channel); fp32 FMA-free sums
__global__ __launch_bounds__(256) void spconv_scalar_kernel(const float *__restrict__ in, int Kdim,
                                                            const float *__restrict__ W, int Cout_w,
                                                            int Cin_w, int transpose, int flip,
                                                            const int *__restrict__ pairs, int ld,
                                                            int KV, int n_rows, int Ndim,
                                                            float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long row = t / Ndim;
  int col = (int)(t - row * Ndim);
  if (row >= n_rows) return;
  float acc = 0.f;
  for (int k = 0; k < KV; ++k) {
    int idx = pairs[(size_t)k * ld + row];
    if (idx < 0) continue;
    const float *x = in + (size_t)idx * Kdim;
    if (!transpose) {
      const float *w = W + ((size_t)col * KV + k) * Cin_w;  // W[co=col][k][ci]
      for (int c = 0; c < Kdim; ++c) acc = fmaf(x[c], w[c], acc);
    } else {
      int ks = flip ? KV - 1 - k : k;
      for (int c = 0; c < Kdim; ++c) acc = fmaf(x[c], W[((size_t)c * KV + ks) * Cin_w + col], acc);  // W[co=c][ks][ci=col]
    }
  }
  out[(size_t)row * Ndim + col] = acc;
}

// -------------------------------------------------------------------------------- wgrad
// dW[co][k][ci] = sum_n dout[n][co] * in[pair[k][n]][ci].
// wave = (k, split s, tile group): TI ci-tiles x TJ co-tiles, K = rows of the split, 4 rows per MFMA.
// A[i=ci][kk=row], B[kk=row][j=co]: lane l reads in[p(row = n0 + (l>>4))][ci0 + (l&15)], dout[row][co0 + (l&15)].
template <int TI, int TJ>
__global__ __launch_bounds__(256) void spconv_wgrad_kernel(const float *__restrict__ in, int Cin,
                                                           const float *__restrict__ dout, int Cout,
                                                           const int *__restrict__ pairs, int ld,
                                                           int KV, int n_rows, int S, int GI, int GJ,
                                                           float *__restrict__ partial) {
  const int lane = threadIdx.x & 63;
  long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long total = (long long)KV * S * GI * GJ;
  if (wave >= total) return;
  const int gj = (int)(wave % GJ); wave /= GJ;
  const int gi = (int)(wave % GI); wave /= GI;
  const int s = (int)(wave % S);
  const int k = (int)(wave / S);
  const int lr = lane & 15, lq = lane >> 4;
  const int rows_per = (((n_rows + S - 1) / S) + 3) & ~3;
  const int r0 = s * rows_per, r1 = min(n_rows, r0 + rows_per);
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ci0 = gi * TI * 16, co0 = gj * TJ * 16;
  constexpr int U = 4;  // K-steps in flight
  for (int n0 = r0; n0 < r1; n0 += 4 * U) {
    int p[U];
    bool any = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int row = n0 + 4 * u + lq;
      p[u] = row < r1 ? pairs[(size_t)k * ld + row] : -1;
      any |= p[u] >= 0;
    }
    if (!__any(any)) continue;
    float av[U][TI], bv[U][TJ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int row = n0 + 4 * u + lq;
#pragma unroll
      for (int a = 0; a < TI; ++a) {
        int ci = ci0 + a * 16 + lr;
        av[u][a] = (p[u] >= 0 && ci < Cin) ? in[(size_t)p[u] * Cin + ci] : 0.f;
      }
#pragma unroll
      for (int b = 0; b < TJ; ++b) {
        int co = co0 + b * 16 + lr;
        bv[u][b] = (p[u] >= 0 && co < Cout) ? dout[(size_t)row * Cout + co] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
  }
  // partial[s][k][ci][co]   (D layout: col = lane&15 -> co, row = (lane>>4)*4 + i -> ci)
  float *dst = partial + ((size_t)s * KV + k) * Cin * Cout;
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ci = ci0 + a * 16 + lq * 4 + i;
      if (ci < Cin) {
#pragma unroll
        for (int b = 0; b < TJ; ++b) {
          int co = co0 + b * 16 + lr;
          if (co < Cout) dst[(size_t)ci * Cout + co] = acc[a][b][i];
        }
      }
    }
}

// 4 consecutive channels of a feature row; features are f32 or (IO16) bf16.  Load and conversion are separate steps so
// that a group of gathers can be issued back to back and converted afterwards: with the conversion next to a predicated
// load the compiler waits for each gather before issuing the next (8 serialised L2 round trips per K group, measured
// 1.8x slower).  Invalid lanes load a valid address (clamped row / channel 0) and are zeroed by a select.
// The MFMA wgrad kernel.  One wave accumulates a 64 x 64 block of dW[k] over a range of rows.  Per K-step
// of 4 (input row, output row) pairs every lane loads ONE 4-channel piece of `in` (pair lq, channels 4 la..4 la+3) and ONE
// of `dout`; the 16 MFMAs (c, d) use element c of the first and element d of the second:
//   D_cd[a][a'] += in[row][4a+c] * dout[row][4a'+d],  a 64 x 64 block with channel index 4 * lane_id + component,
// i.e. 2 coalesced loads per 16 MFMAs.  Holes of the rulebook are compacted away first: the pair entries of the wave's
// whole row range (up to 1024 rows: 16 loads in flight together) go through ballot + prefix popcount into a wave-private
// LDS list, which one software pipeline then consumes -- the gathers of group g+1 are issued before the 64 MFMAs of
// group g.  (An earlier version walked 64-row chunks and paid three dependent round trips per chunk.)
// bf16 rows are read as 8-byte (4-channel) pieces.
template <bool IO16> struct RawRow { typedef uint4 T; };
template <> struct RawRow<true> { typedef uint2 T; };
template <bool IO16>
__device__ __forceinline__ typename RawRow<IO16>::T ldrow4_raw(const void *base, size_t off);
template <>
__device__ __forceinline__ uint4 ldrow4_raw<false>(const void *base, size_t off) { return *(const uint4 *)((const float *)base + off); }
template <>
__device__ __forceinline__ uint2 ldrow4_raw<true>(const void *base, size_t off) { return *(const uint2 *)((const unsigned short *)base + off); }
__device__ __forceinline__ f32x4 ldrow4_cvt(uint4 q, bool ok) {
  f32x4 r = (f32x4){__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
  return ok ? r : (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 ldrow4_cvt(uint2 q, bool ok) {
  f32x4 r = (f32x4){__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                    __uint_as_float(q.y & 0xffff0000u)};
  return ok ? r : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// Work decomposition ("stream-K" over rows, weighted by offset): the T = KV * GI * GJ output tiles (64 x 64 each) times
// Ut = ceil(n_rows / 64) row units form one line, cut into equal runs, one run per workgroup, with exactly as many
// workgroups as the chip holds at once.  (The earlier grid -- one workgroup per (offset, fixed row split, tile) -- left a
// long under-occupied tail: the per-wave timeline (tools/wgrad_trace.py) showed the MFMA saturated while all waves were in
// their loops, but only ~1.2 waves per SIMD alive on average and wave lifetimes comparable to the whole kernel.)
// Offsets differ in how many of their rows are paired (the centre of a submanifold rulebook pairs every row, a corner
// offset ~40 %: 2.4x between the extremes on the nuScenes-like clouds), so a unit of offset k counts w_k = 1..64 virtual
// units, proportional to the offset's pair count (wgrad_offset_counts_kernel, kCountSlices partial counts per offset); a run is q virtual units.  A run covers the
// end of one tile and the start of the next ones.
// XCD-aware order: wgrad walks the feature matrices once per offset (27 passes); in offset-major order every pass misses the
// 4 MiB L2 of the XCD (PMC: 125 MB fetched per 32-channel launch for 31 MB of operands).  So the rows are cut into NR = 8
// regions and the line is ordered (region, offset, tile, unit); workgroups are dispatched to XCDs round-robin (h % 8), and
// the run of workgroup h is run (h % 8) * P/8 + h / 8: XCD x works through region x for all offsets and its L2 only ever
// sees an eighth of the rows (plus the halo of neighbouring rows).  Line tile tl = region * T + t receives the partial sum
// of run b in slab b + tl of partial[][64][64] (b + tl is unique along the staircase of (run, line tile) incidences,
// <= P + NR * T slabs in all).
struct SkGeom { int Ur, M, NR; };  // Ur = units per region, M = GI * GJ * Ur: units per (region, offset), NR regions
constexpr int kSkRegions = 8;

constexpr int kCountSlices = 32;  // row slices per offset in the count pass (one workgroup each)
__global__ __launch_bounds__(256) void wgrad_offset_counts_kernel(const int *__restrict__ pairs, int ld, int n_rows,
                                                                  int *__restrict__ counts /* [KV][kCountSlices] */) {
  const int k = blockIdx.x / kCountSlices, sl = blockIdx.x % kCountSlices;
  const int per = (((n_rows + kCountSlices - 1) / kCountSlices) + 3) & ~3;
  const int i0 = sl * per, i1 = min(n_rows, i0 + per);
  const int *row = pairs + (size_t)k * ld;
  const bool v4 = (ld & 3) == 0 && ((uintptr_t)pairs & 15) == 0;  // 16-byte loads when the table's rows are aligned
  int c = 0;
  for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
    if (v4 && i + 3 < i1) {
      const int4 v = *(const int4 *)(row + i);
      c += (v.x >= 0) + (v.y >= 0) + (v.z >= 0) + (v.w >= 0);
    } else {
      for (int j = i; j < min(i + 4, i1); ++j) c += row[j] >= 0;
    }
  }
  __shared__ int part[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// weights and their prefix in LDS (s_w[64], s_wpre[65]); executed by the first wave, the caller synchronises
__device__ __forceinline__ void sk_make_plan(const int *__restrict__ counts, int KV, int *s_w, int *s_wpre) {
  const int lane = threadIdx.x;
  int c = 0;
  if (counts && lane < KV) {
#pragma unroll
    for (int j = 0; j < kCountSlices; j += 4) {
      const int4 v = *(const int4 *)(counts + lane * kCountSlices + j);
      c += v.x + v.y + v.z + v.w;
    }
  }
  int mx = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
  int w = lane < KV ? ((counts && mx > 0) ? max(1, (int)(((long long)c * 64 + mx - 1) / mx)) : 1) : 0;
  int incl = w;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  s_w[lane] = w;
  s_wpre[lane + 1] = incl;
  if (lane == 0) s_wpre[0] = 0;
}

// R = 1: 64 x 64 tiles of any Cin x Cout.  R = 2 | 4: the narrow stages (Cin == Cout == C = 64 / R), where a 64 x 64 MFMA
// block would be 1/4 or 1/16 useful: the 16 lane columns hold R row groups of C/4 channel vectors -- lane la = (r, a) loads
// channels 4a..4a+3 of pair (lq, r) -- so one K-step consumes 4R pairs and the block's R diagonal C x C sub-blocks each
// accumulate their own pairs (the off-diagonal ones mix different pairs and are dropped); the diagonal blocks are summed
// with cross-lane shuffles in the epilogue.  R times fewer MFMAs and loads per pair.
template <int R, bool IO16>
__global__ __launch_bounds__(256, 3) void spconv_wgrad64p_kernel(const void *__restrict__ in, int Cin,
                                                              const void *__restrict__ dout, int Cout,
                                                              const int *__restrict__ pairs, int ld,
                                                              int KV, int n_rows, SkGeom g, int GI, int GJ,
                                                              const int *__restrict__ perm,
                                                              const int *__restrict__ counts,
                                                              float *__restrict__ partial) {
  typedef typename RawRow<IO16>::T Raw;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int la = lane & 15, lq = lane >> 4;
  constexpr int CH = 1024, U = 4;
  __shared__ int2 s_list[4][CH];  // 32 KB; reused as two 64 x 64 staging tiles by the 4-wave sum
  __shared__ int s_w[64], s_wpre[65];
  int2 *list = s_list[wv];
  if (wv == 0) sk_make_plan(counts, KV, s_w, s_wpre);
  __syncthreads();
  const long long VR = (long long)g.M * s_wpre[KV];                // virtual units of one region
  const long long V = VR * g.NR;                                   // ... in all
  const long long q = (V + gridDim.x - 1) / gridDim.x;            // per workgroup
  // run of this workgroup: XCD x (= blockIdx % 8) takes the x-th eighth of the line when the grid divides evenly
  const unsigned P8 = gridDim.x >> 3;
  const long long bl = (gridDim.x & 7) == 0 ? (long long)(blockIdx.x & 7) * P8 + (blockIdx.x >> 3) : (long long)blockIdx.x;
  const long long x0 = bl * q, x1 = min(x0 + q, V);
#ifdef BFHIP_WGRAD_TRACE
  unsigned long long tr_t0 = wall_clock64(), tr_t1 = 0, tr_t2 = 0;
  int tr_cnt = 0, tr_k = 0;
#endif
  for (long long x = x0; x < x1;) {  // block-uniform: the segments of this run, one per tile touched
    const int rg = (int)(x / VR);                                  // region
    const long long xg = x - (long long)rg * VR;                   // position inside the region
    const int target = (int)(xg / g.M);
    const int k = (int)__popcll(__ballot(lane < KV && s_wpre[lane + 1] <= target));
    const int wk = s_w[k];
    const long long L = (long long)g.Ur * wk;                      // virtual length of one tile of offset k in a region
    const long long xk = xg - (long long)g.M * s_wpre[k];          // position inside (region, offset k)
    const int ti = (int)(xk / L);
    const long long xr = xk - (long long)ti * L, xe = min(L, xr + (x1 - x));
    const int u0 = rg * g.Ur + (int)(xr / wk), u1 = rg * g.Ur + (xe == L ? g.Ur : (int)(xe / wk));
    const int nu = u1 - u0;  // may be 0: the slab is written all the same (the reduction reads it)
    x += xe - xr;
    const int t = (rg * KV + k) * GI * GJ + ti;                    // line tile
    const int gj = ti % GJ, gi = ti / GJ;
    // the segment's units split over the 4 waves (contiguous, the first nu % 4 waves take one more)
    const int ub = nu >> 2, ur = nu & 3;
    const int wu0 = u0 + wv * ub + min(wv, ur), wu1 = wu0 + ub + (wv < ur ? 1 : 0);
    const int r0 = min(wu0 * 64, n_rows), r1 = min(wu1 * 64, n_rows);  // the last region is padded past n_rows
    constexpr int AV = 16 / R;  // channel vectors per row group
    const int ci = R == 1 ? gi * 64 + la * 4 : (la % AV) * 4, co = R == 1 ? gj * 64 + la * 4 : (la % AV) * 4;
    const bool ci_ok = ci < Cin, co_ok = co < Cout;  // Cin, Cout multiples of 4 (checked by the host)
    const int cic = ci_ok ? ci : 0, coc = co_ok ? co : 0;  // lanes past the channel count read channel 0 and are zeroed
    const int jl = R == 1 ? lq : lq * R + la / AV;         // this lane's pair inside a K-step of 4R pairs
    f32x4 acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int d = 0; d < 4; ++d) acc[c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int *prow = pairs + (size_t)k * ld;
    const int rlast = r1 > 0 ? r1 - 1 : 0;
    for (int chunk = r0; chunk < r1; chunk += CH) {  // one pass for the usual run lengths (<= 1024 rows per wave)
      // pair entries of the whole chunk: 16 unconditional loads from clamped rows, back to back -> one round trip
      int pr[CH / 64], orow[CH / 64];
#pragma unroll
      for (int qq = 0; qq < CH / 64; ++qq) orow[qq] = min(chunk + qq * 64 + lane, rlast);
      if (perm) {
#pragma unroll
        for (int qq = 0; qq < CH / 64; ++qq) orow[qq] = perm[orow[qq]];
      }
#pragma unroll
      for (int qq = 0; qq < CH / 64; ++qq) pr[qq] = prow[orow[qq]];
      int cnt = 0;
#pragma unroll
      for (int qq = 0; qq < CH / 64; ++qq) {
        const bool ok1 = chunk + qq * 64 + lane < r1 && pr[qq] >= 0;
        const unsigned long long vmask = __ballot(ok1);
        if (ok1) list[cnt + __popcll(vmask & ((1ull << lane) - 1ull))] = make_int2(pr[qq], orow[qq]);
        cnt += __popcll(vmask);
      }
#ifdef BFHIP_WGRAD_TRACE
      tr_t1 = wall_clock64(); tr_cnt += cnt; tr_k = k;
#endif
      if (cnt == 0) continue;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // one continuous software pipeline over the compacted list: the gathers of group g+1 (U K-steps of 4R pairs) are in
      // flight during the 64 MFMAs of group g
      Raw ra[U], rb[U];
      bool ok[U];
      auto issue = [&](int tt) {  // list entries first (clamped index), then the 2 U gathers back to back
        int2 e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = tt + 4 * R * u + jl;
          ok[u] = j < cnt;
          e[u] = list[min(j, cnt - 1)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ra[u] = ldrow4_raw<IO16>(in, (size_t)e[u].x * Cin + cic);
          rb[u] = ldrow4_raw<IO16>(dout, (size_t)e[u].y * Cout + coc);
        }
      };
      issue(0);
      for (int t0 = 0; t0 < cnt; t0 += 4 * R * U) {
        f32x4 av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          av[u] = ldrow4_cvt(ra[u], ok[u] && ci_ok);
          bv[u] = ldrow4_cvt(rb[u], ok[u] && co_ok);
        }
        if (t0 + 4 * R * U < cnt) issue(t0 + 4 * R * U);  // wave-uniform
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (t0 + 4 * R * u < cnt) {  // wave-uniform: skip K-steps past the end of the compacted list
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int d = 0; d < 4; ++d)
                acc[c][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][c], bv[u][d], acc[c][d], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef BFHIP_WGRAD_TRACE
    tr_t2 = wall_clock64();
#endif
    __syncthreads();  // every wave is done with its list: the space becomes staging tiles
    float *red = (float *)&s_list[0][0];
    float *dst = partial + ((size_t)bl + t) * 4096;  // slab b + tl, laid out [64][64] whatever C is
    if constexpr (R == 1) {
      // Sum of the 4 waves as a two-level tree through LDS (fixed order: (w0 + w1) + (w2 + w3)), then wave 0 stores the
      // 64 x 64 block from its registers.  D layout: row = (lane>>4)*4 + i -> a (ci = 4a + c), col = lane&15 -> a'
      // (co = 4a' + d): for fixed (c, i) a lane's 4 d-values are 4 consecutive co -> one 16-byte store.
      auto stage = [&](float *tile) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            *(f32x4 *)(tile + ((lq * 4 + i) * 4 + c) * 64 + la * 4) = (f32x4){acc[c][0][i], acc[c][1][i], acc[c][2][i], acc[c][3][i]};
      };
      auto fold = [&](const float *tile) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 x4 = *(const f32x4 *)(tile + ((lq * 4 + i) * 4 + c) * 64 + la * 4);
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[c][d][i] += x4[d];
          }
      };
      if (wv & 1) stage(red + (wv >> 1) * 4096);    // w1 -> tile 0, w3 -> tile 1
      __syncthreads();
      if (!(wv & 1)) fold(red + (wv >> 1) * 4096);  // w0 += w1, w2 += w3
      __syncthreads();
      if (wv == 2) stage(red);
      __syncthreads();
      if (wv == 0) {
        fold(red);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            *(f32x4 *)(dst + ((lq * 4 + i) * 4 + c) * 64 + la * 4) = (f32x4){acc[c][0][i], acc[c][1][i], acc[c][2][i], acc[c][3][i]};
      }
    } else {
      // D_cd[m][n]: m = lq*4 + i = (row group m / AV, vector m % AV), n = la = (r', a').  The diagonal block of row group g
      // sits in lanes with r' == g and m / AV == g; group g's copy of element (vector a, vector a') is `g * step` lanes
      // above group 0's.  Each wave leaves its C x C sum in LDS, then the 4 are added in a fixed order.
      constexpr int C = 64 / R, step = 16 * (4 / R) + AV;
      const bool owner = lq < 4 / R && la < AV;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 v4;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const float tv = acc[c][d][i];
            float sum = tv;
#pragma unroll
            for (int gg = 1; gg < R; ++gg) sum += __shfl(tv, (lane + gg * step) & 63);
            v4[d] = sum;
          }
          if (owner) {
            const int cii = ((lq * 4 + i) % AV) * 4 + c;  // input channel 4a + c with a = m % AV
            *(f32x4 *)(red + wv * C * C + cii * C + la * 4) = v4;
          }
        }
      __syncthreads();
      for (int e = threadIdx.x * 4; e < C * C; e += 256 * 4) {
        f32x4 tsum = *(const f32x4 *)(red + e);
#pragma unroll
        for (int w = 1; w < 4; ++w) tsum += *(const f32x4 *)(red + w * C * C + e);  // fixed order
        *(f32x4 *)(dst + (e / C) * 64 + (e % C)) = tsum;
      }
    }
    __syncthreads();  // the staging tiles become lists again
  }
#ifdef BFHIP_WGRAD_TRACE
  if (lane == 0) {  // 6 x u64 per wave behind the slabs
    unsigned long long *tr = (unsigned long long *)(partial + ((size_t)gridDim.x + (size_t)g.NR * KV * GI * GJ) * 4096 + 64 * kCountSlices) + ((size_t)blockIdx.x * 4 + wv) * 6;
    tr[0] = tr_t0; tr[1] = tr_t1; tr[2] = tr_t2; tr[3] = wall_clock64(); tr[4] = (unsigned long long)tr_cnt; tr[5] = ((unsigned long long)tr_k << 32);
  }
#endif
}

// dW[co][k][ci] = sum over the workgroups b whose run touched tile t(k, ci / 64, co / 64) of partial[b + t][ci % 64][co % 64]
__global__ __launch_bounds__(256) void wgrad_reduce_sk_kernel(const float *__restrict__ partial, int KV, int Cin, int Cout,
                                                              int GI, int GJ, SkGeom g, int P, const int *__restrict__ counts,
                                                              float *__restrict__ dW) {
  // 64 consecutive elements per workgroup, the regions of each dealt to 4 thread groups (region = jl, jl + 4, ...) and the
  // 4 sums added in a fixed order: a narrow layer has few elements and ~30 slabs per tile, one thread per element would
  // walk them one dependent load at a time
  __shared__ int s_w[64], s_wpre[65];
  __shared__ float s_sum[4][64];
  if (threadIdx.x < 64) sk_make_plan(counts, KV, s_w, s_wpre);
  __syncthreads();
  const int el = threadIdx.x & 63, jl = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + el;
  const bool live = e < (long long)KV * Cin * Cout;
  float acc = 0.f;
  int co = 0, ci = 0, k = 0;
  if (live) {
    co = (int)(e % Cout);
    const long long r = e / Cout;
    ci = (int)(r % Cin); k = (int)(r / Cin);
    const int ti = (ci >> 6) * GJ + (co >> 6), T = KV * GI * GJ;
    const long long VR = (long long)g.M * s_wpre[KV], V = VR * g.NR, q = (V + P - 1) / P;
    const long long L = (long long)g.Ur * s_w[k];
    const float *base = partial + (ci & 63) * 64 + (co & 63);
    for (int rg = jl; rg < g.NR; rg += 4) {
      const long long vs = (long long)rg * VR + (long long)g.M * s_wpre[k] + (long long)ti * L;
      const long long first = vs / q, last = (vs + L - 1) / q;
      const float *src = base + ((size_t)first + (size_t)rg * T + (size_t)k * GI * GJ + ti) * 4096;
      const int J = (int)(last - first) + 1;
      float a0 = 0.f, a1 = 0.f;
      int j = 0;
      for (; j + 1 < J; j += 2) { a0 += src[(size_t)j * 4096]; a1 += src[(size_t)(j + 1) * 4096]; }
      if (j < J) a0 += src[(size_t)j * 4096];
      acc += a0 + a1;
    }
  }
  s_sum[jl][el] = acc;
  __syncthreads();
  if (live && jl == 0) dW[((size_t)co * KV + k) * Cin + ci] = (s_sum[0][el] + s_sum[1][el]) + (s_sum[2][el] + s_sum[3][el]);
}

// dW[co][k][ci] = sum_s partial[s][k][ci][co]   (fixed order -> deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int S,
                                                           int KV, int Cin, int Cout,
                                                           float *__restrict__ dW) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)KV * Cin * Cout;
  if (t >= total) return;
  int co = (int)(t % Cout);
  long long r = t / Cout;
  int ci = (int)(r % Cin);
  int k = (int)(r / Cin);
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += partial[(size_t)s * total + t];
  dW[((size_t)co * KV + k) * Cin + ci] = acc;
}

// -------------------------------------------------------------------------------- dense + permute
// BF/sparse_encoder.py:147-151: dense [B,C,X,Y,Z] -> permute(0,1,4,2,3) -> view [B, C*Z, X, Y]
__global__ __launch_bounds__(256) void sparse_to_bev_kernel(const float *__restrict__ feats,
                                                            const int4 *__restrict__ indices, int N,
                                                            int C, int X, int Y, int Z,
                                                            float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = t / C;
  int c = (int)(t - n * C);
  if (n >= N) return;
  int4 id = indices[n];
  if (id.x < 0) return;  // inactive row
  out[((((size_t)id.x * C + c) * Z + id.w) * X + id.y) * Y + id.z] = feats[t];
}

__global__ __launch_bounds__(256) void bev_to_sparse_kernel(const float *__restrict__ grad_out,
                                                            const int4 *__restrict__ indices, int N,
                                                            int C, int X, int Y, int Z,
                                                            float *__restrict__ grad_feats) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = t / C;
  int c = (int)(t - n * C);
  if (n >= N) return;
  int4 id = indices[n];
  grad_feats[t] = id.x < 0 ? 0.f : grad_out[((((size_t)id.x * C + c) * Z + id.w) * X + id.y) * Y + id.z];
}

// channels-last variants: out[b][x][y][c*Z + z] (the NHWC memory of the [B, C*Z, X, Y] BEV map), f32 or bf16; a sparse row's
// C channels land in one (Z-interleaved) segment of the pixel instead of C planes X*Y apart
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <typename T>
__global__ __launch_bounds__(256) void sparse_to_bev_nhwc_kernel(const float *__restrict__ feats,
                                                                 const int4 *__restrict__ indices, int N, int C,
                                                                 int X, int Y, int Z, T *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = t / C;
  int c = (int)(t - n * C);
  if (n >= N) return;
  int4 id = indices[n];
  if (id.x < 0) return;  // inactive row
  size_t o = (((size_t)id.x * X + id.y) * Y + id.z) * ((size_t)C * Z) + (size_t)c * Z + id.w;
  if (sizeof(T) == 2) ((unsigned short *)out)[o] = f32_to_bf16_rne(feats[t]);
  else ((float *)out)[o] = feats[t];
}

// grad element (b, ch, x, y) at grad[b*sb + x*sx + y*sy + ch] (channel stride 1; a channel slice of a wider NHWC tensor works)
template <typename T>
__global__ __launch_bounds__(256) void bev_nhwc_to_sparse_kernel(const T *__restrict__ grad, long long sb, long long sx,
                                                                 long long sy, const int4 *__restrict__ indices,
                                                                 int N, int C, int Z, float *__restrict__ grad_feats) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = t / C;
  int c = (int)(t - n * C);
  if (n >= N) return;
  int4 id = indices[n];
  if (id.x < 0) { grad_feats[t] = 0.f; return; }
  size_t o = (size_t)(id.x * sb + id.y * sx + id.z * sy) + (size_t)c * Z + id.w;
  if (sizeof(T) == 2) grad_feats[t] = __uint_as_float((unsigned)((const unsigned short *)grad)[o] << 16);
  else grad_feats[t] = ((const float *)grad)[o];
}

inline unsigned table_cap(int n) {
  unsigned cap = 1024;
  while (cap < 2u * (unsigned)n) cap <<= 1;
  return cap;
}

inline int make_geom(int B, const int *in_shape, const int *ksize, const int *stride,
                     const int *padding, const int *dilation, bool subm, ConvGeom &G) {
  G.B = B;
  G.in0 = in_shape[0]; G.in1 = in_shape[1]; G.in2 = in_shape[2];
  G.k0 = ksize[0]; G.k1 = ksize[1]; G.k2 = ksize[2];
  G.d0 = dilation[0]; G.d1 = dilation[1]; G.d2 = dilation[2];
  if (subm) {
    G.s0 = G.s1 = G.s2 = 1;
    G.p0 = G.d0 * (G.k0 / 2); G.p1 = G.d1 * (G.k1 / 2); G.p2 = G.d2 * (G.k2 / 2);
    G.out0 = G.in0; G.out1 = G.in1; G.out2 = G.in2;
  } else {
    G.s0 = stride[0]; G.s1 = stride[1]; G.s2 = stride[2];
    G.p0 = padding[0]; G.p1 = padding[1]; G.p2 = padding[2];
    // (in + 2p - d(k-1) - 1)//s + 1   (projects/SparseConvolution/sparse_conv.py:88-90)
    G.out0 = (G.in0 + 2 * G.p0 - G.d0 * (G.k0 - 1) - 1) / G.s0 + 1;
    G.out1 = (G.in1 + 2 * G.p1 - G.d1 * (G.k1 - 1) - 1) / G.s1 + 1;
    G.out2 = (G.in2 + 2 * G.p2 - G.d2 * (G.k2 - 1) - 1) / G.s2 + 1;
  }
  G.KV = G.k0 * G.k1 * G.k2;
  if (B <= 0 || G.in0 <= 0 || G.in1 <= 0 || G.in2 <= 0 || G.KV <= 0 || G.KV > 64) return -1;
  if (G.s0 <= 0 || G.s1 <= 0 || G.s2 <= 0 || G.out0 <= 0 || G.out1 <= 0 || G.out2 <= 0) return -1;
  if ((long long)B * G.in0 * G.in1 * G.in2 >= 0x7fffffffLL) return -1;
  return 0;
}

template <int NT>
void launch_gemm(int R, int blocks_rows, hipStream_t stream, const float *in, int Kdim,
                 const f32x4 *Wp, const int *pairs, int ld, int KV, int n_rows, int Ndim, const int *perm,
                 const unsigned *row_mask, float *out) {
  auto grid = [&](int r) { return dim3(ceil_div(n_rows, 4 * r * 16)); };  // one workgroup = 4 waves x r*16 rows
  (void)blocks_rows;
  if (R == 1)
    hipLaunchKernelGGL((spconv_gemm_lds_kernel<NT, 1>), grid(1), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
  else if (R == 2)
    hipLaunchKernelGGL((spconv_gemm_lds_kernel<NT, 2>), grid(2), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
  else
    hipLaunchKernelGGL((spconv_gemm_lds_kernel<NT, 4>), grid(4), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
}

// Pipelined variant of spconv_gemm_bf16_kernel: the same tiling (workgroup = 4 waves x R x 16 output rows, all NT column
// tiles, one (offset, 32-channel chunk) per step with the weight tile shared through LDS), restructured around latency.
// The original issued the gathers and the weight-tile loads of step s+1 during step s: one step is only NT * R MFMAs
// (0.1-0.4 us), so most of every ~1 us round trip was exposed, 27 * CC times per workgroup.  Here
//  * all pair-table entries of the workgroup's rows are fetched once, back to back, into LDS (s_idx) -- no dependent
//    index load remains inside the loop;
//  * loads run TWO steps ahead: three register sets used cyclically (the loop is unrolled by 3 so that no set is copied
//    while its loads are in flight), gathers are unconditional from clamped addresses and masked when consumed.
template <int NT, int R, bool IO16>
__global__ __launch_bounds__(256) void spconv_gemm_bf16p_kernel(const void *__restrict__ in_, int Kdim,
                                                                const bf16x8 *__restrict__ Wp,
                                                                const int *__restrict__ pairs, int ld,
                                                                int KV, int n_rows, int Ndim,
                                                                const int *__restrict__ perm,
                                                                const unsigned *__restrict__ row_mask,
                                                                void *__restrict__ out_) {
  const float *in = (const float *)in_;
  const __bf16 *in16 = (const __bf16 *)in_;
  float *out = (float *)out_;
  __bf16 *out16 = (__bf16 *)out_;
  __shared__ bf16x8 sB[2][NT * 64];
  __shared__ unsigned s_mask;
  __shared__ int s_idx[4][32][16 * R];  // [wave][offset][row of the wave]; KV <= 32 (checked by the host)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  // XCD-chunked block order (common.h): each XCD walks one contiguous eighth of the (region-major sorted) rows
  const long long row_base = (xcd_chunked_block(blockIdx.x, gridDim.x) * 4 + wv) * (R * 16);
  const int lr = lane & 15, lq = lane >> 4;
  const int CC = (Kdim + 31) >> 5;
  if (tid == 0) s_mask = 0u;
  __syncthreads();
  int my_row[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    long long sp = row_base + r * 16 + lr;
    my_row[r] = sp < n_rows ? (perm ? perm[sp] : (int)sp) : -1;
  }
  // pair entries of this wave's rows for every offset: lane (lr, lq) takes offsets lq, lq + 4, ...
  {
    int v[8][R];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int kk = min(lq + 4 * j, KV - 1);
        v[j][r] = pairs[(size_t)kk * ld + max(my_row[r], 0)];
      }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (lq + 4 * j < KV) s_idx[wv][lq + 4 * j][r * 16 + lr] = my_row[r] >= 0 ? v[j][r] : -1;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  unsigned wmask = 0u;
  if (row_mask) {
    unsigned mm = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) mm |= my_row[r] >= 0 ? row_mask[my_row[r]] : 0u;
    for (int o = 32; o > 0; o >>= 1) mm |= __shfl_xor(mm, o);
    wmask = mm;
  } else {
    for (int k = 0; k < KV; ++k) {
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) any |= s_idx[wv][k][r * 16 + lr] >= 0;
      if (__any(any)) wmask |= 1u << k;
    }
  }
  if (lane == 0 && wmask) atomicOr(&s_mask, wmask);
  __syncthreads();
  const unsigned gmask = s_mask;

  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int LOADS = (NT * 64 + 255) / 256;
  struct Set {  // what one step needs from memory
    bf16x8 pre[LOADS];
    f32x4 lo[R], hi[R];
    int ix[R];
  };
  auto next_set = [&](int kk) {
    unsigned rest = gmask & ~((2u << kk) - 1u);
    return rest ? __ffs(rest) - 1 : KV;
  };
  auto advance = [&](int &k, int &cc) {
    if (++cc == CC) { cc = 0; k = next_set(k); }
  };
  auto issue = [&](Set &st, int k, int cc) {  // weight tile of (k, cc) and the gathered operand rows; no waits
    const bf16x8 *wp = Wp + ((size_t)(k * CC + cc) * NT) * 64;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int e = tid + i * 256;
      st.pre[i] = wp[e < NT * 64 ? e : 0];
    }
    const int ch = cc * 32 + lq * 8;
    const int chc = ch < Kdim ? ch : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ix = s_idx[wv][k][r * 16 + lr];
      st.ix[r] = ch < Kdim ? ix : -1;
      const size_t off = (size_t)max(ix, 0) * Kdim + chc;
      if (IO16) {
        st.lo[r] = *(const f32x4 *)(in16 + off);  // 8 bf16 carried in one 16-byte register group
      } else {
        st.lo[r] = *(const f32x4 *)(in + off);
        st.hi[r] = *(const f32x4 *)(in + off + 4);
      }
    }
  };
  int buf = 0;
  auto step = [&](Set &cur, Set &dst, int k, int k2, int cc2) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int e = tid + i * 256;
      if (e < NT * 64) sB[buf][e] = cur.pre[i];
    }
    __syncthreads();
    if (k2 < KV) issue(dst, k2, cc2);  // block-uniform: two steps ahead
    if ((wmask >> k) & 1u) {
      bf16x8 a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (IO16) {
          a[r] = __builtin_bit_cast(bf16x8, cur.ix[r] >= 0 ? cur.lo[r] : (f32x4){0.f, 0.f, 0.f, 0.f});
        } else {
          const f32x4 l = cur.ix[r] >= 0 ? cur.lo[r] : (f32x4){0.f, 0.f, 0.f, 0.f};
          const f32x4 h = cur.ix[r] >= 0 ? cur.hi[r] : (f32x4){0.f, 0.f, 0.f, 0.f};
          a[r][0] = (__bf16)l[0]; a[r][1] = (__bf16)l[1]; a[r][2] = (__bf16)l[2]; a[r][3] = (__bf16)l[3];
          a[r][4] = (__bf16)h[0]; a[r][5] = (__bf16)h[1]; a[r][6] = (__bf16)h[2]; a[r][7] = (__bf16)h[3];
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 b = sB[buf][nt * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r)
          acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b, acc[r][nt], 0, 0, 0);
      }
    }
    buf ^= 1;
  };
  Set s0, s1, s2;
  int k0 = gmask ? __ffs(gmask) - 1 : KV, c0 = 0;
  int k1 = k0, c1 = c0;
  if (k0 < KV) { issue(s0, k0, c0); advance(k1, c1); }
  int k2 = k1, c2 = c1;
  if (k1 < KV) { issue(s1, k1, c1); advance(k2, c2); }
  while (k0 < KV) {  // k0 / k1 / k2: offsets of the current step and the two after it (KV = none)
    step(s0, s2, k0, k2, c2);
    k0 = k1; c0 = c1; k1 = k2; c1 = c2; if (k2 < KV) advance(k2, c2);
    if (k0 >= KV) break;
    step(s1, s0, k0, k2, c2);
    k0 = k1; c0 = c1; k1 = k2; c1 = c2; if (k2 < KV) advance(k2, c2);
    if (k0 >= KV) break;
    step(s2, s1, k0, k2, c2);
    k0 = k1; c0 = c1; k1 = k2; c1 = c2; if (k2 < KV) advance(k2, c2);
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = __shfl(my_row[r], lq * 4 + i);
      if (row >= 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          int col = nt * 16 + lr;
          if (col < Ndim) {
            if (IO16) out16[(size_t)row * Ndim + col] = (__bf16)acc[r][nt][i];
            else out[(size_t)row * Ndim + col] = acc[r][nt][i];
          }
        }
      }
    }
}

template <int NT, bool IO16>
void launch_gemm_bf16(int R, hipStream_t stream, const void *in, int Kdim, const bf16x8 *Wp, const int *pairs,
                      int ld, int KV, int n_rows, int Ndim, const int *perm, const unsigned *row_mask, void *out) {
  auto grid = [&](int r) { return dim3(ceil_div(n_rows, 4 * r * 16)); };
  // measured on the encoder's layers (tools/gemm_micro.py, bf16 storage): 128 -> 128: 88.8 -> 79.4 us, 64 -> 128: 53.1 -> 46.5,
  // 64 -> 64: 53.2 -> 52.6; 32 -> 32: 48.1 -> 59.1 and 16 -> 32: 40.0 -> 59.8 (27 one-chunk steps: the up-front index fetch costs
  // more than the deeper pipeline saves), hence the split at 64 input channels
  if (Kdim >= 64) {
    if (R == 1)
      hipLaunchKernelGGL((spconv_gemm_bf16p_kernel<NT, 1, IO16>), grid(1), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
    else if (R == 2)
      hipLaunchKernelGGL((spconv_gemm_bf16p_kernel<NT, 2, IO16>), grid(2), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
    else
      hipLaunchKernelGGL((spconv_gemm_bf16p_kernel<NT, 4, IO16>), grid(4), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
    return;
  }
  if (R == 1)
    hipLaunchKernelGGL((spconv_gemm_bf16_kernel<NT, 1, IO16>), grid(1), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
  else if (R == 2)
    hipLaunchKernelGGL((spconv_gemm_bf16_kernel<NT, 2, IO16>), grid(2), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
  else
    hipLaunchKernelGGL((spconv_gemm_bf16_kernel<NT, 4, IO16>), grid(4), dim3(256), 0, stream, in, Kdim, Wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out);
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

// ------------------------------------------------------------------------------------ C ABI
BFHIP_EXPORT int bfhip_conv_out_shape(const int *in_shape, const int *ksize, const int *stride,
                                      const int *padding, const int *dilation, int *out_shape) {
  ConvGeom G;
  BFHIP_REQUIRE(make_geom(1, in_shape, ksize, stride, padding, dilation, false, G) == 0, "conv_out_shape: bad geometry");
  out_shape[0] = G.out0; out_shape[1] = G.out1; out_shape[2] = G.out2;
  return BFHIP_OK;
}

// Row masks + mask-sorted row permutation of a pair table (for tile-level offset skipping).
// rocPRIM sorts fewer than 2^20 items with a block sort + log2(n / block) merge passes (10 launches of ~5 us for the 10^5 rows
// of an encoder level).  Forcing its Onesweep radix path (merge limit 0) was measured and is slower here: the 12 sorts of a
// LiDAR pass took 1.29 ms instead of 0.52 ms (tools/sparse_micro.py), so the default configuration stays.
using RowSortConfig = rocprim::default_config;

static inline size_t sort32_bytes(int n, int bits) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<RowSortConfig, unsigned *, unsigned *, unsigned *, unsigned *>(
      nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (size_t)n, 0, bits, 0);
  return bytes;
}

BFHIP_EXPORT size_t bfhip_rulebook_sort_rows_workspace_bytes(int n_rows, int KV) {
  (void)KV;
  if (n_rows <= 0) return 256;
  return 3 * align_up((size_t)n_rows * 4, 256) + align_up(sort32_bytes(n_rows, 32), 256) + 256;
}

namespace {
// Sort key = (region of the row, mask): rows come in voxel order, so a region is a slab of space.  With the gather-GEMM's
// XCD-chunked block order each XCD then walks one region and the neighbour rows it gathers stay in its own 4 MiB L2
// (forward gather-GEMM of the encoder's layers 15-25 % faster than with a pure mask sort, tools/gemm_micro.py; 16 or
// 32 regions measured no better).  Tiles still share their offsets inside a region.
inline int sort_regions(int KV) { return KV + 3 <= 32 ? 8 : 1; }
inline int sort_bits(int KV) { return KV + (KV + 3 <= 32 ? 3 : 0); }

struct SortScratch {
  unsigned *iota, *keys_out, *keys;
  char *tmp;
  size_t tmp_bytes;
  SortScratch(void *workspace, size_t bytes, int n_rows) {
    Workspace ws(workspace, bytes);
    iota = ws.take<unsigned>(n_rows);
    keys_out = ws.take<unsigned>(n_rows);
    keys = ws.take<unsigned>(n_rows);
    tmp_bytes = sort32_bytes(n_rows, 32);
    tmp = ws.take<char>(tmp_bytes);
  }
  hipError_t sort(int n_rows, int KV, int32_t *perm, hipStream_t stream);
};

// Round 3: the row order in ONE launch.  Rows are sorted by offset mask inside consecutive chunks of kSortChunk rows (a
// chunk = a slab of space, rows being in voxel order); one 512-thread workgroup sorts one chunk in LDS (stable LSD radix,
// rocprim::block_radix_sort), instead of a device-wide sort per rulebook -- which for these 10^5-row arrays was a block sort
// plus 6-7 merge passes of ~5 us each, 12 times per LiDAR pass: 0.52 of the 1.15 ms the rulebooks cost.  The chunked order
// keeps what the device-wide (eighth of the row range, mask) order was for: tiles of 16-64 consecutive sorted rows share
// their offsets, and consecutive tiles stay in one slab of space (the gather-GEMM deals contiguous eighths of the tiles to
// the 8 XCDs).
constexpr int kSortChunk = 4096, kSortThreads = 512, kSortItems = kSortChunk / kSortThreads;

__global__ __launch_bounds__(kSortThreads) void sort_rows_chunk_kernel(const unsigned *__restrict__ keys, int n_rows, int KV,
                                                                      int *__restrict__ perm) {
  using BlockSort = rocprim::block_radix_sort<unsigned, kSortThreads, kSortItems, unsigned>;
  __shared__ typename BlockSort::storage_type storage;
  const int base = blockIdx.x * kSortChunk;
  const unsigned invalid = 1u << KV, low = invalid - 1u;  // rows beyond n_rows sort behind every real mask (KV <= 30)
  unsigned k[kSortItems], v[kSortItems];
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const int idx = base + threadIdx.x * kSortItems + i;
    k[i] = idx < n_rows ? (keys[idx] & low) : invalid;
    v[i] = (unsigned)idx;
  }
  BlockSort().sort(k, v, storage, 0, KV + 1);
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const int pos = base + threadIdx.x * kSortItems + i;
    if (pos < n_rows) perm[pos] = (int)v[i];
  }
}

hipError_t SortScratch::sort(int n_rows, int KV, int32_t *perm, hipStream_t stream) {
  static const int chunked = [] { const char *e = getenv("BFHIP_SPCONV_CHUNK_SORT"); return e ? atoi(e) : 1; }();
  if (chunked && KV <= 30) {
    hipLaunchKernelGGL(sort_rows_chunk_kernel, dim3(ceil_div(n_rows, kSortChunk)), dim3(kSortThreads), 0, stream, keys, n_rows, KV,
                       perm);
    return hipGetLastError();
  }
  return rocprim::radix_sort_pairs<RowSortConfig>(tmp, tmp_bytes, keys, keys_out, iota, (unsigned *)perm, (size_t)n_rows, 0, sort_bits(KV), stream);
}
}  // namespace

// SubM rulebook.  row_mask / perm (optional, KV <= 32): the offset masks of the rows and their (region, mask)-sorted order, as
// bfhip_rulebook_sort_rows(pair_fwd) would give them, produced by the same launch that fills the table.
BFHIP_EXPORT size_t bfhip_rulebook_subm_workspace_bytes(int N) {
  return align_up((size_t)table_cap(N > 0 ? N : 1) * sizeof(int2), 256) + bfhip_rulebook_sort_rows_workspace_bytes(N, 27) + 256;
}

BFHIP_EXPORT int bfhip_rulebook_subm(const int32_t *indices, int N, int B, const int *in_shape,
                                     const int *ksize, const int *dilation, int32_t *pair_fwd,
                                     int32_t *n_pairs_dev, uint32_t *row_mask, int32_t *perm, void *workspace,
                                     size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom G;
  BFHIP_REQUIRE(N >= 0, "rulebook_subm: N < 0");
  BFHIP_REQUIRE(make_geom(B, in_shape, ksize, nullptr, nullptr, dilation, true, G) == 0,
                "rulebook_subm: bad geometry (B*X*Y*Z must be < 2^31, kernel volume <= 64)");
  BFHIP_REQUIRE(!(row_mask || perm) || G.KV <= 32, "rulebook_subm: row masks need a kernel volume <= 32");
  BFHIP_REQUIRE(!perm || row_mask, "rulebook_subm: perm without row_mask");
  if (n_pairs_dev && hipMemsetAsync(n_pairs_dev, 0, 64 * sizeof(int), stream) != hipSuccess) return check_launch("rulebook_subm memset");
  if (N == 0) return BFHIP_OK;
  BFHIP_REQUIRE(indices && pair_fwd && ((uintptr_t)indices % 16) == 0, "rulebook_subm: null/unaligned pointer");
  if (workspace_bytes < bfhip_rulebook_subm_workspace_bytes(N) || !workspace) {
    set_error("rulebook_subm: workspace too small");
    return BFHIP_E_WORKSPACE;
  }
  Workspace ws(workspace, workspace_bytes);
  unsigned cap = table_cap(N);
  int2 *table = ws.take<int2>(cap);
  size_t sort_bytes = bfhip_rulebook_sort_rows_workspace_bytes(N, G.KV);
  SortScratch ss(ws.take<char>(sort_bytes), sort_bytes, N);
  ProfScope ps;
  prof_begin(BFHIP_OP_RULEBOOK, stream, &ps);
  hipLaunchKernelGGL(fill_pair_kernel, dim3(ceil_div(cap, 256)), dim3(256), 0, stream, table, (int)cap);
  hipLaunchKernelGGL(subm_insert_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, stream, (const int4 *)indices, N, G,
                     table, cap - 1);
  hipError_t e = hipSuccess;
  if (G.k0 * G.k1 <= 16 && G.KV <= 32) {
    dim3 block(64, G.k0 * G.k1);
    unsigned *iota = perm ? ss.iota : nullptr, *keys = perm ? ss.keys : nullptr;
    if (G.k2 == 3)
      hipLaunchKernelGGL(subm_pairs_rows_kernel<3>, dim3(ceil_div(N, 64)), block, 0, stream, (const int4 *)indices, N, G, table,
                         cap - 1, pair_fwd, row_mask, iota, keys, sort_regions(G.KV), n_pairs_dev);
    else
      hipLaunchKernelGGL(subm_pairs_rows_kernel<0>, dim3(ceil_div(N, 64)), block, 0, stream, (const int4 *)indices, N, G, table,
                         cap - 1, pair_fwd, row_mask, iota, keys, sort_regions(G.KV), n_pairs_dev);
  } else {
    // wide kernels: one thread per (row, offset) over the lower half of the offsets, mirrored writes (pre-filled table)
    (void)hipMemsetAsync(pair_fwd, 0xff, (size_t)G.KV * N * sizeof(int), stream);
    const int symmetric = (G.k0 % 2 == 1) && (G.k1 % 2 == 1) && (G.k2 % 2 == 1);
    const int nk = symmetric ? G.KV / 2 + 1 : G.KV;
    hipLaunchKernelGGL(subm_pairs_kernel, dim3(ceil_div(N, 256), nk), dim3(256), 0, stream, (const int4 *)indices, N,
                       G, table, cap - 1, pair_fwd, n_pairs_dev, symmetric);
    if (row_mask)
      hipLaunchKernelGGL(row_mask_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, stream, pair_fwd, N, G.KV, N, row_mask, ss.iota,
                         ss.keys, sort_regions(G.KV));
  }
  if (perm) e = ss.sort(N, G.KV, perm, stream);
  prof_end(&ps);
  if (e != hipSuccess) { set_error("rulebook_subm: sort: %s", hipGetErrorString(e)); return BFHIP_E_LAUNCH; }
  return check_launch("rulebook_subm");
}

// Strided rulebook, two phases (the output row count must reach the host to size the outputs):
//   count: bitmap over the output grid + prefix popcounts -> counts_dev[0] = N_out
//   fill : out_indices (ascending linear order), pair_fwd[KV, ld_out], pair_bwd[KV, N],
//          sum(counts_dev[1..64]) = pairs (64 spread counters: one hot word would serialise the atomics)
// The workspace must be kept untouched between the two calls.
BFHIP_EXPORT size_t bfhip_rulebook_sparse_workspace_bytes(int B, const int *in_shape, const int *ksize,
                                                          const int *stride, const int *padding,
                                                          const int *dilation) {
  ConvGeom G;
  if (make_geom(B, in_shape, ksize, stride, padding, dilation, false, G) != 0) return 0;
  long long cells = (long long)B * G.out0 * G.out1 * G.out2;
  long long nwords = (cells + 31) / 32;
  size_t nb = (size_t)ceil_div(nwords, kScan);
  return 2 * align_up((size_t)nwords * sizeof(int), 256) + align_up((nb + 1) * sizeof(int), 256) + 256;
}

BFHIP_EXPORT int bfhip_rulebook_sparse_count(const int32_t *indices, int N, const int32_t *n_in_dev, int B,
                                             const int *in_shape, const int *ksize, const int *stride,
                                             const int *padding, const int *dilation, int32_t *counts_dev,
                                             void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom G;
  BFHIP_REQUIRE(N >= 0, "rulebook_sparse: N < 0");
  BFHIP_REQUIRE(make_geom(B, in_shape, ksize, stride, padding, dilation, false, G) == 0, "rulebook_sparse: bad geometry");
  BFHIP_REQUIRE(counts_dev && (N == 0 || (indices && ((uintptr_t)indices % 16) == 0)), "rulebook_sparse: null/unaligned pointer");
  size_t need = bfhip_rulebook_sparse_workspace_bytes(B, in_shape, ksize, stride, padding, dilation);
  if (workspace_bytes < need || !workspace) { set_error("rulebook_sparse: workspace too small (%zu < %zu)", workspace_bytes, need); return BFHIP_E_WORKSPACE; }
  long long cells = (long long)B * G.out0 * G.out1 * G.out2;
  long long nwords = (cells + 31) / 32;
  int nb = ceil_div(nwords, kScan);
  Workspace ws(workspace, workspace_bytes);
  unsigned *bitmap = ws.take<unsigned>(nwords);
  int *word_prefix = ws.take<int>(nwords);
  int *blk = ws.take<int>(nb + 1);
  ProfScope ps;
  prof_begin(BFHIP_OP_RULEBOOK, stream, &ps);
  (void)hipMemsetAsync(bitmap, 0, nwords * sizeof(unsigned), stream);
  (void)hipMemsetAsync(counts_dev, 0, 65 * sizeof(int), stream);
  if (N > 0) {
    hipLaunchKernelGGL(sparse_mark_kernel, dim3(ceil_div(N, 256), G.KV), dim3(256), 0, stream, (const int4 *)indices, N, n_in_dev, G, bitmap);
  }
  hipLaunchKernelGGL(words_count_kernel, dim3(nb), dim3(kScan), 0, stream, bitmap, nwords, blk);
  hipLaunchKernelGGL(blocks_scan_kernel, dim3(1), dim3(kScan), 0, stream, blk, nb, counts_dev);
  hipLaunchKernelGGL(words_prefix_kernel, dim3(nb), dim3(kScan), 0, stream, bitmap, nwords, blk, word_prefix);
  prof_end(&ps);
  return check_launch("rulebook_sparse_count");
}

// Output coordinates of a counted layer into a buffer of `cap` rows, without knowing N_out on the host (rows beyond cap are
// dropped; the caller compares counts_dev[0] with cap after its single read).  Lets a chain of strided layers be counted
// back to back: the next layer's bfhip_rulebook_sparse_count takes this buffer with n_in_dev = counts_dev of this one.
BFHIP_EXPORT int bfhip_rulebook_sparse_out_indices(int B, const int *in_shape, const int *ksize, const int *stride,
                                                   const int *padding, const int *dilation, int cap,
                                                   int32_t *out_indices, void *workspace, size_t workspace_bytes,
                                                   void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom G;
  BFHIP_REQUIRE(make_geom(B, in_shape, ksize, stride, padding, dilation, false, G) == 0, "rulebook_sparse_out_indices: bad geometry");
  BFHIP_REQUIRE(cap > 0 && out_indices && ((uintptr_t)out_indices % 16) == 0, "rulebook_sparse_out_indices: bad output buffer");
  size_t need = bfhip_rulebook_sparse_workspace_bytes(B, in_shape, ksize, stride, padding, dilation);
  if (workspace_bytes < need || !workspace) { set_error("rulebook_sparse_out_indices: workspace too small"); return BFHIP_E_WORKSPACE; }
  long long cells = (long long)B * G.out0 * G.out1 * G.out2;
  long long nwords = (cells + 31) / 32;
  Workspace ws(workspace, workspace_bytes);
  unsigned *bitmap = ws.take<unsigned>(nwords);
  int *word_prefix = ws.take<int>(nwords);
  ProfScope ps;
  prof_begin(BFHIP_OP_RULEBOOK, stream, &ps);
  hipLaunchKernelGGL(sparse_out_indices_kernel, dim3(ceil_div(nwords, 256)), dim3(256), 0, stream, bitmap, word_prefix,
                     nwords, G, cap, (int4 *)out_indices);
  prof_end(&ps);
  return check_launch("rulebook_sparse_out_indices");
}

// mask_fwd / perm_fwd (output rows) and mask_bwd / perm_bwd (input rows), optional, KV <= 32: what bfhip_rulebook_sort_rows gives
// for pair_fwd and pair_bwd; the backward ones leave with the pair kernel itself.  sort_workspace:
// bfhip_rulebook_sort_rows_workspace_bytes(max(N, n_out), KV) bytes (only needed with the masks).
BFHIP_EXPORT int bfhip_rulebook_sparse_fill(const int32_t *indices, int N, int B, const int *in_shape,
                                            const int *ksize, const int *stride, const int *padding,
                                            const int *dilation, int n_out, int32_t *out_indices,
                                            int32_t *pair_fwd, int32_t *pair_bwd, int32_t *counts_dev,
                                            uint32_t *mask_fwd, int32_t *perm_fwd, uint32_t *mask_bwd, int32_t *perm_bwd,
                                            void *sort_workspace, size_t sort_workspace_bytes,
                                            void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom G;
  BFHIP_REQUIRE(make_geom(B, in_shape, ksize, stride, padding, dilation, false, G) == 0, "rulebook_sparse: bad geometry");
  BFHIP_REQUIRE(N >= 0 && n_out >= 0, "rulebook_sparse_fill: bad sizes");
  if (N == 0 || n_out == 0) return BFHIP_OK;
  BFHIP_REQUIRE(indices && out_indices && pair_fwd && pair_bwd, "rulebook_sparse_fill: null pointer");
  BFHIP_REQUIRE(((uintptr_t)indices % 16) == 0 && ((uintptr_t)out_indices % 16) == 0, "rulebook_sparse_fill: indices must be 16-byte aligned");
  const bool masks = mask_fwd || mask_bwd;
  BFHIP_REQUIRE(!masks || (mask_fwd && mask_bwd && G.KV <= 32), "rulebook_sparse_fill: row masks come in pairs and need a kernel volume <= 32");
  BFHIP_REQUIRE((!perm_fwd && !perm_bwd) || (masks && perm_fwd && perm_bwd), "rulebook_sparse_fill: perms need the masks (and come in pairs)");
  size_t need = bfhip_rulebook_sparse_workspace_bytes(B, in_shape, ksize, stride, padding, dilation);
  if (workspace_bytes < need || !workspace) { set_error("rulebook_sparse_fill: workspace too small"); return BFHIP_E_WORKSPACE; }
  const int n_sort = N > n_out ? N : n_out;
  if (masks && (!sort_workspace || sort_workspace_bytes < bfhip_rulebook_sort_rows_workspace_bytes(n_sort, G.KV))) {
    set_error("rulebook_sparse_fill: sort workspace too small");
    return BFHIP_E_WORKSPACE;
  }
  long long cells = (long long)B * G.out0 * G.out1 * G.out2;
  long long nwords = (cells + 31) / 32;
  Workspace ws(workspace, workspace_bytes);
  unsigned *bitmap = ws.take<unsigned>(nwords);
  int *word_prefix = ws.take<int>(nwords);
  int *n_pairs = counts_dev ? counts_dev + 1 : nullptr;
  ProfScope ps;
  prof_begin(BFHIP_OP_RULEBOOK, stream, &ps);
  (void)hipMemsetAsync(pair_fwd, 0xff, (size_t)G.KV * n_out * sizeof(int), stream);
  // n_out may be a capacity (true count only on the device): rows the kernel below does not reach stay inactive (-1)
  (void)hipMemsetAsync(out_indices, 0xff, (size_t)n_out * sizeof(int4), stream);
  hipLaunchKernelGGL(sparse_out_indices_kernel, dim3(ceil_div(nwords, 256)), dim3(256), 0, stream, bitmap, word_prefix,
                     nwords, G, n_out, (int4 *)out_indices);
  hipError_t e = hipSuccess;
  if (G.k0 * G.k1 <= 16 && G.KV <= 32) {
    unsigned *iota = nullptr, *keys = nullptr;
    SortScratch sb(masks ? sort_workspace : nullptr, masks ? sort_workspace_bytes : 0, masks ? N : 0);
    if (perm_bwd) { iota = sb.iota; keys = sb.keys; }
    dim3 block(64, G.k0 * G.k1);
    if (G.k2 == 3)
      hipLaunchKernelGGL(sparse_pairs_rows_kernel<3>, dim3(ceil_div(N, 64)), block, 0, stream, (const int4 *)indices, N, G, bitmap,
                         word_prefix, n_out, pair_fwd, pair_bwd, mask_bwd, iota, keys, sort_regions(G.KV), n_pairs);
    else
      hipLaunchKernelGGL(sparse_pairs_rows_kernel<0>, dim3(ceil_div(N, 64)), block, 0, stream, (const int4 *)indices, N, G, bitmap,
                         word_prefix, n_out, pair_fwd, pair_bwd, mask_bwd, iota, keys, sort_regions(G.KV), n_pairs);
    if (perm_bwd) e = sb.sort(N, G.KV, perm_bwd, stream);
  } else {
    hipLaunchKernelGGL(sparse_pairs_kernel, dim3(ceil_div(N, 256), G.KV), dim3(256), 0, stream, (const int4 *)indices, N, G,
                       bitmap, word_prefix, n_out, pair_fwd, pair_bwd, n_pairs);
    if (masks) {
      SortScratch sb(sort_workspace, sort_workspace_bytes, N);
      hipLaunchKernelGGL(row_mask_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, stream, pair_bwd, N, G.KV, N, mask_bwd, sb.iota,
                         sb.keys, sort_regions(G.KV));
      if (perm_bwd) e = sb.sort(N, G.KV, perm_bwd, stream);
    }
  }
  if (masks && e == hipSuccess) {
    SortScratch sf(sort_workspace, sort_workspace_bytes, n_out);
    hipLaunchKernelGGL(row_mask_kernel, dim3(ceil_div(n_out, 256)), dim3(256), 0, stream, pair_fwd, n_out, G.KV, n_out, mask_fwd,
                       sf.iota, sf.keys, sort_regions(G.KV));
    if (perm_fwd) e = sf.sort(n_out, G.KV, perm_fwd, stream);
  }
  prof_end(&ps);
  if (e != hipSuccess) { set_error("rulebook_sparse_fill: sort: %s", hipGetErrorString(e)); return BFHIP_E_LAUNCH; }
  return check_launch("rulebook_sparse_fill");
}

BFHIP_EXPORT int bfhip_rulebook_sort_rows(const int32_t *pairs, int ld, int KV, int n_rows, uint32_t *row_mask,
                                          int32_t *perm, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(KV > 0 && KV <= 32 && n_rows >= 0 && ld >= n_rows, "rulebook_sort_rows: bad sizes");
  if (n_rows == 0) return BFHIP_OK;
  BFHIP_REQUIRE(pairs && row_mask, "rulebook_sort_rows: null pointer");
  if (workspace_bytes < bfhip_rulebook_sort_rows_workspace_bytes(n_rows, KV) || !workspace) { set_error("rulebook_sort_rows: workspace too small"); return BFHIP_E_WORKSPACE; }
  SortScratch sc(workspace, workspace_bytes, n_rows);
  ProfScope ps;
  prof_begin(BFHIP_OP_RULEBOOK, stream, &ps);
  hipLaunchKernelGGL(row_mask_kernel, dim3(ceil_div(n_rows, 256)), dim3(256), 0, stream, pairs, ld, KV, n_rows, row_mask, sc.iota,
                     sc.keys, sort_regions(KV));
  hipError_t e = hipSuccess;
  if (perm)  // perm == NULL: masks only
    e = sc.sort(n_rows, KV, perm, stream);
  prof_end(&ps);
  if (e != hipSuccess) { set_error("rulebook_sort_rows: sort: %s", hipGetErrorString(e)); return BFHIP_E_LAUNCH; }
  return check_launch("rulebook_sort_rows");
}

// Debug guard of the gather kernels' operands (they dereference pairs[k*ld + perm[i]] and in + pairs[..]*Kdim unchecked):
// status[0] = pair entries outside [-1, n_src), status[1] = perm entries outside [0, n_rows), status[2] = rows that do
// not occur exactly once in perm, status[3] = rows whose row_mask disagrees with the table.  `seen` = i32[n_rows], zeroed.
namespace {
__global__ __launch_bounds__(256) void rulebook_validate_kernel(const int *__restrict__ pairs, int ld, int KV, int n_rows,
                                                                int n_src, const int *__restrict__ perm,
                                                                const unsigned *__restrict__ row_mask,
                                                                int *__restrict__ seen, int *__restrict__ status) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= n_rows) return;
  unsigned m = 0u;
  int bad = 0;
  for (int k = 0; k < KV; ++k) {
    const int v = pairs[(size_t)k * ld + n];
    if (v < -1 || v >= n_src) ++bad;
    if (v >= 0 && k < 32) m |= 1u << k;
  }
  if (bad) atomicAdd(&status[0], bad);
  if (row_mask && KV <= 32 && row_mask[n] != m) atomicAdd(&status[3], 1);
  if (perm) {
    const int p = perm[n];
    if (p < 0 || p >= n_rows) atomicAdd(&status[1], 1);
    else atomicAdd(&seen[p], 1);
  }
}

__global__ __launch_bounds__(256) void rulebook_validate_perm_kernel(const int *__restrict__ seen, int n_rows,
                                                                     int *__restrict__ status) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < n_rows && seen[n] != 1) atomicAdd(&status[2], 1);
}
}  // namespace

BFHIP_EXPORT size_t bfhip_rulebook_validate_workspace_bytes(int n_rows) {
  return align_up((size_t)(n_rows > 0 ? n_rows : 1) * sizeof(int), 256) + 256;
}

BFHIP_EXPORT int bfhip_rulebook_validate(const int32_t *pairs, int ld, int KV, int n_rows, int n_src, const int32_t *perm,
                                         const uint32_t *row_mask, int32_t *status_dev, void *workspace,
                                         size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(KV > 0 && KV <= 64 && n_rows >= 0 && ld >= n_rows && n_src >= 0, "rulebook_validate: bad sizes");
  BFHIP_REQUIRE(status_dev, "rulebook_validate: status_dev is null");
  if (hipMemsetAsync(status_dev, 0, 4 * sizeof(int), stream) != hipSuccess) return check_launch("rulebook_validate memset");
  if (n_rows == 0) return BFHIP_OK;
  BFHIP_REQUIRE(pairs, "rulebook_validate: null pointer");
  if (workspace_bytes < bfhip_rulebook_validate_workspace_bytes(n_rows) || !workspace) { set_error("rulebook_validate: workspace too small"); return BFHIP_E_WORKSPACE; }
  int *seen = (int *)workspace;
  if (perm && hipMemsetAsync(seen, 0, (size_t)n_rows * sizeof(int), stream) != hipSuccess) return check_launch("rulebook_validate memset");
  hipLaunchKernelGGL(rulebook_validate_kernel, dim3(ceil_div(n_rows, 256)), dim3(256), 0, stream, pairs, ld, KV, n_rows, n_src,
                     perm, row_mask, seen, status_dev);
  if (perm)
    hipLaunchKernelGGL(rulebook_validate_perm_kernel, dim3(ceil_div(n_rows, 256)), dim3(256), 0, stream, seen, n_rows, status_dev);
  return check_launch("rulebook_validate");
}

// Gather-GEMM: out[n_rows, Ndim] = sum_k M_k . in[pairs[k][row]]  with M_k derived from W (Cout,KV,Cin):
//   transpose=0: forward  (Kdim = Cin,  Ndim = Cout)
//   transpose=1: dgrad    (Kdim = Cout, Ndim = Cin); flip=1 uses W[KV-1-k] (SubM with pair_fwd as pair_bwd)
BFHIP_EXPORT size_t bfhip_spconv_workspace_bytes(int KV, int Cin, int Cout) {
  size_t cc = (size_t)((Cin > Cout ? Cin : Cout) + 15) / 16;
  return align_up((size_t)KV * cc * cc * 64 * 4 * sizeof(float), 256) + 256;
}

BFHIP_EXPORT int bfhip_spconv_gemm(const float *in, const float *W, const int32_t *pairs, int ld, int KV,
                                   int n_rows, int Cin, int Cout, int transpose, int flip,
                                   const int32_t *perm, const uint32_t *row_mask, float *out,
                                   void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(KV > 0 && Cin > 0 && Cout > 0 && n_rows >= 0 && ld >= n_rows, "spconv_gemm: bad sizes");
  if (n_rows == 0) return BFHIP_OK;
  BFHIP_REQUIRE(in && W && pairs && out, "spconv_gemm: null pointer");
  int Kdim = transpose ? Cout : Cin, Ndim = transpose ? Cin : Cout;
  int NT = (Ndim + 15) / 16, CC = (Kdim + 15) / 16;
  bool mfma_ok = (Kdim % 16 == 0) && ((uintptr_t)in % 16 == 0) && (NT == 1 || NT == 2 || NT == 4 || NT == 8);
  ProfScope ps;
  prof_begin(transpose ? BFHIP_OP_SPCONV_BWD : BFHIP_OP_SPCONV_FWD, stream, &ps);
  if (mfma_ok) {
    if (workspace_bytes < bfhip_spconv_workspace_bytes(KV, Cin, Cout) || !workspace) { set_error("spconv_gemm: workspace too small"); return BFHIP_E_WORKSPACE; }
    float *Wp = (float *)workspace;
    long long total = (long long)KV * CC * NT * 256;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, W, Cout, KV, Cin, transpose,
                       flip, CC, NT, Wp);
    // rows per wave: keep >= ~2 waves per SIMD when the tensor is small
    // rows per wave: aim for >= ~1024 workgroups (4 per CU); a workgroup covers 4*R*16 rows
    int R = n_rows >= 262144 ? 4 : (n_rows >= 131072 ? 2 : 1);
    if (NT == 8 && R == 4) R = 2;
    BFHIP_REQUIRE(KV <= 32, "spconv_gemm: kernel volume > 32 is not supported by the MFMA path");
    const f32x4 *wp = (const f32x4 *)Wp;
    switch (NT) {
      case 1: launch_gemm<1>(R, 0, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      case 2: launch_gemm<2>(R, 0, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      case 4: launch_gemm<4>(R, 0, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      default: launch_gemm<8>(R, 0, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
    }
  } else {
    long long total = (long long)n_rows * Ndim;
    hipLaunchKernelGGL(spconv_scalar_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, in, Kdim, W, Cout, Cin,
                       transpose, flip, pairs, ld, KV, n_rows, Ndim, out);
  }
  prof_end(&ps);
  return check_launch("spconv_gemm");
}

// Same contract as bfhip_spconv_gemm with bf16 MFMA inputs (features/weights rounded to bf16 on load,
// fp32 accumulate and output): the bf16 configs (the reference runs spconv in half precision under AMP).
// Requires Kdim % 8 == 0; other shapes are rejected (use the fp32 entry point).
BFHIP_EXPORT int bfhip_spconv_gemm_bf16(const void *in, const float *W, const int32_t *pairs, int ld, int KV,
                                        int n_rows, int Cin, int Cout, int transpose, int flip,
                                        const int32_t *perm, const uint32_t *row_mask, void *out, int io_bf16,
                                        void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(KV > 0 && KV <= 32 && Cin > 0 && Cout > 0 && n_rows >= 0 && ld >= n_rows, "spconv_gemm_bf16: bad sizes");
  if (n_rows == 0) return BFHIP_OK;
  BFHIP_REQUIRE(in && W && pairs && out, "spconv_gemm_bf16: null pointer");
  int Kdim = transpose ? Cout : Cin, Ndim = transpose ? Cin : Cout;
  int NT = (Ndim + 15) / 16, CC32 = (Kdim + 31) / 32;
  BFHIP_REQUIRE(Kdim % 8 == 0 && ((uintptr_t)in % 16 == 0) && (NT == 1 || NT == 2 || NT == 4 || NT == 8),
                "spconv_gemm_bf16: needs K %% 8 == 0, N in {16,32,64,128} (padded), 16-byte aligned features");
  if (workspace_bytes < bfhip_spconv_workspace_bytes(KV, Cin, Cout) || !workspace) { set_error("spconv_gemm_bf16: workspace too small"); return BFHIP_E_WORKSPACE; }
  __bf16 *Wp = (__bf16 *)workspace;
  ProfScope ps;
  prof_begin(transpose ? BFHIP_OP_SPCONV_BWD : BFHIP_OP_SPCONV_FWD, stream, &ps);
  // (A dense-style implicit GEMM for the wide layers -- 128-row tiles, the tile's pair table and both operands staged in LDS
  // by LDS-DMA, 32x32x16 MFMA, the dense conv's ring -- was built and measured: 64 -> 64: 43.9 vs 41.9 us, 128 -> 128: 75 vs
  // 71.6 us, strided data gradients up to 3x slower (their tables are ~1/8 full and every hole is a zero-page DMA + zero MFMA).
  // With ~190 row tiles the 128-channel stage leaves a quarter of the CUs idle either way; removed.)
  long long total = (long long)KV * CC32 * NT * 512;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, W, Cout, KV, Cin, transpose,
                     flip, CC32, NT, Wp);
  int R = n_rows >= 262144 ? 4 : (n_rows >= 65536 ? 2 : 1);
  const bf16x8 *wp = (const bf16x8 *)Wp;
  if (io_bf16) {
    switch (NT) {
      case 1: launch_gemm_bf16<1, true>(R, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      case 2: launch_gemm_bf16<2, true>(R, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      case 4: launch_gemm_bf16<4, true>(R, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      default: launch_gemm_bf16<8, true>(R, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
    }
  } else {
    switch (NT) {
      case 1: launch_gemm_bf16<1, false>(R, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      case 2: launch_gemm_bf16<2, false>(R, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      case 4: launch_gemm_bf16<4, false>(R, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
      default: launch_gemm_bf16<8, false>(R, stream, in, Kdim, wp, pairs, ld, KV, n_rows, Ndim, perm, row_mask, out); break;
    }
  }
  prof_end(&ps);
  return check_launch("spconv_gemm_bf16");
}

// wgrad: dW (Cout,KV,Cin) = sum_n dout[n] (x) in[pairs[k][n]]
static inline int wgrad_splits(int KV, int GI, int GJ, int n_rows) {
  long long waves_per_split = (long long)KV * GI * GJ;
  int S = (int)(4096 / (waves_per_split > 0 ? waves_per_split : 1));
  if (S < 1) S = 1;
  int max_s = (n_rows + 63) / 64;
  if (S > max_s) S = max_s > 0 ? max_s : 1;
  if (S > 256) S = 256;
  return S;
}

// ---- geometry of the row-streamed kernel (see spconv_wgrad64p_kernel)
static int device_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else { (void)hipGetLastError(); n = 256; }
  }
  return n;
}
constexpr int kSkMaxBlocksPerCu = 4;
template <int R, bool IO16>
static int wgrad_sk_resident_blocks() {  // workgroups the chip holds at once: the grid of the streamed kernel
  static int p = 0;
  if (p == 0) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (spconv_wgrad64p_kernel<R, IO16>), 256, 0) != hipSuccess || occ < 1) {
      (void)hipGetLastError();
      occ = 2;
    }
    const char *e = getenv("BFHIP_WGRAD_BLOCKS_PER_CU");  // tuning knob (tools/wgrad_trace.py)
    if (e && atoi(e) > 0) occ = atoi(e);
    if (occ > kSkMaxBlocksPerCu) occ = kSkMaxBlocksPerCu;
    p = occ * device_cus();
  }
  return p;
}
BFHIP_EXPORT size_t bfhip_spconv_wgrad_workspace_bytes(int KV, int Cin, int Cout, int n_rows) {
  int GI = (Cin + 63) / 64, GJ = (Cout + 63) / 64;
  int S = wgrad_splits(KV, GI, GJ, n_rows);
  size_t uniform = (size_t)S * KV * Cin * Cout * sizeof(float);  // the scalar-load kernel: one slab per row split
  // the streamed 64 x 64 kernel: one slab per (workgroup, tile) incidence + the per-offset pair counts
  size_t streamed = ((size_t)kSkMaxBlocksPerCu * device_cus() + (size_t)kSkRegions * KV * GI * GJ) * 4096 * sizeof(float) + 64 * kCountSlices * sizeof(int);
  size_t need = uniform > streamed ? uniform : streamed;
  if (spconv_wgrad_tr_supported(KV, Cin, Cout)) {
    size_t tr = spconv_wgrad_tr_workspace_bytes(KV, Cin, Cout, n_rows);
    if (tr > need) need = tr;
  }
  return align_up(need, 256) + 256;
}

BFHIP_EXPORT int bfhip_spconv_wgrad(const void *in, const void *dout, const int32_t *pairs, int ld, int KV,
                                    int n_rows, int Cin, int Cout, const int32_t *perm, float *dW, int io_bf16,
                                    void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(KV > 0 && Cin > 0 && Cout > 0 && n_rows >= 0 && ld >= n_rows, "spconv_wgrad: bad sizes");
  BFHIP_REQUIRE(dW, "spconv_wgrad: dW is null");
  if (n_rows == 0) {
    if (hipMemsetAsync(dW, 0, (size_t)KV * Cin * Cout * sizeof(float), stream) != hipSuccess) return check_launch("spconv_wgrad memset");
    return BFHIP_OK;
  }
  BFHIP_REQUIRE(in && dout && pairs, "spconv_wgrad: null pointer");
  const bool vec = (Cin % 4 == 0) && (Cout % 4 == 0) && ((uintptr_t)in % 16 == 0) && ((uintptr_t)dout % 16 == 0);
  BFHIP_REQUIRE(!io_bf16 || vec, "spconv_wgrad: bf16 features need channel counts that are multiples of 4 (Cin=%d Cout=%d)", Cin, Cout);
  int GI = (Cin + 63) / 64, GJ = (Cout + 63) / 64;
  int S = wgrad_splits(KV, GI, GJ, n_rows);
  if (workspace_bytes < bfhip_spconv_wgrad_workspace_bytes(KV, Cin, Cout, n_rows) || !workspace) { set_error("spconv_wgrad: workspace too small"); return BFHIP_E_WORKSPACE; }
  float *partial = (float *)workspace;
  ProfScope ps, ps_op;
  prof_begin(BFHIP_OP_SPCONV_WGRAD, stream, &ps_op);  // the whole op: counts + main + reduce
  static const int use_tr = getenv("BFHIP_SPCONV_WGRAD_FP32MFMA") ? 0 : 1;
  if (io_bf16 && use_tr && !perm && spconv_wgrad_tr_supported(KV, Cin, Cout)) {
    // bf16 features: the bf16-MFMA kernel (16x the matrix rate of the fp32-MFMA kernels below)
    // (the BFHIP_OP_SPCONV_WGRAD_MAIN scope brackets the main kernel only, inside spconv_wgrad_tr: round 2 had it around the
    // slab sum as well, which made the "main kernel" figure 26 % larger than the kernel's trace)
    int rc = spconv_wgrad_tr(in, dout, pairs, ld, KV, n_rows, Cin, Cout, dW, workspace, workspace_bytes, stream);
    prof_end(&ps_op);
    return rc != BFHIP_OK ? rc : check_launch("spconv_wgrad");
  }
  if (vec && KV <= 64) {
    // the row-streamed MFMA kernel: R = 4 | 2 for the narrow square stages (C = 16 | 32), 64 x 64 tiles otherwise
    const int R = (Cin == Cout && Cin == 16) ? 4 : (Cin == Cout && Cin == 32) ? 2 : 1;
    int P = R == 4 ? (io_bf16 ? wgrad_sk_resident_blocks<4, true>() : wgrad_sk_resident_blocks<4, false>())
          : R == 2 ? (io_bf16 ? wgrad_sk_resident_blocks<2, true>() : wgrad_sk_resident_blocks<2, false>())
                   : (io_bf16 ? wgrad_sk_resident_blocks<1, true>() : wgrad_sk_resident_blocks<1, false>());
    const int Ut = (n_rows + 63) / 64;
    SkGeom g;
    // regions only when each has a few units at least; 4 instead of 8 for 128 x 128 layers (4 tiles per offset): with 8 the
    // line tiles get shorter than a run and every workgroup flushes twice (measured 93.9 vs 103.8 us), while a quarter
    // of the rows of the 24 k-row stage (3.1 MB in bf16) still fits the 4 MiB L2
    g.NR = Ut >= 8 * kSkRegions ? (GI * GJ == 4 ? kSkRegions / 2 : kSkRegions) : 1;
    g.Ur = (Ut + g.NR - 1) / g.NR;
    g.M = GI * GJ * g.Ur;
    const long long units = (long long)g.NR * KV * g.M;
    if (units < P) P = (int)units;  // tiny layers: one unit per workgroup at most
    // per-offset pair counts (weights of the decomposition) behind the slabs
    int *counts = (int *)(partial + ((size_t)P + (size_t)g.NR * KV * GI * GJ) * 4096);
    hipLaunchKernelGGL(wgrad_offset_counts_kernel, dim3(KV * kCountSlices), dim3(256), 0, stream, pairs, ld, n_rows, counts);
    prof_begin(BFHIP_OP_SPCONV_WGRAD_MAIN, stream, &ps);
#define BFHIP_WG_LAUNCH(RR, IO)                                                                                          \
  hipLaunchKernelGGL((spconv_wgrad64p_kernel<RR, IO>), dim3(P), dim3(256), 0, stream, in, Cin, dout, Cout, pairs, ld, KV, \
                     n_rows, g, GI, GJ, perm, counts, partial)
    if (R == 4) { if (io_bf16) BFHIP_WG_LAUNCH(4, true); else BFHIP_WG_LAUNCH(4, false); }
    else if (R == 2) { if (io_bf16) BFHIP_WG_LAUNCH(2, true); else BFHIP_WG_LAUNCH(2, false); }
    else { if (io_bf16) BFHIP_WG_LAUNCH(1, true); else BFHIP_WG_LAUNCH(1, false); }
#undef BFHIP_WG_LAUNCH
    prof_end(&ps);  // the events bracket the dominant kernel only, so their average matches rocprof's for that kernel
    hipLaunchKernelGGL(wgrad_reduce_sk_kernel, dim3(ceil_div((long long)KV * Cin * Cout, 64)), dim3(256), 0, stream, partial, KV, Cin, Cout, GI, GJ, g, P, counts, dW);
    prof_end(&ps_op);
    return check_launch("spconv_wgrad");
  }
  // channel counts that are not multiples of 4 (the 5-channel input layer) or more than 64 offsets: scalar-load kernel
  GI = (Cin + 31) / 32;
  const long long waves = (long long)KV * S * GI * GJ;
  prof_begin(BFHIP_OP_SPCONV_WGRAD_MAIN, stream, &ps);
  hipLaunchKernelGGL((spconv_wgrad_kernel<2, 4>), dim3(ceil_div(waves * 64, 256)), dim3(256), 0, stream, (const float *)in, Cin, (const float *)dout, Cout,
                     pairs, ld, KV, n_rows, S, GI, GJ, partial);
  prof_end(&ps);
  long long total = (long long)KV * Cin * Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, partial, S, KV, Cin, Cout, dW);
  prof_end(&ps_op);
  return check_launch("spconv_wgrad");
}

BFHIP_EXPORT int bfhip_sparse_to_bev(const float *feats, const int32_t *indices, int N, int C, int B, int X,
                                     int Y, int Z, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(N >= 0 && C > 0 && B > 0 && X > 0 && Y > 0 && Z > 0 && out, "sparse_to_bev: bad arguments");
  if (hipMemsetAsync(out, 0, (size_t)B * C * Z * X * Y * sizeof(float), stream) != hipSuccess) return check_launch("sparse_to_bev memset");
  if (N == 0) return BFHIP_OK;
  BFHIP_REQUIRE(feats && indices && ((uintptr_t)indices % 16) == 0, "sparse_to_bev: null/unaligned pointer");
  hipLaunchKernelGGL(sparse_to_bev_kernel, dim3(ceil_div((long long)N * C, 256)), dim3(256), 0, stream, feats,
                     (const int4 *)indices, N, C, X, Y, Z, out);
  return check_launch("sparse_to_bev");
}

BFHIP_EXPORT int bfhip_bev_to_sparse(const float *grad_out, const int32_t *indices, int N, int C, int B, int X,
                                     int Y, int Z, float *grad_feats, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)B;
  BFHIP_REQUIRE(N >= 0 && C > 0 && X > 0 && Y > 0 && Z > 0, "bev_to_sparse: bad arguments");
  if (N == 0) return BFHIP_OK;
  BFHIP_REQUIRE(grad_out && grad_feats && indices && ((uintptr_t)indices % 16) == 0, "bev_to_sparse: null/unaligned pointer");
  hipLaunchKernelGGL(bev_to_sparse_kernel, dim3(ceil_div((long long)N * C, 256)), dim3(256), 0, stream, grad_out,
                     (const int4 *)indices, N, C, X, Y, Z, grad_feats);
  return check_launch("bev_to_sparse");
}

BFHIP_EXPORT int bfhip_sparse_to_bev_nhwc(const float *feats, const int32_t *indices, int N, int C, int B, int X,
                                          int Y, int Z, int dtype, void *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(N >= 0 && C > 0 && B > 0 && X > 0 && Y > 0 && Z > 0 && out && (dtype == 0 || dtype == 1),
                "sparse_to_bev_nhwc: bad arguments");
  const size_t es = dtype == 1 ? 2 : 4;
  if (hipMemsetAsync(out, 0, (size_t)B * C * Z * X * Y * es, stream) != hipSuccess) return check_launch("sparse_to_bev_nhwc memset");
  if (N == 0) return BFHIP_OK;
  BFHIP_REQUIRE(feats && indices && ((uintptr_t)indices % 16) == 0, "sparse_to_bev_nhwc: null/unaligned pointer");
  dim3 grid(ceil_div((long long)N * C, 256));
  if (dtype == 1)
    hipLaunchKernelGGL(sparse_to_bev_nhwc_kernel<unsigned short>, grid, dim3(256), 0, stream, feats, (const int4 *)indices,
                       N, C, X, Y, Z, (unsigned short *)out);
  else
    hipLaunchKernelGGL(sparse_to_bev_nhwc_kernel<float>, grid, dim3(256), 0, stream, feats, (const int4 *)indices, N, C, X,
                       Y, Z, (float *)out);
  return check_launch("sparse_to_bev_nhwc");
}

BFHIP_EXPORT int bfhip_bev_nhwc_to_sparse(const void *grad_out, long long stride_b, long long stride_x,
                                          long long stride_y, int dtype, const int32_t *indices, int N, int C,
                                          int Z, float *grad_feats, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BFHIP_REQUIRE(N >= 0 && C > 0 && Z > 0 && (dtype == 0 || dtype == 1) && stride_y >= (long long)C * Z,
                "bev_nhwc_to_sparse: bad arguments");
  if (N == 0) return BFHIP_OK;
  BFHIP_REQUIRE(grad_out && grad_feats && indices && ((uintptr_t)indices % 16) == 0, "bev_nhwc_to_sparse: null/unaligned pointer");
  dim3 grid(ceil_div((long long)N * C, 256));
  if (dtype == 1)
    hipLaunchKernelGGL(bev_nhwc_to_sparse_kernel<unsigned short>, grid, dim3(256), 0, stream, (const unsigned short *)grad_out,
                       stride_b, stride_x, stride_y, (const int4 *)indices, N, C, Z, grad_feats);
  else
    hipLaunchKernelGGL(bev_nhwc_to_sparse_kernel<float>, grid, dim3(256), 0, stream, (const float *)grad_out, stride_b,
                       stride_x, stride_y, (const int4 *)indices, N, C, Z, grad_feats);
  return check_launch("bev_nhwc_to_sparse");
}
