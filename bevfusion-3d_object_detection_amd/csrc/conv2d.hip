// conv2d.hip -- dense 2-D convolution on channels-last (NHWC) bf16 activations as an implicit GEMM on the gfx950 matrix
// cores: forward, data gradient and weight gradient, with BatchNorm statistics accumulated in the forward epilogue.
//
// Layers served (SURVEY 8 a-8 ... a-11): ConvFuser 336->256 3x3 (BF/bevfusion_head.py:26-38), SECOND's twelve 3x3 convs and
// SECONDFPN's 1x1 conv (mmdet3d/models/backbones/second.py:27-95, necks/second_fpn.py:30-94), the head's shared_conv
// (BF/bevfusion_head.py:95-102), depthnet / downsample of the view transform (BF/depth_lss.py:592-620) and the LSS-FPN
// lateral / fpn convs (BF/bevfusion_necks.py:50-72).  Any kernel size / stride / padding / dilation, groups = 1,
// channel counts that are multiples of 8.
//
// Formulation.  y[m][co] = sum_k A[m][k] * Wt[co][k], m = (n, oh, ow), k = (kh, kw, ci): the weight of a channels-last conv,
// [Cout][KH][KW][Cin], IS the row-major B^T operand; A is never materialised -- each 16-byte piece (8 channels of one tap
// of one pixel) is fetched straight from the activation into LDS by `global_load_lds_dwordx4` with a per-lane source
// address (padding / tails read a zero page).  The data gradient is the same kernel in "transposed" mode (rows = input
// pixels, gathered tensor = dy, oh = (ih + pad - kh*dil) / stride when divisible) over the [Cin][KH][KW][Cout] transpose
// of the weight.  The weight gradient dW[co][k] = sum_m dy[m][co] * A[m][k] reduces over pixels: both operands are staged
// pixel-major and read with the transposing LDS read `ds_read_b64_tr_b16`, the pixel range is split over workgroups and
// the fp32 partial slabs are summed in a fixed order.
//
// Tile: 128 rows x (64 | 128) columns per 256-thread workgroup, K step 64 (bf16), 4 waves as 2 x 2, each wave a
// 64 x (32 | 64) block of `v_mfma_f32_32x32x16_bf16` accumulators; two LDS buffers per operand, the next step's loads are
// issued before the current step's MFMAs (one barrier per step).  LDS images are written linearly by the DMA; the bank
// swizzle lives in the SOURCE chunk a lane fetches and in the read address (same involution on both sides).
#include "common.h"
#include <stddef.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace bfhip {
namespace {

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Source of every padded / out-of-range / hole piece.  64 KB, and a wave reads from ITS OWN 64-byte line of it (zero_src()):
// with a single shared line every hole of every workgroup was a request to the same L2 channel -- at ~1 request per clock
// that one channel set the pace of the sparse kernels, where half of the pieces are holes (128 -> 128 SubM layer: 72 us with
// one line, whatever the prefetch depth).
constexpr int kZeroLines = 1024;
__device__ __attribute__((aligned(4096))) unsigned g_zero_page[kZeroLines * 16];

__device__ __forceinline__ const unsigned short *zero_src() {
  const unsigned line = (blockIdx.x * 8u + (threadIdx.x >> 6)) & (unsigned)(kZeroLines - 1);
  return (const unsigned short *)(g_zero_page + line * 16u);
}

struct ConvGeom {
  // gathered tensor [N, H, W, C] (pixel pitch ldx elements); GEMM rows = pixels of an [N, OH, OW] grid
  int N, H, W, C, ldx;
  int OH, OW;
  int KH, KW, stride, pad, dil;
  int transposed;  // 0: src = row * stride - pad + k * dil     1: t = row + pad - k * dil, src = t / stride if divisible
  int sshift, smask;  // transposed mode: stride = 1 << sshift, smask = stride - 1
  int nq;          // KH * KW * C / 8: number of 16-byte pieces along K
  long long M;     // N * OH * OW
  int Kout;        // GEMM columns (output channels of this GEMM)
  int ldw;         // weight row pitch in elements (= KH * KW * C)
  int ldy;         // output pixel pitch in elements
  // parity-class data gradient (MODE 2): class c owns row tiles [cls[c].tile0, cls[c + 1].tile0); its rows are the pixels
  // (h0 + i * stride, w0 + j * stride), i < Hc, j < Wc, of every image, and only the taps kh = kh0 + a * stride (a < nkh),
  // kw = kw0 + b * stride (b < nkw) reach them (none: nkh * nkw = 0, the class's gradient is zero)
  struct ParityClass { int h0, w0, Hc, Wc, kh0, kw0, nkh, nkw, tile0; } cls[17];
  int ncls;
};

__device__ __forceinline__ unsigned rne_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ void glds16(const void *src, void *lds_dst) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                   (void __attribute__((address_space(3))) *)lds_dst, 16, 0, 0);
}

// tap table: piece q -> (kh * dil) << 24 | (kw * dil) << 16 | ci
__device__ __forceinline__ void build_tap_table(unsigned *taps, const ConvGeom &g) {
  for (int q = threadIdx.x; q < g.nq; q += blockDim.x) {
    int k = q * 8;
    int tap = k / g.C, ci = k - tap * g.C;
    int kh = tap / g.KW, kw = tap - kh * g.KW;
    taps[q] = ((unsigned)(kh * g.dil) << 24) | ((unsigned)(kw * g.dil) << 16) | (unsigned)ci;
  }
}

// source address of piece (dh, dw, ci) for the row whose bases are (nb, hb, wb); out of range -> zero page.
// Kept short on purpose: it runs once per 16-byte piece (4 + NB times per wave and K step) beside 16 MFMAs -- no integer
// division (transposed strides are powers of two: shift + mask), 32-bit element offsets (tensors < 2^31 elements).
template <bool TR>
__device__ __forceinline__ const bf16_t *piece_src(const bf16_t *x, const ConvGeom &g, bool row_ok, int nb, int hb, int wb,
                                                   unsigned info, const bf16_t *zsrc) {
  const int dh = info >> 24, dw = (info >> 16) & 0xff, ci = info & 0xffff;
  int ih, iw;
  bool ok = row_ok;
  if (!TR) {
    ih = hb + dh;
    iw = wb + dw;
  } else {
    const int th = hb - dh, tw = wb - dw;
    ok = ok && ((th | tw) >= 0) && (((th | tw) & g.smask) == 0);
    ih = th >> g.sshift;
    iw = tw >> g.sshift;
  }
  ok = ok && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
  const unsigned off = (unsigned)(nb + ih * g.W + iw) * (unsigned)g.ldx + (unsigned)ci;
  return ok ? x + off : zsrc;
}

// ---- epilogue of the implicit-GEMM kernel: (optional) BatchNorm statistics of the raw accumulators, then
// accumulators -> LDS [BM][BN] -> 16-byte coalesced stores.  M rows, Kout columns, row pitch ldy.
// RowMap: GEMM row -> output pixel index (identity except for the parity-class data gradient)
struct RowIdentity {
  __device__ __forceinline__ long long operator()(long long m) const { return m; }
};
struct RowParityClass {
  int Hc, Wc, H, W, h0, w0, stride;
  __device__ __forceinline__ long long operator()(long long m) const {
    // 32-bit arithmetic: every entry point requires N * H * W * pitch < 2^31 (a 64-bit division is ~5x the instructions, and
    // this runs once per 16-byte store)
    const unsigned mu = (unsigned)m, hw = (unsigned)(Hc * Wc);
    const unsigned n = mu / hw, rem = mu - n * hw;
    const unsigned i = rem / (unsigned)Wc, j = rem - i * (unsigned)Wc;
    return (long long)((n * (unsigned)H + h0 + i * stride) * (unsigned)W + w0 + j * stride);
  }
};

// A second gradient path into the tensor the epilogue writes (residual connections): bf16, dense (pixel pitch `ld`), on the
// output's own pixel grid (sub == 1) or on the grid of its even pixels [N, ceil(H/2), ceil(W/2)] (sub == 2: the gradient of a
// stride-2 1x1 shortcut, which only reaches the pixels with even h and w)
struct Addend {
  const bf16_t *p;
  int sub, ld, H, W;
};

template <int WGM, int WGN, int MI, int NI, bool OUT_F32, typename RowMap = RowIdentity>
__device__ __forceinline__ void igemm_epilogue(f32x16 (&acc)[MI][NI], unsigned char *smem, int tm, long long m0, int n0,
                                               long long M, int Kout, int ldy, const float *__restrict__ bias,
                                               void *__restrict__ y, float *__restrict__ stat_partial,
                                               RowMap row_map = RowMap(), Addend add = Addend{nullptr, 0, 0, 0, 0}) {
  constexpr int NTHREADS = WGM * WGN * 64;
  constexpr int BN = WGN * NI * 32, BM = WGM * MI * 32;
  constexpr int PR = BM / 128;   // statistics partial rows of this tile (one per 128 pixels)
  constexpr int G = WGM / PR;    // wave rows that make up one partial row
  static_assert(BM % 128 == 0 && G * PR == WGM, "tile rows must be whole 128-row statistic groups");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w / WGN, wn = w % WGN;
  const int l31 = lane & 31, lh = lane >> 5;
  __syncthreads();  // every wave is done with the staging buffers: reuse them for the epilogue

  // ---- BatchNorm statistics of the raw accumulators (rows beyond M are exact zeros): per-column sum / sum of squares,
  //      one partial row per 128 rows of the tile
  if (stat_partial) {
    float *sred = (float *)(smem);  // [WGM][BN][2]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { float v = acc[mi][ni][r]; s += v; s2 += v * v; }
      s += __shfl_xor(s, 32);
      s2 += __shfl_xor(s2, 32);
      if (lh == 0) {
        int col = wn * (NI * 32) + ni * 32 + l31;
        sred[(wm * BN + col) * 2 + 0] = s;
        sred[(wm * BN + col) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    const int total_rows = (int)((M + 127) / 128);
    for (int e = tid; e < PR * BN; e += NTHREADS) {
      const int pr = e / BN, col = e - pr * BN;
      const int prow = tm * PR + pr;
      if (n0 + col < Kout && prow < total_rows) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const float *q = sred + ((pr * G + gi) * BN + col) * 2;
          t0 += q[0];
          t1 += q[1];
        }
        stat_partial[((size_t)prow * 2 + 0) * Kout + n0 + col] = t0;
        stat_partial[((size_t)prow * 2 + 1) * Kout + n0 + col] = t1;
      }
    }
    __syncthreads();
  }

  // ---- output: accumulators -> LDS [BM][BN] (row-major) -> 16-byte coalesced stores
  constexpr int ESZ = OUT_F32 ? 4 : 2;
  constexpr int ROWB = BN * ESZ;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = wn * (NI * 32) + ni * 32 + l31;
    const float bv = (bias && n0 + col < Kout) ? bias[n0 + col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (MI * 32) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = acc[mi][ni][r] + bv;
        if (OUT_F32) *(float *)(smem + row * ROWB + col * 4) = v;
        else *(bf16_t *)(smem + row * ROWB + col * 2) = (bf16_t)rne_bf16(v);
      }
  }
  __syncthreads();
  constexpr int CPR = ROWB / 16;  // 16-byte pieces per row
  constexpr int EPC = 16 / ESZ;   // elements per piece
  for (int idx = tid; idx < BM * CPR; idx += NTHREADS) {
    const int row = idx / CPR, c = idx - row * CPR;
    const long long m = m0 + row;
    const int col = n0 + c * EPC;
    if (m >= M || col >= Kout) continue;
    unsigned char *dst = (unsigned char *)y + ((size_t)row_map(m) * ldy + col) * ESZ;
    const unsigned char *src = smem + row * ROWB + c * 16;
    const bf16_t *rs = nullptr;
    if (!OUT_F32 && add.p) {
      if (add.sub == 2) {  // addend on the even-pixel grid: rows with odd h or w receive nothing from it
        const unsigned mu = (unsigned)row_map(m), hw = (unsigned)(add.H * add.W);
        const unsigned n = mu / hw, rem = mu - n * hw;
        const unsigned h = rem / (unsigned)add.W, w = rem - h * (unsigned)add.W;
        if (((h | w) & 1u) == 0)
          rs = add.p + ((size_t)((n * (unsigned)((add.H + 1) >> 1) + (h >> 1)) * (unsigned)((add.W + 1) >> 1) + (w >> 1)) * add.ld + col);
      } else {
        rs = add.p + ((size_t)row_map(m) * add.ld + col);
      }
    }
    if (rs) {
      // bf16 output + bf16 addend (the other gradient path into the same tensor): widened, added to the already rounded
      // result in fp32, rounded once more -- what a separate bf16 add kernel computes
      bf16_t *d16 = (bf16_t *)dst;
      const bf16_t *s16 = (const bf16_t *)src;
      if (col + EPC <= Kout && ((((uintptr_t)dst) | ((uintptr_t)rs)) & 15) == 0) {
        const uint4 a4 = *(const uint4 *)src, b4 = *(const uint4 *)rs;
        const unsigned a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = __uint_as_float(a[q] << 16) + __uint_as_float(b[q] << 16);
          const float hi = __uint_as_float(a[q] & 0xffff0000u) + __uint_as_float(b[q] & 0xffff0000u);
          o[q] = rne_bf16(lo) | (rne_bf16(hi) << 16);
        }
        *(uint4 *)dst = make_uint4(o[0], o[1], o[2], o[3]);
        continue;
      }
      for (int e = 0; e < EPC && col + e < Kout; e += 2) {
        if (col + e + 1 < Kout && ((((uintptr_t)(d16 + e)) | ((uintptr_t)(rs + e))) & 3) == 0) {
          const unsigned a = *(const unsigned *)(s16 + e), b = *(const unsigned *)(rs + e);
          const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
          const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
          *(unsigned *)(d16 + e) = rne_bf16(lo) | (rne_bf16(hi) << 16);
        } else {
          for (int q = e; q < e + 2 && col + q < Kout; ++q)
            d16[q] = (bf16_t)rne_bf16(__uint_as_float((unsigned)s16[q] << 16) + __uint_as_float((unsigned)rs[q] << 16));
        }
      }
      continue;
    }
    if (col + EPC <= Kout && (((uintptr_t)dst) & 15) == 0) *(uint4 *)dst = *(const uint4 *)src;
    else
      for (int e = 0; e < EPC && col + e < Kout; ++e) {
        if (OUT_F32) ((float *)dst)[e] = ((const float *)src)[e];
        else ((bf16_t *)dst)[e] = ((const bf16_t *)src)[e];
      }
  }
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
// Tile BM = WGM*MI*32 rows x BN = WGN*NI*32 columns, WGM x WGN waves with (MI*32) x (NI*32) wave tiles of 32x32x16 MFMAs, a
// ring of STAGES LDS stages of one K step (64 bf16) each.  Rows are 128 bytes (8 pieces); piece c of row r sits at position
// c ^ ((r >> 1) & 7): the 16 rows a ds_read_b128 lane group touches land on 16 distinct 16-byte slots.
//   <2, 2, 2, NI, 2>: 128 x (64 | 128) tiles, 256 threads, two stages, two workgroups per CU (small problems, narrow outputs)
//   <2, 4, 4, 2, 2> : 256 x 256 tiles, 512 threads (8 waves as 2 x 4, wave tile 128 x 64), two stages of 64 KB: half the
//                     operand bytes per flop of the 128 x 128 tile (the L2 -> LDS fill rate of a CU, ~40-70 GB/s, is what the
//                     small tile runs into) and 3/4 of its LDS fragment reads per MFMA; for wide outputs with enough tiles
//   <4, 2, 2, 2, 3> : 256 x 128 tiles with 64 x 64 wave tiles and a three-stage ring -- measured, no gain (BFHIP_CONV_BIG_TILES)
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: forward gather; 1: data gradient (transposed gather over all taps, rows = all input pixels); 2: data gradient of a
// strided convolution, one parity class of input pixels per launch: a pixel (ih, iw) is reached only by the taps with
// kh = (ih + pad) mod stride (mod stride), so the class walks KH*KW / stride^2 of the taps instead of meeting holes at the rest
//   <2, 2, 2, NI, 1>: ONE stage, at most 128 registers, four workgroups per CU (the pointwise kernel's recipe with the tap
//                     gather): nothing overlaps inside a workgroup, residency hides the latency -- for the small maps of the
//                     ResNet-50 trunk, where tiles are few and K loops short (BFHIP_CONV_SINGLE_STAGE)
template <int WGM, int WGN, int MI, int NI, int STAGES, bool OUT_F32, int MODE>
__global__ __launch_bounds__(WGM * WGN * 64, (STAGES == 1 ? 4 : (WGM * WGN > 4 ? 1 : 2))) void conv_igemm_kernel(
    const bf16_t *__restrict__ x, const bf16_t *__restrict__ wt, const float *__restrict__ bias, void *__restrict__ y,
    float *__restrict__ stat_partial, ConvGeom g, int tiles_m, int tiles_n) {
  constexpr bool TR = MODE != 0;
  constexpr int NWAVES = WGM * WGN, NTHREADS = NWAVES * 64;
  constexpr int BN = WGN * NI * 32, BM = WGM * MI * 32, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, S_BYTES = A_BYTES + B_BYTES;
  constexpr int NA = (BM / 8) / NWAVES;  // A DMA instructions per wave and stage (8 rows each)
  constexpr int NB = (BN / 8) / NWAVES;  // B DMA instructions per wave and stage
  constexpr int GL = NA + NB;            // DMA instructions per wave and stage
  static_assert(NA >= 1 && NA * 8 * NWAVES == BM && NB >= 1 && NB * 8 * NWAVES == BN, "tiles must split into whole DMA instructions");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned *taps = (unsigned *)(smem + STAGES * S_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long lb = xcd_chunked_block(blockIdx.x, (long long)tiles_m * tiles_n);
  const int tn = (int)(lb % tiles_n);
  int tm = (int)(lb / tiles_n);
  long long M = g.M;  // GEMM rows and K pieces of this workgroup's problem (MODE 2: of its parity class)
  int nq = g.nq;
  ConvGeom::ParityClass pc = {};
  if (MODE == 2) {
    int c = 0;
    while (c + 1 < g.ncls && tm >= g.cls[c + 1].tile0) ++c;
    pc = g.cls[c];
    tm -= pc.tile0;
    M = (long long)g.N * pc.Hc * pc.Wc;
    nq = pc.nkh * pc.nkw * (g.C >> 3);
  }
  const long long m0 = (long long)tm * BM;
  const int n0 = tn * BN;

  if (MODE == 2 && nq == 0) {
    // no tap reaches this parity class (e.g. three of the four classes of a 1x1 stride-2 layer): its gradient is zero -- plain
    // 16-byte zero stores, no staging, no accumulators
    constexpr int ESZ = OUT_F32 ? 4 : 2, EPC = 16 / ESZ, CPR = BN / EPC;
    const RowParityClass rm{pc.Hc, pc.Wc, g.OH, g.OW, pc.h0, pc.w0, g.stride};
    for (int idx = tid; idx < BM * CPR; idx += NTHREADS) {
      const int row = idx / CPR, c = idx - row * CPR;
      const long long m = m0 + row;
      const int col = n0 + c * EPC;
      if (m >= M || col >= g.Kout) continue;
      unsigned char *dst = (unsigned char *)y + ((size_t)rm(m) * g.ldy + col) * ESZ;
      if (col + EPC <= g.Kout && (((uintptr_t)dst) & 15) == 0) *(uint4 *)dst = make_uint4(0u, 0u, 0u, 0u);
      else
        for (int e = 0; e < EPC && col + e < g.Kout; ++e) {
          if (OUT_F32) ((float *)dst)[e] = 0.f;
          else ((bf16_t *)dst)[e] = 0;
        }
    }
    return;
  }

  for (int q = tid; q < nq; q += NTHREADS) {
    int k = q * 8;
    int tap = k / g.C, ci = k - tap * g.C;
    int kh, kw;
    if (MODE == 2) {  // the class's taps only (dilation 1)
      const int a = tap / pc.nkw;
      kh = pc.kh0 + a * g.stride;
      kw = pc.kw0 + (tap - a * pc.nkw) * g.stride;
    } else {
      kh = tap / g.KW;
      kw = tap - kh * g.KW;
    }
    taps[q] = ((unsigned)(kh * g.dil) << 24) | ((unsigned)(kw * g.dil) << 16) | (unsigned)ci;
  }

  // ---- per-lane staging state: NA A rows (one per DMA instruction) and NB B rows; every wave stages NA*8 A rows
  const int lrow = lane >> 3, lpos = lane & 7;
  int nb[NA], hb[NA], wb[NA];
  bool rok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    long long m = m0 + w * (NA * 8) + i * 8 + lrow;
    rok[i] = m < M;
    long long mm = rok[i] ? m : 0;
    int n, oh, ow;
    const unsigned mu = (unsigned)mm;  // rows < 2^31 (entry-point precondition): 32-bit divisions
    if (MODE == 2) {
      const unsigned hw = (unsigned)(pc.Hc * pc.Wc);
      const unsigned nn = mu / hw, rem = mu - nn * hw;
      const unsigned ci_ = rem / (unsigned)pc.Wc;
      n = (int)nn;
      oh = pc.h0 + (int)ci_ * g.stride;
      ow = pc.w0 + (int)(rem - ci_ * (unsigned)pc.Wc) * g.stride;
    } else {
      const unsigned hw = (unsigned)(g.OH * g.OW);
      const unsigned nn = mu / hw, rem = mu - nn * hw;
      n = (int)nn;
      oh = (int)(rem / (unsigned)g.OW);
      ow = (int)(rem - (unsigned)oh * (unsigned)g.OW);
    }
    nb[i] = n * g.H * g.W;
    hb[i] = TR ? oh + g.pad : oh * g.stride - g.pad;
    wb[i] = TR ? ow + g.pad : ow * g.stride - g.pad;
  }
  const bf16_t *wrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int co = n0 + w * (NB * 8) + i * 8 + lrow;
    wrow[i] = co < g.Kout ? wt + (size_t)co * g.ldw : nullptr;
  }
  __syncthreads();  // tap table ready

  const int nt = (nq + 7) >> 3;
  const bf16_t *zsrc = zero_src();
  auto stage = [&](int t, int buf) {  // exactly GL DMA instructions per wave (the counted waits rely on it)
    unsigned char *dA = smem + buf * S_BYTES + (w * (NA * 8)) * 128;
    unsigned char *dB = smem + buf * S_BYTES + A_BYTES + (w * (NB * 8)) * 128;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int swz = ((w * (NA * 8) + i * 8 + lrow) >> 1) & 7;
      const int q = t * 8 + (lpos ^ swz);
      const bf16_t *src = q < nq ? piece_src<TR>(x, g, rok[i], nb[i], hb[i], wb[i], taps[q], zsrc) : zsrc;
      glds16(src, dA + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int swz = ((w * (NB * 8) + i * 8 + lrow) >> 1) & 7;
      const int q = t * 8 + (lpos ^ swz);
      const bf16_t *src = zsrc;
      if (wrow[i] && q < nq) {
        if (MODE == 2) {  // the weight row holds all taps: this piece's tap is (dh, dw) of the table entry
          const unsigned info = taps[q];
          src = wrow[i] + ((size_t)((info >> 24) * g.KW + ((info >> 16) & 0xff)) * g.C + (info & 0xffff));
        } else {
          src = wrow[i] + (size_t)q * 8;
        }
      }
      glds16(src, dB + i * 1024);
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int wm = w / WGN, wn = w % WGN;
  const int l31 = lane & 31, lh = lane >> 5, rswz = (lane >> 1) & 7;  // ((row >> 1) & 7) of row = 32*j + l31
  const int aoff = (wm * (MI * 32) + l31) * 128, boff = A_BYTES + (wn * (NI * 32) + l31) * 128;

#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < nt) stage(s0, s0);
  int buf = 0, nbuf = STAGES - 1;  // stage holding step t / stage the next DMA goes to
  for (int t = 0; t < nt; ++t) {
    if (STAGES == 1) {
      if (t) __syncthreads();  // every wave has consumed step t - 1
      stage(t, 0);
    }
    // step t must have landed; the DMA of the (up to STAGES - 2) later steps stays in flight
    if (STAGES <= 2 || nt - 1 - t == 0) wait_vmcnt<0>();
    else if (STAGES == 3 || nt - 1 - t == 1) wait_vmcnt<GL>();
    else wait_vmcnt<2 * GL>();
    __builtin_amdgcn_s_barrier();  // all of step t is in LDS; every wave is done reading step t - 1
    const unsigned char *pA = smem + buf * S_BYTES + aoff, *pB = smem + buf * S_BYTES + boff;
    if (STAGES > 1 && t + STAGES - 1 < nt) stage(t + STAGES - 1, nbuf);  // before the MFMAs: issuing it after the first K quarter's
                                                           // MFMAs (address generation in their shadow) measured 5-10 % slower
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int pos = ((2 * ks + lh) ^ rswz) << 4;
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *(const bf16x8 *)(pA + mi * 32 * 128 + pos);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *(const bf16x8 *)(pB + ni * 32 * 128 + pos);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    buf = buf + 1 == STAGES ? 0 : buf + 1;
    nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
  }
  if (MODE == 2)
    igemm_epilogue<WGM, WGN, MI, NI, OUT_F32, RowParityClass>(acc, smem, tm, m0, n0, M, g.Kout, g.ldy, bias, y, stat_partial,
                                                              RowParityClass{pc.Hc, pc.Wc, g.OH, g.OW, pc.h0, pc.w0, g.stride});
  else
    igemm_epilogue<WGM, WGN, MI, NI, OUT_F32>(acc, smem, tm, m0, n0, M, g.Kout, g.ldy, bias, y, stat_partial);
}

// ------------------------------------------------------------------------------------------------ pointwise (1x1, stride 1)
// y[m][co] = sum_ci x[m][ci] * Wt[co][ci]: a plain GEMM over a tall activation matrix with a short K (64 ... 256 in the
// ResNet-50 trunk's bottlenecks, where M = 270 k rows): one to four K steps per tile, so a ring of stages has nothing to
// overlap inside a workgroup and the layer is bound by HBM, not by the matrix cores.  What hides the latency here is
// residency: ONE stage of one K step (32 KB for a 128 x 128 tile), at most 128 registers, four to five workgroups per CU
// at different points of load -> MFMA -> epilogue; no tap table, no divisions (row m IS pixel m).  Same tile, LDS image,
// swizzle and epilogue (BatchNorm statistics, LDS transpose, 16-byte stores) as conv_igemm_kernel<2, 2, 2, NI>.
// DIR (0 forward, 1 data gradient) only names the instantiation, so that a kernel trace tells the two uses apart.
template <int NI, bool OUT_F32, int DIR>
__global__ __launch_bounds__(256, 4) void conv_pw_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ wt,
                                                         const float *__restrict__ bias, void *__restrict__ y,
                                                         float *__restrict__ stat_partial, long long M, int C, int ldx, int Kout,
                                                         int ldw, int ldy, int tiles_m, int tiles_n, Addend add) {
  constexpr int WGN = 2, MI = 2;
  constexpr int BM = 128, BN = NI * 64;
  constexpr int A_BYTES = BM * 128;
  constexpr int NA = 4, NB = BN / 32;  // DMA instructions per wave and K step (8 rows of 128 bytes each)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long lb = xcd_chunked_block(blockIdx.x, (long long)tiles_m * tiles_n);
  const int tn = (int)(lb % tiles_n), tm = (int)(lb / tiles_n);
  const long long m0 = (long long)tm * BM;
  const int n0 = tn * BN;
  const int nq = C >> 3, nt = (nq + 7) >> 3;
  const int lrow = lane >> 3, lpos = lane & 7;
  const bf16_t *zsrc = zero_src();
  const bf16_t *arow[NA], *brow[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const long long m = m0 + w * 32 + i * 8 + lrow;
    arow[i] = m < M ? x + (unsigned)m * (unsigned)ldx : nullptr;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int co = n0 + w * (NB * 8) + i * 8 + lrow;
    brow[i] = co < Kout ? wt + (size_t)co * ldw : nullptr;
  }
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int wm = w / WGN, wn = w % WGN;
  const int l31 = lane & 31, lh = lane >> 5, rswz = (lane >> 1) & 7;
  const unsigned char *pA = smem + (wm * 64 + l31) * 128, *pB = smem + A_BYTES + (wn * (NI * 32) + l31) * 128;
  unsigned char *dA = smem + (w * 32) * 128, *dB = smem + A_BYTES + (w * (NB * 8)) * 128;
  for (int t = 0; t < nt; ++t) {
    if (t) __syncthreads();  // every wave has consumed the previous K step
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int q = t * 8 + (lpos ^ (((w * 32 + i * 8 + lrow) >> 1) & 7));
      glds16(arow[i] && q < nq ? arow[i] + q * 8 : zsrc, dA + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = t * 8 + (lpos ^ (((w * (NB * 8) + i * 8 + lrow) >> 1) & 7));
      glds16(brow[i] && q < nq ? brow[i] + q * 8 : zsrc, dB + i * 1024);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int pos = ((2 * ks + lh) ^ rswz) << 4;
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *(const bf16x8 *)(pA + mi * 32 * 128 + pos);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *(const bf16x8 *)(pB + ni * 32 * 128 + pos);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  }
  igemm_epilogue<2, WGN, MI, NI, OUT_F32>(acc, smem, tm, m0, n0, M, Kout, ldy, bias, y, stat_partial, RowIdentity(), add);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][k] = sum over pixels of dy[m][co] * A[m][k].  LDS tiles are pixel-major: dy [64][128 co], A [64][128 k]
// (256-byte rows; piece c of row r at position c ^ (((r & 3) << 2) | ((r >> 2) & 3)), conflict-free for the
// transposing read).  ds_read_b64_tr_b16 hands each lane 4 consecutive pixels of ONE column: the K-major fragment the
// 32x32x16 MFMA wants, for both operands.
struct WgradGeom {
  ConvGeom c;      // forward geometry of the conv (gathered tensor = x)
  int Cout, ldg;   // dy channels and pixel pitch
  int splits;      // pixel range split
  long long rows_per_split;  // multiple of 64
  int tiles_co, tiles_k;
};

__device__ __forceinline__ int tr_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// ---- shared by the dense and the sparse weight-gradient kernels: wave tile = 64 rows (wm) x 64 columns (wn) of dW;
// a 16-lane group reads a 4-pixel x 16-column block with the transposing LDS read
struct TrAddr { int g[2][2], x[2][2]; };  // [tile 0/1][half], for k-step 0; k-step ks adds ks * 16 rows (swizzle period 16)

__device__ __forceinline__ TrAddr tr_addresses(int lane, int wm, int wn) {
  const int grp = (lane >> 4) & 1, li = lane & 15, tq = li >> 2, tp = li & 3, lh = lane >> 5;
  TrAddr a;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int r = 8 * lh + 4 * half + tq;  // row inside a 16-pixel k-step
      const int cg = ((wm * 64 + j * 32) >> 3) + 2 * grp + (tp >> 1);
      const int cx = ((wn * 64 + j * 32) >> 3) + 2 * grp + (tp >> 1);
      a.g[j][half] = r * 256 + ((cg ^ tr_swz(r)) << 4) + 8 * (tp & 1);
      a.x[j][half] = r * 256 + ((cx ^ tr_swz(r)) << 4) + 8 * (tp & 1);
    }
  return a;
}

// LDS byte address of a pointer into the dynamic-LDS region
__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)p;
}

// Transposing LDS read as inline asm, NOT the builtin: behind a `global_load_lds` the compiler's wait-count pass puts an
// `s_waitcnt vmcnt(0)` in front of every `llvm.amdgcn.ds.read.tr16.b64` (it cannot tell the read from the DMA's destination),
// which drains the prefetched stages before the first read of each step -- DMA and MFMAs then never overlap inside a
// workgroup (round 2's kernels ran that way: 30 % MFMA-busy at any tile size or ring depth).  With asm reads the order is
// ours to keep: counted vmcnt + raw s_barrier before the reads, lgkmcnt(0) (tied to the destination registers, so that the
// MFMAs cannot be scheduled above it) before their use.
template <int OFF>
__device__ __forceinline__ short4_t lds_read_tr(unsigned addr) {
  short4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

__device__ __forceinline__ void lds_wait_all(short4_t (&a)[2][2][2], short4_t (&b)[2][2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0][0]), "+v"(a[0][0][1]), "+v"(a[0][1][0]), "+v"(a[0][1][1]), "+v"(a[1][0][0]), "+v"(a[1][0][1]),
                 "+v"(a[1][1][0]), "+v"(a[1][1][1]), "+v"(b[0][0][0]), "+v"(b[0][0][1]), "+v"(b[0][1][0]), "+v"(b[0][1][1]),
                 "+v"(b[1][0][0]), "+v"(b[1][0][1]), "+v"(b[1][1][0]), "+v"(b[1][1][1])
               :
               : "memory");
}

// one 64-pixel step: acc[i][j] += G^T(tile i) . X(tile j).  Reads of k-steps 2-3 are in flight under the MFMAs of k-steps 0-1.
__device__ __forceinline__ void tr_compute_step(const unsigned char *pG, const unsigned char *pX, const TrAddr &ad, f32x16 (&acc)[2][2]) {
  typedef __attribute__((ext_vector_type(8))) short short8_t;
  const unsigned aG = lds_addr(pG), aX = lds_addr(pX);
  short4_t g0[2][2][2], x0[2][2][2], g1[2][2][2], x1[2][2][2];  // [k-step of the pair][tile j][half]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      g0[0][j][h] = lds_read_tr<0>(aG + ad.g[j][h]);
      x0[0][j][h] = lds_read_tr<0>(aX + ad.x[j][h]);
      g0[1][j][h] = lds_read_tr<16 * 256>(aG + ad.g[j][h]);
      x0[1][j][h] = lds_read_tr<16 * 256>(aX + ad.x[j][h]);
    }
  lds_wait_all(g0, x0);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      g1[0][j][h] = lds_read_tr<32 * 256>(aG + ad.g[j][h]);
      x1[0][j][h] = lds_read_tr<32 * 256>(aX + ad.x[j][h]);
      g1[1][j][h] = lds_read_tr<48 * 256>(aG + ad.g[j][h]);
      x1[1][j][h] = lds_read_tr<48 * 256>(aX + ad.x[j][h]);
    }
  auto mfma_pair = [&](short4_t (&g)[2][2][2], short4_t (&x)[2][2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        short8_t av = {g[ks][j][0][0], g[ks][j][0][1], g[ks][j][0][2], g[ks][j][0][3], g[ks][j][1][0], g[ks][j][1][1], g[ks][j][1][2], g[ks][j][1][3]};
        short8_t bv = {x[ks][j][0][0], x[ks][j][0][1], x[ks][j][0][2], x[ks][j][0][3], x[ks][j][1][0], x[ks][j][1][1], x[ks][j][1][2], x[ks][j][1][3]};
        a[j] = __builtin_bit_cast(bf16x8, av);
        b[j] = __builtin_bit_cast(bf16x8, bv);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  mfma_pair(g0, x0);
  lds_wait_all(g1, x1);
  mfma_pair(g1, x1);
}

// partial slab [Cout][Ktot] (fp32) of one split: rows = co, lanes = k columns (contiguous)
__device__ __forceinline__ void tr_store_slab(float *out, int Cout, int Ktot, int co0, int q0, int lane, int wm, int wn,
                                              const f32x16 (&acc)[2][2]) {
  const int lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = q0 * 8 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < Cout && col < Ktot) out[(size_t)co * Ktot + col] = acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ dy,
                                                            float *__restrict__ slab, WgradGeom wg) {
  constexpr int BP = 64, T_BYTES = BP * 256;  // one tile: 64 pixels x 128 columns bf16
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char *sG = smem, *sX = smem + 2 * T_BYTES;
  unsigned *taps = (unsigned *)(smem + 4 * T_BYTES);
  const ConvGeom &g = wg.c;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tiles = wg.tiles_co * wg.tiles_k;
  const long long lb = xcd_chunked_block(blockIdx.x, (long long)tiles * wg.splits);
  const int split = (int)(lb / tiles), tile = (int)(lb - (long long)split * tiles);
  const int tco = tile / wg.tiles_k, tk = tile - tco * wg.tiles_k;
  const int co0 = tco * 128, q0 = tk * 16;  // first dy channel, first K piece of this tile
  build_tap_table(taps, g);

  const long long p_begin = (long long)split * wg.rows_per_split;
  long long p_end = p_begin + wg.rows_per_split;
  if (p_end > g.M) p_end = g.M;
  const int nsteps = p_end > p_begin ? (int)((p_end - p_begin + BP - 1) / BP) : 0;

  // staging: one DMA instruction = 4 rows x 256 B; wave w stages rows [16w, 16w + 16): instruction i -> row 16w + 4i + (lane >> 4)
  const int lrow = lane >> 4, lpos = lane & 15;
  int pn[4], poh[4], pow_[4];  // pixel coordinates of this lane's 4 rows (advanced by 64 pixels per step)
  long long pm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long m = p_begin + w * 16 + i * 4 + lrow;
    pm[i] = m;
    long long mm = m < g.M ? m : 0;
    int n = (int)(mm / ((long long)g.OH * g.OW));
    int rem = (int)(mm - (long long)n * g.OH * g.OW);
    pn[i] = n;
    poh[i] = rem / g.OW;
    pow_[i] = rem - poh[i] * g.OW;
  }
  __syncthreads();

  // a lane's pieces are fixed for the whole kernel: dy channel block / K piece (tap, ci) of row i
  int pdh[4], pdw[4], pcx[4];
  bool qok[4], cok[4];
  const bf16_t *gsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = w * 16 + i * 4 + lrow;
    const int c = lpos ^ tr_swz(r);
    const int co = co0 + c * 8, q = q0 + c;
    cok[i] = co < wg.Cout;
    qok[i] = q < g.nq;
    const unsigned info = qok[i] ? taps[q] : 0u;
    pdh[i] = info >> 24;
    pdw[i] = (info >> 16) & 0xff;
    pcx[i] = info & 0xffff;
    gsrc[i] = dy + ((size_t)pm[i] * wg.ldg + co);
  }
  const bf16_t *zsrc = zero_src();
  auto stage = [&](int buf) {
    unsigned char *dG = sG + buf * T_BYTES + (w * 16) * 256, *dX = sX + buf * T_BYTES + (w * 16) * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool rok = pm[i] < p_end;
      glds16((rok && cok[i]) ? gsrc[i] : zsrc, dG + i * 1024);
      const int ih = poh[i] * g.stride - g.pad + pdh[i], iw = pow_[i] * g.stride - g.pad + pdw[i];
      const bool ok = rok && qok[i] && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      const bf16_t *sx = ok ? x + ((size_t)((pn[i] * g.H + ih) * g.W + iw) * g.ldx + pcx[i]) : zsrc;
      glds16(sx, dX + i * 1024);
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pm[i] += BP;
      gsrc[i] += (size_t)BP * wg.ldg;
      pow_[i] += BP;
      while (pow_[i] >= g.OW) { pow_[i] -= g.OW; ++poh[i]; }
      while (poh[i] >= g.OH) { poh[i] -= g.OH; ++pn[i]; }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int wm = w >> 1, wn = w & 1;
  const TrAddr ad = tr_addresses(lane, wm, wn);

  if (nsteps > 0) stage(0);
  for (int t = 0; t < nsteps; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would add its own vmcnt(0) lgkmcnt(0) (harmless here, not in the ring below)
    if (t + 1 < nsteps) { advance(); stage(buf ^ 1); }
    const unsigned char *pG = sG + buf * T_BYTES, *pX = sX + buf * T_BYTES;
    tr_compute_step(pG, pX, ad, acc);
  }

  const int Ktot = g.nq * 8;
  tr_store_slab(slab + (size_t)split * wg.Cout * Ktot, wg.Cout, Ktot, co0, q0, lane, wm, wn, acc);
}

// Round 3: the same weight gradient on wider tiles with a three-stage ring.  The 128 x 128 kernel above runs two workgroups
// per CU with ONE 32 KB stage in flight each, issued only after the previous one has landed: ~36 KB in flight per CU on
// average, 44 GB/s per CU of L2 -> LDS fill (the guide's gather-into-LDS rate with 72 KB in flight is 66-73), MFMA 30 % busy.
// Here a workgroup owns (PG * 128) dy channels x (PX * 128) K columns -- PG + PX "panels" of 64 pixels x 128 columns per
// stage, each panel laid out exactly like the tiles above, (2 PG) x (2 PX) waves of 64 x 64 -- and keeps TWO stages in
// flight behind the one being consumed (wait = vmcnt(pieces of one stage), one barrier per step): <1, 2> and <2, 1> move
// 3/4 of the operand bytes per flop of the 128 x 128 tile with 96 KB continuously in flight per CU.
// `lb`: this workgroup's index among the tiles_co * tiles_k * splits workgroups of the layer (split-major), `smem`: the dynamic LDS
template <int PG, int PX, int STAGES>
__device__ __forceinline__ void wgrad_wide_body(const bf16_t *__restrict__ x, const bf16_t *__restrict__ dy,
                                                float *__restrict__ slab, const WgradGeom &wg, const long long lb,
                                                unsigned char *smem) {
  constexpr int W = 4 * PG * PX;            // waves
  constexpr int GPW = 16 / W;               // 4-row groups of a 64-pixel stage staged by one wave
  static_assert(GPW >= 1 && GPW * W == 16, "waves must divide the 16 row groups of a stage");
  constexpr int BP = 64, PANEL = BP * 256;  // one panel: 64 pixels x 128 columns bf16
  constexpr int SB = (PG + PX) * PANEL;     // bytes of one stage
  constexpr int PER_STAGE = GPW * (PG + PX);  // DMA instructions per wave and stage
  unsigned *taps = (unsigned *)(smem + STAGES * SB);
  const ConvGeom &g = wg.c;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index in an SGPR
  const int tiles = wg.tiles_co * wg.tiles_k;
  const int split = (int)(lb / tiles), tile = (int)(lb - (long long)split * tiles);
  const int tco = tile / wg.tiles_k, tk = tile - tco * wg.tiles_k;
  const int co0 = tco * (128 * PG), q0 = tk * (16 * PX);  // first dy channel, first K piece of this tile
  build_tap_table(taps, g);

  const long long p_begin = (long long)split * wg.rows_per_split;
  long long p_end = p_begin + wg.rows_per_split;
  if (p_end > g.M) p_end = g.M;
  const int nsteps = p_end > p_begin ? (int)((p_end - p_begin + BP - 1) / BP) : 0;

  // staging: one DMA instruction = 4 rows x 256 B of one panel; wave w stages row groups [GPW * w, GPW * (w + 1)) of EVERY panel.
  // Address generation is incremental and 32-bit (the host checks that both tensors have < 2^31 elements and OH * OW >= 64):
  // a piece's source is x + xb[row group] + xoff[slot] with xb = ((n*H + oh*stride - pad)*W + ow*stride - pad)*ldx moved by a
  // constant per step plus one correction per row / image wrap, and xoff = (dh*W + dw)*ldx + ci fixed for the whole kernel --
  // no multiply and no division inside the loop (round 2's form spent ~150 vector instructions per wave and step here, several
  // of them quarter-rate 64-bit multiplies, beside 16 MFMAs).
  const int lrow = lane >> 4, lpos = lane & 15;
  int rem[GPW], hb[GPW], wb[GPW], xb[GPW], gb[GPW];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const long long m = p_begin + (GPW * w + i) * 4 + lrow;
    rem[i] = (int)(p_end - m);
    const long long mm = m < g.M ? m : 0;
    const int n = (int)(mm / ((long long)g.OH * g.OW));
    const int r2 = (int)(mm - (long long)n * g.OH * g.OW);
    const int oh = r2 / g.OW, ow = r2 - oh * g.OW;
    hb[i] = oh * g.stride - g.pad;
    wb[i] = ow * g.stride - g.pad;
    xb[i] = ((n * g.H + hb[i]) * g.W + wb[i]) * g.ldx;
    gb[i] = (int)mm * wg.ldg;
  }
  const int q64 = BP / g.OW, r64 = BP - q64 * g.OW;
  const int adv_h = q64 * g.stride, adv_w = r64 * g.stride;
  const int adv_x = (adv_h * g.W + adv_w) * g.ldx;
  const int wlim = g.OW * g.stride - g.pad, hlim = g.OH * g.stride - g.pad;
  const int wrap_w = g.OW * g.stride, wrap_h = g.OH * g.stride;
  const int fix_w = (g.stride * g.W - wrap_w) * g.ldx;        // ow: OW -> 0, oh + 1
  const int fix_h = (g.H * g.W - wrap_h * g.W) * g.ldx;       // oh: OH -> 0, n + 1
  const int adv_g = BP * wg.ldg;
  __syncthreads();  // tap table ready

  // a lane's pieces are fixed for the whole kernel: dy channel block of G panel p / K piece (tap, ci) of X panel p, row group i
  int xoff[GPW][PX], xdh[GPW][PX], xdw[GPW][PX], gco[GPW][PG];
  bool qok[GPW][PX], cok[GPW][PG];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int r = (GPW * w + i) * 4 + lrow;
    const int c = lpos ^ tr_swz(r);
#pragma unroll
    for (int p = 0; p < PG; ++p) {
      gco[i][p] = co0 + p * 128 + c * 8;
      cok[i][p] = gco[i][p] < wg.Cout;
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      const int q = q0 + p * 16 + c;
      qok[i][p] = q < g.nq;
      const unsigned info = qok[i][p] ? taps[q] : 0u;
      xdh[i][p] = (int)(info >> 24);
      xdw[i][p] = (int)((info >> 16) & 0xff);
      xoff[i][p] = (xdh[i][p] * g.W + xdw[i][p]) * g.ldx + (int)(info & 0xffff);
    }
  }
  const bf16_t *zsrc = zero_src();
  auto stage = [&](int buf) {
    unsigned char *base = smem + buf * SB;
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      const bool rok = rem[i] > 0;
      const int rowoff = ((GPW * w + i) * 4) * 256;
#pragma unroll
      for (int p = 0; p < PG; ++p)
        glds16((rok && cok[i][p]) ? dy + (unsigned)(gb[i] + gco[i][p]) : zsrc, base + p * PANEL + rowoff);
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        const bool ok = rok && qok[i][p] && (unsigned)(hb[i] + xdh[i][p]) < (unsigned)g.H &&
                        (unsigned)(wb[i] + xdw[i][p]) < (unsigned)g.W;
        glds16(ok ? x + (unsigned)(xb[i] + xoff[i][p]) : zsrc, base + (PG + p) * PANEL + rowoff);
      }
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      rem[i] -= BP;
      gb[i] += adv_g;
      int dx = adv_x;
      wb[i] += adv_w;
      hb[i] += adv_h;
      if (wb[i] >= wlim) { wb[i] -= wrap_w; hb[i] += g.stride; dx += fix_w; }
      if (hb[i] >= hlim) { hb[i] -= wrap_h; dx += fix_h; }
      xb[i] += dx;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int wm = w / (2 * PX), wn = w % (2 * PX);
  const TrAddr ad = tr_addresses(lane, wm & 1, wn & 1);
  const int offG = (wm >> 1) * PANEL, offX = (PG + (wn >> 1)) * PANEL;

  // prologue: STAGES - 1 stages in flight
#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < nsteps) {
      if (s0 > 0) advance();
      stage(s0);
    }
  for (int t = 0; t < nsteps; ++t) {
    const int buf = t % STAGES;
    // stage t has landed when at most the younger stages' pieces are outstanding
    if (STAGES == 3 && t + 1 < nsteps) wait_vmcnt<PER_STAGE>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // raw barrier: __syncthreads() drains vmcnt to 0 and with it the stage just prefetched
    if (t + STAGES - 1 < nsteps) { advance(); stage((t + STAGES - 1) % STAGES); }
    const unsigned char *pS = smem + buf * SB;
    tr_compute_step(pS + offG, pS + offX, ad, acc);
  }

  const int Ktot = g.nq * 8;
  tr_store_slab(slab + (size_t)split * wg.Cout * Ktot, wg.Cout, Ktot, co0, q0, lane, wm, wn, acc);
}

template <int PG, int PX, int STAGES>
__global__ __launch_bounds__(PG * PX * 256, 1) void conv_wgrad_wide_kernel(const bf16_t *__restrict__ x,
                                                                           const bf16_t *__restrict__ dy,
                                                                           float *__restrict__ slab, WgradGeom wg) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const long long lb = xcd_chunked_block(blockIdx.x, (long long)wg.tiles_co * wg.tiles_k * wg.splits);
  wgrad_wide_body<PG, PX, STAGES>(x, dy, slab, wg, lb, smem);
}

// ---- Grouped weight gradients: the layers of a whole backward pass in ONE launch per tile shape.  dW of a layer is a leaf of the
// backward graph, so the host (conv2d.py) only collects (x, dy, dW) during the backward and launches the group when the pass ends.
// Launched one by one, each layer is cut into exactly one residency round (256 workgroups): 6-9 steps of 64 pixels per workgroup
// on the ~50 small layers of the ResNet trunk, where the ring's prologue, the 128 KB slab store of every workgroup and the tail of
// the launch cost more than the steps (28-45 us each at 170-310 TFLOP/s), and splits x dW fp32 slab bytes whatever the layer
// (2.2 GB written per step).  In a group every workgroup runs ~`target_steps` steps of its layer (default 96): slab bytes fall
// with the split count, and there is one tail per step instead of one per layer.
// XCD placement: hardware workgroup b runs on XCD b % 8.  The host cuts the group's workgroup list (layers in descending order
// of steps per workgroup, each layer split-major) into 8 consecutive chunks of equal total STEPS; XCD c works through chunk c in
// order, so the tiles of one split (which share their x / dy rows) meet in one L2 and the 8 XCDs finish together.
struct WgradItem {
  unsigned long long x, dy, dw;          // bf16 [N,H,W,ldx], bf16 [N,OH,OW,ldg], dW (fp32 or bf16) [Cout][KH][KW][Cin]
  unsigned long long slab_off;           // byte offset of this layer's splits x [Cout][Ktot] fp32 slabs in the group's workspace
  long long M, rows_per_split, total;    // pixels, pixels per split (multiple of 64), Cout * Ktot
  int N, H, W, C, ldx, OH, OW, KH, KW, stride, pad, dil, nq;
  int Cout, ldg, splits, tiles_co, tiles_k;
  int dw_bf16, shape;                    // shape: 1 = 128 co x 256 k, 2 = 256 co x 128 k
  int first_block, n_blocks;             // in the launch of its shape (layer-local index = logical index - first_block)
  int first_rblock, n_rblocks;           // in the reduce launch (one block = 1024 elements of dW)
};
struct WgradGroupHeader {                // first 256 bytes of the table image; the items follow
  int n_items, n_shape[3], first_item[3], blocks[3], grid[3], max_nq[3], rblocks, target_steps;
  int chunk_start[3][9];                 // logical block range of XCD c in the launch of shape s: [chunk_start[s][c], chunk_start[s][c + 1])
  unsigned long long slab_bytes;
};
static_assert(sizeof(WgradGroupHeader) <= 256, "group header must fit its 256-byte slot");

template <int PG, int PX, int STAGES>
__global__ __launch_bounds__(PG * PX * 256, 1) void conv_wgrad_group_kernel(const WgradItem *__restrict__ items, int n_items,
                                                                            const int *__restrict__ chunk_start,
                                                                            unsigned char *__restrict__ slab_base) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int lbg = chunk_start[xcd] + slot;
  if (lbg >= chunk_start[xcd + 1]) return;   // whole workgroup (uniform)
  int lo = 0, hi = n_items - 1;              // last item with first_block <= lbg
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_block <= lbg) lo = mid;
    else hi = mid - 1;
  }
  const WgradItem &it = items[lo];
  WgradGeom wg;
  ConvGeom &g = wg.c;
  g.N = it.N; g.H = it.H; g.W = it.W; g.C = it.C; g.ldx = it.ldx; g.OH = it.OH; g.OW = it.OW;
  g.KH = it.KH; g.KW = it.KW; g.stride = it.stride; g.pad = it.pad; g.dil = it.dil; g.nq = it.nq; g.M = it.M;
  wg.Cout = it.Cout; wg.ldg = it.ldg; wg.splits = it.splits; wg.rows_per_split = it.rows_per_split;
  wg.tiles_co = it.tiles_co; wg.tiles_k = it.tiles_k;
  wgrad_wide_body<PG, PX, STAGES>((const bf16_t *)it.x, (const bf16_t *)it.dy, (float *)(slab_base + it.slab_off), wg,
                                  (long long)(lbg - it.first_block), smem);
}

// ------------------------------------------------------------------------------------------------ sparse weight gradient
// dW[co][k][ci] = sum over output rows of dout[row][co] * in[pairs[k][row]][ci]  (SubMConv3d / SparseConv3d, spconv's
// (out, kD, kH, kW, in) weight layout): the dense kernel above with the rulebook as the gather -- column piece q of a tile is
// (offset k = q*8 / Cin, channels q*8 % Cin ..+8), its source row is pairs[k][row] (-1: no neighbour -> zero piece).  The
// pair index of the NEXT 64-row step is loaded while the current step computes, so the index -> row chain costs one
// round trip per step, not two.  bf16 features in, fp32 accumulate: 16x the matrix rate of the fp32-MFMA kernel in
// spconv.hip, which stays for fp32 features.
struct SpWgradGeom {
  int Cin, Cout, KV, ld, n_rows, nq;  // nq = KV * Cin / 8
  int splits, tiles_co, tiles_k;
  int rows_per_split;                  // multiple of 64
};

__global__ __launch_bounds__(256, 2) void spconv_wgrad_tr_kernel(const bf16_t *__restrict__ in, const bf16_t *__restrict__ dout,
                                                                 const int *__restrict__ pairs, float *__restrict__ slab,
                                                                 SpWgradGeom sg) {
  constexpr int BP = 64, T_BYTES = BP * 256;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char *sG = smem, *sX = smem + 2 * T_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tiles = sg.tiles_co * sg.tiles_k;
  const long long lb = xcd_chunked_block(blockIdx.x, (long long)tiles * sg.splits);
  const int split = (int)(lb / tiles), tile = (int)(lb - (long long)split * tiles);
  const int tco = tile / sg.tiles_k, tk = tile - tco * sg.tiles_k;
  const int co0 = tco * 128, q0 = tk * 16;
  const int p_begin = split * sg.rows_per_split;
  const int p_end = min(p_begin + sg.rows_per_split, sg.n_rows);
  const int nsteps = p_end > p_begin ? (p_end - p_begin + BP - 1) / BP : 0;

  const int lrow = lane >> 4, lpos = lane & 15;
  int row[4], pk[4], pci[4], pidx[4];  // this lane's 4 rows, the (offset, channel) of its piece of each, the prefetched pair
  bool qok[4], cok[4];
  int gco[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = w * 16 + i * 4 + lrow;
    const int c = lpos ^ tr_swz(r);
    const int q = q0 + c;
    row[i] = p_begin + r;
    gco[i] = co0 + c * 8;
    cok[i] = gco[i] < sg.Cout;
    qok[i] = q < sg.nq;
    const int k8 = (qok[i] ? q : 0) * 8;
    pk[i] = k8 / sg.Cin;
    pci[i] = k8 - pk[i] * sg.Cin;
    pidx[i] = (qok[i] && row[i] < p_end) ? pairs[(size_t)pk[i] * sg.ld + row[i]] : -1;
  }
  const bf16_t *zsrc = zero_src();
  auto stage = [&](int buf) {  // rows `row[]`, pairs `pidx[]` (already loaded)
    unsigned char *dG = sG + buf * T_BYTES + (w * 16) * 256, *dX = sX + buf * T_BYTES + (w * 16) * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool rok = row[i] < p_end;
      glds16((rok && cok[i]) ? dout + ((size_t)row[i] * sg.Cout + gco[i]) : zsrc, dG + i * 1024);
      glds16(pidx[i] >= 0 ? in + ((size_t)pidx[i] * sg.Cin + pci[i]) : zsrc, dX + i * 1024);
    }
  };
  auto advance = [&]() {  // next step's rows and their pair indices (global loads issued here, consumed by the next stage())
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      row[i] += BP;
      pidx[i] = (qok[i] && row[i] < p_end) ? pairs[(size_t)pk[i] * sg.ld + row[i]] : -1;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wm = w >> 1, wn = w & 1;
  const TrAddr ad = tr_addresses(lane, wm, wn);

  if (nsteps > 0) { stage(0); advance(); }
  for (int t = 0; t < nsteps; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // step t in LDS, step t + 1's pair indices in registers
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nsteps) { stage(buf ^ 1); advance(); }
    tr_compute_step(sG + buf * T_BYTES, sX + buf * T_BYTES, ad, acc);
  }
  const int Ktot = sg.nq * 8;
  tr_store_slab(slab + (size_t)split * sg.Cout * Ktot, sg.Cout, Ktot, co0, q0, lane, wm, wn, acc);
}

// dW = sum over splits (fixed order), written as fp32 or bf16.  The loads of 8 slabs are issued before their adds: one
// dependent round trip per 8 slabs instead of one per slab (18 slabs of the 128 -> 128 sparse layers: 31 -> ~8 us).
__device__ __forceinline__ void wgrad_reduce_body(const float *__restrict__ slab, int splits, long long total,
                                                  void *__restrict__ dw, int out_bf16, long long block) {
  long long i = (block * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  if (i + 4 <= total) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const float4 *)(slab + (size_t)(k + u) * total + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < splits; ++k) {
      float4 v = *(const float4 *)(slab + (size_t)k * total + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (out_bf16) {
      uint2 o;
      o.x = rne_bf16(s.x) | (rne_bf16(s.y) << 16);
      o.y = rne_bf16(s.z) | (rne_bf16(s.w) << 16);
      *(uint2 *)((bf16_t *)dw + i) = o;
    } else *(float4 *)((float *)dw + i) = s;
  } else {
    for (long long e = i; e < total; ++e) {
      float a = 0.f;
      for (int k = 0; k < splits; ++k) a += slab[(size_t)k * total + e];
      if (out_bf16) ((bf16_t *)dw)[e] = (bf16_t)rne_bf16(a);
      else ((float *)dw)[e] = a;
    }
  }
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ slab, int splits, long long total,
                                                                void *__restrict__ dw, int out_bf16) {
  wgrad_reduce_body(slab, splits, total, dw, out_bf16, blockIdx.x);
}

// every layer of a group in one launch: block -> layer by its first reduce block
__global__ __launch_bounds__(256) void conv_wgrad_group_reduce_kernel(const WgradItem *__restrict__ items, int n_items,
                                                                      const unsigned char *__restrict__ slab_base) {
  const int b = blockIdx.x;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_rblock <= b) lo = mid;
    else hi = mid - 1;
  }
  const WgradItem &it = items[lo];
  wgrad_reduce_body((const float *)(slab_base + it.slab_off), it.splits, it.total, (void *)it.dw, it.dw_bf16, b - it.first_rblock);
}

// Wt'[ci][kh][kw][co] = W[co][kh][kw][ci]  (the dgrad's B^T operand)
__global__ __launch_bounds__(256) void conv_weight_transpose_kernel(const bf16_t *__restrict__ w, bf16_t *__restrict__ wt,
                                                                    int Cout, int taps, int Cin) {
  __shared__ bf16_t tile[32][33];
  const int tap = blockIdx.z, c0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int co = o0 + r, ci = c0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * taps + tap) * Cin + ci] : (bf16_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int ci = c0 + r, co = o0 + tx;
    if (ci < Cin && co < Cout) wt[((size_t)ci * taps + tap) * Cout + co] = tile[tx][r];
  }
}

// The same for a table of weights in ONE launch (all layers of a model after the optimizer step: a training step otherwise
// pays one ~5 us transpose launch per data gradient, 69 per step).  Segment i owns blocks [blk0[i], blk0[i + 1]).
struct WtSeg {
  const void *src;   // [Cout][taps][Cin], bf16 or fp32 (src_f32)
  bf16_t *dst;       // [Cin][taps][Cout]
  int Cout, taps, Cin, src_f32;
  long long blk0;
};
__global__ __launch_bounds__(256) void conv_weight_transpose_batched_kernel(const WtSeg *__restrict__ segs, int nseg) {
  __shared__ bf16_t tile[32][33];
  const long long b = blockIdx.x;
  int lo = 0, hi = nseg - 1;  // largest segment with blk0 <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].blk0 <= b) lo = mid; else hi = mid - 1;
  }
  const WtSeg sg = segs[lo];
  const int nx = (sg.Cin + 31) >> 5, ny = (sg.Cout + 31) >> 5;
  int local = (int)(b - sg.blk0);
  const int bx = local % nx; local /= nx;
  const int by = local % ny;
  const int tap = local / ny;
  if (tap >= sg.taps) return;
  const int c0 = bx * 32, o0 = by * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = o0 + r, ci = c0 + tx;
    bf16_t v = 0;
    if (co < sg.Cout && ci < sg.Cin) {
      const size_t i = ((size_t)co * sg.taps + tap) * sg.Cin + ci;
      v = sg.src_f32 ? (bf16_t)rne_bf16(((const float *)sg.src)[i]) : ((const bf16_t *)sg.src)[i];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = c0 + r, co = o0 + tx;
    if (ci < sg.Cin && co < sg.Cout) sg.dst[((size_t)ci * sg.taps + tap) * sg.Cout + co] = tile[tx][r];
  }
}

// fp32 operand -> bf16 pair (hi = bf16(v), lo = bf16(v - hi)), written in the two layouts the three-product fp32 convolution
// consumes (conv2d.py: _Conv2dSplitFunction): channel blocks [P][3C] and batch blocks [3][P][C]; bit k of an order word says
// whether block k holds hi (0) or lo (1).  One thread = 8 channels of one pixel.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ src, long long P, int C, bf16_t *__restrict__ chan,
                                                           int order_chan, bf16_t *__restrict__ batch, int order_batch) {
  const int cv = C >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * cv) return;
  const long long p = t / cv;
  const int c = (int)(t - p * cv) << 3;
  const float4 a = *(const float4 *)(src + p * C + c), b = *(const float4 *)(src + p * C + c + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned hi[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hi[j] = rne_bf16(v[j]);
    lo[j] = rne_bf16(v[j] - __uint_as_float(hi[j] << 16));
  }
  const uint4 H4 = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
  const uint4 L4 = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (chan) *(uint4 *)(chan + p * (3LL * C) + (long long)k * C + c) = ((order_chan >> k) & 1) ? L4 : H4;
    if (batch) *(uint4 *)(batch + ((long long)k * P + p) * C + c) = ((order_batch >> k) & 1) ? L4 : H4;
  }
}

// K pieces (16-byte = 8-channel pieces of one tap) a kernel can hold a tap table for: 4 bytes per piece beside 64 KB of
// stages under the 80 KB dynamic-LDS attribute of the two-workgroups-per-CU tiles (the 256-wide tiles have 31 KB beside
// 128 KB, the wide weight-gradient tiles 16 KB beside 144 KB).  C is Cin for forward / weight gradient and Cout for the data
// gradient, so bfhip_conv2d_supported checks both (round 2 checked Cin only and allowed 8192 pieces: such calls passed
// `supported` and then failed at launch instead of falling back to the library).
constexpr int kMaxPieces = 3584;
bool geom_ok(int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil) {
  return N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C < 65536 && KH > 0 && KW > 0 && stride > 0 &&
         (stride & (stride - 1)) == 0 /* the data gradient shifts instead of dividing */ && pad >= 0 && dil > 0 &&
         (KH - 1) * dil < 256 && (KW - 1) * dil < 256 && (long long)KH * KW * C / 8 <= kMaxPieces;
}

size_t igemm_lds_bytes(int BM, int BN, int stages, int nq) { return (size_t)stages * (BM + BN) * 128 + (size_t)nq * 4; }


// ---- internal entry points for spconv.hip (declared in common.h)
static void sp_wgrad_plan(int n_rows, int Cout, int Ktot, SpWgradGeom &sg) {
  sg.tiles_co = ceil_div(Cout, 128);
  sg.tiles_k = ceil_div(Ktot, 128);
  const int tiles = sg.tiles_co * sg.tiles_k;
  const int steps = ceil_div(n_rows, 64);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
    (void)hipGetLastError();
    cus = 256;
  }
  const int slots = 2 * cus;                         // one residency round (see wgrad_plan)
  int want = tiles >= slots ? 1 : slots / tiles;
  if (want > steps / 4) want = steps / 4;
  if (want < 1) want = 1;
  const int per = ceil_div(steps, want);
  sg.splits = ceil_div(steps, per);
  sg.rows_per_split = per * 64;
}
}  // namespace

size_t spconv_wgrad_tr_workspace_bytes(int KV, int Cin, int Cout, int n_rows) {
  SpWgradGeom sg;
  sp_wgrad_plan(n_rows > 0 ? n_rows : 1, Cout, KV * Cin, sg);
  return align_up((size_t)sg.splits * Cout * KV * Cin * sizeof(float), 256);
}

bool spconv_wgrad_tr_supported(int KV, int Cin, int Cout) { return Cin % 8 == 0 && Cout % 8 == 0 && Cin >= 8 && KV >= 1; }

int spconv_wgrad_tr(const void *in, const void *dout, const int32_t *pairs, int ld, int KV, int n_rows, int Cin, int Cout,
                    float *dW, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  SpWgradGeom sg;
  sg.Cin = Cin; sg.Cout = Cout; sg.KV = KV; sg.ld = ld; sg.n_rows = n_rows; sg.nq = KV * Cin / 8;
  sp_wgrad_plan(n_rows, Cout, KV * Cin, sg);
  if (workspace_bytes < spconv_wgrad_tr_workspace_bytes(KV, Cin, Cout, n_rows)) { set_error("spconv_wgrad: workspace too small"); return BFHIP_E_WORKSPACE; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)spconv_wgrad_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set = true;
  }
  ProfScope ps_main;
  prof_begin(BFHIP_OP_SPCONV_WGRAD_MAIN, stream, &ps_main);
  hipLaunchKernelGGL(spconv_wgrad_tr_kernel, dim3((unsigned)(sg.tiles_co * sg.tiles_k * sg.splits)), dim3(256), (size_t)4 * 64 * 256, stream,
                     (const bf16_t *)in, (const bf16_t *)dout, pairs, (float *)workspace, sg);
  prof_end(&ps_main);
  const long long total = (long long)Cout * KV * Cin;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(ceil_div(total, 1024)), dim3(256), 0, stream, (const float *)workspace, sg.splits,
                     total, (void *)dW, 0);
  return BFHIP_OK;
}

}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_conv2d_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil) {
  return geom_ok(N, H, W, Cin, KH, KW, stride, pad, dil) && Cout > 0 && Cout % 8 == 0 && Cout < 65536 &&
                 (long long)KH * KW * Cout / 8 <= kMaxPieces
             ? 1
             : 0;
}

// rows of the BN-statistics partial buffer the forward writes: stat_partial f32[conv2d_stat_rows][2][Cout]
BFHIP_EXPORT int bfhip_conv2d_stat_rows(int N, int OH, int OW) { return ceil_div((long long)N * OH * OW, 128); }

static bool pointwise_geom(const ConvGeom &g) { return g.KH == 1 && g.KW == 1 && g.stride == 1 && g.pad == 0 && g.transposed != 2; }

// largest channel count of the gathered tensor the pointwise kernel takes (BFHIP_CONV_PW_MAXC; 0 = never)
static int pw_max_channels() {
  static const int v = [] { const char *e = getenv("BFHIP_CONV_PW_MAXC"); return e ? atoi(e) : 4096; }();
  return v;
}
static bool takes_pointwise(const ConvGeom &g) { return pointwise_geom(g) && g.C <= pw_max_channels(); }

static int launch_igemm(const void *x, const void *wt, const float *bias, void *y, float *stat_partial, ConvGeom g, int out_f32,
                        hipStream_t s, const char *what, Addend add = Addend{nullptr, 0, 0, 0, 0}) {
  // tile shapes (see conv_igemm_kernel): 0 = 128 x 64, 1 = 128 x 128, 2 = 256 x 256 (bf16 output, wide GEMMs with at least
  // ~1.5 tiles per CU), 3 = 256 x 128 with 64 x 64 wave tiles and three stages (experiment switch only)
  static const int force_big = getenv("BFHIP_CONV_BIG_TILES") ? 1 : 0;
  // BFHIP_CONV_TILE256: 0 = never, 1 = by the rule below, 2 = whenever the output is bf16 and wider than 128 (tests)
  static const int tile256 = [] { const char *e = getenv("BFHIP_CONV_TILE256"); return e ? atoi(e) : 1; }();
  static const int tile256_min = [] { const char *e = getenv("BFHIP_CONV_TILE256_MIN"); return e ? atoi(e) : 384; }();
  // 1x1, stride 1, no padding (forward, or the data gradient of such a layer: both are plain GEMMs over the pixel matrix)
  // with a short K: conv_pw_kernel.  BFHIP_CONV_PW_MAXC: largest channel count of the gathered tensor it takes (0 = never)
  const int pw_maxc = pw_max_channels();
  BFHIP_REQUIRE(!add.p || (pointwise_geom(g) && g.C <= pw_maxc && !out_f32),
                "%s: an addend is only fused into the pointwise kernel (1x1, stride 1, no padding, bf16 output)", what);
  if (pointwise_geom(g) && g.C <= pw_maxc) {
    const int ni = g.Kout > 64 ? 2 : 1, BN = ni * 64;
    const int tiles_m = ceil_div(g.M, 128), tiles_n = ceil_div(g.Kout, BN);
    const size_t stage = (size_t)(128 + BN) * 128, epi = (size_t)128 * BN * (out_f32 ? 4 : 2);
    const size_t lds = stage > epi ? stage : epi;
    dim3 grid((unsigned)((long long)tiles_m * tiles_n));
#define BFHIP_PW(NIV, F32, DIRV)                                                                                        \
  do {                                                                                                                 \
    static bool attr_set = false;                                                                                      \
    if (!attr_set) {                                                                                                   \
      (void)hipFuncSetAttribute((const void *)conv_pw_kernel<NIV, F32, DIRV>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_pw_kernel<NIV, F32, DIRV>), grid, dim3(256), lds, s, (const bf16_t *)x, (const bf16_t *)wt, bias, y, \
                       stat_partial, g.M, g.C, g.ldx, g.Kout, g.ldw, g.ldy, tiles_m, tiles_n, add);                   \
  } while (0)
#define BFHIP_PW2(NIV, F32) do { if (g.transposed) BFHIP_PW(NIV, F32, 1); else BFHIP_PW(NIV, F32, 0); } while (0)
    if (ni == 2) { if (out_f32) BFHIP_PW2(2, true); else BFHIP_PW2(2, false); }
    else { if (out_f32) BFHIP_PW2(1, true); else BFHIP_PW2(1, false); }
#undef BFHIP_PW2
#undef BFHIP_PW
    return check_launch(what);
  }
  int shape = g.Kout > 64 ? 1 : 0;
  const long long t256 = (long long)ceil_div(g.M, 256) * ceil_div(g.Kout, 256);
  // enough tiles for the 256 one-workgroup CUs, and at most 1/8 of the 256-wide column tiles wasted
  const bool fits256 = t256 >= tile256_min && (long long)ceil_div(g.Kout, 256) * 256 * 8 <= (long long)g.Kout * 9;
  if (tile256 && !out_f32 && g.Kout > 128 && (fits256 || tile256 == 2)) shape = 2;
  else if (shape == 1 && force_big && ceil_div(g.M, 256) * ceil_div(g.Kout, 128) >= 192) shape = 3;
  static const int quant = [] { const char *e = getenv("BFHIP_CONV_QUANT"); return e ? atoi(e) : 1; }();
  if (quant && shape == 1) {
    // 128 x 128 tiles run two workgroups per CU: a tile count just above a whole number of rounds leaves the last round almost
    // empty; 128 x 64 tiles (three per CU) quantise finer (depthnet / LSS-FPN 3x3 on the 24 x 32 x 88 maps, 1 056 tiles = 2.06
    // rounds: forward 0.167 -> 0.161 ms, backward 0.344 -> 0.318 ms)
    const long long t = (long long)ceil_div(g.M, 128) * ceil_div(g.Kout, 128);
    const double rounds = (double)t / 512.0;
    if (rounds > 1.0 && rounds < 4.0 && rounds - (long long)rounds < 0.15) shape = 0;
  }
  // less than half a residency round of 128 x 128 tiles (small maps with long K: ResNet layer4's 3x3 layers are 132 tiles of 72 K
  // steps): 128 x 64 tiles double the workgroups; BFHIP_CONV_SMALL_GRID=0 switches the rule off
  static const int small_grid = [] { const char *e = getenv("BFHIP_CONV_SMALL_GRID"); return e ? atoi(e) : 1; }();
  if (small_grid && shape == 1 && g.transposed != 2 && (long long)ceil_div(g.M, 128) * ceil_div(g.Kout, 128) < 256) shape = 0;
  // One stage + four workgroups per CU for the 128-row tiles (BFHIP_CONV_SINGLE_STAGE: 0 never, 1 by rule, 2 always).  Residency
  // hides the load latency when there are enough workgroups to fill it (>= 3 per CU) or the K loop is too short for a ring to reach
  // steady state (<= 18 steps; the parity classes of a strided data gradient: 1/4 ... 1/stride^2 of the taps each); few tiles
  // with a long K loop keep the two-stage ring.  Measured, same box (tools/resnet_conv_micro.py / conv_micro.py, us, two-stage ->
  // one stage): ResNet 3x3 64 ch fwd 48.5 -> 42.8, dgrad 51.6 -> 44.9; 128 ch 44.9 -> 39.5, 52.2 -> 43.8; stride-2 data gradients
  // 93.5 -> 72.7, 82.2 -> 69.9, 88.1 -> 74.9; SECOND 128 -> 128 fwd 63.6 -> 55.1; shared_conv 198.5 -> 184.9; downsample 80 -> 80
  // 151.8 -> 128.6; against that 256 ch on 16 x 44 maps (264 tiles, 36 steps) 47.6 -> 52.7 and 512 ch on 8 x 22 66.0 -> 84.4
  static const int single = [] { const char *e = getenv("BFHIP_CONV_SINGLE_STAGE"); return e ? atoi(e) : 1; }();
  const long long tiles128 = (long long)ceil_div(g.M, 128) * ceil_div(g.Kout, shape == 0 ? 64 : 128);
  const bool one_stage = shape <= 1 && single && (single == 2 || g.transposed == 2 || tiles128 >= 768 || (g.nq + 7) / 8 <= 18);
  const int BM = shape >= 2 ? 256 : 128, BN = shape == 2 ? 256 : (shape == 0 ? 64 : 128), stages = shape == 3 ? 3 : (one_stage ? 1 : 2);
  int tiles_m = ceil_div(g.M, BM);
  const int tiles_n = ceil_div(g.Kout, BN);
  if (g.transposed == 2) {  // row tiles class by class
    tiles_m = 0;
    for (int c = 0; c < g.ncls; ++c) {
      g.cls[c].tile0 = tiles_m;
      tiles_m += ceil_div((long long)g.N * g.cls[c].Hc * g.cls[c].Wc, BM);
    }
    g.cls[g.ncls].tile0 = tiles_m;
  }
  size_t lds = igemm_lds_bytes(BM, BN, stages, g.nq);
  if (stages == 1) {  // the epilogue stages the output tile in the same LDS: BM x BN elements
    const size_t epi = (size_t)BM * BN * (out_f32 ? 4 : 2);
    if (lds < epi) lds = epi;
  }
  dim3 grid((unsigned)((long long)tiles_m * tiles_n));
#define BFHIP_IG(WGMV, WGNV, MIV, NIV, ST, F32, TRV)                                                                    \
  do {                                                                                                                 \
    static bool attr_set = false;                                                                                      \
    if (!attr_set) {                                                                                                   \
      (void)hipFuncSetAttribute((const void *)conv_igemm_kernel<WGMV, WGNV, MIV, NIV, ST, F32, TRV>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (WGMV * WGNV > 4) ? 159 * 1024 : 80 * 1024); \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_igemm_kernel<WGMV, WGNV, MIV, NIV, ST, F32, TRV>), grid, dim3(WGMV * WGNV * 64), lds, s,   \
                       (const bf16_t *)x, (const bf16_t *)wt, bias, y, stat_partial, g, tiles_m, tiles_n);             \
  } while (0)
#define BFHIP_IG2(WGMV, WGNV, MIV, NIV, ST, F32) do { if (g.transposed == 2) BFHIP_IG(WGMV, WGNV, MIV, NIV, ST, F32, 2); else if (g.transposed) BFHIP_IG(WGMV, WGNV, MIV, NIV, ST, F32, 1); else BFHIP_IG(WGMV, WGNV, MIV, NIV, ST, F32, 0); } while (0)
  if (shape == 2) BFHIP_IG2(2, 4, 4, 2, 2, false);
  else if (shape == 3) { if (out_f32) BFHIP_IG2(4, 2, 2, 2, 3, true); else BFHIP_IG2(4, 2, 2, 2, 3, false); }
  else if (shape == 1 && one_stage) { if (out_f32) BFHIP_IG2(2, 2, 2, 2, 1, true); else BFHIP_IG2(2, 2, 2, 2, 1, false); }
  else if (shape == 1) { if (out_f32) BFHIP_IG2(2, 2, 2, 2, 2, true); else BFHIP_IG2(2, 2, 2, 2, 2, false); }
  else if (one_stage) { if (out_f32) BFHIP_IG2(2, 2, 2, 1, 1, true); else BFHIP_IG2(2, 2, 2, 1, 1, false); }
  else { if (out_f32) BFHIP_IG2(2, 2, 2, 1, 2, true); else BFHIP_IG2(2, 2, 2, 1, 2, false); }
#undef BFHIP_IG2
#undef BFHIP_IG
  return check_launch(what);
}

// y[N, OH, OW, Cout] (pixel pitch ldy) = conv(x[N, H, W, Cin] (pixel pitch ldx), w[Cout][KH][KW][Cin]) (+ bias); bf16 in,
// bf16 or fp32 out.  stat_partial (optional): f32[ceil(M / 128)][2][Cout] per-row-block column sums / sums of squares of the
// un-biased fp32 accumulators (the `partial` input of bfhip_bn2d_fwd_partials).
BFHIP_EXPORT int bfhip_conv2d_fwd(const void *x, int ldx, const void *w, const float *bias, void *y, int ldy, int N, int H, int W,
                                  int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int out_f32,
                                  float *stat_partial, void *stream_) {
  BFHIP_REQUIRE(bfhip_conv2d_supported(N, H, W, Cin, Cout, KH, KW, stride, pad, dil), "conv2d_fwd: unsupported geometry");
  BFHIP_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
  BFHIP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ldx % 8 == 0 && ldx >= Cin && ldy >= Cout,
                "conv2d_fwd: operands must be 16-byte aligned with pitches that are multiples of 8 elements");
  ConvGeom g = {};
  g.N = N; g.H = H; g.W = W; g.C = Cin; g.ldx = ldx;
  g.OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  g.OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  BFHIP_REQUIRE(g.OH > 0 && g.OW > 0, "conv2d_fwd: empty output");
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.dil = dil; g.transposed = 0; g.sshift = 0; g.smask = 0;
  g.nq = KH * KW * Cin / 8;
  g.M = (long long)N * g.OH * g.OW;
  g.Kout = Cout; g.ldw = KH * KW * Cin; g.ldy = ldy;
  BFHIP_REQUIRE((long long)N * H * W * ldx < (1LL << 31), "conv2d_fwd: tensors of 2^31 elements or more are not supported");
  ProfScope ps;
  prof_begin(takes_pointwise(g) ? BFHIP_OP_CONV2D_PW_FWD : BFHIP_OP_CONV2D_FWD, (hipStream_t)stream_, &ps);
  const int rc = launch_igemm(x, w, bias, y, stat_partial, g, out_f32, (hipStream_t)stream_, "conv2d_fwd");
  prof_end(&ps);
  return rc;
}

BFHIP_EXPORT size_t bfhip_conv2d_dgrad_workspace_bytes(int Cin, int Cout, int KH, int KW) {
  return align_up((size_t)Cin * KH * KW * Cout * 2, 256);
}

// dx[N, H, W, Cin] = conv_transpose(dy[N, OH, OW, Cout], w): the forward kernel in transposed-gather mode over the
// [Cin][KH][KW][Cout] transpose of the weight (built in `workspace`).
// w: the convolution's weight (transposed into `workspace` first) or, with w == nullptr, `workspace` IS the transposed weight
static int conv2d_dgrad_impl(const void *dy, int ldg, const void *w, void *dx, int ldx, int N, int H, int W, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, int dil, int out_f32, void *workspace, size_t workspace_bytes,
                             hipStream_t s, const void *addend = nullptr, int addend_stride = 1) {
  BFHIP_REQUIRE(bfhip_conv2d_supported(N, H, W, Cin, Cout, KH, KW, stride, pad, dil), "conv2d_dgrad: unsupported geometry");
  BFHIP_REQUIRE(dy && dx && workspace, "conv2d_dgrad: null pointer");
  BFHIP_REQUIRE(workspace_bytes >= bfhip_conv2d_dgrad_workspace_bytes(Cin, Cout, KH, KW), "conv2d_dgrad: workspace too small");
  BFHIP_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)workspace % 16) == 0 && ldg % 8 == 0 && ldg >= Cout && ldx >= Cin,
                "conv2d_dgrad: operands must be 16-byte aligned with pitches that are multiples of 8 elements");
  const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  ProfScope ps;
  prof_begin(KH == 1 && KW == 1 && stride == 1 && pad == 0 && Cout <= pw_max_channels() ? BFHIP_OP_CONV2D_PW_DGRAD : BFHIP_OP_CONV2D_DGRAD, s,
             &ps);
  if (w)
    hipLaunchKernelGGL(conv_weight_transpose_kernel, dim3(ceil_div(Cin, 32), ceil_div(Cout, 32), KH * KW), dim3(256), 0, s,
                       (const bf16_t *)w, (bf16_t *)workspace, Cout, KH * KW, Cin);
  ConvGeom g = {};
  g.N = N; g.H = OH; g.W = OW; g.C = Cout; g.ldx = ldg;   // gathered tensor = dy
  g.OH = H; g.OW = W;                                    // GEMM rows = input pixels
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.dil = dil; g.transposed = 1;
  BFHIP_REQUIRE((stride & (stride - 1)) == 0, "conv2d_dgrad: the stride must be a power of two (got %d)", stride);
  g.sshift = __builtin_ctz((unsigned)stride); g.smask = stride - 1;
  BFHIP_REQUIRE((long long)N * OH * OW * ldg < (1LL << 31), "conv2d_dgrad: tensors of 2^31 elements or more are not supported");
  g.nq = KH * KW * Cout / 8;
  g.M = (long long)N * H * W;
  g.Kout = Cin; g.ldw = KH * KW * Cout; g.ldy = ldx;
  static const int parity = [] { const char *e = getenv("BFHIP_CONV_DGRAD_PARITY"); return e ? atoi(e) : 1; }();
  if (stride > 1 && stride <= 4 && dil == 1 && parity) {
    // parity classes of the input pixels: (ih + pad) mod stride selects the kh that reach a pixel (ConvGeom::cls); one launch,
    // row tiles class by class; g.nq stays the full tap count (it sizes the tap table), g.M the full row count (tile shape)
    g.transposed = 2;
    g.ncls = 0;
    for (int ph = 0; ph < stride; ++ph)
      for (int pw = 0; pw < stride; ++pw) {
        ConvGeom::ParityClass &c = g.cls[g.ncls];
        c.kh0 = ph; c.kw0 = pw;
        c.nkh = ph < KH ? (KH - ph + stride - 1) / stride : 0;
        c.nkw = pw < KW ? (KW - pw + stride - 1) / stride : 0;
        if (c.nkh == 0 || c.nkw == 0) c.nkh = c.nkw = 0;
        c.h0 = ((ph - pad) % stride + stride) % stride;
        c.w0 = ((pw - pad) % stride + stride) % stride;
        c.Hc = c.h0 < H ? (H - c.h0 + stride - 1) / stride : 0;
        c.Wc = c.w0 < W ? (W - c.w0 + stride - 1) / stride : 0;
        c.tile0 = 0;
        if (c.Hc > 0 && c.Wc > 0) ++g.ncls;
      }
  }
  BFHIP_REQUIRE(((uintptr_t)addend % 4) == 0 && (addend_stride == 1 || addend_stride == 2), "conv2d_dgrad: bad addend");
  const int rc = launch_igemm(dy, workspace, nullptr, dx, nullptr, g, out_f32, s, "conv2d_dgrad",
                              Addend{(const bf16_t *)addend, addend ? addend_stride : 0, Cin, H, W});
  prof_end(&ps);
  return rc;
}

BFHIP_EXPORT int bfhip_conv2d_dgrad(const void *dy, int ldg, const void *w, void *dx, int ldx, int N, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad, int dil, int out_f32, void *workspace,
                                    size_t workspace_bytes, void *stream_) {
  BFHIP_REQUIRE(w, "conv2d_dgrad: null pointer");
  return conv2d_dgrad_impl(dy, ldg, w, dx, ldx, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, out_f32, workspace, workspace_bytes,
                           (hipStream_t)stream_);
}

// the same with the weight already transposed: wt bf16 [Cin][KH][KW][Cout] (bfhip_conv2d_weight_transpose_batched); read-only.
// addend (optional, bf16, dense: pixel pitch Cin): dx = data gradient + addend in the kernel's epilogue -- the other gradient path
// into the same tensor (a residual connection).  addend_stride 1: addend is [N, H, W, Cin]; 2: addend is [N, ceil(H/2), ceil(W/2), Cin],
// the gradient of a stride-2 1x1 shortcut over x, and reaches the pixels with even h and w only.  Only for calls the pointwise
// kernel serves (bfhip_conv2d_dgrad_fuses_addend)
BFHIP_EXPORT int bfhip_conv2d_dgrad_wt(const void *dy, int ldg, const void *wt, const void *addend, int addend_stride, void *dx,
                                       int ldx, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil,
                                       int out_f32, void *stream_) {
  return conv2d_dgrad_impl(dy, ldg, nullptr, dx, ldx, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, out_f32, (void *)wt,
                           bfhip_conv2d_dgrad_workspace_bytes(Cin, Cout, KH, KW), (hipStream_t)stream_, addend, addend_stride);
}

BFHIP_EXPORT int bfhip_conv2d_dgrad_fuses_addend(int KH, int KW, int stride, int pad, int out_f32) {
  return KH == 1 && KW == 1 && stride == 1 && pad == 0 && !out_f32 ? 1 : 0;
}

// src f32 [P][C] (dense) -> bf16 pairs hi / lo (v ~ hi + lo to 2^-16 relative): chan (optional) bf16 [P][3C], batch (optional)
// bf16 [3][P][C]; bit k of order_* = block k holds lo.  C % 8 == 0.
BFHIP_EXPORT int bfhip_split_bf16x3(const float *src, long long P, int C, void *chan, int order_chan, void *batch, int order_batch,
                                    void *stream_) {
  BFHIP_REQUIRE(src && (chan || batch) && P > 0 && C > 0 && C % 8 == 0, "split_bf16x3: bad arguments");
  BFHIP_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)chan % 16) == 0 && ((uintptr_t)batch % 16) == 0, "split_bf16x3: misaligned tensor");
  const long long total = P * (C / 8);
  BFHIP_REQUIRE(total < (1LL << 31) * 256, "split_bf16x3: tensor too large");
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream_, src, P, C,
                     (bf16_t *)chan, order_chan, (bf16_t *)batch, order_batch);
  return check_launch("split_bf16x3");
}

BFHIP_EXPORT int bfhip_conv2d_wt_segment_bytes(void) { return (int)sizeof(WtSeg); }

// segs_dev: nseg device records {u64 src, u64 dst, i32 Cout, i32 taps, i32 Cin, i32 src_f32, i64 first block}, blocks of a
// segment = ceil(Cin / 32) * ceil(Cout / 32) * taps, first blocks ascending from 0; total_blocks = their sum
BFHIP_EXPORT int bfhip_conv2d_weight_transpose_batched(const void *segs_dev, int nseg, long long total_blocks, void *stream_) {
  BFHIP_REQUIRE(segs_dev && nseg > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "conv2d_weight_transpose_batched: bad table");
  hipLaunchKernelGGL(conv_weight_transpose_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream_,
                     (const WtSeg *)segs_dev, nseg);
  return check_launch("conv2d_weight_transpose_batched");
}

static int resident_blocks() {  // workgroups the chip holds at once (2 per CU: 66 KB of LDS each)
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = 2 * v;
    else { (void)hipGetLastError(); n = 512; }
  }
  return n;
}

// The launch must fit ONE residency round: every workgroup runs the same number of K-steps, so a grid of 513 workgroups on
// 512 slots takes twice as long as one of 512 (measured: 130 vs 66 us on the 128 -> 128 layer).
static void wgrad_plan(long long M, int Cout, int Ktot, int *splits, long long *rows_per_split, int *tiles_co, int *tiles_k) {
  *tiles_co = ceil_div(Cout, 128);
  *tiles_k = ceil_div(Ktot, 128);
  const int tiles = *tiles_co * *tiles_k;
  long long steps = (M + 63) / 64;
  const int slots = resident_blocks();
  int want = tiles >= slots ? 1 : slots / tiles;    // each split costs a full fp32 slab of dW
  if (want > steps / 8) want = (int)(steps / 8);    // at least 8 K-steps per workgroup
  if (want < 1) want = 1;
  long long per = (steps + want - 1) / want;
  *splits = (int)((steps + per - 1) / per);
  *rows_per_split = per * 64;
}

// Tile shape of the weight gradient: 0 = 128 x 128 (conv_wgrad_kernel, two workgroups per CU), 1 = 128 co x 256 k and
// 2 = 256 co x 128 k (conv_wgrad_wide_kernel, three stages, one 512-thread workgroup per CU).  The wide tiles need K (resp.
// Cout) beyond one 128-column panel, a tap table that fits beside three 48 KB stages, and enough pixels for >= 6 steps.
static int wgrad_shape(long long M, int Cout, int Ktot, long long ohow) {
  static const int wide = [] { const char *e = getenv("BFHIP_WGRAD_WIDE"); return e ? atoi(e) : 1; }();
  // ohow >= 64: the wide kernel's incremental addressing assumes at most one row wrap and one image wrap per 64-pixel step
  if (!wide || Ktot / 8 > 3584 || M < 64 * 6 || ohow < 64) return 0;
  if (Ktot > 128) return 1;
  if (Cout > 128) return 2;
  // a single 128 x 128 tile of dW over very many pixels (1x1 layers with <= 128 channels on both sides; x^T y of the decoder's
  // projections): the wide kernel with half of its tile empty still beats the two-stage 128 x 128 kernel, whose 253 workgroups of
  // 4 waves leave one wave per SIMD (BFHIP_WGRAD_WIDE_SMALL_M: fewest pixels for that, 0 = never)
  static const long long small_m = [] { const char *e = getenv("BFHIP_WGRAD_WIDE_SMALL_M"); return e ? atoll(e) : 1024LL; }();
  if (small_m > 0 && M >= small_m) return 1;
  return 0;
}

static void wgrad_plan_wide(long long M, int Cout, int Ktot, int shape, int *splits, long long *rows_per_split, int *tiles_co,
                            int *tiles_k) {
  *tiles_co = ceil_div(Cout, shape == 2 ? 256 : 128);
  *tiles_k = ceil_div(Ktot, shape == 1 ? 256 : 128);
  const int tiles = *tiles_co * *tiles_k;
  long long steps = (M + 63) / 64;
  const int slots = resident_blocks() / 2;          // one workgroup per CU (3 x 48 KB of LDS), one residency round
  int want = tiles >= slots ? 1 : slots / tiles;
  static const int min_steps = [] { const char *e = getenv("BFHIP_WGRAD_MIN_STEPS"); return e && atoi(e) > 0 ? atoi(e) : 6; }();
  if (want > steps / min_steps) want = (int)(steps / min_steps);  // at least 6 steps per workgroup (the ring is 3 deep)
  if (want < 1) want = 1;
  long long per = (steps + want - 1) / want;
  *splits = (int)((steps + per - 1) / per);
  *rows_per_split = per * 64;
}

static void wgrad_plan_any(long long M, long long ohow, int Cout, int Ktot, int *shape, int *splits, long long *rps, int *tco,
                           int *tk) {
  *shape = wgrad_shape(M, Cout, Ktot, ohow);
  if (*shape) wgrad_plan_wide(M, Cout, Ktot, *shape, splits, rps, tco, tk);
  else wgrad_plan(M, Cout, Ktot, splits, rps, tco, tk);
}

BFHIP_EXPORT size_t bfhip_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW) {
  int shape, splits, tco, tk;
  long long rps;
  wgrad_plan_any((long long)N * OH * OW, (long long)OH * OW, Cout, KH * KW * Cin, &shape, &splits, &rps, &tco, &tk);
  return align_up((size_t)splits * Cout * KH * KW * Cin * sizeof(float), 256);
}

// dw[Cout][KH][KW][Cin] (fp32 or bf16) = sum over pixels of dy x gathered x
BFHIP_EXPORT int bfhip_conv2d_wgrad(const void *x, int ldx, const void *dy, int ldg, void *dw, int N, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad, int dil, int dw_bf16, void *workspace,
                                    size_t workspace_bytes, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(bfhip_conv2d_supported(N, H, W, Cin, Cout, KH, KW, stride, pad, dil), "conv2d_wgrad: unsupported geometry");
  BFHIP_REQUIRE(x && dy && dw && workspace, "conv2d_wgrad: null pointer");
  BFHIP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ldx % 8 == 0 && ldg % 8 == 0 && ldx >= Cin && ldg >= Cout,
                "conv2d_wgrad: operands must be 16-byte aligned with pitches that are multiples of 8 elements");
  WgradGeom wg;
  ConvGeom &g = wg.c;
  g.N = N; g.H = H; g.W = W; g.C = Cin; g.ldx = ldx;
  g.OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  g.OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.dil = dil; g.transposed = 0; g.sshift = 0; g.smask = 0;
  g.nq = KH * KW * Cin / 8;
  g.M = (long long)N * g.OH * g.OW;
  g.Kout = Cout; g.ldw = 0; g.ldy = 0;
  wg.Cout = Cout; wg.ldg = ldg;
  int shape;
  BFHIP_REQUIRE((long long)N * H * W * ldx < (1LL << 31) && g.M * ldg < (1LL << 31),
                "conv2d_wgrad: tensors of 2^31 elements or more are not supported");
  wgrad_plan_any(g.M, (long long)g.OH * g.OW, Cout, KH * KW * Cin, &shape, &wg.splits, &wg.rows_per_split, &wg.tiles_co, &wg.tiles_k);
  BFHIP_REQUIRE(workspace_bytes >= bfhip_conv2d_wgrad_workspace_bytes(N, g.OH, g.OW, Cin, Cout, KH, KW), "conv2d_wgrad: workspace too small");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)conv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / 2);
    (void)hipFuncSetAttribute((const void *)conv_wgrad_wide_kernel<1, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)conv_wgrad_wide_kernel<2, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  ProfScope ps;
  prof_begin(BFHIP_OP_CONV2D_WGRAD, s, &ps);
  const dim3 grid((unsigned)((long long)wg.tiles_co * wg.tiles_k * wg.splits));
  if (shape == 0) {
    const size_t lds = (size_t)4 * 64 * 256 + (size_t)g.nq * 4;
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), lds, s, (const bf16_t *)x, (const bf16_t *)dy, (float *)workspace, wg);
  } else {
    const size_t lds = (size_t)3 * 3 * 64 * 256 + (size_t)g.nq * 4;
    if (shape == 1)
      hipLaunchKernelGGL((conv_wgrad_wide_kernel<1, 2, 3>), grid, dim3(512), lds, s, (const bf16_t *)x, (const bf16_t *)dy,
                         (float *)workspace, wg);
    else
      hipLaunchKernelGGL((conv_wgrad_wide_kernel<2, 1, 3>), grid, dim3(512), lds, s, (const bf16_t *)x, (const bf16_t *)dy,
                         (float *)workspace, wg);
  }
  const long long total = (long long)Cout * KH * KW * Cin;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(ceil_div(total, 1024)), dim3(256), 0, s, (const float *)workspace, wg.splits,
                     total, dw, dw_bf16);
  prof_end(&ps);
  return check_launch("conv2d_wgrad");
}

// ---------------------------------------------------------------------------------- grouped weight gradients (host side)
// One record per layer, filled by the caller in host memory (include/bevfusion_hip.h: bfhip_wgrad_layer).
struct WgradLayerDesc {
  const void *x, *dy;
  void *dw;
  int ldx, ldg, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, dw_bf16, reserved;
};
static_assert(sizeof(WgradLayerDesc) == 80, "bfhip_wgrad_layer layout");

BFHIP_EXPORT size_t bfhip_conv2d_wgrad_group_table_bytes(int n_layers) {
  return n_layers > 0 ? 256 + (size_t)n_layers * sizeof(WgradItem) : 0;
}

// 1 when the layer can join a group: a geometry the wide weight-gradient kernels take (everything else keeps bfhip_conv2d_wgrad)
BFHIP_EXPORT int bfhip_conv2d_wgrad_groupable(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil) {
  if (!bfhip_conv2d_supported(N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return 0;
  const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  return wgrad_shape((long long)N * OH * OW, Cout, KH * KW * Cin, (long long)OH * OW) != 0;
}

// Plans the group and writes the image of its device table (header + one item per layer) into table_host; the caller copies
// the image to the device (stream-ordered, before the launch) and provides *slab_bytes of workspace.  target_steps: 64-pixel
// steps per workgroup to aim for (<= 0: BFHIP_WGRAD_GROUP_STEPS or 96: the 77 layers of the full step at 32 / 64 / 96 / 128 / 192 / 256 steps take 3.29 / 2.95 /
// 2.80 / 2.80 / 2.83 / 2.84 ms with 1.9 / 1.0 / 0.73 / 0.58 / 0.44 / 0.36 GB of slabs; inside the full step, ten alternating pairs: 96
// steps 25.9 ms, 192 steps 26.3 ms (median) -- the last residency round of 192-step workgroups is a 260 us tail).
BFHIP_EXPORT int bfhip_conv2d_wgrad_group_plan(const void *layers_, int n, int target_steps, void *table_host, size_t table_bytes,
                                               size_t *slab_bytes) {
  const WgradLayerDesc *L = (const WgradLayerDesc *)layers_;
  BFHIP_REQUIRE(L && n > 0 && table_host && slab_bytes, "conv2d_wgrad_group_plan: bad arguments");
  BFHIP_REQUIRE(table_bytes >= bfhip_conv2d_wgrad_group_table_bytes(n), "conv2d_wgrad_group_plan: table too small");
  if (target_steps <= 0) {
    static const int env = [] { const char *e = getenv("BFHIP_WGRAD_GROUP_STEPS"); return e && atoi(e) > 0 ? atoi(e) : 96; }();
    target_steps = env;
  }
  WgradGroupHeader hd;
  memset(&hd, 0, sizeof hd);
  hd.n_items = n;
  hd.target_steps = target_steps;
  std::vector<WgradItem> items((size_t)n);
  std::vector<long long> per((size_t)n);
  long long shape_steps[3] = {0, 0, 0};
  int shape_target[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const WgradLayerDesc &d = L[i];
    BFHIP_REQUIRE(d.x && d.dy && d.dw, "conv2d_wgrad_group_plan: null pointer in a layer");
    BFHIP_REQUIRE(bfhip_conv2d_wgrad_groupable(d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.dil),
                  "conv2d_wgrad_group_plan: a layer is not groupable (ask bfhip_conv2d_wgrad_groupable first)");
    BFHIP_REQUIRE(((uintptr_t)d.x % 16) == 0 && ((uintptr_t)d.dy % 16) == 0 && ((uintptr_t)d.dw % 16) == 0 && d.ldx % 8 == 0 &&
                  d.ldg % 8 == 0 && d.ldx >= d.Cin && d.ldg >= d.Cout,
                  "conv2d_wgrad_group_plan: operands must be 16-byte aligned with pitches that are multiples of 8 elements");
    WgradItem &it = items[i];
    memset(&it, 0, sizeof it);
    it.x = (unsigned long long)(uintptr_t)d.x; it.dy = (unsigned long long)(uintptr_t)d.dy; it.dw = (unsigned long long)(uintptr_t)d.dw;
    it.N = d.N; it.H = d.H; it.W = d.W; it.C = d.Cin; it.ldx = d.ldx;
    it.OH = (d.H + 2 * d.pad - d.dil * (d.KH - 1) - 1) / d.stride + 1;
    it.OW = (d.W + 2 * d.pad - d.dil * (d.KW - 1) - 1) / d.stride + 1;
    it.KH = d.KH; it.KW = d.KW; it.stride = d.stride; it.pad = d.pad; it.dil = d.dil;
    const int Ktot = d.KH * d.KW * d.Cin;
    it.nq = Ktot / 8;
    it.M = (long long)d.N * it.OH * it.OW;
    it.Cout = d.Cout; it.ldg = d.ldg; it.dw_bf16 = d.dw_bf16;
    BFHIP_REQUIRE((long long)d.N * d.H * d.W * d.ldx < (1LL << 31) && it.M * d.ldg < (1LL << 31),
                  "conv2d_wgrad_group_plan: tensors of 2^31 elements or more are not supported");
    it.shape = wgrad_shape(it.M, d.Cout, Ktot, (long long)it.OH * it.OW);
    it.tiles_co = ceil_div(d.Cout, it.shape == 2 ? 256 : 128);
    it.tiles_k = ceil_div(Ktot, it.shape == 1 ? 256 : 128);
    it.total = (long long)d.Cout * Ktot;
    it.n_rblocks = (int)ceil_div(it.total, 1024);
    shape_steps[it.shape] += (long long)it.tiles_co * it.tiles_k * ((it.M + 63) / 64);
  }
  // steps per workgroup: the target, lowered for a launch that would otherwise have fewer than ~4 residency rounds of workgroups
  // (the 9 layers with K <= 128 of the ResNet trunk at 96 steps: 285 workgroups on 256 CUs = two rounds, the second one empty)
  for (int sh = 1; sh <= 2; ++sh) {
    long long t = shape_steps[sh] / 1024;
    shape_target[sh] = (int)std::min<long long>(target_steps, std::max<long long>(8, t));
  }
  for (int i = 0; i < n; ++i) {
    WgradItem &it = items[i];
    const int tgt = shape_target[it.shape];
    const long long steps = (it.M + 63) / 64;
    long long want = (steps + tgt / 2) / tgt;
    if (want > steps / 6) want = steps / 6;  // the ring is three deep: at least 6 steps per workgroup
    if (want < 1) want = 1;
    per[i] = (steps + want - 1) / want;
    it.splits = (int)((steps + per[i] - 1) / per[i]);
    it.rows_per_split = per[i] * 64;
    it.n_blocks = it.tiles_co * it.tiles_k * it.splits;
  }
  // table order: shape 1 then shape 2, inside a shape by descending steps per workgroup (ties: caller's order)
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (items[a].shape != items[b].shape) return items[a].shape < items[b].shape;
    return per[a] > per[b];
  });
  WgradItem *out = (WgradItem *)((unsigned char *)table_host + 256);
  size_t slab = 0;
  long long rb = 0;
  for (int sh = 1; sh <= 2; ++sh) hd.first_item[sh] = -1;
  for (int k = 0; k < n; ++k) {
    WgradItem it = items[order[k]];
    const int sh = it.shape;
    if (hd.first_item[sh] < 0) hd.first_item[sh] = k;
    ++hd.n_shape[sh];
    it.first_block = hd.blocks[sh];
    BFHIP_REQUIRE((long long)hd.blocks[sh] + it.n_blocks < (1LL << 30), "conv2d_wgrad_group_plan: too many workgroups");
    hd.blocks[sh] += it.n_blocks;
    if (it.nq > hd.max_nq[sh]) hd.max_nq[sh] = it.nq;
    it.slab_off = slab;
    slab += align_up((size_t)it.splits * it.total * sizeof(float), 256);
    it.first_rblock = (int)rb;
    rb += it.n_rblocks;
    BFHIP_REQUIRE(rb < (1LL << 30), "conv2d_wgrad_group_plan: too many reduce blocks");
    out[k] = it;
  }
  hd.rblocks = (int)rb;
  hd.slab_bytes = slab;
  // XCD chunks of equal total steps (a workgroup's cost = its step count; all workgroups of one launch have the same tile shape)
  for (int sh = 1; sh <= 2; ++sh) {
    if (!hd.n_shape[sh]) continue;
    const WgradItem *its = out + hd.first_item[sh];
    long long total_steps = 0;
    for (int k = 0; k < hd.n_shape[sh]; ++k) total_steps += (long long)its[k].n_blocks * (its[k].rows_per_split / 64);
    int c = 1, longest = 0;
    long long acc = 0;
    hd.chunk_start[sh][0] = 0;
    for (int k = 0; k < hd.n_shape[sh]; ++k) {
      const long long w = its[k].rows_per_split / 64;
      for (int b = 0; b < its[k].n_blocks; ++b) {
        // block (first_block + b) opens chunk c when the steps before it reach c/8 of the total
        while (c < 8 && acc * 8 >= total_steps * c) hd.chunk_start[sh][c++] = its[k].first_block + b;
        acc += w;
      }
    }
    while (c <= 8) hd.chunk_start[sh][c++] = hd.blocks[sh];
    for (int x = 0; x < 8; ++x) longest = std::max(longest, hd.chunk_start[sh][x + 1] - hd.chunk_start[sh][x]);
    hd.grid[sh] = 8 * longest;
  }
  memcpy(table_host, &hd, sizeof hd);
  *slab_bytes = slab;
  return 0;
}

// table_host: the image bfhip_conv2d_wgrad_group_plan wrote (its header is read here), table_dev: its device copy
BFHIP_EXPORT int bfhip_conv2d_wgrad_group_launch(const void *table_host, const void *table_dev, void *slab, size_t slab_bytes,
                                                 void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(table_host && table_dev && slab, "conv2d_wgrad_group_launch: null pointer");
  WgradGroupHeader hd;
  memcpy(&hd, table_host, sizeof hd);
  BFHIP_REQUIRE(hd.n_items > 0 && hd.n_items == hd.n_shape[1] + hd.n_shape[2], "conv2d_wgrad_group_launch: not a planned table");
  BFHIP_REQUIRE(slab_bytes >= hd.slab_bytes && ((uintptr_t)slab % 256) == 0, "conv2d_wgrad_group_launch: workspace too small or misaligned");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)conv_wgrad_group_kernel<1, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)conv_wgrad_group_kernel<2, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const WgradItem *items = (const WgradItem *)((const unsigned char *)table_dev + 256);
  const int *chunks = (const int *)((const unsigned char *)table_dev + offsetof(WgradGroupHeader, chunk_start));
  ProfScope ps;
  prof_begin(BFHIP_OP_CONV2D_WGRAD, s, &ps);
  for (int sh = 1; sh <= 2; ++sh) {
    if (!hd.n_shape[sh]) continue;
    const size_t lds = (size_t)3 * 3 * 64 * 256 + (size_t)hd.max_nq[sh] * 4;
    if (sh == 1)
      hipLaunchKernelGGL((conv_wgrad_group_kernel<1, 2, 3>), dim3((unsigned)hd.grid[sh]), dim3(512), lds, s, items + hd.first_item[sh],
                         hd.n_shape[sh], chunks + sh * 9, (unsigned char *)slab);
    else
      hipLaunchKernelGGL((conv_wgrad_group_kernel<2, 1, 3>), dim3((unsigned)hd.grid[sh]), dim3(512), lds, s, items + hd.first_item[sh],
                         hd.n_shape[sh], chunks + sh * 9, (unsigned char *)slab);
  }
  hipLaunchKernelGGL(conv_wgrad_group_reduce_kernel, dim3((unsigned)hd.rblocks), dim3(256), 0, s, items, hd.n_items,
                     (const unsigned char *)slab);
  prof_end(&ps);
  return check_launch("conv2d_wgrad_group");
}
