// attn.hip -- cross attention of a few queries over very many keys (gfx950): the TransFusion decoder layer attends
// from 200 object queries to all 180 x 180 = 32 400 BEV cells (BF/transformer.py:60-105; 8 heads of 16 channels, dropout 0.1
// on the attention weights).  Generic fused-attention kernels parallelise over query tiles: with 200 queries that is
// ~128 workgroups walking 32 400 keys each (1.2 ms forward, 1.6 ms for dQ here).  These kernels split the KEY axis over
// the chip instead and keep every query of a (batch, head) in one workgroup:
//   attn_lse_kernel     per (b, h, key chunk): S^T = K Q^T on the bf16 MFMA, online (max, sum) per query -> partials;
//   attn_lse_combine    log-sum-exp per query over the chunks
//   attn_out_kernel     P^T = exp(S^T - lse) (x dropout mask / (1 - p)) feeds the second MFMA as its A operand straight from
//                       the accumulator layout (no transposition: computing S TRANSPOSED puts the 4 keys a lane holds on the
//                       MFMA's K axis); O partial per chunk; attn_out_combine sums the chunks in a fixed order (bit-reproducible)
//   attn_bwd_kernel     per (b, h, key chunk): dS^T and A^T tiles in the accumulator layout feed dQ (reduction over keys)
//                       directly; for dK, dV (reductions over queries) the two 16 x 16 tiles go through 1 KB of wave-private
//                       LDS to swap their axes; dK, dV are final per key, dQ partial per chunk -> attn_out_combine
// Tensors are [B, L, E] row-major with E = H * 16 (head h = channels 16h .. 16h+15), bf16; softmax statistics fp32.
// Dropout uses a counter hash of (seed, element index): the same mask is regenerated in the backward.
#include "common.h"

namespace bfhip {
namespace {

typedef unsigned short bf16_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 16;        // head dim
constexpr int kMaxQT = 16;    // up to 256 queries
constexpr int kChunk = 512;   // keys per workgroup (4 waves x 8 key tiles)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ s16x4 ld4(const bf16_t *p) {  // 4 consecutive bf16 (8-byte aligned)
  uint2 u = *(const uint2 *)p;
  s16x4 r;
  r[0] = (short)(u.x & 0xffff); r[1] = (short)(u.x >> 16); r[2] = (short)(u.y & 0xffff); r[3] = (short)(u.y >> 16);
  return r;
}
__device__ __forceinline__ s16x4 zero4() { s16x4 r = {0, 0, 0, 0}; return r; }

// keep-probability test: 32-bit counter hash of the element index ((b*H + h)*Lq + q)*Lk + key, mixed with the seed
// (three 32-bit multiplies; a 64-bit mixer costs several times more ALU and these kernels are ALU-bound on it)
__device__ __forceinline__ bool keep(unsigned long long seed, unsigned row_base, unsigned key, unsigned thresh24) {
  unsigned x = (row_base + key) ^ (unsigned)seed;
  x *= 0x9E3779B1u; x ^= x >> 15;
  x += (unsigned)(seed >> 32);
  x *= 0x85EBCA77u; x ^= x >> 13;
  x *= 0xC2B2AE3Du; x ^= x >> 16;
  return (x >> 8) >= thresh24;
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)

struct Dims { int B, H, Lq, Lk, E, nchunk; float scale; unsigned thresh24; float inv_keep; unsigned long long seed;
              const unsigned long long *seed_dev; };
// seed_dev (optional): a device-side call counter mixed into the host seed -- a captured hipGraph replays the same host
// arguments every step, the counter (advanced inside the graph) still gives every step its own dropout mask
__device__ __forceinline__ unsigned long long eff_seed(const Dims &dm) {
  return dm.seed_dev ? dm.seed + *dm.seed_dev * 0xD1B54A32D192ED03ull : dm.seed;
}

// ---------------------------------------------------------------------------------------------------- LSE
// partial[(bh * nchunk + c) * Lq + q] = (m, l) of chunk c
__global__ __launch_bounds__(256) void attn_lse_kernel(const bf16_t *__restrict__ Q, const bf16_t *__restrict__ K, Dims dm,
                                                       float2 *__restrict__ partial) {
  __shared__ float2 red[4][kMaxQT * 16];
  const int c = blockIdx.x, bh = blockIdx.y, b = bh / dm.H, h = bh - b * dm.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ln = lane & 15, lg = lane >> 4;
  const int NQ = (dm.Lq + 15) >> 4;
  const bf16_t *Qb = Q + (size_t)b * dm.Lq * dm.E + h * kD, *Kb = K + (size_t)b * dm.Lk * dm.E + h * kD;
  float m[kMaxQT], l[kMaxQT];
  for (int t = 0; t < kMaxQT; ++t) { m[t] = -INFINITY; l[t] = 0.f; }
  // the workgroup keeps ALL queries: their MFMA operands stay in registers for the whole key loop (2 VGPRs per 16-query tile)
  s16x4 qreg[kMaxQT];
#pragma unroll
  for (int t = 0; t < kMaxQT; ++t) qreg[t] = (t < NQ && t * 16 + ln < dm.Lq) ? ld4(Qb + (size_t)(t * 16 + ln) * dm.E + lg * 4) : zero4();
  const int key0 = c * kChunk + wave * (kChunk / 4);
  for (int kt = 0; kt < kChunk / 64; ++kt) {
    const int key = key0 + kt * 16 + ln;       // A operand row (key) of this lane
    const s16x4 ka = key < dm.Lk ? ld4(Kb + (size_t)key * dm.E + lg * 4) : zero4();
    const int kbase = key0 + kt * 16 + lg * 4;  // the 4 keys of this lane's accumulator rows
#pragma unroll
    for (int t = 0; t < kMaxQT; ++t) {
      if (t >= NQ) continue;
      f32x4 s = MFMA16(ka, qreg[t], ((f32x4){0.f, 0.f, 0.f, 0.f}));  // S^T[key = kbase + i][q = t*16 + ln]
      float mx = m[t];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[i] = kbase + i < dm.Lk ? s[i] * dm.scale : -INFINITY;
        mx = fmaxf(mx, s[i]);
      }
      if (mx == -INFINITY) continue;
      float acc = l[t] * __expf(m[t] - mx);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += __expf(s[i] - mx);
      m[t] = mx; l[t] = acc;
    }
  }
  // merge the 4 key groups of the wave (lanes ln, ln+16, ln+32, ln+48), then the 4 waves
#pragma unroll
  for (int t = 0; t < kMaxQT; ++t) {
    if (t >= NQ) continue;
    float mm = m[t], ll = l[t];
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      float om = __shfl_xor(mm, off), ol = __shfl_xor(ll, off);
      float M = fmaxf(mm, om);
      ll = (M == -INFINITY) ? 0.f : ll * __expf(mm - M) + ol * __expf(om - M);
      mm = M;
    }
    if (lg == 0) red[wave][t * 16 + ln] = make_float2(mm, ll);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < dm.Lq; q += 256) {
    float mm = -INFINITY, ll = 0.f;
    for (int w = 0; w < 4; ++w) {
      float2 v = red[w][q];
      float M = fmaxf(mm, v.x);
      ll = (M == -INFINITY) ? 0.f : ll * __expf(mm - M) + v.y * __expf(v.x - M);
      mm = M;
    }
    partial[((size_t)bh * dm.nchunk + c) * dm.Lq + q] = make_float2(mm, ll);
  }
}

__global__ __launch_bounds__(256) void attn_lse_combine(const float2 *__restrict__ partial, Dims dm, float *__restrict__ lse) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= dm.B * dm.H * dm.Lq) return;
  int bh = t / dm.Lq, q = t - bh * dm.Lq;
  float mm = -INFINITY, ll = 0.f;
  for (int c = 0; c < dm.nchunk; ++c) {
    float2 v = partial[((size_t)bh * dm.nchunk + c) * dm.Lq + q];
    float M = fmaxf(mm, v.x);
    ll = (M == -INFINITY) ? 0.f : ll * __expf(mm - M) + v.y * __expf(v.x - M);
    mm = M;
  }
  lse[t] = mm + __logf(ll);
}

// ---------------------------------------------------------------------------------------------------- O
// opart[((bh * nchunk + c) * Lq + q) * 16 + dv]
template <bool DROP>
__global__ __launch_bounds__(256) void attn_out_kernel(const bf16_t *__restrict__ Q, const bf16_t *__restrict__ K,
                                                       const bf16_t *__restrict__ V, const float *__restrict__ lse, Dims dm,
                                                       float *__restrict__ opart) {
  __shared__ bf16_t vt[4][16][kD + 2];          // per wave: the current V key tile (transposed reads)
  __shared__ float osum[kMaxQT * 16 * kD];      // cross-wave reduction of O
  const int c = blockIdx.x, bh = blockIdx.y, b = bh / dm.H, h = bh - b * dm.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ln = lane & 15, lg = lane >> 4;
  const int NQ = (dm.Lq + 15) >> 4;
  const bf16_t *Qb = Q + (size_t)b * dm.Lq * dm.E + h * kD, *Kb = K + (size_t)b * dm.Lk * dm.E + h * kD,
               *Vb = V + (size_t)b * dm.Lk * dm.E + h * kD;
  const float *lb = lse + (size_t)bh * dm.Lq;
  for (int i = threadIdx.x; i < NQ * 16 * kD; i += 256) osum[i] = 0.f;
  f32x4 o[kMaxQT];
  for (int t = 0; t < kMaxQT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float lq_[kMaxQT];
  for (int t = 0; t < kMaxQT; ++t) lq_[t] = (t < NQ && t * 16 + ln < dm.Lq) ? lb[t * 16 + ln] : 0.f;
  s16x4 qreg[kMaxQT];  // all query tiles in registers for the whole key loop
#pragma unroll
  for (int t = 0; t < kMaxQT; ++t) qreg[t] = (t < NQ && t * 16 + ln < dm.Lq) ? ld4(Qb + (size_t)(t * 16 + ln) * dm.E + lg * 4) : zero4();
  const unsigned long long seed = eff_seed(dm);
  const int key0 = c * kChunk + wave * (kChunk / 4);
  for (int kt = 0; kt < kChunk / 64; ++kt) {
    const int key = key0 + kt * 16 + ln;
    const s16x4 ka = key < dm.Lk ? ld4(Kb + (size_t)key * dm.E + lg * 4) : zero4();
    // stage V[16 keys][16] of this tile: lane (ln = key, lg = 4-channel group)
    {
      s16x4 vv = key < dm.Lk ? ld4(Vb + (size_t)key * dm.E + lg * 4) : zero4();
#pragma unroll
      for (int j = 0; j < 4; ++j) vt[wave][ln][lg * 4 + j] = (bf16_t)vv[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // B operand of P V: B[k = key][n = dv]: lane (n = ln, keys lg*4 .. +3)
    s16x4 vb;
#pragma unroll
    for (int j = 0; j < 4; ++j) vb[j] = (short)vt[wave][lg * 4 + j][ln];
    const int kbase = key0 + kt * 16 + lg * 4;
#pragma unroll
    for (int t = 0; t < kMaxQT; ++t) {
      if (t >= NQ) continue;
      const int q = t * 16 + ln;
      f32x4 s = MFMA16(ka, qreg[t], ((f32x4){0.f, 0.f, 0.f, 0.f}));  // S^T[key][q]
      s16x4 pa;                                                  // A operand of P V: A[m = q = ln][k = keys lg*4 + i]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float p = (kbase + i < dm.Lk && q < dm.Lq) ? __expf(s[i] * dm.scale - lq_[t]) : 0.f;
        if (DROP) p = keep(seed, ((unsigned)bh * dm.Lq + q) * dm.Lk, kbase + i, dm.thresh24) ? p * dm.inv_keep : 0.f;
        pa[i] = (short)f2bf(p);
      }
      o[t] = MFMA16(pa, vb, o[t]);  // O[q = t*16 + lg*4 + i][dv = ln]
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (int w = 0; w < 4; ++w) {  // the 4 waves add in turn: fixed order -> bit-reproducible
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < kMaxQT; ++t) {
        if (t >= NQ) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) osum[(t * 16 + lg * 4 + i) * kD + ln] += o[t][i];
      }
    }
    __syncthreads();
  }
  float *dst = opart + ((size_t)bh * dm.nchunk + c) * dm.Lq * kD;
  for (int i = threadIdx.x; i < dm.Lq * kD; i += 256) dst[i] = osum[i];
}

// O[b][q][h*16 + dv] = sum_c opart (bf16 out)
__global__ __launch_bounds__(256) void attn_out_combine(const float *__restrict__ opart, Dims dm, bf16_t *__restrict__ O) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= dm.B * dm.H * dm.Lq * kD) return;
  int dv = t % kD, r = t / kD, q = r % dm.Lq, bh = r / dm.Lq, b = bh / dm.H, h = bh - b * dm.H;
  float a0 = 0.f, a1 = 0.f;
  int c = 0;
  for (; c + 1 < dm.nchunk; c += 2) {
    a0 += opart[(((size_t)bh * dm.nchunk + c) * dm.Lq + q) * kD + dv];
    a1 += opart[(((size_t)bh * dm.nchunk + c + 1) * dm.Lq + q) * kD + dv];
  }
  if (c < dm.nchunk) a0 += opart[(((size_t)bh * dm.nchunk + c) * dm.Lq + q) * kD + dv];
  O[((size_t)b * dm.Lq + q) * dm.E + h * kD + dv] = f2bf(a0 + a1);
}

// ---------------------------------------------------------------------------------------------------- backward
// dqpart[((bh * nchunk + c) * Lq + q) * 16 + d]; dK, dV final.
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const bf16_t *__restrict__ Q, const bf16_t *__restrict__ K,
                                                       const bf16_t *__restrict__ V, const bf16_t *__restrict__ O,
                                                       const bf16_t *__restrict__ dO, const float *__restrict__ lse,
                                                       Dims dm, float *__restrict__ dqpart, bf16_t *__restrict__ dK,
                                                       bf16_t *__restrict__ dV) {
  __shared__ bf16_t qs[kMaxQT * 16][kD + 2], dos[kMaxQT * 16][kD + 2];  // Q, dO of this (b, h): transposed reads
  __shared__ float dsum[kMaxQT * 16], ls[kMaxQT * 16];                   // D_q = rowsum(dO o O), lse
  __shared__ bf16_t kts[4][16][kD + 2];                                  // per wave: current K tile (transposed reads)
  __shared__ __attribute__((aligned(8))) bf16_t tds[4][16][16], tas[4][16][16];  // per wave: dS and A tiles, [key][query]
  __shared__ float dqs[kMaxQT * 16 * kD];
  const int c = blockIdx.x, bh = blockIdx.y, b = bh / dm.H, h = bh - b * dm.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ln = lane & 15, lg = lane >> 4;
  const int NQ = (dm.Lq + 15) >> 4;
  const bf16_t *Qb = Q + (size_t)b * dm.Lq * dm.E + h * kD, *Kb = K + (size_t)b * dm.Lk * dm.E + h * kD,
               *Vb = V + (size_t)b * dm.Lk * dm.E + h * kD, *Ob = O + (size_t)b * dm.Lq * dm.E + h * kD,
               *dOb = dO + (size_t)b * dm.Lq * dm.E + h * kD;
  for (int i = threadIdx.x; i < NQ * 16; i += 256) {
    float acc = 0.f;
    for (int j = 0; j < kD; ++j) {
      bf16_t qv = i < dm.Lq ? Qb[(size_t)i * dm.E + j] : (bf16_t)0, gv = i < dm.Lq ? dOb[(size_t)i * dm.E + j] : (bf16_t)0;
      qs[i][j] = qv; dos[i][j] = gv;
      if (i < dm.Lq) acc += bf2f(gv) * bf2f(Ob[(size_t)i * dm.E + j]);
    }
    dsum[i] = acc;
    ls[i] = i < dm.Lq ? lse[(size_t)bh * dm.Lq + i] : 0.f;
  }
  for (int i = threadIdx.x; i < NQ * 16 * kD; i += 256) dqs[i] = 0.f;
  __syncthreads();
  f32x4 dq[kMaxQT];
  for (int t = 0; t < kMaxQT; ++t) dq[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long seed = eff_seed(dm);
  const int key0 = c * kChunk + wave * (kChunk / 4);
  for (int kt = 0; kt < kChunk / 64; ++kt) {
    const int key = key0 + kt * 16 + ln;  // this lane's key as an MFMA row / column index
    const bool kok = key < dm.Lk;
    const s16x4 kr = kok ? ld4(Kb + (size_t)key * dm.E + lg * 4) : zero4();  // K[key][4lg..]: A of K Q^T, B of Q K^T
    const s16x4 vr = kok ? ld4(Vb + (size_t)key * dm.E + lg * 4) : zero4();  // V[key][4lg..]: A of V dO^T, B of dO V^T
#pragma unroll
    for (int j = 0; j < 4; ++j) kts[wave][ln][lg * 4 + j] = (bf16_t)kr[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    s16x4 kcol;  // B of dS K: B[k = key lg*4 + j][n = d = ln]
#pragma unroll
    for (int j = 0; j < 4; ++j) kcol[j] = (short)kts[wave][lg * 4 + j][ln];
    const int kbase = key0 + kt * 16 + lg * 4;
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};  // [key = lg*4 + i][d = ln]
#pragma unroll
    for (int t = 0; t < kMaxQT; ++t) {
      if (t >= NQ) continue;
      const int q = t * 16 + ln;  // this lane's query as an MFMA row / column index
      s16x4 qr, gr;               // Q[q][4lg..], dO[q][4lg..] (row-major vectors)
#pragma unroll
      for (int j = 0; j < 4; ++j) { qr[j] = (short)qs[q][lg * 4 + j]; gr[j] = (short)dos[q][lg * 4 + j]; }
      // rows = keys (lg*4 + i), columns = queries (ln): the accumulator layout of K Q^T
      f32x4 st = MFMA16(kr, qr, ((f32x4){0.f, 0.f, 0.f, 0.f}));   // S^T
      f32x4 dat = MFMA16(vr, gr, ((f32x4){0.f, 0.f, 0.f, 0.f}));  // dA^T = V dO^T
      s16x4 dsa;                                                   // A of dS K: A[m = q = ln][k = key lg*4 + i]
      const float lq1 = ls[q], dq1 = dsum[q];
      const unsigned rb = ((unsigned)bh * dm.Lq + q) * dm.Lk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = kbase + i < dm.Lk && q < dm.Lq;
        float p = ok ? __expf(st[i] * dm.scale - lq1) : 0.f;
        float da = dat[i], a = p;
        if (DROP) {
          const bool kp = keep(seed, rb, kbase + i, dm.thresh24);
          da = kp ? da * dm.inv_keep : 0.f;
          a = kp ? p * dm.inv_keep : 0.f;
        }
        const bf16_t dsv = f2bf(p * (da - dq1) * dm.scale);
        dsa[i] = (short)dsv;
        // the same two tiles transposed (rows = keys) for the reductions over queries: through wave-private LDS
        tds[wave][lg * 4 + i][ln] = dsv;
        tas[wave][lg * 4 + i][ln] = f2bf(a);
      }
      dq[t] = MFMA16(dsa, kcol, dq[t]);  // dQ[q = t*16 + lg*4 + i][d = ln]
      s16x4 qcol, gcol;  // B of dS^T Q / A^T dO: B[k = q lg*4 + j][n = d = ln]
#pragma unroll
      for (int j = 0; j < 4; ++j) { qcol[j] = (short)qs[t * 16 + lg * 4 + j][ln]; gcol[j] = (short)dos[t * 16 + lg * 4 + j][ln]; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const s16x4 dst_ = ld4(&tds[wave][ln][lg * 4]);  // A[m = key = ln][k = q lg*4 + j]
      const s16x4 at_ = ld4(&tas[wave][ln][lg * 4]);
      __builtin_amdgcn_wave_barrier();
      dk = MFMA16(dst_, qcol, dk);  // output rows = the A operand's m axis = keys lg*4 + i of this tile
      dv = MFMA16(at_, gcol, dv);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = key0 + kt * 16 + lg * 4 + i;
      if (kk < dm.Lk) {
        dK[((size_t)b * dm.Lk + kk) * dm.E + h * kD + ln] = f2bf(dk[i]);
        dV[((size_t)b * dm.Lk + kk) * dm.E + h * kD + ln] = f2bf(dv[i]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < kMaxQT; ++t) {
        if (t >= NQ) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) dqs[(t * 16 + lg * 4 + i) * kD + ln] += dq[t][i];
      }
    }
    __syncthreads();
  }
  float *dst = dqpart + ((size_t)bh * dm.nchunk + c) * dm.Lq * kD;
  for (int i = threadIdx.x; i < dm.Lq * kD; i += 256) dst[i] = dqs[i];
}

// debug / test helper: the dropout keep mask as bytes [B*H, Lq, Lk]
__global__ __launch_bounds__(256) void attn_mask_kernel(Dims dm, unsigned char *__restrict__ mask) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)dm.B * dm.H * dm.Lq * dm.Lk;
  if (t >= total) return;
  int key = (int)(t % dm.Lk);
  long long r = t / dm.Lk;
  int q = (int)(r % dm.Lq), bh = (int)(r / dm.Lq);
  mask[t] = keep(eff_seed(dm), ((unsigned)bh * dm.Lq + q) * dm.Lk, key, dm.thresh24) ? 1 : 0;
}

inline int make_dims(int B, int H, int Lq, int Lk, float scale, float dropout_p, unsigned long long seed,
                     const unsigned long long *seed_dev, Dims &dm) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lq > kMaxQT * 16 || Lk <= 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return -1;
  dm.B = B; dm.H = H; dm.Lq = Lq; dm.Lk = Lk; dm.E = H * kD;
  dm.nchunk = (Lk + kChunk - 1) / kChunk;
  dm.scale = scale;
  dm.thresh24 = (unsigned)(dropout_p * 16777216.0f);
  dm.inv_keep = 1.0f / (1.0f - dropout_p);
  dm.seed = seed;
  dm.seed_dev = seed_dev;
  return 0;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT size_t bfhip_attn_workspace_bytes(int B, int H, int Lq, int Lk) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  size_t nchunk = (Lk + kChunk - 1) / kChunk;
  size_t a = align_up((size_t)B * H * nchunk * Lq * sizeof(float2), 256);
  size_t o = align_up((size_t)B * H * nchunk * Lq * kD * sizeof(float), 256);
  return a + o;
}

BFHIP_EXPORT int bfhip_attn_fwd(const void *Q, const void *K, const void *V, int B, int H, int Lq, int Lk, float scale,
                                float dropout_p, unsigned long long seed, const unsigned long long *seed_dev, void *O, float *lse, void *workspace,
                                size_t workspace_bytes, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  Dims dm;
  BFHIP_REQUIRE(make_dims(B, H, Lq, Lk, scale, dropout_p, seed, seed_dev, dm) == 0, "attn_fwd: unsupported sizes B=%d H=%d Lq=%d Lk=%d (Lq <= 256, head dim 16)", B, H, Lq, Lk);
  BFHIP_REQUIRE(Q && K && V && O && lse, "attn_fwd: null pointer");
  BFHIP_REQUIRE(((uintptr_t)Q % 8) == 0 && ((uintptr_t)K % 8) == 0 && ((uintptr_t)V % 8) == 0, "attn_fwd: tensors must be 8-byte aligned");
  if (!workspace || workspace_bytes < bfhip_attn_workspace_bytes(B, H, Lq, Lk)) { set_error("attn_fwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  float2 *lpart = (float2 *)workspace;
  float *opart = (float *)((char *)workspace + align_up((size_t)B * H * dm.nchunk * Lq * sizeof(float2), 256));
  dim3 grid(dm.nchunk, B * H);
  hipLaunchKernelGGL(attn_lse_kernel, grid, dim3(256), 0, s, (const bf16_t *)Q, (const bf16_t *)K, dm, lpart);
  hipLaunchKernelGGL(attn_lse_combine, dim3(ceil_div((long long)B * H * Lq, 256)), dim3(256), 0, s, lpart, dm, lse);
  if (dropout_p > 0.f)
    hipLaunchKernelGGL(attn_out_kernel<true>, grid, dim3(256), 0, s, (const bf16_t *)Q, (const bf16_t *)K, (const bf16_t *)V, lse, dm, opart);
  else
    hipLaunchKernelGGL(attn_out_kernel<false>, grid, dim3(256), 0, s, (const bf16_t *)Q, (const bf16_t *)K, (const bf16_t *)V, lse, dm, opart);
  hipLaunchKernelGGL(attn_out_combine, dim3(ceil_div((long long)B * H * Lq * kD, 256)), dim3(256), 0, s, opart, dm, (bf16_t *)O);
  return check_launch("attn_fwd");
}

BFHIP_EXPORT int bfhip_attn_bwd(const void *Q, const void *K, const void *V, const void *O, const void *dO, const float *lse,
                                int B, int H, int Lq, int Lk, float scale, float dropout_p, unsigned long long seed,
                                const unsigned long long *seed_dev, void *dQ, void *dK, void *dV, void *workspace,
                                size_t workspace_bytes, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  Dims dm;
  BFHIP_REQUIRE(make_dims(B, H, Lq, Lk, scale, dropout_p, seed, seed_dev, dm) == 0, "attn_bwd: unsupported sizes");
  BFHIP_REQUIRE(Q && K && V && O && dO && lse && dQ && dK && dV, "attn_bwd: null pointer");
  if (!workspace || workspace_bytes < bfhip_attn_workspace_bytes(B, H, Lq, Lk)) { set_error("attn_bwd: workspace too small"); return BFHIP_E_WORKSPACE; }
  float *dqpart = (float *)((char *)workspace + align_up((size_t)B * H * dm.nchunk * Lq * sizeof(float2), 256));
  dim3 grid(dm.nchunk, B * H);
  if (dropout_p > 0.f)
    hipLaunchKernelGGL(attn_bwd_kernel<true>, grid, dim3(256), 0, s, (const bf16_t *)Q, (const bf16_t *)K, (const bf16_t *)V, (const bf16_t *)O, (const bf16_t *)dO, lse, dm, dqpart, (bf16_t *)dK, (bf16_t *)dV);
  else
    hipLaunchKernelGGL(attn_bwd_kernel<false>, grid, dim3(256), 0, s, (const bf16_t *)Q, (const bf16_t *)K, (const bf16_t *)V, (const bf16_t *)O, (const bf16_t *)dO, lse, dm, dqpart, (bf16_t *)dK, (bf16_t *)dV);
  hipLaunchKernelGGL(attn_out_combine, dim3(ceil_div((long long)B * H * Lq * kD, 256)), dim3(256), 0, s, dqpart, dm, (bf16_t *)dQ);
  return check_launch("attn_bwd");
}

BFHIP_EXPORT int bfhip_attn_dropout_mask(int B, int H, int Lq, int Lk, float dropout_p, unsigned long long seed,
                                         const unsigned long long *seed_dev,
                                         unsigned char *mask, void *stream_) {
  Dims dm;
  BFHIP_REQUIRE(make_dims(B, H, Lq, Lk, 1.f, dropout_p, seed, seed_dev, dm) == 0 && mask, "attn_dropout_mask: bad arguments");
  long long total = (long long)B * H * Lq * Lk;
  hipLaunchKernelGGL(attn_mask_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream_, dm, mask);
  return check_launch("attn_dropout_mask");
}
