// optim.hip -- gradient-norm clipping + AdamW + bf16 parameter refresh of ALL parameter tensors of the model in three launches.
//
// The training step of the benchmark ends with clip_grad_norm_(35) + AdamW (reference config: optim_wrapper = AdamW lr 2e-4,
// wd 0.01, clip_grad max_norm 35, projects/BEVFusion/configs/nuscenes/bevfusion_lidar_voxel0075...py:369-372).  Through torch
// that is ~40 multi-tensor launches per step for this model's ~450 parameter tensors (bf16 -> fp32 gradient copies, per-tensor
// norms, the clip scale, the fused AdamW chunks, fp32 -> bf16 parameter copies: 0.65 ms of GPU time moving ~2 GB) plus the
// Python that prepares their lists.  Here a device-resident table describes every tensor once (fp32 master weights, the two
// moments, the bf16 copy the kernels consume, the length); per step only the gradient pointers change (autograd hands over
// fresh tensors), and
//   1. adamw_sumsq_kernel  : per 4096-element chunk, sum of squares of the gradient (bf16 or fp32)        -> partial[chunk]
//   2. adamw_scalars_kernel: one block adds the partials in a FIXED order (fp64), forms the clip scale max_norm / (norm + 1e-6)
//                            clamped to 1, advances the step counter unless the norm is NaN / inf, and derives the bias
//                            corrections
//   3. adamw_update_kernel : per chunk, torch's fused AdamW arithmetic (decoupled weight decay, lerp form of the first
//                            moment) on the clipped gradient, master weight + moments written back, bf16 copy rounded once.
// A non-finite gradient norm leaves parameters, moments and the step counter untouched (GradScaler's found_inf semantics).
// A tensor without a gradient this step (pointer 0) is treated as having a zero gradient.
#include "common.h"

namespace bfhip {
namespace {

struct AdamSeg {            // one parameter tensor; all arrays element-for-element in the parameter's own memory order
  float *master, *m, *v;    // fp32 master weight (the parameter itself when it is fp32), first / second moment
  unsigned short *lowp;     // bf16 copy the forward / backward kernels read (NULL for fp32 parameters)
  long long n;
  int grad_bf16, pad;
};

constexpr int kChunk = 4096, kThreads = 256, kPer = kChunk / kThreads;  // 16 elements per thread

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ float load_grad(const void *g, int bf16, long long i) {
  return bf16 ? bf16_to_f32(((const unsigned short *)g)[i]) : ((const float *)g)[i];
}

__global__ __launch_bounds__(kThreads) void adamw_sumsq_kernel(const AdamSeg *__restrict__ segs, const long long *__restrict__ grads,
                                                              const int2 *__restrict__ chunks, float *__restrict__ partial) {
  __shared__ float red[kThreads / 64];
  const int2 ch = chunks[blockIdx.x];  // (tensor, first element / kChunk)
  const AdamSeg s = segs[ch.x];
  const void *g = (const void *)grads[ch.x];
  float acc = 0.f;
  if (g) {
    const long long base = (long long)ch.y * kChunk;
#pragma unroll 4
    for (int j = 0; j < kPer; ++j) {
      const long long i = base + j * kThreads + threadIdx.x;
      if (i < s.n) {
        const float x = load_grad(g, s.grad_bf16, i);
        acc += x * x;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < kThreads / 64; ++k) t += red[k];
    partial[blockIdx.x] = t;
  }
}

// scalars: [0] clip scale, [1] found_inf (0 / 1), [2] step (float, as torch keeps it), [3] 1 - beta1^step, [4] sqrt(1 - beta2^step),
//          [5] total gradient norm
__global__ __launch_bounds__(1024) void adamw_scalars_kernel(const float *__restrict__ partial, int n_chunks, float max_norm,
                                                             float beta1, float beta2, float *__restrict__ scalars) {
  __shared__ double red[1024];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_chunks; i += 1024) acc += (double)partial[i];  // fixed assignment of chunks to lanes
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {  // fixed-shape tree
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    const bool bad = !(norm == norm) || norm > 3.0e38f;
    float clip = 1.f;
    if (max_norm > 0.f) {
      clip = max_norm / (norm + 1e-6f);
      if (clip > 1.f) clip = 1.f;
    }
    float step = scalars[2];
    if (!bad) step += 1.f;
    scalars[0] = clip;
    scalars[1] = bad ? 1.f : 0.f;
    scalars[2] = step;
    scalars[3] = 1.f - powf(beta1, step);
    scalars[4] = sqrtf(1.f - powf(beta2, step));
    scalars[5] = norm;
  }
}

__global__ __launch_bounds__(kThreads) void adamw_update_kernel(const AdamSeg *__restrict__ segs, const long long *__restrict__ grads,
                                                               const int2 *__restrict__ chunks, const float *__restrict__ scalars,
                                                               float lr, float beta1, float beta2, float eps, float weight_decay) {
  if (scalars[1] != 0.f) return;  // non-finite gradient norm: the whole step is skipped
  const int2 ch = chunks[blockIdx.x];
  const AdamSeg s = segs[ch.x];
  const void *g = (const void *)grads[ch.x];
  const float clip = scalars[0], bc1 = scalars[3], bc2_sqrt = scalars[4];
  const float step_size = lr / bc1;
  const long long base = (long long)ch.y * kChunk;
#pragma unroll 4
  for (int j = 0; j < kPer; ++j) {
    const long long i = base + j * kThreads + threadIdx.x;
    if (i >= s.n) continue;
    const float grad = g ? load_grad(g, s.grad_bf16, i) * clip : 0.f;
    float p = s.master[i], m = s.m[i], v = s.v[i];
    p -= lr * weight_decay * p;                              // decoupled weight decay
    m = m + (1.f - beta1) * (grad - m);                      // lerp(m, grad, 1 - beta1), as torch's fused kernel
    v = beta2 * v + (1.f - beta2) * grad * grad;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p -= step_size * m / denom;
    s.master[i] = p;
    s.m[i] = m;
    s.v[i] = v;
    if (s.lowp) s.lowp[i] = f32_to_bf16(p);
  }
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_adamw_segment_bytes(void) { return (int)sizeof(AdamSeg); }
BFHIP_EXPORT int bfhip_adamw_chunk_elems(void) { return kChunk; }

BFHIP_EXPORT int bfhip_adamw_step(const void *segs_dev, const int64_t *grad_ptrs_dev, const int32_t *chunks_dev, int n_chunks,
                                  float *partial_dev, float *scalars_dev, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, float max_norm, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  BFHIP_REQUIRE(segs_dev && grad_ptrs_dev && chunks_dev && partial_dev && scalars_dev, "adamw_step: null pointer");
  BFHIP_REQUIRE(n_chunks > 0 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f,
                "adamw_step: bad hyper-parameters");
  hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(n_chunks), dim3(kThreads), 0, s, (const AdamSeg *)segs_dev,
                     (const long long *)grad_ptrs_dev, (const int2 *)chunks_dev, partial_dev);
  hipLaunchKernelGGL(adamw_scalars_kernel, dim3(1), dim3(1024), 0, s, partial_dev, n_chunks, max_norm, beta1, beta2, scalars_dev);
  hipLaunchKernelGGL(adamw_update_kernel, dim3(n_chunks), dim3(kThreads), 0, s, (const AdamSeg *)segs_dev,
                     (const long long *)grad_ptrs_dev, (const int2 *)chunks_dev, scalars_dev, lr, beta1, beta2, eps, weight_decay);
  return check_launch("adamw_step");
}
