// head.hip -- TransFusion head: box decoding, target assignment and losses on the device (gfx950).
// SURVEY 8 row f-3.  The reference does this per sample with CPU round trips: decoded boxes and costs are built
// with torch ops, the cost matrix goes `.cpu()` into scipy's linear_sum_assignment (BF/utils.py:266-272), the
// heat-map target is drawn box by box from numpy gaussians (BF/bevfusion_head.py:643-662) and the averaging
// factor is read back with `.item()` (:718).  Here the whole batch is handled by a handful of launches with no
// host read:
//   decode_boxes_kernel      BF/utils.py:72-85 (TransFusionBBoxCoder.decode, filter=False)
//   assign_cost_kernel       FocalLossCost + BBoxBEVL1Cost + IoU3DCost (BF/utils.py:128-151,254-264); rotated BEV
//                            IoU x height overlap as M3D/structures/bbox_3d/base_box3d.py:529-590
//   hungarian_kernel         one wave per sample: shortest-augmenting-path assignment in fp64, the algorithm of
//                            scipy.optimize.linear_sum_assignment (rectangular LAPJV, Crouse 2016) with the column
//                            scan of every Dijkstra step spread over the 64 lanes
//   assign_targets_kernel    BF/bevfusion_head.py:604-633 (labels, weights, encoded boxes BF/utils.py:33-46, ious)
//   draw_heatmap_kernel      BF/bevfusion_head.py:636-662 + M3D/models/utils/gaussian.py:9-92, max-combined with
//                            an integer atomicMax (values >= 0) -> order-independent
//   gaussian_focal_*         clip_sigmoid + mmdet GaussianFocalLoss, loss and d/dlogit in one pass (:714-719)
//   query_losses_kernel      mmdet FocalLoss (sigmoid) + L1Loss over the 200 queries (:729-791)
#include <math.h>

#include "common.h"

namespace bfhip {
namespace {

// ------------------------------------------------------------------------------------ decode
// center f32[B,2,P], height [B,1,P], dim [B,3,P], rot [B,2,P], vel [B,2,P] or null -> boxes [B,P,W], W = 7 | 9
__global__ __launch_bounds__(256) void decode_boxes_kernel(const float *__restrict__ center,
                                                           const float *__restrict__ height,
                                                           const float *__restrict__ dim,
                                                           const float *__restrict__ rot,
                                                           const float *__restrict__ vel, int B, int P, int ld,
                                                           int p_off, float osf, float vx, float vy, float x0,
                                                           float y0, float *__restrict__ boxes) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * P) return;
  int b = t / P, p = t - b * P, q = p + p_off;  // ld = proposals per sample in the inputs (layers concatenated)
  const int W = vel ? 9 : 7;
  float *o = boxes + (size_t)t * W;
  o[0] = __fadd_rn(__fmul_rn(__fmul_rn(center[((size_t)b * 2 + 0) * ld + q], osf), vx), x0);
  o[1] = __fadd_rn(__fmul_rn(__fmul_rn(center[((size_t)b * 2 + 1) * ld + q], osf), vy), y0);
  float d0 = expf(dim[((size_t)b * 3 + 0) * ld + q]), d1 = expf(dim[((size_t)b * 3 + 1) * ld + q]),
        d2 = expf(dim[((size_t)b * 3 + 2) * ld + q]);
  o[2] = __fsub_rn(height[(size_t)b * ld + q], __fmul_rn(d2, 0.5f));  // gravity centre -> bottom centre
  o[3] = d0; o[4] = d1; o[5] = d2;
  o[6] = atan2f(rot[((size_t)b * 2 + 0) * ld + q], rot[((size_t)b * 2 + 1) * ld + q]);
  if (vel) {
    o[7] = vel[((size_t)b * 2 + 0) * ld + q];
    o[8] = vel[((size_t)b * 2 + 1) * ld + q];
  }
}

// ------------------------------------------------------------------------------------ rotated IoU
struct P2 { float x, y; };
__device__ __forceinline__ float cross2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float dot2(P2 a, P2 b) { return a.x * b.x + a.y * b.y; }

__device__ __forceinline__ void box_corners(float xc, float yc, float w, float h, float a, P2 *c, P2 &u, P2 &v) {
  float cs = cosf(a), sn = sinf(a);
  u = {cs, sn}; v = {-sn, cs};
  float ux = cs * w * 0.5f, uy = sn * w * 0.5f, vx = -sn * h * 0.5f, vy = cs * h * 0.5f;
  c[0] = {xc - ux - vx, yc - uy - vy};
  c[1] = {xc + ux - vx, yc + uy - vy};
  c[2] = {xc + ux + vx, yc + uy + vy};
  c[3] = {xc - ux + vx, yc - uy + vy};
}

// area of the intersection of two rotated rectangles (x, y, w, h, angle)
__device__ float rotated_intersection(float x1, float y1, float w1, float h1, float a1, float x2, float y2,
                                      float w2, float h2, float a2) {
  const float mx = (x1 + x2) * 0.5f, my = (y1 + y2) * 0.5f;  // centre the pair: keeps fp32 cancellation small
  x1 -= mx; y1 -= my; x2 -= mx; y2 -= my;
  P2 c1[4], c2[4], u1, v1, u2, v2;
  box_corners(x1, y1, w1, h1, a1, c1, u1, v1);
  box_corners(x2, y2, w2, h2, a2, c2, u2, v2);
  P2 pts[24];
  int n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    P2 A = c1[i], d1 = {c1[(i + 1) & 3].x - A.x, c1[(i + 1) & 3].y - A.y};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      P2 C = c2[j], d2 = {c2[(j + 1) & 3].x - C.x, c2[(j + 1) & 3].y - C.y};
      float det = cross2(d1, d2);
      if (fabsf(det) <= 1e-14f) continue;
      P2 ac = {C.x - A.x, C.y - A.y};
      float t1 = cross2(ac, d2) / det, t2 = cross2(ac, d1) / det;
      if (t1 >= 0.f && t1 <= 1.f && t2 >= 0.f && t2 <= 1.f) pts[n++] = {A.x + t1 * d1.x, A.y + t1 * d1.y};
    }
  }
  const float tol = 1e-5f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    P2 d = {c1[i].x - x2, c1[i].y - y2};
    if (fabsf(dot2(d, u2)) <= w2 * 0.5f + tol && fabsf(dot2(d, v2)) <= h2 * 0.5f + tol) pts[n++] = c1[i];
    P2 e = {c2[i].x - x1, c2[i].y - y1};
    if (fabsf(dot2(e, u1)) <= w1 * 0.5f + tol && fabsf(dot2(e, v1)) <= h1 * 0.5f + tol) pts[n++] = c2[i];
  }
  if (n < 3) return 0.f;
  float cx = 0.f, cy = 0.f;
  for (int i = 0; i < n; ++i) { cx += pts[i].x; cy += pts[i].y; }
  cx /= (float)n; cy /= (float)n;
  float ang[24];
  for (int i = 0; i < n; ++i) ang[i] = atan2f(pts[i].y - cy, pts[i].x - cx);
  for (int i = 1; i < n; ++i) {  // insertion sort by angle: the points lie on the boundary of a convex polygon
    float a = ang[i];
    P2 p = pts[i];
    int j = i - 1;
    while (j >= 0 && ang[j] > a) { ang[j + 1] = ang[j]; pts[j + 1] = pts[j]; --j; }
    ang[j + 1] = a; pts[j + 1] = p;
  }
  float area = 0.f;
  for (int i = 0; i < n; ++i) {
    P2 p = pts[i], q = pts[i + 1 == n ? 0 : i + 1];
    area += p.x * q.y - p.y * q.x;
  }
  return fabsf(area) * 0.5f;
}

// base_box3d.py:560-590 on (x, y, z_bottom, dx, dy, dz, yaw)
__device__ float iou3d_lidar(const float *a, const float *b) {
  float top = fminf(a[2] + a[5], b[2] + b[5]), bot = fmaxf(a[2], b[2]);
  float oh = fmaxf(top - bot, 0.f);
  float w1 = fmaxf(a[3], 1e-4f), h1 = fmaxf(a[4], 1e-4f), w2 = fmaxf(b[3], 1e-4f), h2 = fmaxf(b[4], 1e-4f);
  float inter = rotated_intersection(a[0], a[1], w1, h1, a[6], b[0], b[1], w2, h2, b[6]);
  float ar1 = w1 * h1, ar2 = w2 * h2;
  float iou2d = inter > 0.f ? inter / (ar1 + ar2 - inter) : 0.f;
  float obev = iou2d * (ar1 + ar2) / (1.f + iou2d);
  float o3d = obev * oh;
  float v1 = a[3] * a[4] * a[5], v2 = b[3] * b[4] * b[5];
  return o3d / fmaxf(v1 + v2 - o3d, 1e-8f);
}

struct CostCfg { float cls_w, alpha, gamma, eps, reg_w, iou_w, x0, y0, rx, ry; };

// one thread per (b, p, g): cost[b][p][g], iou[b][p][g]; padded columns (g >= n_gt[b]) get 0
__global__ __launch_bounds__(128) void assign_cost_kernel(const float *__restrict__ boxes, int W,
                                                          const float *__restrict__ logits, int C, int ld,
                                                          int p_off, const float *__restrict__ gt, int Wg,
                                                          const int *__restrict__ gt_labels,
                                                          const int *__restrict__ n_gt, int B, int P, int G,
                                                          CostCfg cfg, float *__restrict__ cost,
                                                          float *__restrict__ iou) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * P * G) return;
  int g = (int)(t % G);
  long long r = t / G;
  int p = (int)(r % P), b = (int)(r / P);
  if (g >= n_gt[b]) { cost[t] = 0.f; iou[t] = 0.f; return; }
  const float *pb = boxes + ((size_t)b * P + p) * W;
  const float *gb = gt + ((size_t)b * G + g) * Wg;
  int lab = gt_labels[(size_t)b * G + g];
  lab = lab < 0 ? 0 : (lab >= C ? C - 1 : lab);
  // FocalLossCost
  float x = logits[((size_t)b * C + lab) * ld + p_off + p];
  float s = 1.f / (1.f + expf(-x));
  float sp = cfg.gamma == 2.f ? s * s : powf(s, cfg.gamma);
  float sn = cfg.gamma == 2.f ? (1.f - s) * (1.f - s) : powf(1.f - s, cfg.gamma);
  float neg = -logf(1.f - s + cfg.eps) * (1.f - cfg.alpha) * sp;
  float pos = -logf(s + cfg.eps) * cfg.alpha * sn;
  float cls_cost = (pos - neg) * cfg.cls_w;
  // BBoxBEVL1Cost
  float ax = (pb[0] - cfg.x0) / cfg.rx, ay = (pb[1] - cfg.y0) / cfg.ry;
  float bx = (gb[0] - cfg.x0) / cfg.rx, by = (gb[1] - cfg.y0) / cfg.ry;
  float reg_cost = (fabsf(ax - bx) + fabsf(ay - by)) * cfg.reg_w;
  float v = iou3d_lidar(pb, gb);
  iou[t] = v;
  cost[t] = cls_cost + reg_cost + (-v * cfg.iou_w);
}

// ------------------------------------------------------------------------------------ Hungarian
// One wave per sample.  rows = the smaller side (scipy transposes when it has more rows than columns).
// stage_cap: floats of LDS behind the work arrays for a copy of the sample's cost matrix (0 = none).  Every Dijkstra step reads
// one row of the matrix; from global memory that is a dependent ~1.5 us round trip per step, ~200 steps per sample (0.26 ms for the
// batch, one wave per sample, on the critical path between the decoder and the losses); from LDS ~0.1 us.
__global__ __launch_bounds__(64) void hungarian_kernel(const float *__restrict__ cost,
                                                       const int *__restrict__ n_gt, int P, int G, int M,
                                                       int *__restrict__ assigned, int *__restrict__ status, int stage_cap) {
  extern __shared__ double smem[];
  const int b = blockIdx.x, lane = threadIdx.x;
  double *u = smem, *v = u + M, *spc = v + M;
  int *path = (int *)(spc + M), *row4col = path + M, *col4row = row4col + M, *SR = col4row + M, *SC = SR + M;
  float *cs = (float *)(SC + M);  // [nr][nc] when staged
  int g = n_gt[b];
  g = g < 0 ? 0 : (g > G ? G : g);
  const float *Cb = cost + (size_t)b * P * G;
  int *out = assigned + (size_t)b * P;
  for (int p = lane; p < P; p += 64) out[p] = 0;
  if (g == 0) {
    if (lane == 0) status[b] = 0;
    return;
  }
  const bool tr = P > g;
  const int nr = tr ? g : P, nc = tr ? P : g;
  const bool staged = (long long)P * g <= stage_cap;
  for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; }
  __syncthreads();
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  int err = 0;
  {  // scipy's linear_sum_assignment rejects a matrix with ANY NaN or -inf entry ("matrix contains invalid numeric entries")
    int bad = 0;
    for (int e = lane; e < P * g; e += 64) {
      const int pp = e / g, gg = e - pp * g;
      const float c = Cb[(size_t)pp * G + gg];
      bad |= (c != c) || (c == -__builtin_inff());
      if (staged) cs[tr ? gg * nc + pp : e] = c;  // rows = the smaller side
    }
    err = __any(bad) ? 1 : 0;
    __syncthreads();
  }
  for (int cur = 0; cur < nr && !err; ++cur) {
    for (int j = lane; j < nc; j += 64) { spc[j] = INF; SC[j] = 0; }
    for (int i = lane; i < nr; i += 64) SR[i] = 0;
    __syncthreads();
    double minVal = 0.0;
    int i = cur, sink = -1;
    for (int it = 0; it < nc && sink < 0; ++it) {  // every step closes one column: at most nc steps
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      double best = INF;
      int bestj = -1, bestfree = 0;
      for (int j = lane; j < nc; j += 64) {
        if (SC[j]) continue;
        double c = staged ? (double)cs[i * nc + j] : (tr ? (double)Cb[(size_t)j * G + i] : (double)Cb[(size_t)i * G + j]);
        double r = minVal + c - ui - v[j];
        double s = spc[j];
        if (r < s) { spc[j] = r; path[j] = i; s = r; }
        int fr = row4col[j] < 0;
        // ties: a column that ends the path first (as scipy), then the lowest index
        if (bestj < 0 ? (s < INF) : (s < best || (s == best && fr && !bestfree))) { best = s; bestj = j; bestfree = fr; }
      }
      for (int off = 32; off; off >>= 1) {
        double ob = __shfl_xor(best, off);
        int oj = __shfl_xor(bestj, off), of = __shfl_xor(bestfree, off);
        bool take = oj >= 0 && (bestj < 0 || ob < best ||
                                (ob == best && (of > bestfree || (of == bestfree && oj < bestj))));
        if (take) { best = ob; bestj = oj; bestfree = of; }
      }
      if (bestj < 0) { err = 1; break; }  // NaN / inf costs: no admissible column (wave-uniform)
      minVal = best;
      const int r4 = row4col[bestj];
      __syncthreads();
      if (lane == 0) SC[bestj] = 1;
      if (r4 < 0) sink = bestj; else i = r4;
      __syncthreads();
    }
    if (err || sink < 0) { err = 1; break; }
    if (lane == 0) u[cur] += minVal;
    for (int i2 = lane; i2 < nr; i2 += 64)
      if (SR[i2] && i2 != cur) u[i2] += minVal - spc[col4row[i2]];
    for (int j = lane; j < nc; j += 64)
      if (SC[j]) v[j] -= minVal - spc[j];
    __syncthreads();
    if (lane == 0) {
      int j = sink;
      for (int guard = 0; guard <= nr; ++guard) {
        int i3 = path[j];
        row4col[j] = i3;
        int t = col4row[i3];
        col4row[i3] = j;
        j = t;
        if (i3 == cur) break;
      }
    }
    __syncthreads();
  }
  if (!err) {
    for (int i = lane; i < nr; i += 64) {
      int j = col4row[i];
      if (j < 0) continue;
      if (tr) out[j] = i + 1; else out[i] = j + 1;  // 0 = background, g + 1 = matched GT (AssignResult)
    }
  }
  if (lane == 0) status[b] = err;
}

// ------------------------------------------------------------------------------------ targets
struct EncCfg { float x0, y0, divx, divy; int num_classes, code; float pos_weight; };

__global__ __launch_bounds__(256) void assign_targets_kernel(const int *__restrict__ assigned,
                                                             const float *__restrict__ iou,
                                                             const float *__restrict__ gt, int Wg,
                                                             const int *__restrict__ gt_labels, int B, int P,
                                                             int G, EncCfg cfg, int *__restrict__ labels,
                                                             float *__restrict__ label_weights,
                                                             float *__restrict__ bbox_targets,
                                                             float *__restrict__ bbox_weights,
                                                             float *__restrict__ ious) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * P) return;
  int b = t / P;
  int a = assigned[t];
  float *bt = bbox_targets + (size_t)t * cfg.code, *bw = bbox_weights + (size_t)t * cfg.code;
  if (a <= 0) {
    labels[t] = cfg.num_classes;
    label_weights[t] = a == 0 ? 1.f : 0.f;
    ious[t] = 0.f;
    for (int k = 0; k < cfg.code; ++k) { bt[k] = 0.f; bw[k] = 0.f; }
    return;
  }
  int g = a - 1;
  const float *gb = gt + ((size_t)b * G + g) * Wg;
  labels[t] = gt_labels[(size_t)b * G + g];
  label_weights[t] = cfg.pos_weight <= 0.f ? 1.f : cfg.pos_weight;
  float v = iou[(size_t)t * G + g];
  ious[t] = fminf(fmaxf(v, 0.f), 1.f);
  bt[0] = __fdiv_rn(__fsub_rn(gb[0], cfg.x0), cfg.divx);
  bt[1] = __fdiv_rn(__fsub_rn(gb[1], cfg.y0), cfg.divy);
  bt[2] = __fadd_rn(gb[2], __fmul_rn(gb[5], 0.5f));
  bt[3] = logf(gb[3]); bt[4] = logf(gb[4]); bt[5] = logf(gb[5]);
  bt[6] = sinf(gb[6]); bt[7] = cosf(gb[6]);
  if (cfg.code == 10) { bt[8] = Wg > 7 ? gb[7] : 0.f; bt[9] = Wg > 8 ? gb[8] : 0.f; }
  for (int k = 0; k < cfg.code; ++k) bw[k] = 1.f;
}

// ------------------------------------------------------------------------------------ heat-map target
struct HeatCfg {
  float x0, y0, vx, vy, osf;
  float c1m, c1p, c2m, b3m, c3m, a3x4;  // (1-o), (1+o), (1-o), -2o, (o-1), 16o  as fp32 constants
  int min_radius;
};

// one block per (b, g); heat[b][cls][row = x cell][col = y cell]  (the reference's center_int[[1, 0]] fix, :662)
__global__ __launch_bounds__(256) void draw_heatmap_kernel(const float *__restrict__ gt, int Wg,
                                                           const int *__restrict__ gt_labels,
                                                           const int *__restrict__ n_gt, int G, int NC, int H,
                                                           int Wd, HeatCfg c, float *__restrict__ heat) {
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  if (g >= n_gt[b]) return;
  const float *gb = gt + ((size_t)b * G + g) * Wg;
  int lab = gt_labels[(size_t)b * G + g];
  if (lab < 0 || lab >= NC) return;
  const float width = __fdiv_rn(__fdiv_rn(gb[3], c.vx), c.osf), length = __fdiv_rn(__fdiv_rn(gb[4], c.vy), c.osf);
  if (!(width > 0.f && length > 0.f)) return;
  // gaussian_radius((length, width)) -> height = length, width = width   (gaussian.py:62-92, fp32)
  const float hh = length, ww = width;
  float b1 = __fadd_rn(hh, ww);
  float c1 = __fdiv_rn(__fmul_rn(__fmul_rn(ww, hh), c.c1m), c.c1p);
  float r1 = __fdiv_rn(__fadd_rn(b1, __fsqrt_rn(__fsub_rn(__fmul_rn(b1, b1), __fmul_rn(4.f, c1)))), 2.f);
  float b2 = __fmul_rn(2.f, __fadd_rn(hh, ww));
  float c2 = __fmul_rn(__fmul_rn(c.c2m, ww), hh);
  float r2 = __fdiv_rn(__fadd_rn(b2, __fsqrt_rn(__fsub_rn(__fmul_rn(b2, b2), __fmul_rn(16.f, c2)))), 2.f);
  float b3 = __fmul_rn(c.b3m, __fadd_rn(hh, ww));
  float c3 = __fmul_rn(__fmul_rn(c.c3m, ww), hh);
  float r3 = __fdiv_rn(__fadd_rn(b3, __fsqrt_rn(__fsub_rn(__fmul_rn(b3, b3), __fmul_rn(c.a3x4, c3)))), 2.f);
  float rf = fminf(r1, fminf(r2, r3));
  int radius = (int)rf;
  radius = radius < c.min_radius ? c.min_radius : radius;
  const int cx = (int)__fdiv_rn(__fdiv_rn(__fsub_rn(gb[0], c.x0), c.vx), c.osf);
  const int cy = (int)__fdiv_rn(__fdiv_rn(__fsub_rn(gb[1], c.y0), c.vy), c.osf);
  const int d = 2 * radius + 1;
  const double sigma = (double)d / 6.0, den = 2.0 * sigma * sigma;
  int *plane = (int *)(heat + ((size_t)b * NC + lab) * H * Wd);
  for (int t = threadIdx.x; t < d * d; t += blockDim.x) {
    int dr = t / d - radius, dc = t % d - radius;  // dr along the row axis (x cells), dc along columns (y cells)
    int row = cx + dr, col = cy + dc;
    if (row < 0 || row >= H || col < 0 || col >= Wd) continue;
    float val = (float)exp(-((double)(dc * dc) + (double)(dr * dr)) / den);
    atomicMax(&plane[(size_t)row * Wd + col], __float_as_int(val));  // val >= 0: int order == float order
  }
}

// ------------------------------------------------------------------------------------ losses
__device__ __forceinline__ double block_sum(double v, double *s) {
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s[w] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += s[i];  // fixed order -> deterministic
  return r;
}

// clip_sigmoid (BF/bevfusion_head.py:20-23) + mmdet GaussianFocalLoss(alpha 2, gamma 4), elementwise part.
// grad[i] = d loss_i / d logit_i (unscaled); partial[block] = (sum of loss, count of target == 1)
__global__ __launch_bounds__(256) void gaussian_focal_kernel(const float *__restrict__ logits,
                                                             const float *__restrict__ target, long long n,
                                                             float clip, float *__restrict__ grad,
                                                             double *__restrict__ partial) {
  __shared__ double s[4];
  double acc = 0.0, cnt = 0.0;
  const float eps = 1e-12f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float x = logits[i], t = target[i];
    float sg = 1.f / (1.f + expf(-x));
    float p = fminf(fmaxf(sg, clip), 1.f - clip);
    float dpdx = (sg > clip && sg < 1.f - clip) ? sg * (1.f - sg) : 0.f;
    float loss, dl;
    if (t == 1.f) {
      float q = 1.f - p, lg = logf(p + eps);
      loss = -lg * q * q;
      dl = -q * q / (p + eps) + 2.f * q * lg;
      cnt += 1.0;
    } else {
      float q = 1.f - t, w = (q * q) * (q * q), lg = logf(1.f - p + eps);
      loss = -lg * p * p * w;
      dl = (p * p / (1.f - p + eps) - 2.f * p * lg) * w;
    }
    acc += (double)loss;
    grad[i] = dl * dpdx;
  }
  double a = block_sum(acc, s), c = block_sum(cnt, s);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = c; }
}

__global__ __launch_bounds__(256) void reduce_pairs_kernel(const double *__restrict__ partial, int nblocks,
                                                           float *__restrict__ out) {
  __shared__ double s[4];
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { a += partial[2 * i]; c += partial[2 * i + 1]; }
  a = block_sum(a, s);
  c = block_sum(c, s);
  if (threadIdx.x == 0) { out[0] = (float)a; out[1] = (float)c; }
}

// mmdet FocalLoss(use_sigmoid) on logits [B,C,ld] (queries p_off .. p_off+P) with labels [B,P] (C = background) and
// row weights [B,P]; mmdet L1Loss on pred [B,K,ld] vs targets [B,P,K] with weights [B,P,K] * code_weights[K].
// Single block: sums in out[0] (cls), out[1] (bbox); unscaled gradients in the layouts of the inputs.
__global__ __launch_bounds__(1024) void query_losses_kernel(const float *__restrict__ logits,
                                                            const int *__restrict__ labels,
                                                            const float *__restrict__ label_weights,
                                                            const float *__restrict__ pred,
                                                            const float *__restrict__ bbox_targets,
                                                            const float *__restrict__ bbox_weights,
                                                            const float *__restrict__ code_weights, int B, int C,
                                                            int P, int K, int ld, int p_off, float gamma,
                                                            float alpha, float *__restrict__ grad_cls,
                                                            float *__restrict__ grad_box,
                                                            float *__restrict__ out) {
  __shared__ double s[16];
  double acc = 0.0;
  for (int t = threadIdx.x; t < B * C * P; t += blockDim.x) {
    int p = t % P, c = (t / P) % C, b = t / (P * C);
    size_t src = ((size_t)b * C + c) * ld + p_off + p;
    float x = logits[src];
    float w = label_weights[b * P + p];
    bool tgt = labels[b * P + p] == c;
    float sg = 1.f / (1.f + expf(-x));
    // softplus forms keep log(sigmoid) finite for large |x|
    float log_p = -(fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))));   // log sigmoid(x)
    float log_q = -(fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))));    // log (1 - sigmoid(x))
    float loss, g;
    if (tgt) {
      float q = 1.f - sg, qg = gamma == 2.f ? q * q : powf(q, gamma);
      loss = -alpha * qg * log_p;
      // d/dx: -alpha * [ -gamma q^(g-1) * sg q * log_p + q^g * q ]
      float qg1 = gamma == 2.f ? q : powf(q, gamma - 1.f);
      g = -alpha * (-gamma * qg1 * sg * q * log_p + qg * q);
    } else {
      float pg = gamma == 2.f ? sg * sg : powf(sg, gamma);
      loss = -(1.f - alpha) * pg * log_q;
      float pg1 = gamma == 2.f ? sg : powf(sg, gamma - 1.f);
      g = -(1.f - alpha) * (gamma * pg1 * sg * (1.f - sg) * log_q - pg * sg);
    }
    acc += (double)(loss * w);
    grad_cls[src] = g * w;
  }
  double cls_sum = block_sum(acc, s);
  acc = 0.0;
  for (int t = threadIdx.x; t < B * K * P; t += blockDim.x) {
    int p = t % P, k = (t / P) % K, b = t / (P * K);
    size_t src = ((size_t)b * K + k) * ld + p_off + p;
    size_t ti = ((size_t)b * P + p) * K + k;
    float w = bbox_weights[ti] * code_weights[k];
    float d = pred[src] - bbox_targets[ti];
    acc += (double)(fabsf(d) * w);
    grad_box[src] = d > 0.f ? w : (d < 0.f ? -w : 0.f);
  }
  double box_sum = block_sum(acc, s);
  if (threadIdx.x == 0) { out[0] = (float)cls_sum; out[1] = (float)box_sum; }
}


// ------------------------------------------------------------------------------------ circle NMS
// mmdet3d/models/layers/box3d_nms.py:186-228: greedy, highest score first; j is suppressed when the SQUARED centre distance is
// <= thresh (the reference compares the squared distance with the radius as given; kept).  One workgroup: rank by
// counting (ties -> lower index first), then the sequential sweep with the inner loop spread over the threads.
__global__ __launch_bounds__(256) void circle_nms_kernel(const float *__restrict__ dets, int n, float thresh,
                                                         int post_max, int *__restrict__ keep, int *__restrict__ n_keep) {
  extern __shared__ int sm_i[];
  int *order = sm_i;                 // [n] index of the i-th highest score
  int *supp = order + n;             // [n] by sorted position
  float *sx = (float *)(supp + n), *sy = sx + n;  // [n] centres in sorted order
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float si = dets[i * 3 + 2];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float sj = dets[j * 3 + 2];
      rank += (sj > si) || (sj == si && j < i);
    }
    order[rank] = i;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    supp[i] = 0;
    sx[i] = dets[order[i] * 3];
    sy[i] = dets[order[i] * 3 + 1];
  }
  __syncthreads();
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    if (supp[i]) continue;  // uniform: every thread reads the same LDS word after the barrier below
    if (threadIdx.x == 0 && kept < post_max) keep[kept] = order[i];
    ++kept;
    const float xi = sx[i], yi = sy[i];
    for (int j = i + 1 + threadIdx.x; j < n; j += blockDim.x) {
      const float dx = xi - sx[j], dy = yi - sy[j];
      if (dx * dx + dy * dy <= thresh) supp[j] = 1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_keep = kept < post_max ? kept : post_max;
}

// ---------------------------------------------------------------------------------------- rotate NMS
// nms_bev (mmdet3d/models/layers/box3d_nms.py:234-275) -> mmcv.ops.nms_rotated (third-party, mmcv 2.x; restated from its
// published algorithm): sort by score, keep the first pre_max, IoU of rotated rectangles (x, y, w, h, angle) on the
// exact intersection polygon, a box is suppressed by an earlier kept box when IoU > thresh.  Three launches:
//   rank+stage (one thread per box)  ->  suppression bit matrix (one thread per (row, 64-column word))  ->  serial sweep (one wave).
__global__ __launch_bounds__(256) void rnms_rank_kernel(const float *__restrict__ boxes, const float *__restrict__ scores,
                                                        int n, int m, int *__restrict__ order, float *__restrict__ sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float si = scores[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float sj = scores[j];
      rank += (sj > si) || (sj == si && j < i);
    }
    if (rank < m) {
      order[rank] = i;
#pragma unroll
      for (int c = 0; c < 5; ++c) sorted[rank * 5 + c] = boxes[i * 5 + c];
    }
  }
}

__device__ __forceinline__ float iou_rotated(const float *a, const float *b) {
  const float ar1 = a[2] * a[3], ar2 = b[2] * b[3];
  if (ar1 < 1e-14f || ar2 < 1e-14f) return 0.f;
  const float inter = rotated_intersection(a[0], a[1], a[2], a[3], a[4], b[0], b[1], b[2], b[3], b[4]);
  return inter / (ar1 + ar2 - inter);
}

// mask[i][w] bit b: sorted box j = 64 w + b (j > i) overlaps sorted box i by more than thresh
__global__ __launch_bounds__(64) void rnms_mask_kernel(const float *__restrict__ sorted, int m, int words, float thresh,
                                                       unsigned long long *__restrict__ mask) {
  __shared__ float cols[64 * 5];
  const int w = blockIdx.x, i = blockIdx.y * 64 + threadIdx.x;
  if (w * 64 + 63 <= (int)blockIdx.y * 64) {  // whole word at or left of the diagonal for every row of this block
    if (i < m) mask[(size_t)i * words + w] = 0ull;
    return;
  }
  const int ncol = min(64, m - w * 64);
  for (int t = threadIdx.x; t < ncol * 5; t += 64) cols[t] = sorted[(size_t)w * 64 * 5 + t];
  __syncthreads();
  if (i >= m) return;
  float a[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) a[c] = sorted[(size_t)i * 5 + c];
  unsigned long long bits = 0ull;
  for (int b = 0; b < ncol; ++b) {
    const int j = w * 64 + b;
    if (j <= i) continue;
    if (iou_rotated(a, cols + b * 5) > thresh) bits |= 1ull << b;
  }
  mask[(size_t)i * words + w] = bits;
}

// one wave: lane l owns removed-word l (+64, ... for m > 4096 is excluded by the host check)
__global__ __launch_bounds__(64) void rnms_sweep_kernel(const unsigned long long *__restrict__ mask,
                                                        const int *__restrict__ order, int m, int words, int post_max,
                                                        int *__restrict__ keep, int *__restrict__ n_keep) {
  const int lane = threadIdx.x;
  unsigned long long removed = 0ull;
  int kept = 0;
  for (int i = 0; i < m; ++i) {
    const unsigned long long wi = __shfl(removed, i >> 6);
    if ((wi >> (i & 63)) & 1ull) continue;  // wave-uniform
    if (lane == 0 && kept < post_max) keep[kept] = order[i];
    ++kept;
    if (lane < words) removed |= mask[(size_t)i * words + lane];
  }
  if (lane == 0) *n_keep = kept < post_max ? kept : post_max;
}

}  // namespace
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_circle_nms(const float *dets, int n, float thresh, int post_max_size, int32_t *keep,
                                  int32_t *n_keep, void *stream) {
  BFHIP_REQUIRE(n_keep && (n == 0 || (dets && keep)) && n >= 0 && n <= 4096 && post_max_size >= 0, "circle_nms: bad arguments (n <= 4096)");
  if (n == 0) {
    if (hipMemsetAsync(n_keep, 0, sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return check_launch("circle_nms memset");
    return BFHIP_OK;
  }
  hipLaunchKernelGGL(circle_nms_kernel, dim3(1), dim3(256), (size_t)n * 16, (hipStream_t)stream, dets, n, thresh, post_max_size,
                     keep, n_keep);
  return check_launch("circle_nms");
}

BFHIP_EXPORT size_t bfhip_rotate_nms_workspace_bytes(int n, int pre_max_size) {
  const size_t m = (size_t)(n < pre_max_size ? n : pre_max_size), words = (m + 63) / 64;
  return align_up(m * sizeof(int32_t), 256) + align_up(m * 5 * sizeof(float), 256) + align_up(m * words * 8, 256);
}

BFHIP_EXPORT int bfhip_rotate_nms(const float *boxes, const float *scores, int n, float thresh, int pre_max_size,
                                  int post_max_size, int32_t *keep, int32_t *n_keep, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  BFHIP_REQUIRE(n_keep && n >= 0 && pre_max_size >= 0 && post_max_size >= 0, "rotate_nms: bad arguments");
  const int m = n < pre_max_size ? n : pre_max_size;
  hipStream_t s = (hipStream_t)stream;
  if (m == 0) {
    if (hipMemsetAsync(n_keep, 0, sizeof(int32_t), s) != hipSuccess) return check_launch("rotate_nms memset");
    return BFHIP_OK;
  }
  BFHIP_REQUIRE(boxes && scores && keep && workspace, "rotate_nms: null pointer");
  BFHIP_REQUIRE(n <= 16384 && m <= 4096, "rotate_nms: n=%d (<= 16384), min(n, pre_max_size)=%d (<= 4096)", n, m);
  BFHIP_REQUIRE(workspace_bytes >= bfhip_rotate_nms_workspace_bytes(n, pre_max_size), "rotate_nms: workspace too small");
  const int words = (m + 63) / 64;
  char *w = (char *)workspace;
  int *order = (int *)w;                w += align_up((size_t)m * sizeof(int32_t), 256);
  float *sorted = (float *)w;           w += align_up((size_t)m * 5 * sizeof(float), 256);
  unsigned long long *mask = (unsigned long long *)w;
  hipLaunchKernelGGL(rnms_rank_kernel, dim3((n + 255) / 256), dim3(256), 0, s, boxes, scores, n, m, order, sorted);
  hipLaunchKernelGGL(rnms_mask_kernel, dim3(words, words), dim3(64), 0, s, sorted, m, words, thresh, mask);
  hipLaunchKernelGGL(rnms_sweep_kernel, dim3(1), dim3(64), 0, s, mask, order, m, words, post_max_size, keep, n_keep);
  return check_launch("rotate_nms");
}

BFHIP_EXPORT int bfhip_decode_boxes(const float *center, const float *height, const float *dim, const float *rot,
                                    const float *vel, int B, int P, int ld, int p_off,
                                    const float *cfg_host, float *boxes, void *stream) {
  BFHIP_REQUIRE(center && height && dim && rot && boxes && cfg_host, "decode_boxes: null pointer");
  BFHIP_REQUIRE(B > 0 && P > 0 && ld >= P && p_off >= 0 && p_off + P <= ld, "decode_boxes: bad sizes B=%d P=%d ld=%d off=%d", B, P, ld, p_off);
  hipLaunchKernelGGL(decode_boxes_kernel, dim3(ceil_div((long long)B * P, 256)), dim3(256), 0, (hipStream_t)stream,
                     center, height, dim, rot, vel, B, P, ld, p_off, cfg_host[0], cfg_host[1], cfg_host[2],
                     cfg_host[3], cfg_host[4], boxes);
  return check_launch("decode_boxes");
}

BFHIP_EXPORT int bfhip_assign_cost(const float *boxes, int W, const float *cls_logits, int C, int ld, int p_off,
                                   const float *gt_boxes, int Wg, const int32_t *gt_labels, const int32_t *n_gt,
                                   int B, int P, int G, const float *cfg_host, float *cost, float *iou,
                                   void *stream) {
  BFHIP_REQUIRE(boxes && cls_logits && gt_boxes && gt_labels && n_gt && cfg_host && cost && iou, "assign_cost: null pointer");
  BFHIP_REQUIRE(B > 0 && P > 0 && G > 0 && W >= 7 && Wg >= 7 && C > 0 && ld >= P && p_off >= 0 && p_off + P <= ld,
                "assign_cost: bad sizes B=%d P=%d G=%d W=%d Wg=%d C=%d", B, P, G, W, Wg, C);
  CostCfg c;
  c.cls_w = cfg_host[0]; c.alpha = cfg_host[1]; c.gamma = cfg_host[2]; c.eps = cfg_host[3];
  c.reg_w = cfg_host[4]; c.iou_w = cfg_host[5];
  c.x0 = cfg_host[6]; c.y0 = cfg_host[7]; c.rx = cfg_host[8] - cfg_host[6]; c.ry = cfg_host[9] - cfg_host[7];
  BFHIP_REQUIRE(c.rx > 0.f && c.ry > 0.f, "assign_cost: empty point cloud range");
  hipLaunchKernelGGL(assign_cost_kernel, dim3(ceil_div((long long)B * P * G, 128)), dim3(128), 0, (hipStream_t)stream,
                     boxes, W, cls_logits, C, ld, p_off, gt_boxes, Wg, gt_labels, n_gt, B, P, G, c, cost, iou);
  return check_launch("assign_cost");
}

BFHIP_EXPORT int bfhip_hungarian(const float *cost, const int32_t *n_gt, int B, int P, int G, int32_t *assigned,
                                 int32_t *status, void *stream) {
  BFHIP_REQUIRE(cost && n_gt && assigned && status, "hungarian: null pointer");
  const int M = P > G ? P : G;
  BFHIP_REQUIRE(B > 0 && P > 0 && G > 0 && M <= 1024, "hungarian: bad sizes B=%d P=%d G=%d (max side 1024)", B, P, G);
  size_t lds = (size_t)M * (3 * sizeof(double) + 5 * sizeof(int));
  // the cost matrix of a sample in LDS when it fits beside the work arrays (nuScenes: 200 x <= 100 floats = 80 KB)
  int stage_cap = 0;
  if (lds + (size_t)P * G * sizeof(float) <= 150 * 1024) {
    stage_cap = P * G;
    lds += (size_t)P * G * sizeof(float);
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)hungarian_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(hungarian_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, cost, n_gt, P, G, M, assigned,
                     status, stage_cap);
  return check_launch("hungarian");
}

BFHIP_EXPORT int bfhip_assign_targets(const int32_t *assigned, const float *iou, const float *gt_boxes, int Wg,
                                      const int32_t *gt_labels, int B, int P, int G, int num_classes, int code_size,
                                      const float *cfg_host, int32_t *labels, float *label_weights,
                                      float *bbox_targets, float *bbox_weights, float *ious, void *stream) {
  BFHIP_REQUIRE(assigned && iou && gt_boxes && gt_labels && cfg_host && labels && label_weights && bbox_targets &&
                    bbox_weights && ious, "assign_targets: null pointer");
  BFHIP_REQUIRE(B > 0 && P > 0 && G > 0 && Wg >= 7 && (code_size == 8 || code_size == 10),
                "assign_targets: bad sizes B=%d P=%d G=%d Wg=%d code=%d", B, P, G, Wg, code_size);
  EncCfg c;
  c.x0 = cfg_host[0]; c.y0 = cfg_host[1]; c.divx = cfg_host[2]; c.divy = cfg_host[3];
  c.pos_weight = cfg_host[4]; c.num_classes = num_classes; c.code = code_size;
  hipLaunchKernelGGL(assign_targets_kernel, dim3(ceil_div((long long)B * P, 256)), dim3(256), 0, (hipStream_t)stream,
                     assigned, iou, gt_boxes, Wg, gt_labels, B, P, G, c, labels, label_weights, bbox_targets,
                     bbox_weights, ious);
  return check_launch("assign_targets");
}

BFHIP_EXPORT int bfhip_draw_heatmap(const float *gt_boxes, int Wg, const int32_t *gt_labels, const int32_t *n_gt,
                                    int B, int G, int num_classes, int H, int W, const float *cfg_host,
                                    double gaussian_overlap, int min_radius, float *heatmap, void *stream) {
  BFHIP_REQUIRE(gt_boxes && gt_labels && n_gt && cfg_host && heatmap, "draw_heatmap: null pointer");
  BFHIP_REQUIRE(B > 0 && G > 0 && Wg >= 5 && num_classes > 0 && H > 0 && W > 0, "draw_heatmap: bad sizes");
  HeatCfg c;
  c.x0 = cfg_host[0]; c.y0 = cfg_host[1]; c.vx = cfg_host[2]; c.vy = cfg_host[3]; c.osf = cfg_host[4];
  const double o = gaussian_overlap;  // the fp32 constants below are rounded from double, as python does
  BFHIP_REQUIRE(c.vx > 0.f && c.vy > 0.f && c.osf > 0.f && o > 0.0 && o < 1.0, "draw_heatmap: bad config");
  c.c1m = (float)(1 - o); c.c1p = (float)(1 + o); c.c2m = (float)(1 - o);
  c.b3m = (float)(-2 * o); c.c3m = (float)(o - 1); c.a3x4 = (float)(4 * (4 * o));
  c.min_radius = min_radius;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(heatmap, 0, (size_t)B * num_classes * H * W * sizeof(float), s) != hipSuccess) {
    set_error("draw_heatmap: memset failed");
    return BFHIP_E_LAUNCH;
  }
  hipLaunchKernelGGL(draw_heatmap_kernel, dim3(B * G), dim3(256), 0, s, gt_boxes, Wg, gt_labels, n_gt, G, num_classes,
                     H, W, c, heatmap);
  return check_launch("draw_heatmap");
}

static int focal_blocks(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

BFHIP_EXPORT size_t bfhip_gaussian_focal_loss_workspace_bytes(long long n) {
  return align_up((size_t)focal_blocks(n) * 2 * sizeof(double), 256);
}

BFHIP_EXPORT int bfhip_gaussian_focal_loss(const float *logits, const float *target, long long n, float clip_eps,
                                           float *loss_sum_npos, float *grad, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  BFHIP_REQUIRE(logits && target && loss_sum_npos && grad && n > 0, "gaussian_focal_loss: null pointer / n");
  Workspace ws(workspace, workspace_bytes);
  const int nb = focal_blocks(n);
  double *partial = ws.take<double>((size_t)nb * 2);
  if (!ws.ok() || !workspace) { set_error("gaussian_focal_loss: workspace too small"); return BFHIP_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gaussian_focal_kernel, dim3(nb), dim3(256), 0, s, logits, target, n, clip_eps, grad, partial);
  hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(256), 0, s, partial, nb, loss_sum_npos);
  return check_launch("gaussian_focal_loss");
}

BFHIP_EXPORT int bfhip_query_losses(const float *cls_logits, const int32_t *labels, const float *label_weights,
                                    const float *box_pred, const float *bbox_targets, const float *bbox_weights,
                                    const float *code_weights, int B, int C, int P, int K, int ld, int p_off,
                                    float gamma, float alpha, float *grad_cls, float *grad_box, float *loss_sums,
                                    void *stream) {
  BFHIP_REQUIRE(cls_logits && labels && label_weights && box_pred && bbox_targets && bbox_weights && code_weights &&
                    grad_cls && grad_box && loss_sums, "query_losses: null pointer");
  BFHIP_REQUIRE(B > 0 && C > 0 && P > 0 && K > 0 && ld >= P && p_off >= 0 && p_off + P <= ld, "query_losses: bad sizes");
  hipLaunchKernelGGL(query_losses_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cls_logits, labels,
                     label_weights, box_pred, bbox_targets, bbox_weights, code_weights, B, C, P, K, ld, p_off, gamma,
                     alpha, grad_cls, grad_box, loss_sums);
  return check_launch("query_losses");
}
