/*
 * bevfusion_hip.h -- C ABI of libbevfusion_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the native operators of the reference
 * (lhn0323/BEVFUSION-3D_object_detection, paths relative to the reference root,
 *  BF/ = projects/BEVFusion/bevfusion/).  Each entry point names the reference interface it
 * replaces.  The reference binds its ops with pybind11 + at::Tensor
 * (BF/ops/bev_pool/src/bev_pool.cpp:89-94, BF/ops/voxel/src/voxelization.cpp:6-11); this
 * library exposes the same operations with plain pointers and sizes so that any host
 * (ctypes, pybind, C++) can bind them.  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream)
 *   - return value: 0 = ok, negative = error (BFHIP_E_*); message via bfhip_last_error()
 *   - no entry point allocates, frees or synchronises the device; temporaries live in the
 *     caller-provided workspace (size from the matching *_workspace_bytes()).  All entry
 *     points are therefore hipGraph-capturable.
 *   - outputs are caller-allocated and written in place, like the reference's
 *     hard_voxelize/dynamic_voxelize (BF/ops/voxel/voxelize.py:47-66)
 */
#ifndef BEVFUSION_HIP_H_
#define BEVFUSION_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFHIP_OK 0
#define BFHIP_E_INVALID (-1)   /* bad argument (shape, alignment, unsupported size) */
#define BFHIP_E_WORKSPACE (-2) /* workspace too small */
#define BFHIP_E_LAUNCH (-3)    /* hipGetLastError() after a launch was not hipSuccess */

#define BFHIP_REDUCE_SUM 0  /* enum order of BF/ops/voxel/src/scatter_points_cuda.cu:7 */
#define BFHIP_REDUCE_MEAN 1
#define BFHIP_REDUCE_MAX 2

/* op ids of the optional profiler */
#define BFHIP_OP_BEV_POOL_FWD 0
#define BFHIP_OP_BEV_POOL_BWD 1
#define BFHIP_OP_HARD_VOXELIZE 2 /* whole 7-launch pipeline */
#define BFHIP_OP_DYNAMIC_VOXELIZE 3
#define BFHIP_OP_LIFT_SPLAT_FWD 4
#define BFHIP_OP_LIFT_SPLAT_BWD 5
#define BFHIP_OP_SPCONV_FWD 6
#define BFHIP_OP_SPCONV_BWD 7 /* dgrad */
#define BFHIP_OP_RULEBOOK 8
#define BFHIP_OP_BEV_AUX 9
#define BFHIP_OP_SCATTER_FWD 10
#define BFHIP_OP_SCATTER_BWD 11
#define BFHIP_OP_SPCONV_WGRAD 12 /* whole op: offset counts + main kernel + partial-slab reduce */
#define BFHIP_OP_RASTER 13
#define BFHIP_OP_SPCONV_WGRAD_MAIN 14 /* the dominant kernel alone (what rocprofv3 lists as spconv_wgrad64p_kernel) */
/* dense ops: recorded only at profile level 2 (bfhip_profile_enable(2)): ~280 scopes per training step */
#define BFHIP_OP_CONV2D_FWD 15   /* bfhip_conv2d_fwd */
#define BFHIP_OP_CONV2D_DGRAD 16 /* bfhip_conv2d_dgrad: weight transpose + implicit GEMM */
#define BFHIP_OP_CONV2D_WGRAD 17 /* bfhip_conv2d_wgrad: main kernel + slab reduce */
#define BFHIP_OP_BN2D_FWD 18     /* bfhip_bn2d_fwd / _fwd_partials: statistics, finalize, apply */
#define BFHIP_OP_BN2D_BWD 19     /* bfhip_bn2d_bwd: reduce, finalize, apply */
#define BFHIP_OP_CONV2D_PW_FWD 20   /* bfhip_conv2d_fwd calls served by the pointwise kernel (1x1, stride 1): HBM-bound GEMMs over the pixel matrix */
#define BFHIP_OP_CONV2D_PW_DGRAD 21 /* bfhip_conv2d_dgrad(_wt) calls served by the pointwise kernel */
#define BFHIP_OP_COUNT 24

int bfhip_abi_version(void);
/* thread-local, valid until the next failing call on the same thread */
const char *bfhip_last_error(void);

/* Optional profiler used by bench.py for the roofline line: when enabled, every entry point
 * records a HIP event pair ON ITS OWN STREAM around its dominant kernel launch (memsets and
 * helper launches excluded).  bfhip_profile_read() synchronises the recorded events (the only
 * call in this library that blocks) and returns the accumulated milliseconds and launch count. */
void bfhip_profile_enable(int on);
int bfhip_profile_read(int op, double *sum_ms_host, long long *count_host, int reset);

/* ---------------------------------------------------------------------------------------
 * bev_pool  (replaces bev_pool_ext.bev_pool_forward / bev_pool_backward,
 *            BF/ops/bev_pool/src/bev_pool.cpp:22-87, kernels BF/ops/bev_pool/src/bev_pool_cuda.cu:20-98)
 *   x        f32[n, c]   rows sorted so that each interval is contiguous
 *   geom     i32[n, 4]   (x, y, z, b) per row; only the first row of an interval is read
 *   starts   i32[m], lengths i32[m]
 *   out      f32[b, d, h, w, c]  cell (b, z, x, y) <- sum of the interval's rows, other cells 0
 * fwd zero-fills `out` itself (the reference's wrapper does torch::zeros, bev_pool.cpp:38-40).
 * bwd: x_grad[row] = out_grad[cell(interval(row))].  If `intervals_cover_all_rows` is 0 the
 * whole x_grad is zero-filled first (reference: torch::zeros, bev_pool.cpp:76-78); pass 1 when
 * starts/lengths partition [0, n) (always true for intervals built from ranks) to skip it.
 * m_dev: optional device int holding the live interval count (<= m); NULL = use m.
 * --------------------------------------------------------------------------------------- */
int bfhip_bev_pool_fwd(const float *x, const int32_t *geom, const int32_t *starts,
                       const int32_t *lengths, float *out, int n, int c, int m, int b, int d,
                       int h, int w, const int32_t *m_dev, void *stream);
int bfhip_bev_pool_bwd(const float *out_grad, const int32_t *geom, const int32_t *starts,
                       const int32_t *lengths, float *x_grad, int n, int c, int m, int b, int d,
                       int h, int w, int intervals_cover_all_rows, const int32_t *m_dev,
                       void *stream);

/* ---------------------------------------------------------------------------------------
 * voxelization  (replaces voxel_layer.dynamic_voxelize / hard_voxelize,
 *                BF/ops/voxel/src/voxelization.h:58-96; CPU kernels voxelization_cpu.cpp:8-144,
 *                CUDA kernels voxelization_cuda.cu:24-373)
 *   points f32[n, f] (f >= 3), voxel_size_host f32[3], coors_range_host f32[6] (xyzxyz min,max)
 * dynamic: coors i32[n,3] = floor((p - min)/voxel) per axis in IEEE fp32 (subtract, divide),
 *          (-1,-1,-1) when any axis is outside [0, grid) -- the CPU path's convention
 *          (voxelization_cpu.cpp:34-39).
 * hard   : first-come grouping, deterministic, identical to hard_voxelize_cpu:
 *          voxel id = order of the voxel's first point; <= max_points points kept per voxel in
 *          point order; voxels beyond max_voxels dropped.  voxels f32[max_voxels, max_points, f]
 *          and num_points_per_voxel must be zero-filled by the caller (voxelize.py:51-53);
 *          coors i32[max_voxels,3] in (x,y,z).  The voxel count is written to *voxel_num_dev.
 * --------------------------------------------------------------------------------------- */
int bfhip_dynamic_voxelize(const float *points, int32_t *coors, int n, int f,
                           const float *voxel_size_host, const float *coors_range_host,
                           void *stream);
size_t bfhip_hard_voxelize_workspace_bytes(int n, int max_points, int max_voxels);
int bfhip_hard_voxelize(const float *points, int n, int f, float *voxels, int32_t *coors,
                        int32_t *num_points_per_voxel, const float *voxel_size_host,
                        const float *coors_range_host, int max_points, int max_voxels,
                        void *workspace, size_t workspace_bytes, int32_t *voxel_num_dev,
                        void *stream);

/* ---------------------------------------------------------------------------------------
 * dynamic scatter  (replaces voxel_layer.dynamic_point_to_voxel_forward / _backward,
 *   BF/ops/voxel/src/scatter_points_cuda.cu:183-308) and the hard-voxel mean (BF/bevfusion.py:251-253)
 *   feats f32[N,C], coors i32[N,3]; rows with any negative coordinate are dropped (:202).
 *   voxel rows come out in ascending lexicographic (c0,c1,c2) order, like at::unique_dim(sorted).
 *   fwd outputs are sized for N rows; counts_dev i32[3] = {M voxels, valid points, key-overflow flag
 *   (coordinates must be < 2^21)}.  Sums run in point order (deterministic).
 *   bwd: sum/mean -> gather (mean divides by count); max -> gradient to the lowest-index arg-max.
 * --------------------------------------------------------------------------------------- */
size_t bfhip_dynamic_scatter_workspace_bytes(int N);
int bfhip_dynamic_scatter_fwd(const float *feats, const int32_t *coors, int N, int C, int reduce_type,
                              float *voxel_feats, int32_t *voxel_coors, int32_t *point2voxel,
                              int32_t *voxel_count, int32_t *counts_dev, void *workspace,
                              size_t workspace_bytes, void *stream);
size_t bfhip_dynamic_scatter_bwd_workspace_bytes(int M, int C);
int bfhip_dynamic_scatter_bwd(float *grad_feats, const float *grad_voxel_feats, const float *feats,
                              const float *voxel_feats, const int32_t *point2voxel,
                              const int32_t *voxel_count, int N, int M, int C, int reduce_type,
                              void *workspace, size_t workspace_bytes, void *stream);
int bfhip_voxel_mean(const float *voxels, const int32_t *num_points, int M, int P, int F, float *out,
                     void *stream);

/* ---------------------------------------------------------------------------------------
 * camera frustum -> BEV plan  (replaces BaseViewTransform.get_geometry + bev_pool_aux,
 *   BF/depth_lss.py:68-112,118-176, and the interval construction of
 *   BF/ops/bev_pool/bev_pool.py:48-54; sync-free: all counts stay on the device)
 *   frustum        f32[D*HW, 3]  pixel-depth grid (depth_lss.py:53-66)
 *   per camera (B*N rows): post_trans f32[3], post_rots_inv f32[9], combine f32[9]
 *                          (= camera2lidar_rots @ intrins_inverse, depth_lss.py:93), c2l_trans f32[3]
 *   per sample (B rows)  : extra_rots f32[9], extra_trans f32[3]   (identity / 0 when absent)
 *   origin_host = bx - dx/2, dx_host, nx_host = (nx[0], nx[1], nx[2])   (depth_lss.py:14-18)
 * geometry is evaluated per point in fp32 with the fixed association ((m0*p0+m1*p1)+m2*p2), no
 * fma; cell = trunc((p - origin)/dx); kept = inside the grid; rank = x*(nx1*nx2*B) + y*(nx2*B)
 * + z*B + b; kept points are STABLY sorted by rank (the reference's argsort is unstable).
 * outputs (N' = B*N*D*HW rows where sized by points; mmax rows where sized by intervals):
 *   sorted_pd        u32[N']  (pixel_index << 8 | depth_bin) of the k-th sorted kept point
 *   starts, lengths  i32[mmax], cell_of_interval i32[mmax] (offset of the cell in out[b][z][x][y])
 *   counts_dev       i32[2] = {n_kept, n_intervals}
 *   interval_order   i32[mmax] (may be NULL): the intervals in camera-major order -- stably sorted by (sample, camera) of
 *                    their first member; bfhip_lift_splat_fwd walks them in this order so that every XCD gathers the feature
 *                    rows of "its" cameras from its own L2 (the cells' sums are unchanged, only their production order)
 *   optional (may be NULL): cell_of_point i32[N'] (out cell of every frustum point or -1; needed by
 *   lift_splat_bwd), geom_sorted i32[N',4] (x,y,z,b), ranks_sorted i64[N'], kept u8[N'],
 *   geom_xyz f32[N',3] (the materialised get_geometry output, for parity tests)
 * --------------------------------------------------------------------------------------- */
size_t bfhip_bev_plan_workspace_bytes(long long nprime, long long ncells_times_b);
int bfhip_bev_plan(const float *frustum, const float *post_trans, const float *post_rots_inv,
                   const float *combine, const float *c2l_trans, const float *extra_rots,
                   const float *extra_trans, int B, int N, int D, int HW,
                   const float *origin_host, const float *dx_host, const int32_t *nx_host,
                   uint32_t *sorted_pd, int32_t *starts, int32_t *lengths,
                   int32_t *cell_of_interval, int32_t *interval_order, int32_t *counts_dev, int32_t *cell_of_point,
                   int32_t *geom_sorted, int64_t *ranks_sorted, uint8_t *kept, float *geom_xyz,
                   int mmax, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * fused lift-splat  (replaces the outer product BF/depth_lss.py:723-725, the two gathers
 *   x[kept][indices] :190-194 and the bev_pool op, without materialising x[N', C])
 *   depth f32[P, depth_pitch] (softmax over D bins, pixel-major), feat f32[P, feat_pitch]
 *   (C channels, pixel-major), P = B*N*HW pixels; plan arrays from bfhip_bev_plan.
 *   out f32[out_cells, C] (out_bf16 = 1: bf16[out_cells, C], the fp32 sums rounded once on store) with
 *   out_cells = B*nx2*nx0*nx1, zero-filled by the call.
 * bwd: d_depth[p,d] = <out_grad[cell(p,d)], feat[p]>, d_feat[p] = sum_d depth[p,d]*out_grad[cell(p,d)]
 * feat_bf16 = 1: feat is bf16[P, feat_pitch] (pitches in ELEMENTS, C % 8 == 0) -- the depthnet's output as the bf16
 *   convolution left it (the reference widens it with x.float(), BF/depth_lss.py:467-468: same values); the gathered rows are
 *   half as long, the arithmetic is unchanged (fp32 products and sums in the same order), and the backward writes d_feat as
 *   bf16[P, d_feat_pitch] (the fp32 sums rounded once, what the backward of that x.float() does).
 * --------------------------------------------------------------------------------------- */
int bfhip_lift_splat_fwd(const float *depth, int depth_pitch, const void *feat, int feat_bf16, int feat_pitch,
                         const uint32_t *sorted_pd, const int32_t *starts, const int32_t *lengths,
                         const int32_t *cell_of_interval, const int32_t *interval_order /* may be NULL: rank order */,
                         const int32_t *counts_dev, int mmax, int C, long long out_cells, void *out, int out_bf16,
                         void *stream);
int bfhip_lift_splat_bwd(const void *out_grad, int grad_bf16 /* out_grad is bf16[out_cells, C] */, const float *depth,
                         int depth_pitch,
                         const void *feat, int feat_bf16, int feat_pitch, const int32_t *cell_of_point,
                         int num_cams, int D, int HW, int C, float *d_depth, int d_depth_pitch,
                         void *d_feat, int d_feat_pitch, void *stream);

/* ---------------------------------------------------------------------------------------
 * sparse 3-D convolution  (replaces what the reference delegates to spconv 2.x:
 *   SpconvOps.get_indice_pairs_implicit_gemm, projects/SparseConvolution/sparse_functional.py:118-137;
 *   ConvGemmOps.implicit_gemm :287-314; SparseConvTensor.dense(), BF/sparse_encoder.py:147)
 *   indices i32[N,4] = (b, x, y, z), 16-byte aligned; shapes/ksize/stride/padding/dilation are host int[3]
 *   pair table i32[KV, ld]: pair[k*ld + out_row] = input row or -1; offset k = (i*k1 + j)*k2 + l
 *   weights f32 (Cout, k0, k1, k2, Cin) (mmdet3d/models/layers/spconv/overwrite_spconv/write_spconv2.py:50-51)
 * SubM    : out rows = in rows; pad = dil*(k//2).
 * strided : out shape (in + 2p - d(k-1) - 1)//s + 1 (projects/SparseConvolution/sparse_conv.py:88-90);
 *           output rows in ASCENDING LINEAR ORDER ((b*X+x)*Y+y)*Z+z (canonical; spconv's is hash order).
 *           Two phases because N_out must reach the host: _count writes counts_dev[0] = N_out;
 *           _fill (same workspace, untouched in between) writes out_indices i32[n_out,4],
 *           pair_fwd i32[KV,n_out], pair_bwd i32[KV,N]; the number of pairs is sum(counts_dev[1..64])
 *           (counts_dev is i32[65]; rulebook_subm's n_pairs_dev is i32[64], same convention: 64 spread
 *           counters, because one hot word serialises the atomics).
 * duplicate input coordinates (never produced by the voxelizers; the reference's own encoder test feeds them,
 *           tests/test_models/test_middle_encoders/test_sparse_encoders.py:22-25, where spconv's result is whichever
 *           row its hash insert kept): every duplicate stays a row; a neighbour lookup resolves a coordinate to its LOWEST
 *           row (SubM table) / a strided output takes the HIGHEST row per (offset, output) slot -- deterministic both.
 * gemm    : out[n_rows, Ndim] = sum_k M_k . in[pairs[k][row]];  transpose=0 forward (M_k = W[:,k,:]^T),
 *           transpose=1 dgrad (M_k = W[:,k',:], k' = KV-1-k when flip else k; flip=1 lets a SubM layer
 *           reuse pair_fwd as its backward table).  Exact-fp32 MFMA, deterministic, no atomics.
 * wgrad   : dW (Cout,KV,Cin) = sum_n dout[n] (x) in[pairs[k][n]], fixed-order reduction.
 * sparse_to_bev: out f32[B, C*Z, X, Y] (zero-filled here) <- feats[n][c] at (b, c*Z+z, x, y);
 * bev_to_sparse: its gather (backward).
 * --------------------------------------------------------------------------------------- */
int bfhip_conv_out_shape(const int *in_shape, const int *ksize, const int *stride,
                         const int *padding, const int *dilation, int *out_shape);
size_t bfhip_rulebook_subm_workspace_bytes(int N);
int bfhip_rulebook_subm(const int32_t *indices, int N, int B, const int *in_shape, const int *ksize,
                        const int *dilation, int32_t *pair_fwd, int32_t *n_pairs_dev, uint32_t *row_mask,
                        int32_t *perm, void *workspace, size_t workspace_bytes, void *stream);
/* row_mask / perm (optional, kernel volume <= 32): the table's row masks and sorted row order (see
 * bfhip_rulebook_sort_rows below), produced by the launch that fills the table instead of a second pass over it.
 * n_pairs_dev is optional (NULL: no pair statistics). */
size_t bfhip_rulebook_sparse_workspace_bytes(int B, const int *in_shape, const int *ksize,
                                             const int *stride, const int *padding,
                                             const int *dilation);
/* n_in_dev (optional): the true number of input rows on the device; N is then only the launch bound.  With
 * bfhip_rulebook_sparse_out_indices (output coordinates into a buffer of `cap` rows, no host-side N_out) a chain of strided
 * layers is counted back to back and all N_out values are read in ONE host read. */
int bfhip_rulebook_sparse_count(const int32_t *indices, int N, const int32_t *n_in_dev, int B, const int *in_shape,
                                const int *ksize, const int *stride, const int *padding, const int *dilation,
                                int32_t *counts_dev, void *workspace, size_t workspace_bytes, void *stream);
int bfhip_rulebook_sparse_out_indices(int B, const int *in_shape, const int *ksize, const int *stride,
                                      const int *padding, const int *dilation, int cap, int32_t *out_indices,
                                      void *workspace, size_t workspace_bytes, void *stream);
int bfhip_rulebook_sparse_fill(const int32_t *indices, int N, int B, const int *in_shape,
                               const int *ksize, const int *stride, const int *padding,
                               const int *dilation, int n_out, int32_t *out_indices,
                               int32_t *pair_fwd, int32_t *pair_bwd, int32_t *counts_dev,
                               uint32_t *mask_fwd, int32_t *perm_fwd, uint32_t *mask_bwd, int32_t *perm_bwd,
                               void *sort_workspace, size_t sort_workspace_bytes,
                               void *workspace, size_t workspace_bytes, void *stream);
/* mask_fwd / perm_fwd (over the output rows, from pair_fwd) and mask_bwd / perm_bwd (over the input rows, from pair_bwd):
 * optional, kernel volume <= 32; sort_workspace = bfhip_rulebook_sort_rows_workspace_bytes(max(N, n_out), KV) bytes. */
/* row_mask[n] bit k = (pairs[k][n] >= 0); perm = rows stably sorted by mask inside consecutive chunks of 4096 rows
 * (key = (n / 4096, mask); one LDS sort per chunk, one launch), so that the 16 rows of an MFMA tile share their kernel
 * offsets and whole offsets are skipped per tile (cf. spconv's mask_argsort,
 * projects/SparseConvolution/sparse_functional.py:139-162) while consecutive tiles stay in one slab of space (the
 * gather-GEMM deals contiguous eighths of the tiles to the 8 XCDs).  BFHIP_SPCONV_CHUNK_SORT=0: the round-2 order
 * (eighth of the row range, mask) from a device-wide sort.  perm/row_mask are optional
 * (NULL) inputs of gemm / wgrad; results do not depend on them (only the fp32 summation order of wgrad). */
size_t bfhip_rulebook_sort_rows_workspace_bytes(int n_rows, int KV);
int bfhip_rulebook_sort_rows(const int32_t *pairs, int ld, int KV, int n_rows, uint32_t *row_mask,
                             int32_t *perm, void *workspace, size_t workspace_bytes, void *stream);
/* Debug guard: the gather kernels below dereference pairs[k*ld + perm[i]] and in + pairs[..]*K unchecked (an index that is
 * out of range is a memory-aperture fault).  Counts into status_dev i32[4]: [0] pair entries outside
 * [-1, n_src), [1] perm entries outside [0, n_rows), [2] rows not occurring exactly once in perm, [3] rows whose row_mask
 * disagrees with the table.  perm / row_mask optional.  The Python host runs it before every gather launch under
 * BFHIP_SPCONV_VALIDATE=1 and in the parity tests; it is not part of the timed path. */
size_t bfhip_rulebook_validate_workspace_bytes(int n_rows);
int bfhip_rulebook_validate(const int32_t *pairs, int ld, int KV, int n_rows, int n_src, const int32_t *perm,
                            const uint32_t *row_mask, int32_t *status_dev, void *workspace, size_t workspace_bytes,
                            void *stream);
size_t bfhip_spconv_workspace_bytes(int KV, int Cin, int Cout);
int bfhip_spconv_gemm(const float *in, const float *W, const int32_t *pairs, int ld, int KV,
                      int n_rows, int Cin, int Cout, int transpose, int flip, const int32_t *perm,
                      const uint32_t *row_mask, float *out, void *workspace, size_t workspace_bytes,
                      void *stream);
/* bf16-MFMA variant of bfhip_spconv_gemm (same workspace size): features and weights enter the MFMA as bf16, products
 * accumulate in fp32.  io_bf16 = 0: features f32 (rounded on load), output f32.  io_bf16 = 1: features STORED in bf16
 * (the gathered 16-byte row segments are the MFMA operand as they are: half the gather traffic) and the output rounded
 * to bf16 once -- the reference runs spconv in half precision under AMP.  Requires K % 8 == 0. */
int bfhip_spconv_gemm_bf16(const void *in, const float *W, const int32_t *pairs, int ld, int KV,
                           int n_rows, int Cin, int Cout, int transpose, int flip, const int32_t *perm,
                           const uint32_t *row_mask, void *out, int io_bf16, void *workspace, size_t workspace_bytes,
                           void *stream);
size_t bfhip_spconv_wgrad_workspace_bytes(int KV, int Cin, int Cout, int n_rows);
/* io_bf16 = 1: `in` and `dout` are bf16 feature matrices (widened on load; Cin, Cout multiples of 4); dW is always f32 */
int bfhip_spconv_wgrad(const void *in, const void *dout, const int32_t *pairs, int ld, int KV,
                       int n_rows, int Cin, int Cout, const int32_t *perm, float *dW, int io_bf16, void *workspace,
                       size_t workspace_bytes, void *stream);
int bfhip_sparse_to_bev(const float *feats, const int32_t *indices, int N, int C, int B, int X,
                        int Y, int Z, float *out, void *stream);
int bfhip_bev_to_sparse(const float *grad_out, const int32_t *indices, int N, int C, int B, int X,
                        int Y, int Z, float *grad_feats, void *stream);
/* channels-last variants: out (f32 | bf16, dtype 0 | 1) is the NHWC memory of the [B, C*Z, X, Y] map, out[b][x][y][c*Z+z];
 * the backward reads element (b, ch, x, y) at grad_out[b*stride_b + x*stride_x + y*stride_y + ch] (strides in elements,
 * channel stride 1), so a channel slice of a wider channels-last gradient is consumed in place. */
int bfhip_sparse_to_bev_nhwc(const float *feats, const int32_t *indices, int N, int C, int B, int X, int Y, int Z,
                             int dtype, void *out, void *stream);
int bfhip_bev_nhwc_to_sparse(const void *grad_out, long long stride_b, long long stride_x, long long stride_y,
                             int dtype, const int32_t *indices, int N, int C, int Z, float *grad_feats, void *stream);

/* ---------------------------------------------------------------------------------------
 * sparse LiDAR depth images + GT depth histogram  (replaces the per-sample torch loop of
 *   BaseDepthTransform.forward, BF/depth_lss.py:372-449, and the scatter_add_ histogram of
 *   DepthLSSTransform.get_cam_feats, :636-686)
 *   points f32[n,f]; inv_rot f32[9] = lidar_aug_matrix_inverse[:3,:3]; aug_trans f32[3] = lidar_aug_matrix[:3,3];
 *   lidar2image, img_aug f32[ncam,16] (row-major 4x4).  depth f32[ncam,iH,iW] is fully written.
 *   A pixel hit by several points keeps the LAST point (torch's scatter_ leaves it unspecified, :410-417).
 *   counts (optional, f32[ncam,fH,fW,D], cleared by the caller) receives the depth-bin histogram of the hit
 *   pixels; bfhip_depth_histogram normalises it (bin 0 excluded, as :670-674) or rebuilds it from a depth image.
 * --------------------------------------------------------------------------------------- */
/* Gradient of the first dtransform layer, Conv2d(1, 8, 1) on the one-channel depth image (BF/depth_lss.py:592-594; its
 * autograd backward in the reference): y[m][c] = b[c] + d[m] * w[c] over the M = BN * iH * iW pixels.  dy bf16 [M][8] dense,
 * d bf16 [M] -> out f32[16] = {db[0..8), dw[0..8)}; fp32 accumulation, fixed-order sums (deterministic). */
size_t bfhip_depth_lift_bwd_workspace_bytes(void);
int bfhip_depth_lift_bwd(const void *dy, const void *d, long long M, float *out, void *workspace, size_t workspace_bytes, void *stream);
size_t bfhip_rasterise_depth_workspace_bytes(int ncam, int iH, int iW);
int bfhip_rasterise_depth(const float *points, int n, int f, const float *inv_rot, const float *aug_trans,
                          const float *lidar2image, const float *img_aug, int ncam, int iH, int iW,
                          float *depth, float *counts, int fH, int fW, int D, const float *dbound_host,
                          void *workspace, size_t workspace_bytes, void *stream);
int bfhip_depth_histogram(const float *depth, int BN, int iH, int iW, int fH, int fW, int D,
                          const float *dbound_host, float *counts, float *distr, void *stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm1d (+ residual) (+ ReLU) on sparse feature matrices f32[N, C], training mode  (SURVEY 8 f-4;
 *   replaces the BN1d / add / ReLU torch kernels after every sparse conv,
 *   mmdet3d/models/layers/sparse_block.py:135-154,157-224; norm_cfg BN1d eps 1e-3 momentum 0.01)
 *   fwd: stats f32[2C] <- (batch mean, 1/sqrt(biased var + eps)); running stats updated like torch
 *        (running_var with the unbiased variance); y = act(gamma * (x - mean) * invstd + beta [+ residual])
 *   bwd: dgb f32[2C] <- (dgamma, dbeta); dx; dres (optional) = gradient of the residual input
 *   C must divide 256 and be a multiple of 4; tensors 16-byte aligned.
 * --------------------------------------------------------------------------------------- */
size_t bfhip_bn1d_workspace_bytes(int N, int C);
/* n_rows_dev / m_dev (optional, both BatchNorm families): the number of ACTIVE rows when N / M is only a capacity (feature
 * matrices of the sparse encoder sized by bounds, true counts on the device: no host read); the rows beyond it must be zero. */
int bfhip_bn1d_fwd(const float *x, const float *residual, const float *gamma, const float *beta, int N, int C,
                   float eps, float momentum, int relu, float *running_mean, float *running_var, float *stats,
                   float *y, const int32_t *n_rows_dev, void *workspace, size_t workspace_bytes, void *stream);
int bfhip_bn1d_bwd(const float *dy, const float *y, const float *x, const float *stats, const float *gamma, int N,
                   int C, int relu, float *dx, float *dres, float *dgb, const int32_t *n_rows_dev, void *workspace,
                   size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm2d (+ residual) (+ ReLU), training mode, on channels-last activations viewed as [M = N*H*W, C]
 *   (the BN / add / ReLU after every dense conv: ResNet bottlenecks, BF/depth_lss.py:581-620,
 *    BF/bevfusion_head.py:25-38,104-126, mmdet3d/models/backbones/second.py:70-95, necks/second_fpn.py:60-84)
 *   dtype: 0 = f32, 1 = bf16 (x, residual, y, dy, dx, dres share it); gamma/beta/stats/running stats are f32.
 *   C must be a multiple of the 16-byte vector (4 f32 / 8 bf16) with at most 256 vectors per row:
 *   bfhip_bn2d_supported() tells.  stats f32[4C] <- (mean, invstd, a = gamma*invstd, b = beta - mean*a).
 *   bwd: pass y only for layers that added a residual before the ReLU (otherwise the ReLU mask is recomputed
 *   from x and y may be NULL); dgb f32[2C] <- (dgamma, dbeta); dres optional.
 * --------------------------------------------------------------------------------------- */
int bfhip_bn2d_supported(long long M, int C, int dtype);
/* out f32[C] = column sums of x [M][C] (dtype 0 f32 | 1 bf16, dense; shapes as bfhip_bn2d_supported, workspace as
 * bfhip_bn2d_workspace_bytes): the bias gradient of a convolution / linear layer (sum of dy over pixels / rows), fixed order. */
int bfhip_colsum(const void *x, long long M, int C, int dtype, float *out, void *workspace, size_t workspace_bytes, void *stream);
size_t bfhip_bn2d_workspace_bytes(long long M, int C, int dtype);
int bfhip_bn2d_fwd(const void *x, const void *residual, const float *gamma, const float *beta, long long M, int C,
                   int dtype, float eps, float momentum, int relu, float *running_mean, float *running_var,
                   float *stats, void *y, const int32_t *m_dev, void *workspace, size_t workspace_bytes, void *stream);
int bfhip_bn2d_bwd(const void *dy, const void *x, const void *y, const float *stats, const float *gamma, long long M,
                   int C, int dtype, int relu, void *dx, void *dres, float *dgb, const int32_t *m_dev, void *workspace,
                   size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * out f32[M, N] = X^T Y for tall-skinny row-major X [K, M], Y [K, N] (f32 | bf16, dtype 0 | 1), K >> M, N: the weight
 *   gradient of the linear layers that act on every BEV cell (cross-attention K / V projections and the position
 *   embedding MLP of the TransFusion decoder layer, BF/transformer.py:10-23,60-105: dW = dY^T X with K = B*180*180).
 *   K is split over the chip, products and sums in fp32, fixed-order combine (deterministic).
 * --------------------------------------------------------------------------------------- */
size_t bfhip_xty_workspace_bytes(long long K, int M, int N);
int bfhip_xty(const void *X, const void *Y, long long K, int M, int N, int dtype, float *out, void *workspace,
              size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Cross attention of few queries over very many keys (the TransFusion decoder layer: 200 queries x 32 400 BEV cells,
 *   8 heads of 16 channels, dropout on the attention weights; BF/transformer.py:60-105, mmcv MultiheadAttention ->
 *   torch scaled_dot_product_attention).  Q [B, Lq, H*16], K / V [B, Lk, H*16], O, dO, dQ, dK, dV alike: bf16, row-major;
 *   head h = channels 16h..16h+15; Lq <= 256.  lse f32[B*H, Lq] (log-sum-exp of the scaled scores) links fwd and bwd.
 *   The key axis is split over the chip; combines run in a fixed order (bit-reproducible).  dropout_p in [0, 1): keep
 *   mask = counter hash of (seed, b, h, query, key), regenerated in the backward; bfhip_attn_dropout_mask writes it
 *   out (u8[B*H, Lq, Lk]) for tests.  The workspace is shared by fwd and bwd.
 * --------------------------------------------------------------------------------------- */
size_t bfhip_attn_workspace_bytes(int B, int H, int Lq, int Lk);
int bfhip_attn_fwd(const void *Q, const void *K, const void *V, int B, int H, int Lq, int Lk, float scale,
                   float dropout_p, unsigned long long seed, const unsigned long long *seed_dev, void *O, float *lse,
                   void *workspace, size_t workspace_bytes, void *stream);
int bfhip_attn_bwd(const void *Q, const void *K, const void *V, const void *O, const void *dO, const float *lse, int B,
                   int H, int Lq, int Lk, float scale, float dropout_p, unsigned long long seed,
                   const unsigned long long *seed_dev, void *dQ, void *dK, void *dV, void *workspace, size_t workspace_bytes,
                   void *stream);
/* seed_dev (optional, all three): device u64 call counter mixed into the seed (a replayed hipGraph repeats its host arguments) */
int bfhip_attn_dropout_mask(int B, int H, int Lq, int Lk, float dropout_p, unsigned long long seed,
                            const unsigned long long *seed_dev, unsigned char *mask, void *stream);

/* ---------------------------------------------------------------------------------------
 * Bilinear 2x upsampling of channels-last maps [B, H, W, C] -> [B, 2H, 2W, C] (f32 | bf16), torch's align_corners=False
 *   index rule; the LSS-FPN's top-down path (BF/bevfusion_necks.py:76-88).  dir 0 = forward; dir 1 = backward as a gather
 *   (src = grad of the output, dst = grad of the input): no atomics, deterministic.  C % (4 | 8) == 0.
 * --------------------------------------------------------------------------------------- */
int bfhip_upsample2x_nhwc(const void *src, void *dst, int B, int H, int W, int C, int dtype, int dir, void *stream);

/* ---------------------------------------------------------------------------------------
 * 3x3 stride-2 pad-1 max pooling of a channels-last bf16 map [N, H, W, C] -> [N, OH, OW, C], OH = (H - 1) / 2 + 1 (the ResNet-50
 *   stem's nn.MaxPool2d(3, 2, 1); img_backbone = mmdet.ResNet, an external dependency of the reference).  Forward stores the
 *   winning tap (0..8, torch's scan order and tie / NaN rule) of every element in `tap` (u8, same shape as y); backward is a
 *   gather over the <= 2 x 2 windows of each input pixel: no atomics, no zero-fill, deterministic.  C % 8 == 0.
 * --------------------------------------------------------------------------------------- */
int bfhip_maxpool3x3s2_fwd(const void *x, int N, int H, int W, int C, void *y, unsigned char *tap, void *stream);
int bfhip_maxpool3x3s2_bwd(const void *dy, const unsigned char *tap, int N, int H, int W, int C, void *dx, void *stream);

/* ---------------------------------------------------------------------------------------
 * TransFusion head: box decoding, target assignment and losses on the device  (SURVEY 8 f-3).
 *   Replaces TransFusionBBoxCoder.decode/encode (BF/utils.py:33-96), HungarianAssigner3D.assign with its three costs
 *   and the `.cpu()` + scipy.optimize.linear_sum_assignment round trip (BF/utils.py:128-151,241-284; IoU as
 *   mmdet3d/structures/bbox_3d/base_box3d.py:529-590), the target scatter and the box-by-box heat-map drawing of
 *   BEVFusionHead.get_targets_single (BF/bevfusion_head.py:514-674, mmdet3d/models/utils/gaussian.py:9-92) and the
 *   three loss terms of loss_by_feat (:696-796: mmdet GaussianFocalLoss / FocalLoss / L1Loss).
 *   Head outputs keep the reference layout [B, channels, ld] (ld = proposals of all decoder layers concatenated);
 *   one call handles the P proposals starting at p_off.  Ground truth is padded to G boxes per sample:
 *   gt_boxes f32[B,G,Wg] = (x, y, z_bottom, dx, dy, dz, yaw[, vx, vy]), gt_labels i32[B,G], n_gt i32[B].
 *
 *   decode_boxes   cfg_host = {out_size_factor, voxel_x, voxel_y, pc_x0, pc_y0}; vel may be NULL (-> 7 columns);
 *                  boxes f32[B,P,7|9] bottom-centre
 *   assign_cost    cfg_host = {cls_weight, alpha, gamma, eps, reg_weight, iou_weight, pc_x0, pc_y0, pc_x1, pc_y1};
 *                  cost, iou f32[B,P,G] (columns >= n_gt[b] are written as 0 and never read)
 *   hungarian      minimum-cost assignment per sample, fp64, max(P, G) <= 1024.  assigned i32[B,P]: 0 = background,
 *                  g + 1 = matched (AssignResult.gt_inds); status i32[B]: 0 ok, 1 = non-finite costs (all background)
 *   assign_targets cfg_host = {pc_x0, pc_y0, (float)(out_size_factor*voxel_x), (float)(out_size_factor*voxel_y),
 *                  pos_weight}; labels i32[B,P] (num_classes = background), label_weights f32[B,P],
 *                  bbox_targets / bbox_weights f32[B,P,code_size], ious f32[B,P] (clamped matched IoU)
 *   draw_heatmap   cfg_host = {pc_x0, pc_y0, voxel_x, voxel_y, out_size_factor}; heatmap f32[B,num_classes,H,W] is
 *                  cleared and drawn; box (x, y) lands at [cls][x cell][y cell] (the `center_int[[1, 0]]` fix, :662)
 *   gaussian_focal_loss  loss_sum_npos f32[2] <- (sum of element losses on clip_sigmoid(logits), count of target == 1);
 *                  grad f32[n] <- d element loss / d logit (unscaled)
 *   query_losses   loss_sums f32[2] <- (weighted focal sum over [B,P,C], weighted L1 sum over [B,P,K]); grad_cls /
 *                  grad_box in the layout of the inputs (entries outside [p_off, p_off+P) untouched)
 * --------------------------------------------------------------------------------------- */
int bfhip_decode_boxes(const float *center, const float *height, const float *dim, const float *rot,
                       const float *vel, int B, int P, int ld, int p_off, const float *cfg_host, float *boxes,
                       void *stream);
int bfhip_assign_cost(const float *boxes, int W, const float *cls_logits, int C, int ld, int p_off,
                      const float *gt_boxes, int Wg, const int32_t *gt_labels, const int32_t *n_gt, int B, int P,
                      int G, const float *cfg_host, float *cost, float *iou, void *stream);
int bfhip_hungarian(const float *cost, const int32_t *n_gt, int B, int P, int G, int32_t *assigned,
                    int32_t *status, void *stream);
int bfhip_assign_targets(const int32_t *assigned, const float *iou, const float *gt_boxes, int Wg,
                         const int32_t *gt_labels, int B, int P, int G, int num_classes, int code_size,
                         const float *cfg_host, int32_t *labels, float *label_weights, float *bbox_targets,
                         float *bbox_weights, float *ious, void *stream);
int bfhip_draw_heatmap(const float *gt_boxes, int Wg, const int32_t *gt_labels, const int32_t *n_gt, int B, int G,
                       int num_classes, int H, int W, const float *cfg_host, double gaussian_overlap,
                       int min_radius, float *heatmap, void *stream);
size_t bfhip_gaussian_focal_loss_workspace_bytes(long long n);
int bfhip_gaussian_focal_loss(const float *logits, const float *target, long long n, float clip_eps,
                              float *loss_sum_npos, float *grad, void *workspace, size_t workspace_bytes,
                              void *stream);
/* circle NMS (mmdet3d/models/layers/box3d_nms.py:186-228, called from BEVFusionHead.predict_by_feat :399-413 on the CPU):
 * dets f32[n,3] = (x, y, score); keep i32[min(n, post_max_size)] receives the kept indices, highest score first;
 * n_keep i32[1].  n <= 4096. */
int bfhip_circle_nms(const float *dets, int n, float thresh, int post_max_size, int32_t *keep, int32_t *n_keep,
                     void *stream);
/* rotate NMS: nms_bev (mmdet3d/models/layers/box3d_nms.py:234-275) -> mmcv.ops.nms_rotated, called from
 * BEVFusionHead.predict_by_feat :414-423 when test_cfg.nms_type is neither None nor 'circle'.
 * boxes f32[n,5] = (x, y, w, h, angle), scores f32[n]; the min(n, pre_max_size) highest-scoring boxes take part, box j is
 * dropped when an earlier kept box overlaps it with IoU > thresh; keep i32[min(n, pre_max_size, post_max_size)] receives
 * the kept indices (into boxes), highest score first; n_keep i32[1].  Equal scores: lower index first.
 * n <= 16384, min(n, pre_max_size) <= 4096. */
size_t bfhip_rotate_nms_workspace_bytes(int n, int pre_max_size);
int bfhip_rotate_nms(const float *boxes, const float *scores, int n, float thresh, int pre_max_size, int post_max_size,
                     int32_t *keep, int32_t *n_keep, void *workspace, size_t workspace_bytes, void *stream);
int bfhip_query_losses(const float *cls_logits, const int32_t *labels, const float *label_weights,
                       const float *box_pred, const float *bbox_targets, const float *bbox_weights,
                       const float *code_weights, int B, int C, int P, int K, int ld, int p_off, float gamma,
                       float alpha, float *grad_cls, float *grad_box, float *loss_sums, void *stream);

/* BEVFusion.voxelize (projects/BEVFusion/bevfusion/bevfusion.py:227-255: per-sample voxelize, F.pad batch id, cat, mean) for B
 * samples without the per-sample host reads: voxels f32[B][max_voxels][P][F], coors i32[B][max_voxels][3], num_points
 * i32[B][max_voxels], counts_dev i32[B] (what bfhip_hard_voxelize left on the device) -> feats f32[cap][F] (per-voxel means),
 * out_coords i32[cap][4] = (b, x, y, z); active rows are the prefix, the rest are zeros with b = -1 (inactive rows: every
 * index-driven kernel of this library skips them); n_total_dev i32[2] = (active rows, sum of counts). */
int bfhip_voxel_compact_mean(const float *voxels, const int32_t *coors, const int32_t *num_points, const int32_t *counts_dev,
                             int B, int max_voxels, int P, int F, int cap, float *feats, int32_t *out_coords,
                             int32_t *n_total_dev, void *stream);

/* ---- dense 2-D convolution, channels-last bf16, implicit GEMM on the matrix cores (csrc/conv2d.hip).
 * Replaces the cuDNN convolutions behind torch.nn.Conv2d of ConvFuser (projects/BEVFusion/bevfusion/bevfusion_head.py:26-38),
 * SECOND / SECONDFPN (mmdet3d/models/backbones/second.py:27-95, necks/second_fpn.py:30-94), shared_conv (:95-102),
 * depthnet / downsample (projects/BEVFusion/bevfusion/depth_lss.py:592-620) and GeneralizedLSSFPN
 * (projects/BEVFusion/bevfusion/bevfusion_necks.py:50-72).  x bf16 [N,H,W,Cin] with pixel pitch ldx, w bf16
 * [Cout][KH][KW][Cin] (the memory of a channels-last Conv2d weight), y bf16|f32 [N,OH,OW,Cout] with pixel pitch ldy; Cin, Cout
 * multiples of 8.  stat_partial (optional) f32[bfhip_conv2d_stat_rows()][2][Cout]: per-row-block column sums and sums of
 * squares of the fp32 accumulators (without bias) = the `partial` input of bfhip_bn2d_fwd_partials. */
int bfhip_conv2d_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil);
int bfhip_conv2d_stat_rows(int N, int OH, int OW);
int bfhip_conv2d_fwd(const void *x, int ldx, const void *w, const float *bias, void *y, int ldy, int N, int H, int W, int Cin,
                     int Cout, int KH, int KW, int stride, int pad, int dil, int out_f32, float *stat_partial, void *stream);
size_t bfhip_conv2d_dgrad_workspace_bytes(int Cin, int Cout, int KH, int KW);
int bfhip_conv2d_dgrad(const void *dy, int ldg, const void *w, void *dx, int ldx, int N, int H, int W, int Cin, int Cout,
                       int KH, int KW, int stride, int pad, int dil, int out_f32, void *workspace, size_t workspace_bytes,
                       void *stream);
/* The data gradient with the weight already transposed (wt bf16 [Cin][KH][KW][Cout], read-only), and the transposition of a
 * whole table of weights in one launch -- a training step refreshes all layers' copies once, after the optimizer, instead of
 * paying a transpose launch inside every bfhip_conv2d_dgrad.  segs_dev: nseg device records of bfhip_conv2d_wt_segment_bytes()
 * bytes each {u64 src ([Cout][taps][Cin], bf16 or f32), u64 dst, i32 Cout, i32 taps, i32 Cin, i32 src_f32, i64 first_block};
 * a segment owns ceil(Cin/32) * ceil(Cout/32) * taps blocks, first_block ascending from 0, total_blocks = their sum. */
int bfhip_conv2d_dgrad_wt(const void *dy, int ldg, const void *wt, const void *addend, int addend_stride, void *dx, int ldx, int N,
                          int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int out_f32, void *stream);
/* addend (optional, bf16, dense with pixel pitch Cin): dx = data gradient + addend in the kernel's epilogue (the second gradient path
 * of a residual connection).  addend_stride 1: [N,H,W,Cin]; 2: [N,ceil(H/2),ceil(W/2),Cin], the gradient of a stride-2 1x1 shortcut
 * over the same tensor -- it reaches the pixels with even h and w only.  Accepted only when bfhip_conv2d_dgrad_fuses_addend() says
 * so (1x1, stride 1, no padding, bf16 output) */
int bfhip_conv2d_dgrad_fuses_addend(int KH, int KW, int stride, int pad, int out_f32);
int bfhip_conv2d_wt_segment_bytes(void);
int bfhip_conv2d_weight_transpose_batched(const void *segs_dev, int nseg, long long total_blocks, void *stream);
/* fp32 operand -> bf16 pair hi = bf16(v), lo = bf16(v - hi) (v = hi + lo to 2^-16 relative), in the layouts of the three-product
 * fp32 convolution (a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix cores, fp32 accumulation): src f32 [P][C] dense ->
 * chan (optional) bf16 [P][3C] and batch (optional) bf16 [3][P][C]; bit k of order_* set = block k holds lo.  C % 8 == 0. */
int bfhip_split_bf16x3(const float *src, long long P, int C, void *chan, int order_chan, void *batch, int order_batch, void *stream);
size_t bfhip_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW);
int bfhip_conv2d_wgrad(const void *x, int ldx, const void *dy, int ldg, void *dw, int N, int H, int W, int Cin, int Cout,
                       int KH, int KW, int stride, int pad, int dil, int dw_bf16, void *workspace, size_t workspace_bytes,
                       void *stream);
/* Grouped weight gradients: dW of MANY layers in one launch per tile shape plus one slab-sum launch.  dW of a layer is a leaf of
 * the backward graph (torch/nn/modules/conv.py's weight gradient, reached through BF/bevfusion.py:143-171 and every ConvModule of
 * the dense path), so a caller may collect (x, dy, dW) during the backward and launch the group at its end: every workgroup then
 * runs ~target_steps 64-pixel steps of its layer instead of the 6-9 a one-residency-round launch of a small layer leaves it,
 * and the fp32 split slabs shrink with the split count.  Same kernels, same fixed-order slab sum as bfhip_conv2d_wgrad; only the
 * number of splits (= the fp32 summation order) differs.
 *   bfhip_wgrad_layer                    one record per layer, host memory, filled by the caller
 *   bfhip_conv2d_wgrad_groupable()       1 if the layer's geometry can join a group (else: bfhip_conv2d_wgrad)
 *   bfhip_conv2d_wgrad_group_table_bytes size of the table image for n layers
 *   bfhip_conv2d_wgrad_group_plan()      host only: plans splits / XCD chunks, writes the table image into table_host (host memory,
 *                                        e.g. pinned) and the slab workspace size into *slab_bytes; target_steps <= 0: default 96
 *   bfhip_conv2d_wgrad_group_launch()    table_dev = device copy of the image (copied by the caller, stream-ordered before this
 *                                        call); slab: 256-byte aligned device workspace of >= *slab_bytes */
typedef struct bfhip_wgrad_layer {
  const void *x, *dy; /* bf16 [N,H,W,ldx] and bf16 [N,OH,OW,ldg], 16-byte aligned */
  void *dw;           /* [Cout][KH][KW][Cin], fp32 or bf16 (dw_bf16) */
  int32_t ldx, ldg, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, dw_bf16, reserved;
} bfhip_wgrad_layer;
size_t bfhip_conv2d_wgrad_group_table_bytes(int n_layers);
int bfhip_conv2d_wgrad_groupable(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil);
int bfhip_conv2d_wgrad_group_plan(const void *layers, int n_layers, int target_steps, void *table_host, size_t table_bytes,
                                  size_t *slab_bytes);
int bfhip_conv2d_wgrad_group_launch(const void *table_host, const void *table_dev, void *slab, size_t slab_bytes, void *stream);
/* BatchNorm2d forward whose statistics pass already happened in the producing convolution's epilogue: `partial`
 * f32[nblk][2][C] (column sums, sums of squares per row block).  Otherwise as bfhip_bn2d_fwd. */
int bfhip_bn2d_fwd_partials(const void *x, const void *residual, const float *gamma, const float *beta, long long M, int C,
                            int dtype, float eps, float momentum, int relu, float *running_mean, float *running_var,
                            float *stats, void *y, const float *partial, int nblk, const int32_t *m_dev, void *stream);
/* The same, also storing the ReLU decision of every element as one bit (relu_mask u8[M][C/8]; bf16 with residual and ReLU only),
 * and the backward that reads those bits instead of the saved output y (1/16 of its bytes, in both passes of the backward). */
int bfhip_bn2d_fwd_partials_mask(const void *x, const void *residual, const float *gamma, const float *beta, long long M, int C,
                                 int dtype, float eps, float momentum, int relu, float *running_mean, float *running_var,
                                 float *stats, void *y, const float *partial, int nblk, const int32_t *m_dev,
                                 unsigned char *relu_mask, void *stream);
int bfhip_bn2d_bwd_mask(const void *dy, const void *x, const unsigned char *relu_mask, const float *stats, const float *gamma,
                        long long M, int C, int dtype, void *dx, void *dres, float *dgb, const int32_t *m_dev, void *workspace,
                        size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * gradient clipping + AdamW + bf16 parameter refresh of all parameter tensors in three launches
 *   (replaces, in the benchmarked training step, clip_grad_norm_ + torch.optim.AdamW of the reference's optim_wrapper,
 *   projects/BEVFusion/configs/nuscenes/bevfusion_lidar_voxel0075_second_secfpn_8xb4-cyclic-20e_nus-3d.py:369-372)
 *   segs_dev      : n_tensors records of bfhip_adamw_segment_bytes() bytes each:
 *                   { float *master, *m, *v; uint16_t *lowp (bf16 copy or NULL); int64 n; int32 grad_bf16, pad }
 *   grad_ptrs_dev : int64[n_tensors] device addresses of this step's gradients (0: no gradient = zero gradient), element
 *                   order = the parameter's own memory order
 *   chunks_dev    : int32[n_chunks][2] = (tensor, chunk index) covering every tensor in chunks of bfhip_adamw_chunk_elems()
 *   partial_dev   : f32[n_chunks] scratch; scalars_dev : f32[8] persistent state ([2] = step count; zero it once):
 *                   after the call [0] clip scale, [1] 1 if the gradient norm was NaN / inf (nothing was updated), [5] the norm
 *   max_norm <= 0 : no clipping.  Arithmetic: torch's fused AdamW (decoupled weight decay, fp32), bf16 copy rounded once.
 * --------------------------------------------------------------------------------------- */
int bfhip_adamw_segment_bytes(void);
int bfhip_adamw_chunk_elems(void);
int bfhip_adamw_step(const void *segs_dev, const int64_t *grad_ptrs_dev, const int32_t *chunks_dev, int n_chunks,
                     float *partial_dev, float *scalars_dev, float lr, float beta1, float beta2, float eps,
                     float weight_decay, float max_norm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVFUSION_HIP_H_ */
