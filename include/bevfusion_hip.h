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
#define BFHIP_OP_SPCONV_BWD 7
#define BFHIP_OP_RULEBOOK 8
#define BFHIP_OP_BEV_AUX 9
#define BFHIP_OP_SCATTER_FWD 10
#define BFHIP_OP_SCATTER_BWD 11
#define BFHIP_OP_COUNT 16

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

#ifdef __cplusplus
}
#endif
#endif /* BEVFUSION_HIP_H_ */
