/*
 * r3d.h -- C ABI of libr3d_hip.so: the MI355X (gfx950) depth -> world point-cloud
 * fusion path.  This is the drop-in boundary: plain pointers and sizes, no C++ or
 * torch types.  A Python host binds it with ctypes; the
 * reference (rainfall1998/3D_reconstruction_system) has no FFI of its own, so each
 * entry point names the reference *function* whose arithmetic it replaces.
 *
 * Conventions
 *   - every function returns an int status: R3D_OK (0) or a negative R3D_ERR_*;
 *     r3d_last_error() returns a thread-local human-readable message for the last
 *     failure on the calling thread.  Nothing throws, nothing aborts.
 *   - "d_" pointers are device (HBM) addresses valid on the ctx's GPU; "h_" pointers
 *     are host addresses owned by the caller.  The library never frees caller memory.
 *   - device-pointer entry points are ASYNCHRONOUS on the ctx's HIP stream; *_host
 *     entry points are synchronous (H2D, kernel, D2H, stream sync).
 *   - one ctx = one GPU + one stream.  Calls on one ctx are not thread-safe; different
 *     ctxs are independent.  Objects created from a ctx (r3d_camera, r3d_voxelset, r3d_nn_index,
 *     r3d_dev_alloc / r3d_host_alloc memory) must not be USED after r3d_ctx_destroy; destroying
 *     them afterwards is allowed (their destroy functions do not touch the ctx).
 *   - point clouds are AoS xyz, row-major [n][3], float32 (R3D_F32) or float64
 *     (R3D_F64).  All arithmetic is done in fp64 registers in the reference's evaluation
 *     order -- products and differences exactly as written there; the 3-term dot products of the
 *     SE(3) / 4x4 apply are an fma chain fma(r2,dz, fma(r1,dy, r0*dx)) -- and rounded ONCE on
 *     store, so R3D_F32 output is the correctly rounded fp64 result (<= 6e-8 relative per
 *     component), R3D_F64 unprojection is bit-identical to the reference and R3D_F64 world
 *     points equal the reference's fp64 up to that dot-product's rounding (~1e-16 relative).
 */
#ifndef R3D_H
#define R3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R3D_VERSION 200 /* 0.2.0 */

/* status codes */
#define R3D_OK 0
#define R3D_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, unknown enum) */
#define R3D_ERR_HIP (-2)         /* a HIP runtime call failed; see r3d_last_error() */
#define R3D_ERR_NOMEM (-3)       /* device or host allocation failed */
#define R3D_ERR_NODEVICE (-4)    /* no usable gfx950 device / device index out of range */
#define R3D_ERR_UNSUPPORTED (-5) /* valid request the library does not implement */

/* depth raster element types (reference: uint8 from cv.imread(...,IMREAD_GRAYSCALE),
 * camera_to_world.py:160; u16 / f32 cover 16-bit PNG and metric depth rasters) */
#define R3D_DEPTH_U8 0
#define R3D_DEPTH_U16 1
#define R3D_DEPTH_F32 2

/* point-cloud element types */
#define R3D_F32 0
#define R3D_F64 1

typedef struct r3d_ctx r3d_ctx;       /* one GPU + one HIP stream + scratch buffers */
typedef struct r3d_camera r3d_camera; /* pinhole intrinsics + per-column/per-row ray tables in HBM */

/* ---- library / context ------------------------------------------------------------ */
int r3d_version(void);
const char* r3d_last_error(void);
int r3d_device_count(int* n_out);
/* flags = 0: the ctx creates and owns a non-blocking stream (`stream` is ignored).
 * flags & R3D_CTX_EXTERNAL_STREAM: launch on the caller's hipStream_t `stream` (e.g. torch's current
 * stream); NULL then means the device's default stream.  The ctx never destroys an external stream. */
#define R3D_CTX_EXTERNAL_STREAM 1
int r3d_ctx_create(int device, void* stream, int flags, r3d_ctx** ctx_out);
int r3d_ctx_destroy(r3d_ctx* ctx);
int r3d_ctx_sync(r3d_ctx* ctx);
int r3d_ctx_stream(r3d_ctx* ctx, void** stream_out);
/* tuning knobs (integers; 0 = auto everywhere): launch geometry only -- "fuse_blocks", "apply_blocks"
 * (workgroup counts), "nn_variant" (sources per lane of the NN sweeps: 1/2/4), "nn_warm" (0 auto: a presorted
 * index query with the same source / output buffers as the previous one starts from that one's matches as bounds, and the ICP
 * loops run their later iterations on the wave-local kernel; 1 off; 2 never the wave-local kernel; 3 always when there are bounds: A/B), "voxel_dedupe" (0 auto, 1 off, 2 on, 3 = the pre-round-3 form with the flush barrier inside its `if`: A/B only),
 * "fuse_prefetch" (0 auto, 1 off, 2 on: a read-only sweep stages a fused launch's inputs in the 256 MiB Infinity Cache first,
 * chunk by chunk -- "fuse_chunk_mb", 0 = 96 -- so that the kernel's reads do not mix with its write stream at the DRAM.  auto
 * stages BY PROVENANCE: a launch whose small-share inputs exceed "fuse_stage_auto_mb" (default 8) MB is staged unless those very
 * bytes are presumed to be in the cache already -- i.e. a launch of this library on this device read them and fewer than
 * "fuse_resident_mb" (default 128) MB of other inputs went through since.  r3d_memcpy_h2d / _d2d / r3d_memset, the *_host
 * pipeline's uploads, r3d_comm_allgather's receives and r3d_dev_free make the library forget the ranges they touch (measured:
 * a raster an H2D copy has just written runs at 0.49 of the HBM peak plain, 0.82 staged; staging a cached one costs 8 %).
 * A FOREIGN producer (torch, another library) that rewrites an input buffer in place says so with
 * r3d_ctx_set_tuning(ctx, "fuse_inputs_fresh", 1): an event, not a state -- nothing on the device is presumed cached any more;
 * reading the key back gives the number of ranges on record; "fuse_sweeps" counts the staging sweeps enqueued so far).
 * No knob changes any result bit.  Unknown key -> R3D_ERR_INVALID.  (Kernel A/B variants live in tools/ab_kernels.hip,
 * not in the library.) */
int r3d_ctx_set_tuning(r3d_ctx* ctx, const char* key, int value);
int r3d_ctx_get_tuning(r3d_ctx* ctx, const char* key, int* value_out);

/* Read-only sweep of a device buffer on the ctx stream: leaves (up to ~100 MB of) it in the 256 MiB Infinity Cache, so that a
 * write-heavy kernel enqueued next reads it from there instead of mixing reads into its HBM write stream (the fused kernels do
 * this themselves where it pays, see "fuse_prefetch" -- and they take a range swept here as cached).  Asynchronous; changes nothing. */
int r3d_cache_prefetch(r3d_ctx* ctx, const void* d_ptr, size_t bytes);

/* ---- device memory + timing helpers (so a ctypes host needs nothing but this library) */
int r3d_dev_alloc(r3d_ctx* ctx, size_t bytes, void** d_ptr_out);
int r3d_dev_free(r3d_ctx* ctx, void* d_ptr);
int r3d_memcpy_h2d(r3d_ctx* ctx, void* d_dst, const void* h_src, size_t bytes); /* async on ctx stream */
int r3d_memcpy_d2h(r3d_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* async on ctx stream */
int r3d_memcpy_d2d(r3d_ctx* ctx, void* d_dst, const void* d_src, size_t bytes); /* async on ctx stream */
/* Synchronous device -> host copy that reaches the pinned PCIe rate into PAGEABLE memory (a NumPy array): 32 MiB chunks through
 * pinned staging buffers, the pageable copies spread over host threads (a plain copy runs at ~12 GB/s on this host).  Returns
 * when h_dst holds the data; everything enqueued on the ctx stream before it has completed by then. */
int r3d_download(r3d_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int r3d_memset(r3d_ctx* ctx, void* d_dst, int byte_value, size_t bytes);
/* Pinned (page-locked) host memory.  The *_host entry points detect pinned buffers and DMA straight from/to them;
 * pageable buffers go through the library's own pinned staging ring with multi-threaded copies. */
int r3d_host_alloc(r3d_ctx* ctx, size_t bytes, void** h_ptr_out);
int r3d_host_free(r3d_ctx* ctx, void* h_ptr);
/* HIP-event stopwatch on the ctx's stream: start records an event, stop records a second
 * one, synchronises on it and returns the elapsed milliseconds between them. */
int r3d_timer_start(r3d_ctx* ctx);
int r3d_timer_stop(r3d_ctx* ctx, float* ms_out);

/* ---- camera (replaces the hard-coded fx,fy,cx,cy of pixel_to_camera.py:25-28 and
 * camera_to_world.py:68-71).  Builds u[i]=(i-cx)/fx, v[j]=(j-cy)/fy in fp64 on the host
 * with the reference's evaluation order and keeps them in HBM. */
int r3d_camera_create(r3d_ctx* ctx, int height, int width, double fx, double fy, double cx, double cy,
                      r3d_camera** cam_out);
int r3d_camera_destroy(r3d_camera* cam);


/* ---- a1/a2: per-pixel back-projection.  Replaces gentxtcord() (pixel_to_camera.py:24-44,
 * camera_to_world.py:67-83): Z=depth[j,i]*depth_scale, X=(i-cx)/fx*Z, Y=(j-cy)/fy*Z, row-major,
 * every pixel emitted, no masking.  n_frames rasters of cam's HxW -> [n_frames*H*W][3]. */
int r3d_unproject(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                  double depth_scale, void* d_xyz_out, int out_dtype);
int r3d_unproject_host(r3d_ctx* ctx, const r3d_camera* cam, const void* h_depth, int depth_dtype, int n_frames,
                       double depth_scale, void* h_xyz_out, int out_dtype);

/* ---- a2+a3+a4 fused, batched: unproject + per-frame SE(3), frames concatenated in order.
 * Replaces the frame loop of get_file_name() (camera_to_world.py:149-172) =
 * gentxtcord (c2w:67-83) + get_pointdata/point_camera (c2w:86-105, 57-59):
 *     p_world = Rinv_f . (p_cam - t_f)
 * pose: n_frames x 12 doubles, [Rinv row-major (9), t (3)] per frame; Rinv is what
 * scipy_transfer() (c2w:53-55) returns -- computed on the host in fp64. */
int r3d_fuse_frames(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                    double depth_scale, const double* d_pose, void* d_xyz_out, int out_dtype);
int r3d_fuse_frames_host(r3d_ctx* ctx, const r3d_camera* cam, const void* h_depth, int depth_dtype, int n_frames,
                         double depth_scale, const double* h_pose, void* h_xyz_out, int out_dtype);

/* f4 colour attach (genply_noRGB, pixel_to_camera.py:55-91; BASELINE config 5 "RGBD"): the same fused launch also
 * carries the frames' colour.  d_rgb: [n_frames][H][W][3] uint8, R,G,B per pixel (the image the depth raster belongs
 * to).  d_rgba_out: [n_frames*H*W] uint32 = r | g<<8 | b<<16, i.e. bytes R,G,B,0 -- the "R G B 0" of the reference's
 * PLY rows -- point k takes the colour of pixel k.  d_pose == NULL: camera-frame points (pixel_to_camera.py).
 * +7 B/point of HBM traffic (3 read, 4 written). */
int r3d_fuse_frames_rgb(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                        double depth_scale, const double* d_pose, const unsigned char* d_rgb, void* d_xyz_out,
                        int out_dtype, uint32_t* d_rgba_out);
/* The same from / to host arrays (pageable or pinned): depth + colour stream in and xyz + rgba stream out chunk by chunk
 * through the library's pinned staging pipeline, both PCIe directions busy at once.  Synchronous. */
int r3d_fuse_frames_rgb_host(r3d_ctx* ctx, const r3d_camera* cam, const void* h_depth, int depth_dtype, int n_frames,
                             double depth_scale, const double* h_pose, const unsigned char* h_rgb, void* h_xyz_out,
                             int out_dtype, uint32_t* h_rgba_out);

/* f4, torch-facing: the BackprojectDepth layer the reference's trainer calls (monodepth2/trainer.py:150-160, 387-390;
 * upstream monodepth2 layers.py, not vendored by the reference):
 *     cam_points[b][c][p] = depth[b][p] * (inv_K[b][c][0]*x + inv_K[b][c][1]*y + inv_K[b][c][2]),  c = 0..2;  [b][3][p] = 1
 * with p = y*W + x, all fp32 like the layer.  d_depth [B][H*W], d_inv_K [B][4][4] row-major, d_cam_points [B][4][H*W].
 * The _grad entry point is the layer's backward with respect to depth: grad_depth[b][p] = sum_c grad_cam[b][c][p]*ray_c. */
int r3d_backproject_depth_f32(r3d_ctx* ctx, const float* d_depth, const float* d_inv_K, int batch, int height, int width,
                              float* d_cam_points);
int r3d_backproject_depth_grad_f32(r3d_ctx* ctx, const float* d_grad_cam_points, const float* d_inv_K, int batch, int height,
                                   int width, float* d_grad_depth);

/* Its partner in the same trainer lines, Project3D (trainer.py:150-160, 389-390; upstream layers.py):
 *     c = P[b] . points[b][:, p]   (P = (K @ T)[:, :3, :], [B][3][4] row-major, formed by the caller: sixteen numbers per image)
 *     pix[b][p] = ( (c0 / (c2 + eps) / (W-1) - 0.5) * 2,  (c1 / (c2 + eps) / (H-1) - 0.5) * 2 )        grid_sample coordinates
 * d_points [B][4][H*W] (the BackprojectDepth layout), d_pix [B][H][W][2], fp32; upstream eps = 1e-7.
 * The _grad entry point is the backward pass: d_grad_points [B][4][H*W] and/or d_grad_P [B][3][4] (either may be NULL);
 * d_grad_P is a two-stage fixed-order reduction (bitwise repeatable). */
int r3d_project3d_f32(r3d_ctx* ctx, const float* d_points, const float* d_P, int batch, int height, int width, float eps,
                      float* d_pix);
int r3d_project3d_grad_f32(r3d_ctx* ctx, const float* d_grad_pix, const float* d_points, const float* d_P, int batch,
                           int height, int width, float eps, float* d_grad_points, float* d_grad_P);

/* ---- a4 on an existing cloud: p_world = Rinv . (p_cam - t), the evaluation order of point_camera()
 * (camera_to_world.py:57-59) and of the fused kernel, so fuse_frames(depth) == se3_apply(unproject(depth))
 * bit for bit.  h_pose is ALWAYS a host pointer: 12 doubles [Rinv row-major (9), t (3)].  In-place allowed. */
int r3d_se3_apply(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* h_pose,
                  void* d_xyz_out, int out_dtype);
int r3d_se3_apply_host(r3d_ctx* ctx, const void* h_xyz_in, int in_dtype, int64_t n_points, const double* h_pose,
                       void* h_xyz_out, int out_dtype);

/* ---- a7: p' = (T . [x,y,z,1]^T)[0:3] for a general row-major 4x4.
 * Replaces local_world(flag=True)/point_camera (transfer_T_icp.py:71-97, 10-12).
 * h_T is ALWAYS a host pointer (16 doubles); in-place (d_xyz_out == d_xyz_in) is allowed. */
int r3d_apply_T(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* h_T,
                void* d_xyz_out, int out_dtype);
int r3d_apply_T_host(r3d_ctx* ctx, const void* h_xyz_in, int in_dtype, int64_t n_points, const double* h_T,
                     void* h_xyz_out, int out_dtype);

/* ---- a8: ICP estimation kernels (NOT in the reference -- transfer_T_icp.py only consumes a
 * T_data.txt made by an external tool; build-defined per SURVEY.md 8(a8)).
 * r3d_icp_nn: for each source point the index of the nearest target point under squared L2
 * computed in fp32 as d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx)) with dx = sx-tx, dy = sy-ty, dz = sz-tz (one rounding
 * for dx*dx, one per fused multiply-add; no other contraction), lowest index wins exact ties.  This expression IS
 * the specification: brute force, culled index and oracle all evaluate exactly it.
 * Non-finite input: a pair whose d2 is NaN or +inf never wins; a source with no finite d2 at all (it, or every target,
 * has a NaN / inf coordinate) gets index 0 and d2 = +inf; the pair sums leave rows with non-finite coordinates out.
 * src/tgt are float32 xyz AoS.  d_idx_out [n_src] uint32, d_d2_out [n_src] float32 (may be NULL). */
int r3d_icp_nn(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
               uint32_t* d_idx_out, float* d_d2_out);
int r3d_icp_nn_host(r3d_ctx* ctx, const float* h_src, int64_t n_src, const float* h_tgt, int64_t n_tgt,
                    uint32_t* h_idx_out, float* h_d2_out);
/* The same answer as r3d_icp_nn with spatial culling: the target cloud is Morton-sorted once into 1024-point tiles
 * with bounding boxes; a query sorts its sources the same way and sweeps only tiles whose box can still hold a
 * closer point.  Exactness is kept: identical fp32 distance expression, lowest original target index on ties
 * (a source that meets its minimum again in another group of targets looks through that group on the spot).  create and query
 * are asynchronous on the ctx stream
 * unless h_tiles_swept != NULL (then it synchronises and reports how many tile sweeps all workgroups did). */
typedef struct r3d_nn_index r3d_nn_index;
int r3d_nn_index_create(r3d_ctx* ctx, const float* d_tgt, int64_t n_tgt, r3d_nn_index** index_out);
int r3d_nn_index_destroy(r3d_nn_index* index);
/* presorted = 0: the call sorts a copy of the sources into index order itself (any source order is fine).
 * presorted = 1: the caller keeps the sources spatially coherent -- r3d_nn_index_sort_cloud once, then rigid /
 * similarity moves of the whole cloud (ICP) preserve it -- and the sort is skipped.  Results are the same either way
 * and always land at the source's position in d_src. */
int r3d_nn_index_query(r3d_nn_index* index, const float* d_src, int64_t n_src, uint32_t* d_idx_out, float* d_d2_out,
                       int presorted, int64_t* h_tiles_swept);
/* Permutes an xyz cloud in place into the index's Morton order; d_perm_out (optional, [n] uint32) receives the original
 * position of every row.  Asynchronous on the ctx stream. */
int r3d_nn_index_sort_cloud(r3d_nn_index* index, float* d_xyz, int64_t n_points, uint32_t* d_perm_out);
/* r3d_icp_accumulate: the 18 fp64 sums Umeyama needs over the matched pairs (p=src[k], q=tgt[idx[k]]),
 * pairs with d2 > max_d2 skipped when max_d2 >= 0 (d_d2 may be NULL when max_d2 < 0):
 *   sums[0]=n, [1..3]=sum p, [4..6]=sum q, [7..15]=sum p_a*q_b (a major), [16]=sum |p|^2, [17]=sum |q|^2.
 * Deterministic (fixed two-stage tree, no float atomics).  h_sums is a host pointer; synchronous.
 * Pairs whose idx[k] >= n_tgt are skipped (the index array is data; the kernel never reads past the target cloud), and so
 * are pairs in which p or q has a NaN / inf coordinate (in every sums pass of this library, fused ones included). */
int r3d_icp_accumulate(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
                       const uint32_t* d_idx, const float* d_d2, float max_d2, double* h_sums);

/* Closed-form similarity from the 18 sums (Umeyama 1991): h_T (16 doubles, row-major 4x4) = [sR t; 0 1] minimising
 * sum w |q - (sRp + t)|^2; with_scale = 0 pins s = 1.  h_rms_out (optional) = sqrt(sum w|p-q|^2 / sum w) before the
 * step.  Pure host arithmetic (3x3 one-sided Jacobi SVD, fp64), no GPU needed -- and the very code the device-side
 * solve runs.  R3D_ERR_INVALID when the fit is undefined (weight sum < 3, no spread). */
int r3d_umeyama_from_sums(const double* h_sums, int with_scale, double* h_T, double* h_rms_out);
/* Device-resident ICP state: R3D_ICP_STATE_DOUBLES doubles in HBM.
 *   [0..15] T_total (row-major 4x4, maps the ORIGINAL source onto the target), [16..31] the last step,
 *   [32] steps solved, [33] status (1 = some step was degenerate and skipped), [34] rms seen by the last step,
 *   [35] its weight sum, [48+k] rms seen by step k (while it fits). */
#define R3D_ICP_STATE_DOUBLES 512
#define R3D_ICP_STATE_HISTORY 48
int r3d_icp_state_reset(r3d_ctx* ctx, double* d_state);
/* n_iters whole ICP iterations enqueued back to back with NO host round trip: nearest neighbours (+ fused sums) ->
 * device solve -> source cloud moved in place by the step (r3d_apply_T_dev).  index != NULL: culled exact NN, d_src must
 * be in the index's order (r3d_nn_index_sort_cloud; similarity moves preserve it); index == NULL: brute force against
 * d_tgt.  The host reads d_state whenever it wants to look at the rms history / convergence. */
int r3d_icp_iterate(r3d_ctx* ctx, r3d_nn_index* index, float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
                    uint32_t* d_idx, float* d_d2, int n_iters, int with_scale, float max_d2, double* d_state);

/* ---- a8 continued: RIGID POINT-TO-PLANE registration of two partially overlapping single-view clouds -- the reference's
 * own use of ICP ("match the point clouds corresponding to two images", readme.md:25; transfer_T_icp.py:107-108 merges the
 * camera clouds ./point/0.txt and ./point/24.txt with the T this produces).  Build-defined; this text is the specification.
 *
 * r3d_normals_organized: unit normals of an ORGANISED cloud [n_frames][height][width][3] f32 -- the row-major raster order
 * gentxtcord emits (pixel_to_camera.py:34-44).  For an interior pixel: a = p[j][i+1] - p[j][i-1], b = p[j+1][i] - p[j-1][i],
 * n = a x b normalised (fp64 throughout, rounded once to f32), turned towards the viewpoint (h_viewpoint, 3 doubles; NULL =
 * the origin, i.e. a camera-frame cloud).  The zero vector ("no plane here") is written for raster-border pixels, when the
 * centre or a neighbour is non-finite or AT the viewpoint (Z = 0 pixels: the reference emits every pixel, p2c:24-44), when a
 * neighbour's range differs from the centre's by more than max_jump x that range (a depth edge), and when a x b vanishes. */
int r3d_normals_organized(r3d_ctx* ctx, const float* d_xyz, int64_t n_frames, int height, int width, float max_jump,
                          const double* h_viewpoint, float* d_normals_out);
/* n_iters whole point-to-plane iterations with NO host round trip: culled exact NN (d_src in the index's order,
 * r3d_nn_index_sort_cloud; rigid moves preserve it) -> residuals + classes -> per-class selection -> 29 sums -> device solve
 * -> source moved.  d_tgt_normals: normals of the index's target cloud in its ORIGINAL order.  d_src_orig != NULL: the cloud
 * as it was when d_state was reset (same order as d_src); every iteration then writes d_src = T_total . d_src_orig (one
 * rounding per point however many steps were taken); NULL: d_src is moved in place step by step.  d_state: the
 * R3D_ICP_STATE_DOUBLES layout above ([34] / history = rms of r over the kept pairs).  Asynchronous. */
int r3d_icp_iterate_plane(r3d_ctx* ctx, r3d_nn_index* index, const float* d_src_orig, float* d_src, int64_t n_src,
                          const float* d_tgt_normals, uint32_t* d_idx, float* d_d2, int n_iters, float trim_q, float gate_scale,
                          float max_d2, double* d_state);

/* ---- (e) multi-GPU: one process per GPU, frames sharded in contiguous blocks (the frame loop of camera_to_world.py:149-172
 * carries no state between frames), ONE exchange step: an all-gather over RCCL / xGMI.  The reference has no
 * counterpart (single process, no collective); these entry points let a ctypes or plain-C host shard without torch.
 * librccl is dlopen'ed on first use (an RCCL already in the process is reused); without it they return
 * R3D_ERR_UNSUPPORTED.  All transfers are asynchronous on the ctx's stream. */
typedef struct r3d_comm r3d_comm;
#define R3D_COMM_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId) and hands the 128 bytes to every rank by any means (file, pipe, MPI, torch). */
int r3d_comm_unique_id(void* id_out);
/* collective over all ranks (ncclCommInitRank on the ctx's GPU); one GPU per rank. */
int r3d_comm_create(r3d_ctx* ctx, const void* id, int rank, int world, r3d_comm** comm_out);
int r3d_comm_destroy(r3d_comm* comm);
/* any out pointer may be NULL; *rccl_origin_out says which librccl was bound */
int r3d_comm_info(const r3d_comm* comm, int* rank_out, int* world_out, const char** rccl_origin_out);
/* What the communicator says about ITSELF: ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion (-1 where the
 * bound library lacks the call); any out pointer may be NULL. */
int r3d_comm_rccl_report(const r3d_comm* comm, int* count_out, int* user_rank_out, int* device_out, int* version_out);
/* All-gather of byte shards of possibly UNEQUAL length: rank r's h_counts[r] bytes land at offset sum(h_counts[0..r))
 * of d_recv on every rank (rank order = pose-file order).  d_send may already be this rank's slot of d_recv (in place).
 * algo: R3D_GATHER_AUTO (ncclAllGather when the shards are equal, else direct), R3D_GATHER_NCCL (equal shards only),
 * R3D_GATHER_DIRECT (one grouped ncclSend/ncclRecv per peer: each shard crosses its own xGMI link once). */
#define R3D_GATHER_AUTO 0
#define R3D_GATHER_NCCL 1
#define R3D_GATHER_DIRECT 2
int r3d_comm_allgather(r3d_comm* comm, const void* d_send, const int64_t* h_counts, void* d_recv, int algo);
/* "outputs" assembly: all-gather of the world-frame xyz shards (12 or 24 B/point over the fabric). */
int r3d_allgather_xyz(r3d_comm* comm, const void* d_shard, const int64_t* h_points_per_rank, int dtype, void* d_full,
                      int algo);
/* "inputs" assembly: all-gather of the depth rasters (+ pose rows when d_pose_all != NULL), 1-4 B/point over the fabric;
 * every rank then runs r3d_fuse_frames over all frames locally (same kernel, same bits as the xyz all-gather). */
int r3d_allgather_inputs(r3d_comm* comm, const void* d_depth, int depth_dtype, const int64_t* h_frames_per_rank, int height,
                         int width, const double* d_pose, void* d_depth_all, double* d_pose_all, int algo);
/* in-place sum over ranks (e.g. the 18 ICP sums of a sharded source cloud) */
int r3d_comm_allreduce_sum_f64(r3d_comm* comm, double* d_buf, int64_t n);

/* ---- f1: reference-layout ASCII serialisation on the host (multi-threaded C++).
 * r3d_format_ply: the byte layout of genply() (camera_to_world.py:112-134): header with 4-space
 * indents, "%.4f %.4f %.4f \n" rows, "\n    " trailer.  Two-call protocol: with h_buf == NULL
 * returns the exact byte count in *n_bytes_out; otherwise writes at most buf_cap bytes. */
int r3d_format_ply(const void* h_xyz, int dtype, int64_t n_points, char* h_buf, size_t buf_cap,
                   size_t* n_bytes_out);
/* Same bytes straight to a file (formatted and written in slabs; replaces the open/write of c2w:122-132). */
int r3d_write_ply(const char* path, const void* h_xyz, int dtype, int64_t n_points);
/* f1's optional binary flag: the same vertices as a STANDARD binary_little_endian PLY (float x, y, z; the header without the
 * reference template's indents).  12 bytes per vertex; not the reference's bytes -- an opt-in. */
int r3d_write_ply_binary(const char* path, const void* h_xyz, int dtype, int64_t n_points);
/* f4: the coloured layout of genply_noRGB() (pixel_to_camera.py:55-91): uchar red/green/blue/alpha header lines and
 * "%.4f %.4f %.4f R G B 0\n" rows; h_rgb is [n][3] uint8 in R,G,B order, point k takes colour k. */
int r3d_write_ply_rgb(const char* path, const void* h_xyz, int dtype, const unsigned char* h_rgb, int64_t n_points);
/* The same file from the rgba words r3d_fuse_frames_rgb produces ([n] uint32, bytes R,G,B,0 in memory). */
int r3d_write_ply_rgba(const char* path, const void* h_xyz, int dtype, const uint32_t* h_rgba, int64_t n_points);
/* "X,Y,Z\n" lines with Python repr() float formatting -- the camera / world txt of
 * camera_to_world.py:80-81, 103-104 and transfer_T_icp.py:87,93.  If h_z_raw != NULL the third column
 * is printed as that raw integer raster value (the reference's camera txt prints str(np.uint8));
 * z_raw_dtype is R3D_DEPTH_U8 or R3D_DEPTH_U16.  append: 0 truncates ('w'), 1 appends ('a'). */
int r3d_write_xyz_txt(const char* path, const void* h_xyz, int dtype, int64_t n_points, const void* h_z_raw,
                      int z_raw_dtype, int append);
/* One such file per frame -- the ./point/<stem>.txt that camera_to_world.py:163-165 leaves for every pose line: file k takes
 * points [k * points_per_file, (k + 1) * points_per_file) of h_xyz (and of h_z_raw).  Files are spread over the host threads
 * the process may use; same bytes as n_files calls of r3d_write_xyz_txt(..., append = 0). */
int r3d_write_xyz_txt_batch(const char* const* paths, int n_files, const void* h_xyz, int dtype, int64_t points_per_file,
                            const void* h_z_raw, int z_raw_dtype);
/* The same text into a caller buffer; two-call protocol like r3d_format_ply. */
int r3d_format_xyz_txt(const void* h_xyz, int dtype, int64_t n_points, const void* h_z_raw, int z_raw_dtype,
                       char* h_buf, size_t buf_cap, size_t* n_bytes_out);
/* The way back (f1: "a fast X,Y,Z parser"): rows of text -> [n][3] fp64 on host threads.  separator ',' reads the txt
 * lines above the way get_pointdata / local_world do (camera_to_world.py:92-98, transfer_T_icp.py:74-80): the first three
 * comma-separated fields of every non-blank line, extra fields ignored; separator ' ' reads blank-separated rows (the
 * body of the PLY layout).  Numbers are parsed correctly rounded (= Python float()).  h_xyz_out == NULL only counts.
 * A line that does not parse returns R3D_ERR_INVALID with its 1-based number in *bad_line_out (may be NULL). */
int r3d_parse_xyz_text(const char* h_text, size_t n_bytes, int separator, double* h_xyz_out, int64_t cap_points,
                       int64_t* n_points_out, int64_t* bad_line_out);

/* ---- f1 on the device: the same text formatted by the GPU (csrc/r3d_textfmt.hip), so that the TEXT leaves the device
 * (41 B/point of camera txt) instead of fp64 clouds into pageable memory, and the host only copies it into files.
 * kind R3D_TEXT_XYZ_TXT: the lines of r3d_write_xyz_txt (d_aux = the integer third column, u8 / u16 per aux_dtype, or NULL);
 * R3D_TEXT_PLY_ROWS: the vertex rows of r3d_write_ply ("%.4f %.4f %.4f \n", no header / trailer); R3D_TEXT_PLY_ROWS_RGB: the
 * rows of r3d_write_ply_rgb / _rgba (d_aux = colours, aux_dtype = 3 or 4 bytes apart).  Byte-identical to the host formatter.
 * d_xyz is a device [n][3] cloud; rows appear in point order in d_text (device, text_cap bytes).  Two-call protocol:
 * d_text == NULL only computes *n_bytes_out and the segment offsets.  segment_points > 0: h_segment_offsets_out
 * [ceil(n / segment_points) + 1] receives the byte offset of the first row of every block of segment_points points (a frame's
 * camera txt) and, last, the total.  Synchronises the ctx's stream once (the sizes come to the host); the rows themselves are
 * enqueued.  R3D_ERR_UNSUPPORTED when a "%.4f" coordinate has magnitude >= 2^40 (more digits than a device row holds): format
 * that cloud on the host. */
#define R3D_TEXT_XYZ_TXT 0
#define R3D_TEXT_PLY_ROWS 1
#define R3D_TEXT_PLY_ROWS_RGB 2
int r3d_format_text_device(r3d_ctx* ctx, int kind, const void* d_xyz, int dtype, int64_t n_points, const void* d_aux, int aux_dtype,
                           int64_t segment_points, char* d_text, size_t text_cap, int64_t* h_segment_offsets_out,
                           int64_t* n_bytes_out);
/* Device bytes -> files: file k = head (host bytes), then d_bytes[0 .. n_bytes) (device memory: text made by
 * r3d_format_text_device, or any other bytes -- the f32 cloud itself for a binary PLY), then tail (host bytes).
 * Host threads take the files largest first, each through its own pinned 1 MiB pieces and stream (PCIe and write() overlap);
 * one file is one sequential write stream, different files are written side by side.  Waits for the ctx's stream first. */
typedef struct r3d_text_file {
  const char* path;
  const char* head;
  size_t head_bytes;
  const void* d_bytes;
  size_t n_bytes;
  const char* tail;
  size_t tail_bytes;
} r3d_text_file;
int r3d_write_device_text_files(r3d_ctx* ctx, const r3d_text_file* files, int n_files);

/* ---- f3 ingestion: the depth rasters of camera_to_world.py:160 (`cv.imread(path, IMREAD_GRAYSCALE)`) decoded by host
 * threads.  Native path: non-interlaced greyscale PNG, 8 bits (-> uint8, same bytes as OpenCV) or 16 bits (-> uint16);
 * any other PNG flavour returns R3D_ERR_UNSUPPORTED (the Python host then falls back to cv2 / PIL).
 * r3d_png_gray_info: header only.  r3d_png_gray_decode_batch: n files of identical height x width x bit_depth into one
 * contiguous [n][height][width] buffer (pageable or pinned). */
int r3d_png_gray_info(const char* path, int* height, int* width, int* bit_depth);
int r3d_png_gray_decode_batch(const char* const* paths, int n_files, void* h_out, int height, int width, int bit_depth);
/* The SAME raster cv.imread(path, IMREAD_GRAYSCALE) returns, for every non-interlaced 8/16-bit PNG: uint8 [n][height][width].
 * 8-bit grey: as stored; 16-bit: the high byte (libpng's strip_16, OpenCV's choice); alpha: dropped; RGB(A): converted by
 * `rule` --
 *   R3D_GRAY_OPENCV_PNG  (9797 R + 19234 G + 3737 B) >> 15, a pixel with R = G = B keeps its value: libpng's
 *                        png_set_rgb_to_gray(1, 0.299, 0.587), which is what OpenCV's PNG reader (4.2.0, the reference's pin,
 *                        and later) calls instead of cvtColor; the reference's camera_to_world.py:160 on a PNG file;
 *   R3D_GRAY_CVTCOLOR    (4899 R + 9617 G + 1868 B + 8192) >> 14: cv.cvtColor(BGR2GRAY), imread's rule for the formats whose
 *                        decoders deliver colour (BMP, TIFF, WebP).
 * (Files carrying gamma information -- gAMA / sRGB / iCCP chunks -- make libpng convert in linear light: not restated; a pixel
 * with differing channels in such a file -> R3D_ERR_UNSUPPORTED under R3D_GRAY_OPENCV_PNG, equal channels pass through as in libpng.
 * Palette, interlaced and sub-byte PNGs: R3D_ERR_UNSUPPORTED.)  r3d_rgb_to_gray_u8: the rule alone, on [n][channels] 8-bit
 * R,G,B(,A) pixels already in memory. */
#define R3D_GRAY_OPENCV_PNG 0
#define R3D_GRAY_CVTCOLOR 1
int r3d_png_gray8_info(const char* path, int* height, int* width);
int r3d_png_gray8_decode_batch(const char* const* paths, int n_files, unsigned char* h_out, int height, int width, int rule);
int r3d_rgb_to_gray_u8(const unsigned char* pixels, int64_t n_pixels, int channels, int rule, unsigned char* gray_out);
/* The same for JPEG depth files (AirSim writes its depth images as 3-channel JPG, airsim/main.cpp:1369-1392): OpenCV's JPEG reader
 * asks libjpeg for GREY output, i.e. the luma component alone through libjpeg's default integer IDCT ("islow", jidctint.c), + 128,
 * clamped -- restated here, pinned byte for byte against libjpeg-turbo (PIL's draft('L') decode makes the same request).
 * Sequential Huffman JPEGs of 1 or 3 (YCbCr) components with full-resolution luma, restart intervals included; progressive,
 * arithmetic-coded, 12-bit, CMYK and RGB-tagged files: R3D_ERR_UNSUPPORTED.  n files into one [n][height][width] buffer. */
int r3d_jpeg_gray_info(const char* path, int* height, int* width);
int r3d_jpeg_gray_decode_batch(const char* const* paths, int n_files, unsigned char* h_out, int height, int width);
/* The colour images of the RGBD path (the `Image.open(imgpath)` of genply_noRGB, pixel_to_camera.py:58-60): non-interlaced
 * 8-bit PNGs -- RGB, RGBA (alpha dropped) or grey (replicated) -- as R,G,B bytes, n files into one [n][height][width][3]
 * buffer, which is what r3d_fuse_frames_rgb takes.  *channels = samples per pixel in the file. */
int r3d_png_rgb_info(const char* path, int* height, int* width, int* channels);
int r3d_png_rgb_decode_batch(const char* const* paths, int n_files, unsigned char* h_out, int height, int width);
/* The same for JPEG colour files (AirSim's scene images, airsim/main.cpp:1369-1392): the bytes PIL / libjpeg(-turbo) give by
 * default -- every component through the islow IDCT, chroma to full resolution by libjpeg's "fancy" triangle upsampling
 * (jdsample.c h2v1 / h2v2; replication when a chroma row has fewer than three samples), YCbCr -> RGB through jdcolor.c's
 * 16-bit fixed-point tables; a grey JPEG is replicated.  Pinned byte for byte against PIL over sizes, qualities and
 * 4:4:4 / 4:2:2 / 4:2:0.  (That is libjpeg-turbo's / libjpeg 6b's decoder, the one in current Pillow wheels and in OpenCV; IJG
 * libjpeg 7+ upsamples subsampled chroma in the DCT domain and gives slightly different COLOURS for 4:2:x files.)
 * Same refusals as the grey reader, plus any other chroma layout: R3D_ERR_UNSUPPORTED (the Python
 * host then lets PIL decode, which IS the reference's reader for colour).  *components = 1 or 3. */
int r3d_jpeg_rgb_info(const char* path, int* height, int* width, int* components);
int r3d_jpeg_rgb_decode_batch(const char* const* paths, int n_files, unsigned char* h_out, int height, int width);

/* ---- f2: occupied-voxel set + OctoMap binary export.  Replaces the per-point tree.updateNode(xyz, True) loop,
 * updateInnerOccupancy() and writeBinary() of octomap/txt_transfer_octomap.py:16-36 and
 * octomap/ply_transfer_octomap.py:16-48 (arithmetic in the un-vendored OctoMap library; restated from its
 * published semantics: float coordinates, key = (int)floor((1/res)*x) + 32768 per axis, depth 16, hits only).
 * The set lives in HBM as an open-addressing hash table of 48-bit Morton codes; `capacity` = slots
 * (rounded up to a power of two; >= 2x the expected number of distinct voxels keeps probing short). */
typedef struct r3d_voxelset r3d_voxelset;
struct r3d_comm; /* multi-GPU communicator, declared below */
int r3d_voxelset_create(r3d_ctx* ctx, double resolution, int64_t capacity, r3d_voxelset** vs_out);
int r3d_voxelset_destroy(r3d_voxelset* vs);
int r3d_voxelset_clear(r3d_voxelset* vs);
/* Insert float32 xyz points (on the ctx stream; may be called once per frame batch).  Two paths with the same result:
 *   1  per-workgroup LDS set in front of 64-bit CAS into the table: for clouds whose neighbouring points share voxels (scans);
 *   2  sort-merge: region-tagged keys, two radix passes, every table region updated in LDS and streamed back -- no random
 *      access to HBM: for clouds where nearly every point has a voxel of its own (2-3x faster there; needs a table of
 *      2^16..2^29 slots).
 * Tuning key "voxel_path": 0 (default) = inserts of >= 2^22 points into a table of <= 16 slots per point are SAMPLED first
 * (256 groups of 4096 neighbouring points: distinct voxels per point; path 2 when its cost -- per point and per table slot -- comes
 * out below path 1's, which on a 2-slots-per-point table is the case below ~7 points per voxel among neighbours) -- that sample
 * synchronises the stream once (16 bytes come back); smaller inserts take path 1 without asking.  1 / 2 force a path
 * (asynchronous).  "voxel_last_path" reads back which one the last insert took. */
int r3d_voxelset_insert(r3d_voxelset* vs, const float* d_xyz, int64_t n_points);
/* The cloud AND the map in one launch: r3d_fuse_frames_rgb (f32 xyz; d_pose NULL = camera frame; d_rgb and d_rgba_out both
 * NULL = no colour) followed by r3d_voxelset_insert of the points it wrote, without reading them back: the keys are taken
 * from the very floats the store writes.  Same cloud bytes, same set, same counters as the two calls.  (camera_to_world.py
 * :77-86 feeding txt_transfer_octomap.py:16-26 in the reference.)  The call picks the faster FORM itself ("voxel_path" 0): a
 * big batch is probed -- its first ~2^18 points are fused by the plain kernel and sampled as above -- and the rest goes
 * either through the one-launch kernel (scans) or through plain fuse + sort-merge insert of the whole cloud (no surfaces);
 * "voxel_path" 1 = always the one-launch kernel, 2 = always fuse + sort-merge. */
int r3d_fuse_frames_voxel(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                          double depth_scale, const double* d_pose, const unsigned char* d_rgb, float* d_xyz_out,
                          uint32_t* d_rgba_out, r3d_voxelset* vs);
int r3d_voxelset_insert_host(r3d_voxelset* vs, const float* h_xyz, int64_t n_points);
/* synchronises; any pointer may be NULL.  n_ignored = points outside the 2^16 key range or non-finite
 * (OctoMap drops them); n_overflow > 0 means the table was too small and the set is incomplete. */
int r3d_voxelset_stats(r3d_voxelset* vs, int64_t* n_voxels, int64_t* n_ignored, int64_t* n_overflow);
/* distinct voxels as ascending 48-bit Morton codes (3 bits per level, x lowest: OctoMap's child index order).
 * h_codes_sorted == NULL only reports the count. */
int r3d_voxelset_codes(r3d_voxelset* vs, uint64_t* h_codes_sorted, int64_t cap, int64_t* n_out);
/* Insert ready-made 48-bit Morton codes (e.g. another rank's distinct voxels); codes with bits above 48 count as ignored. */
int r3d_voxelset_insert_codes(r3d_voxelset* vs, const uint64_t* d_codes, int64_t n_codes);
/* Config 5 (frames sharded over the GPUs, ONE map): collective over `comm`.  Every rank has voxelised its own shard of the
 * world cloud into its own set; the ranks all-gather only their DISTINCT codes (8 B/voxel, unequal shards -- the 12 B/point
 * of the clouds never leave their GPU) and fold them in.  Afterwards every rank's set is the union.  Synchronises.
 * `comm` must live on the set's context (one stream orders the set's kernels and the exchange).  If ANY rank's set has
 * overflowed, every rank takes part in the first exchange and then returns R3D_ERR_NOMEM: no rank is left waiting. */
int r3d_voxelset_union(r3d_voxelset* vs, struct r3d_comm* comm);
/* In-place ascending sort of 64-bit keys in HBM by their low key_bits bits (stable LSD radix sort, 8-bit digits;
 * asynchronous on the ctx stream).  Building block of r3d_voxelset_codes, exported for tests and reuse. */
int r3d_sort_u64(r3d_ctx* ctx, uint64_t* d_keys, int64_t n_keys, int key_bits);
/* OctoMap ".bt" bytes (header + pruned maximum-likelihood tree, depth first) for ascending unique codes. */
int r3d_octree_format_bt(const uint64_t* h_codes_sorted, int64_t n_codes, double resolution, char* h_buf,
                         size_t buf_cap, size_t* n_bytes_out, int64_t* n_nodes_out);
int r3d_octree_write_bt(const char* path, const uint64_t* h_codes_sorted, int64_t n_codes, double resolution,
                        int64_t* n_nodes_out);

#ifdef __cplusplus
}
#endif
#endif /* R3D_H */
