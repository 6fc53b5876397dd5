/* r3d_internal_api.h -- entry points of libr3d_hip.so that are NOT part of the drop-in boundary.
 *
 * The building blocks of the ICP estimators' Python driver (3d_reconstruction_system_amd/icp.py: selection, partial sums,
 * cloud reordering, device-side solves, multi-start moves) and a host self-test hook.  They are exported extern "C" with the
 * conventions of r3d.h -- status codes, r3d_last_error(), caller-owned buffers, asynchronous on the ctx's stream unless a
 * comment says otherwise -- because that driver binds them through ctypes and the tests call them one by one; they may
 * change between versions without notice.  A host that wants ICP uses r3d_icp_iterate / r3d_icp_iterate_plane and the
 * r3d_nn_index_* calls of r3d.h.
 */
#ifndef R3D_INTERNAL_API_H
#define R3D_INTERNAL_API_H

#include "r3d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Self-test hook (no GPU needed): floor(x / d) computed with the host-made magic number the kernels use for
 * pixel -> (row, column) and tile -> (frame, tile) splits.  d >= 1, x < 2^31. */
int r3d_selftest_magic_div(uint32_t d, uint32_t x, uint32_t* q_out);


/* The same with the matrix in HBM (16 doubles, row-major; e.g. the step a device-side ICP solve just wrote). */
int r3d_apply_T_dev(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* d_T,
                    void* d_xyz_out, int out_dtype);
/* n_transforms copies of one cloud, copy k moved by h_Ts[16 k .. 16 k + 15] (row-major 4x4) and written to block k of
 * d_xyz_out (n_points rows each): the candidates of a multi-start in one call.  Asynchronous. */
int r3d_apply_T_many(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* h_Ts, int n_transforms,
                     void* d_xyz_out, int out_dtype);

/* Builds the index anew for another target cloud of n_tgt <= the size it was created with, reusing its allocations (an
 * estimator that probes several clouds keeps one index object instead of paying eight hipMalloc / hipFree pairs per probe).
 * Asynchronous. */
int r3d_nn_index_rebuild(r3d_nn_index* index, const float* d_tgt, int64_t n_tgt);
/* Rows d_rows[0], d_rows[1], ... (n_out uint32 row numbers; a number >= n_points yields a NaN row) of a device xyz cloud into
 * d_xyz_out -- e.g. the permutation r3d_nn_index_sort_cloud reports, applied to a second cloud.  Asynchronous. */
int r3d_gather_rows(r3d_ctx* ctx, const float* d_xyz, int64_t n_points, const uint32_t* d_rows, int64_t n_out, float* d_xyz_out);
/* d_inverse_out[d_perm[j]] = j for a permutation of 0..n-1 (uint32), and d_values[k] <- d_table[d_values[k]] in place
 * (0xffffffff where d_values[k] >= n_table): row numbers reported against one ordering of a cloud, re-expressed in another --
 * e.g. neighbours found through an index built before r3d_nn_index_sort_cloud rearranged the cloud.  Asynchronous. */
int r3d_permutation_invert(r3d_ctx* ctx, const uint32_t* d_perm, int64_t n, uint32_t* d_inverse_out);
int r3d_remap_u32(r3d_ctx* ctx, uint32_t* d_values, int64_t n, const uint32_t* d_table, int64_t n_table);
/* Rows first, first + step, ... (n_out of them) of a device xyz cloud into d_xyz_out: strided samples without a trip to the
 * host.  Asynchronous. */
int r3d_gather_rows_strided(r3d_ctx* ctx, const float* d_xyz, int64_t n_points, int64_t first, int64_t step, int64_t n_out,
                            float* d_xyz_out);

/* The same, for a cloud that still holds rows which are no points: rows with a NaN / inf coordinate end up BEHIND the
 * valid ones (in their input order), *n_valid_out = the number of valid rows in front (synchronous: it waits for the
 * count).  With r3d_cloud_zero_rows_to_nan first -- the (0,0,0) rows gentxtcord emits for pixels without depth
 * (pixel_to_camera.py:34-44) -- this replaces the host-side row filter in front of an ICP (4 ms of NumPy for 307k rows). */
int r3d_nn_index_sort_cloud_valid(r3d_nn_index* index, float* d_xyz, int64_t n_points, uint32_t* d_perm_out,
                                  int64_t* n_valid_out);
int r3d_cloud_zero_rows_to_nan(r3d_ctx* ctx, float* d_xyz, int64_t n_points);

/* Weighted, device-resident variant: the 18 sums go to d_sums_out (HBM, 18 doubles), asynchronously on the ctx stream.
 * dead_zone > 0 weights every pair by w = max(0, 1 - dead_zone/d), d = sqrt(d2) -- the IRLS weight of the cost
 * max(0, d - dead_zone)^2, which treats a densely sampled cloud as the solid it samples (a match closer than the
 * sampling resolution carries no information about the transform); every sum above is then a weighted sum and
 * sums[0] the weight total.  dead_zone <= 0: w = 1.  d_d2 is required when max_d2 >= 0 or dead_zone > 0.
 * d_idx == NULL pairs row k with row k (with d_tgt == d_src: the moments sum p, sum p p^T, sum |p|^2 of one cloud). */
int r3d_icp_accumulate_dev(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
                           const uint32_t* d_idx, const float* d_d2, float max_d2, float dead_zone, double* d_sums_out);
/* r3d_nn_index_query with the pair sums taken in the query kernel's own epilogue (each lane holds its source and its
 * winner there): one pass instead of NN + gather.  Same idx / d2 as r3d_nn_index_query, same sums as
 * r3d_icp_accumulate_dev up to fp64 summation order; bitwise repeatable run to run.  d_d2_out must not be NULL. */
int r3d_nn_index_query_sums(r3d_nn_index* index, const float* d_src, int64_t n_src, uint32_t* d_idx_out,
                            float* d_d2_out, int presorted, float max_d2, float dead_zone, double* d_sums_out);

/* One solve on the GPU (single thread, fp64): step = umeyama(d_sums); T_total <- step . T_total; history. Asynchronous. */
int r3d_icp_solve_dev(r3d_ctx* ctx, const double* d_sums, int with_scale, double* d_state);

/* Exact order statistic of a device fp32 array: the finite values (NaN, +-inf never count) sorted ascending, the element of
 * rank floor(q (m - 1)) of the m finite ones (numpy.quantile(..., method="lower")); +inf and count 0 when there are none.
 * Three histogram passes on the GPU, 8 bytes come back.  Synchronous. */
int r3d_select_quantile_f32(r3d_ctx* ctx, const float* d_values, int64_t n, double q, float* h_value_out, int64_t* h_count_out);
/* The same selection left in HBM: d_out8 receives {float value; uint32 count} (8 bytes).  Asynchronous. */
int r3d_select_quantile_f32_dev(r3d_ctx* ctx, const float* d_values, int64_t n, double q, void* d_out8);
/* Robust means of n_classes (<= 32) consecutive blocks of per_class device floats: per block the fp64 mean of the finite values
 * that are <= the block's `keep` order statistic (the rule above); +inf for a block without finite values.  The estimator's
 * multi-start judges all its candidate poses with one such call per direction.  Synchronous; n_classes doubles come back. */
int r3d_trimmed_means_f32(r3d_ctx* ctx, const float* d_values, int n_classes, int64_t per_class, double keep, double* h_means_out);
/* One matched pair = (p = src[k], q = tgt[idx[k]], n = tgt_normals[idx[k]]); it is ADMISSIBLE when idx[k] < n_tgt, n is not
 * the zero vector, p, q, n are finite, and (max_d2 < 0 or d2[k] <= max_d2).  Residual r = n.x (p.x - q.x) + n.y (p.y - q.y)
 * + n.z (p.z - q.z) in fp64, left to right.  r3d_icp_plane_residuals writes (float)(r r) per source row, +inf for pairs that
 * are not admissible, and (d_class_out, optional) the DIRECTION CLASS of the pair's target normal, 255 for such pairs:
 *   class = 8 major + 4 [n_major < 0] + 2 [n_(major+1) < 0] + [n_(major+2) < 0]   (24 classes; indices cyclic in x, y, z)
 * where major is the axis of largest |component| of the stored f32 normal (lowest axis on ties).  Asynchronous. */
int r3d_icp_plane_residuals(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, const float* d_tgt_normals,
                            int64_t n_tgt, const uint32_t* d_idx, const float* d_d2, float max_d2, float* d_r2_out,
                            unsigned char* d_class_out);
/* The 29 fp64 sums of the linearised point-to-plane normal equations over the admissible pairs -- with 0 < trim_q < 1 only
 * over those whose (float)(r r) is <= g_c x gate_scale (one fp32 multiply), g_c = the trim_q order statistic
 * (r3d_select_quantile_f32's rule) of that value over the admissible pairs OF THE SAME DIRECTION CLASS c.  The statistic is
 * taken per class so that a wall whose pairs all disagree with the current pose keeps its say against walls that already
 * fit (ranking all pairs together drops exactly the family that carries the missing constraint, and the pose slides along
 * it); trim_q = 0.5, gate_scale = 20 keeps what lies within ~3 sigma of each class's median-based scale; gate_scale = 1 is
 * plain rank trimming.  All selections run on the GPU.  With J = [p x n ; n]:
 *   sums[0] = pairs, [1] = sum r^2, [2..7] = sum J r, [8..28] = upper triangle of sum J J^T, row-major.
 * Deterministic (wave shuffle tree -> LDS -> fixed-order second stage, no float atomics).  d_sums_out: 29 doubles in HBM.
 * Asynchronous. */
#define R3D_PLANE_SUMS 29
int r3d_icp_plane_accumulate(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, const float* d_tgt_normals,
                             int64_t n_tgt, const uint32_t* d_idx, const float* d_d2, float max_d2, float trim_q,
                             float gate_scale, double* d_sums_out);
/* The rigid step from the 29 sums: solve (sum J J^T) x = - sum J r (Cholesky, fp64, unknowns scaled to one length unit),
 * x = (omega, v); h_T = [exp([omega]x) v; 0 1] (Rodrigues: exactly a rotation).  h_rms_out (optional) = sqrt(sum r^2 / pairs)
 * before the step.  Pure host arithmetic -- and the very code the device-side solve runs.  R3D_ERR_INVALID (h_T = identity)
 * with fewer than 6 pairs or when the matched normals leave a freedom unconstrained (one plane, two parallel walls ...). */
int r3d_plane_step_from_sums(const double* h_sums, double* h_T, double* h_rms_out);

#ifdef __cplusplus
}
#endif

#endif /* R3D_INTERNAL_API_H */
