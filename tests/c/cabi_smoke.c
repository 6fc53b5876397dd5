/* Plain C consumer of include/r3d.h: proves the boundary is a C ABI (no C++, no Python, no torch).
 * Build: gcc -std=c99 -Iinclude tests/c/cabi_smoke.c -o cabi_smoke -L3d_reconstruction_system_amd -lr3d_hip -lm
 * Fuses 2 frames of a 4x6 raster (SURVEY KAT-1 depth pattern) and checks two hand-computed points. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "r3d.h"

#define CK(call)                                                                \
  do {                                                                          \
    int rc_ = (call);                                                           \
    if (rc_ != R3D_OK) {                                                        \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, r3d_last_error());          \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

int main(void) {
  enum { H = 4, W = 6, F = 2, N = F * H * W };
  unsigned char depth[N];
  double pose[F * 12];
  float xyz[N * 3];
  double xyz64[N * 3];
  int f, j, i, k, n_dev = 0;
  for (f = 0; f < F; ++f)
    for (j = 0; j < H; ++j)
      for (i = 0; i < W; ++i) depth[(f * H + j) * W + i] = (unsigned char)((7 * j + 3 * i + 1 + 11 * f) % 256);
  /* frame 0: identity pose; frame 1: Rinv = rotation by 90 deg about z, t = (1,2,3) */
  memset(pose, 0, sizeof(pose));
  pose[0] = pose[4] = pose[8] = 1.0;
  pose[12 + 1] = -1.0; pose[12 + 3] = 1.0; pose[12 + 8] = 1.0;
  pose[12 + 9] = 1.0; pose[12 + 10] = 2.0; pose[12 + 11] = 3.0;

  printf("r3d version %d\n", r3d_version());
  if (r3d_device_count(&n_dev) != R3D_OK || n_dev < 1) {
    printf("no GPU visible: %s\n", r3d_last_error());
    return 77; /* skipped */
  }
  r3d_ctx* ctx = NULL;
  r3d_camera* cam = NULL;
  CK(r3d_ctx_create(0, NULL, 0, &ctx));
  CK(r3d_camera_create(ctx, H, W, 600.391, 600.079, 320.0, 240.0, &cam));
  CK(r3d_fuse_frames_host(ctx, cam, depth, R3D_DEPTH_U8, F, 1.0, pose, xyz, R3D_F32));
  CK(r3d_fuse_frames_host(ctx, cam, depth, R3D_DEPTH_U8, F, 1.0, pose, xyz64, R3D_F64));
  /* frame 0, pixel (0,0): Z=1 -> (-320/600.391, -240/600.079, 1) */
  {
    const double ex = (0 - 320.0) / 600.391 * 1.0, ey = (0 - 240.0) / 600.079 * 1.0;
    if (xyz64[0] != ex || xyz64[1] != ey || xyz64[2] != 1.0) { fprintf(stderr, "frame 0 pixel 0 mismatch\n"); return 1; }
    if (xyz[0] != (float)ex || xyz[1] != (float)ey) { fprintf(stderr, "f32 rounding mismatch\n"); return 1; }
  }
  /* frame 1, last pixel (j=3,i=5): Z=(21+15+1+11)=48; p-t then rotate: (x,y,z)->(-y', x', z') */
  {
    const double Z = 48.0, X = (5 - 320.0) / 600.391 * Z, Y = (3 - 240.0) / 600.079 * Z;
    const double dx = X - 1.0, dy = Y - 2.0, dz = Z - 3.0;
    const double* p = xyz64 + (size_t)(N - 1) * 3;
    if (fabs(p[0] - (-dy)) > 1e-12 || fabs(p[1] - dx) > 1e-12 || fabs(p[2] - dz) > 1e-12) {
      fprintf(stderr, "frame 1 last pixel mismatch: %.17g %.17g %.17g\n", p[0], p[1], p[2]);
      return 1;
    }
  }
  /* apply-T in place on the host copy, then the reference PLY bytes */
  {
    double T[16] = {2, 0, 0, 1, 0, 2, 0, 2, 0, 0, 2, 3, 0, 0, 0, 1};
    size_t n_bytes = 0;
    char* buf;
    CK(r3d_apply_T_host(ctx, xyz, R3D_F32, N, T, xyz, R3D_F32));
    CK(r3d_format_ply(xyz, R3D_F32, N, NULL, 0, &n_bytes));
    buf = (char*)malloc(n_bytes + 1);
    CK(r3d_format_ply(xyz, R3D_F32, N, buf, n_bytes, &n_bytes));
    buf[n_bytes] = 0;
    if (strncmp(buf, "ply\n    format ascii 1.0\n    element vertex 48\n", 47) != 0) { fprintf(stderr, "PLY header mismatch\n"); return 1; }
    free(buf);
  }
  /* the text round trip a maintainer of camera_to_world.py / transfer_T_icp.py cares about: write `X,Y,Z` lines with
   * repr() floats, read them back (first three fields of every line), same doubles */
  {
    size_t n_bytes = 0;
    char* txt;
    double back[N * 3];
    long long n_pts = 0, bad_line = 0;
    CK(r3d_format_xyz_txt(xyz64, R3D_F64, N, NULL, 0, NULL, 0, &n_bytes));
    txt = (char*)malloc(n_bytes + 1);
    CK(r3d_format_xyz_txt(xyz64, R3D_F64, N, NULL, 0, txt, n_bytes, &n_bytes));
    CK(r3d_parse_xyz_text(txt, n_bytes, ',', NULL, 0, (int64_t*)&n_pts, NULL));
    if (n_pts != N) { fprintf(stderr, "parser counted %lld lines, expected %d\n", n_pts, (int)N); return 1; }
    CK(r3d_parse_xyz_text(txt, n_bytes, ',', back, N, (int64_t*)&n_pts, (int64_t*)&bad_line));
    if (memcmp(back, xyz64, sizeof(back)) != 0) { fprintf(stderr, "txt round trip changed a double\n"); return 1; }
    txt[5] = 'x'; /* damage the first line */
    if (r3d_parse_xyz_text(txt, n_bytes, ',', back, N, (int64_t*)&n_pts, (int64_t*)&bad_line) != R3D_ERR_INVALID || bad_line != 1) {
      fprintf(stderr, "damaged line not reported (line %lld)\n", bad_line);
      return 1;
    }
    free(txt);
  }
  /* the colour -> grey rules of the depth-file readers, straight on pixels: known answers (include/r3d.h, R3D_GRAY_*) */
  {
    const unsigned char px[4 * 3] = {255, 0, 0, 0, 255, 0, 12, 200, 77, 9, 9, 9};
    unsigned char g[4];
    CK(r3d_rgb_to_gray_u8(px, 4, 3, R3D_GRAY_OPENCV_PNG, g));
    if (g[0] != 76 || g[1] != 149 || g[2] != 129 || g[3] != 9) { fprintf(stderr, "libpng grey rule: %d %d %d %d\n", g[0], g[1], g[2], g[3]); return 1; }
    CK(r3d_rgb_to_gray_u8(px, 4, 3, R3D_GRAY_CVTCOLOR, g));
    if (g[0] != 76 || g[1] != 150 || g[2] != 130 || g[3] != 9) { fprintf(stderr, "cvtColor grey rule: %d %d %d %d\n", g[0], g[1], g[2], g[3]); return 1; }
    if (r3d_rgb_to_gray_u8(px, 4, 3, 5, g) != R3D_ERR_INVALID) { fprintf(stderr, "unknown grey rule accepted\n"); return 1; }
    {
      int h = 0, w = 0;
      if (r3d_jpeg_gray_info("/nonexistent/depth.jpg", &h, &w) != R3D_ERR_INVALID || r3d_png_gray8_info("/nonexistent/depth.png", &h, &w) == R3D_OK) {
        fprintf(stderr, "missing depth file not reported\n");
        return 1;
      }
    }
  }
  /* staging sweep + colour-carrying launch on device memory */
  {
    void *d_depth = NULL, *d_pose = NULL, *d_rgb = NULL, *d_xyz = NULL, *d_rgba = NULL;
    unsigned char rgb[N * 3];
    unsigned int rgba[N];
    for (k = 0; k < N * 3; ++k) rgb[k] = (unsigned char)(k * 7);
    CK(r3d_dev_alloc(ctx, N, &d_depth));
    CK(r3d_dev_alloc(ctx, sizeof(pose), &d_pose));
    CK(r3d_dev_alloc(ctx, N * 3, &d_rgb));
    CK(r3d_dev_alloc(ctx, N * 12, &d_xyz));
    CK(r3d_dev_alloc(ctx, N * 4, &d_rgba));
    CK(r3d_memcpy_h2d(ctx, d_depth, depth, N));
    CK(r3d_memcpy_h2d(ctx, d_pose, pose, sizeof(pose)));
    CK(r3d_memcpy_h2d(ctx, d_rgb, rgb, N * 3));
    CK(r3d_cache_prefetch(ctx, d_rgb, N * 3));
    CK(r3d_fuse_frames_rgb(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, (const double*)d_pose, (const unsigned char*)d_rgb, d_xyz, R3D_F32,
                           (uint32_t*)d_rgba));
    CK(r3d_memcpy_d2h(ctx, rgba, d_rgba, N * 4));
    CK(r3d_ctx_sync(ctx));
    for (k = 0; k < N; ++k)
      if (rgba[k] != ((unsigned)rgb[3 * k] | ((unsigned)rgb[3 * k + 1] << 8) | ((unsigned)rgb[3 * k + 2] << 16))) {
        fprintf(stderr, "colour of point %d wrong\n", k);
        return 1;
      }
    /* config 5's pair in one launch: the same cloud and colours, and the occupied voxels of exactly that cloud */
    {
      r3d_voxelset *two = NULL, *one = NULL;
      int64_t v2 = 0, i2 = 0, o2 = 0, v1 = 0, i1 = 0, o1 = 0;
      float first[N * 3], again[N * 3];
      CK(r3d_voxelset_create(ctx, 0.1, 1 << 12, &two));
      CK(r3d_voxelset_create(ctx, 0.1, 1 << 12, &one));
      CK(r3d_memcpy_d2h(ctx, first, d_xyz, N * 12));
      CK(r3d_voxelset_insert(two, (const float*)d_xyz, N));
      CK(r3d_memset(ctx, d_xyz, 0, N * 12));
      CK(r3d_memset(ctx, d_rgba, 0, N * 4));
      CK(r3d_fuse_frames_voxel(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, (const double*)d_pose, (const unsigned char*)d_rgb, (float*)d_xyz,
                               (uint32_t*)d_rgba, one));
      CK(r3d_memcpy_d2h(ctx, again, d_xyz, N * 12));
      CK(r3d_memcpy_d2h(ctx, rgba, d_rgba, N * 4));
      CK(r3d_voxelset_stats(two, &v2, &i2, &o2));
      CK(r3d_voxelset_stats(one, &v1, &i1, &o1));
      if (memcmp(first, again, sizeof(first)) != 0 || v1 != v2 || i1 != i2 || o1 != o2 || v1 < 1 ||
          rgba[N - 1] != ((unsigned)rgb[3 * N - 3] | ((unsigned)rgb[3 * N - 2] << 8) | ((unsigned)rgb[3 * N - 1] << 16))) {
        fprintf(stderr, "one-launch cloud + voxels differs from the two calls (%lld vs %lld voxels)\n", (long long)v1, (long long)v2);
        return 1;
      }
      printf("cloud + map in one launch: %lld voxels, same as the two calls\n", (long long)v1);
      CK(r3d_voxelset_destroy(two));
      CK(r3d_voxelset_destroy(one));
    }
    CK(r3d_dev_free(ctx, d_depth));
    CK(r3d_dev_free(ctx, d_pose));
    CK(r3d_dev_free(ctx, d_rgb));
    CK(r3d_dev_free(ctx, d_xyz));
    CK(r3d_dev_free(ctx, d_rgba));
  }
  /* the registration the reference left to CloudCompare (readme.md:25): a 48x64 single view of a room corner (two walls and
   * the floor) against a copy of itself moved by a known small rigid motion -- normals from the raster, index, ten
   * point-to-plane iterations enqueued without a host round trip, then the pose is read back from the ICP state */
  {
    enum { RH = 48, RW = 64, RN = RH * RW };
    static float tgt[RN * 3], src[RN * 3];
    double st[R3D_ICP_STATE_DOUBLES];
    const double a = 2.0 * 3.14159265358979323846 / 180.0, ca = cos(a), sa = sin(a);
    const double Rm[9] = {ca, 0, sa, 0, 1, 0, -sa, 0, ca}, tv[3] = {0.03, -0.02, 0.04};
    void *d_tgt = NULL, *d_src = NULL, *d_src0 = NULL, *d_nrm = NULL, *d_idx = NULL, *d_d2 = NULL, *d_state = NULL;
    r3d_nn_index* ix = NULL;
    double worst = 0.0;
    for (j = 0; j < RH; ++j)
      for (i = 0; i < RW; ++i) {
        const double u = (i - 20.0) / 50.0, v = (j - 14.0) / 50.0;
        double z = 3.0; /* wall z = 3; wall x = 2 where u > 0; floor y = 1.2 where v > 0 */
        float* q = tgt + (size_t)(j * RW + i) * 3;
        float* p = src + (size_t)(j * RW + i) * 3;
        double d[3];
        if (u > 0 && 2.0 / u < z) z = 2.0 / u;
        if (v > 0 && 1.2 / v < z) z = 1.2 / v;
        q[0] = (float)(u * z); q[1] = (float)(v * z); q[2] = (float)z;
        d[0] = q[0] - tv[0]; d[1] = q[1] - tv[1]; d[2] = q[2] - tv[2];          /* src = R^T (tgt - t): T maps src onto tgt */
        for (k = 0; k < 3; ++k) p[k] = (float)(Rm[0 + k] * d[0] + Rm[3 + k] * d[1] + Rm[6 + k] * d[2]);
      }
    CK(r3d_dev_alloc(ctx, sizeof(tgt), &d_tgt));
    CK(r3d_dev_alloc(ctx, sizeof(src), &d_src));
    CK(r3d_dev_alloc(ctx, sizeof(src), &d_src0));
    CK(r3d_dev_alloc(ctx, sizeof(tgt), &d_nrm));
    CK(r3d_dev_alloc(ctx, RN * 4, &d_idx));
    CK(r3d_dev_alloc(ctx, RN * 4, &d_d2));
    CK(r3d_dev_alloc(ctx, sizeof(st), &d_state));
    CK(r3d_memcpy_h2d(ctx, d_tgt, tgt, sizeof(tgt)));
    CK(r3d_memcpy_h2d(ctx, d_src, src, sizeof(src)));
    CK(r3d_normals_organized(ctx, (const float*)d_tgt, 1, RH, RW, 0.05f, NULL, (float*)d_nrm));
    CK(r3d_nn_index_create(ctx, (const float*)d_tgt, RN, &ix));
    CK(r3d_nn_index_sort_cloud(ix, (float*)d_src, RN, NULL));
    CK(r3d_memcpy_d2d(ctx, d_src0, d_src, sizeof(src)));
    CK(r3d_icp_state_reset(ctx, (double*)d_state));
    CK(r3d_icp_iterate_plane(ctx, ix, (const float*)d_src0, (float*)d_src, RN, (const float*)d_nrm, (uint32_t*)d_idx, (float*)d_d2, 10,
                             0.5f, 20.0f, -1.0f, (double*)d_state));
    CK(r3d_memcpy_d2h(ctx, st, d_state, sizeof(st)));
    CK(r3d_ctx_sync(ctx));
    for (j = 0; j < 3; ++j) {
      for (i = 0; i < 3; ++i) if (fabs(st[4 * j + i] - Rm[3 * j + i]) > worst) worst = fabs(st[4 * j + i] - Rm[3 * j + i]);
      if (fabs(st[4 * j + 3] - tv[j]) > worst) worst = fabs(st[4 * j + 3] - tv[j]);
    }
    printf("point-to-plane: %d iterations, status %g, |T - T_true| max %.2e\n", (int)st[32], st[33], worst);
    if (st[32] != 10.0 || st[33] != 0.0 || worst > 1e-3) { fprintf(stderr, "two-view registration did not come back\n"); return 1; }
    CK(r3d_nn_index_destroy(ix));
    CK(r3d_dev_free(ctx, d_tgt)); CK(r3d_dev_free(ctx, d_src)); CK(r3d_dev_free(ctx, d_src0)); CK(r3d_dev_free(ctx, d_nrm));
    CK(r3d_dev_free(ctx, d_idx)); CK(r3d_dev_free(ctx, d_d2)); CK(r3d_dev_free(ctx, d_state));
  }
  /* error convention: bad arguments come back as codes with a message, nothing aborts */
  if (r3d_fuse_frames_host(ctx, cam, NULL, R3D_DEPTH_U8, F, 1.0, pose, xyz, R3D_F32) != R3D_ERR_INVALID) return 1;
  if (r3d_ctx_set_tuning(ctx, "no_such_knob", 1) != R3D_ERR_INVALID) return 1;
  for (k = 0; k < 3; ++k) printf("xyz[%d] = %.6f\n", k, xyz[k]);
  CK(r3d_camera_destroy(cam));
  CK(r3d_ctx_destroy(ctx));
  printf("C ABI smoke OK\n");
  return 0;
}
