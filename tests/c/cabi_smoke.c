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
  /* error convention: bad arguments come back as codes with a message, nothing aborts */
  if (r3d_fuse_frames_host(ctx, cam, NULL, R3D_DEPTH_U8, F, 1.0, pose, xyz, R3D_F32) != R3D_ERR_INVALID) return 1;
  if (r3d_ctx_set_tuning(ctx, "no_such_knob", 1) != R3D_ERR_INVALID) return 1;
  for (k = 0; k < 3; ++k) printf("xyz[%d] = %.6f\n", k, xyz[k]);
  CK(r3d_camera_destroy(cam));
  CK(r3d_ctx_destroy(ctx));
  printf("C ABI smoke OK\n");
  return 0;
}
