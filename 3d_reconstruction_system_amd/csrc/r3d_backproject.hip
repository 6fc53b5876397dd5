// BackprojectDepth for the depth-estimation trainer (f4): the torch cousin of the unprojection.
//
// The reference's trainer (monodepth2/trainer.py:150-160, 387-390) calls upstream monodepth2's layer
// `BackprojectDepth(batch, h, w)(depth, inv_K)` (upstream layers.py; that file is NOT in the reference repo):
//     cam_points = cat([depth.view(B,1,-1) * (inv_K[:, :3, :3] @ [x; y; 1]), ones], 1)          -> [B, 4, H*W], fp32
// with pixel p = y*W + x.  Here as two HIP kernels behind the C ABI (forward and the depth gradient), fp32 like the
// layer.  HBM-bound, 20 B/pixel forward (4 read + 16 written as four coalesced planes), 20 B/pixel backward.
#include "r3d_internal.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void backproject_kernel(const float* __restrict__ depth, const float* __restrict__ inv_K,
                                                             int hw, int width, float* __restrict__ cam) {
  const int b = blockIdx.y;
  const float* K = inv_K + (size_t)b * 16;  // wave-uniform: scalar loads
  const float k00 = K[0], k01 = K[1], k02 = K[2], k10 = K[4], k11 = K[5], k12 = K[6], k20 = K[8], k21 = K[9], k22 = K[10];
  const float* d = depth + (size_t)b * hw;
  float* o = cam + (size_t)b * 4 * hw;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < hw; p += gridDim.x * kThreads) {
    const int y = p / width, x = p - y * width;
    const float fx = (float)x, fy = (float)y, z = d[p];
    // row . [x y 1], left to right like a plain matmul
    const float r0 = k00 * fx + k01 * fy + k02;
    const float r1 = k10 * fx + k11 * fy + k12;
    const float r2 = k20 * fx + k21 * fy + k22;
    __builtin_nontemporal_store(z * r0, o + p);
    __builtin_nontemporal_store(z * r1, o + hw + p);
    __builtin_nontemporal_store(z * r2, o + 2 * (size_t)hw + p);
    __builtin_nontemporal_store(1.0f, o + 3 * (size_t)hw + p);
  }
}

// d(loss)/d(depth[p]) = sum_c grad_cam[c][p] * ray_c(p), c = 0..2 (the ones row carries no gradient)
__global__ __launch_bounds__(kThreads) void backproject_grad_kernel(const float* __restrict__ grad_cam,
                                                                  const float* __restrict__ inv_K, int hw, int width,
                                                                  float* __restrict__ grad_depth) {
  const int b = blockIdx.y;
  const float* K = inv_K + (size_t)b * 16;
  const float k00 = K[0], k01 = K[1], k02 = K[2], k10 = K[4], k11 = K[5], k12 = K[6], k20 = K[8], k21 = K[9], k22 = K[10];
  const float* g = grad_cam + (size_t)b * 4 * hw;
  float* o = grad_depth + (size_t)b * hw;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < hw; p += gridDim.x * kThreads) {
    const int y = p / width, x = p - y * width;
    const float fx = (float)x, fy = (float)y;
    const float r0 = k00 * fx + k01 * fy + k02;
    const float r1 = k10 * fx + k11 * fy + k12;
    const float r2 = k20 * fx + k21 * fy + k22;
    o[p] = g[p] * r0 + g[hw + p] * r1 + g[2 * (size_t)hw + p] * r2;
  }
}

int check(r3d_ctx* ctx, const void* a, const void* b, const void* c, int batch, int height, int width) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(batch >= 0 && height > 0 && width > 0, "bad shape %d x %d x %d", batch, height, width);
  R3D_REQUIRE((int64_t)height * width < ((int64_t)1 << 30), "raster too large");
  R3D_REQUIRE(batch <= 65535, "batch too large for one launch");
  R3D_REQUIRE(batch == 0 || (a && b && c), "NULL device pointer");
  return R3D_OK;
}

}  // namespace

extern "C" {

int r3d_backproject_depth_f32(r3d_ctx* ctx, const float* d_depth, const float* d_inv_K, int batch, int height, int width,
                              float* d_cam_points) {
  int rc = check(ctx, d_depth, d_inv_K, d_cam_points, batch, height, width);
  if (rc || batch == 0) return rc;
  const int hw = height * width;
  int bx = (hw + kThreads - 1) / kThreads;
  const int cap = ctx->num_cus * 16;
  if (bx > cap) bx = cap;
  hipLaunchKernelGGL(backproject_kernel, dim3(bx, batch), dim3(kThreads), 0, ctx->stream, d_depth, d_inv_K, hw, width,
                     d_cam_points);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_backproject_depth_grad_f32(r3d_ctx* ctx, const float* d_grad_cam_points, const float* d_inv_K, int batch, int height,
                                   int width, float* d_grad_depth) {
  int rc = check(ctx, d_grad_cam_points, d_inv_K, d_grad_depth, batch, height, width);
  if (rc || batch == 0) return rc;
  const int hw = height * width;
  int bx = (hw + kThreads - 1) / kThreads;
  const int cap = ctx->num_cus * 16;
  if (bx > cap) bx = cap;
  hipLaunchKernelGGL(backproject_grad_kernel, dim3(bx, batch), dim3(kThreads), 0, ctx->stream, d_grad_cam_points, d_inv_K, hw,
                     width, d_grad_depth);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

}  // extern "C"
