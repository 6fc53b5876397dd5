// BackprojectDepth for the depth-estimation trainer (f4): the torch cousin of the unprojection.
//
// The reference's trainer (monodepth2/trainer.py:150-160, 387-390) calls upstream monodepth2's layer
// `BackprojectDepth(batch, h, w)(depth, inv_K)` (upstream layers.py; that file is NOT in the reference repo):
//     cam_points = cat([depth.view(B,1,-1) * (inv_K[:, :3, :3] @ [x; y; 1]), ones], 1)          -> [B, 4, H*W], fp32
// with pixel p = y*W + x.  Here as two HIP kernels behind the C ABI (forward and the depth gradient), fp32 like the
// layer.  HBM-bound, 20 B/pixel forward (4 read + 16 written as four coalesced planes), 20 B/pixel backward.
//
// Its partner in the same trainer lines, `Project3D(batch, h, w)(cam_points, K, T)` (upstream layers.py as well):
//     P = (K @ T)[:, :3, :];  c = P @ points;  pix = c[:, :2] / (c[:, 2:3] + eps)          (eps = 1e-7)
//     pix -> [B, H, W, 2];  pix[..., 0] /= W - 1;  pix[..., 1] /= H - 1;  pix = (pix - 0.5) * 2    (grid_sample coords)
// Forward 24 B/pixel (four planes read, one float2 written).  Backward: d/d(points) per pixel and d/d(P) -- twelve sums
// per image, wave-shuffle + LDS tree per workgroup, then a fixed-order second stage (bitwise repeatable, no atomics).
// The 4x4 product K @ T and its gradient stay in torch (sixteen numbers per image).
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

struct Proj {
  float p[12];
};

__device__ __forceinline__ Proj load_P(const float* __restrict__ P, int b) {
  Proj r;
  const float* q = P + (size_t)b * 12;  // wave-uniform: scalar loads
#pragma unroll
  for (int i = 0; i < 12; ++i) r.p[i] = q[i];
  return r;
}

__global__ __launch_bounds__(kThreads) void project3d_kernel(const float* __restrict__ points, const float* __restrict__ P,
                                                           int hw, float eps, float w1, float h1,
                                                           float2* __restrict__ pix) {
  const int b = blockIdx.y;
  const Proj m = load_P(P, b);
  const float* x = points + (size_t)b * 4 * hw;
  float2* o = pix + (size_t)b * hw;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < hw; p += gridDim.x * kThreads) {
    const float px = x[p], py = x[hw + p], pz = x[2 * (size_t)hw + p], pw = x[3 * (size_t)hw + p];
    const float c0 = m.p[0] * px + m.p[1] * py + m.p[2] * pz + m.p[3] * pw;
    const float c1 = m.p[4] * px + m.p[5] * py + m.p[6] * pz + m.p[7] * pw;
    const float c2 = m.p[8] * px + m.p[9] * py + m.p[10] * pz + m.p[11] * pw;
    const float den = c2 + eps;
    float2 r;
    r.x = ((c0 / den) / w1 - 0.5f) * 2.0f;   // the layer's own order: divide, divide by (W-1), shift, scale
    r.y = ((c1 / den) / h1 - 0.5f) * 2.0f;
    __builtin_nontemporal_store(r.x, &o[p].x);
    __builtin_nontemporal_store(r.y, &o[p].y);
  }
}

constexpr int kProjSums = 12;

// grad wrt points (per pixel, four planes; optional) and per-workgroup partial sums of grad wrt P (optional)
__global__ __launch_bounds__(kThreads) void project3d_grad_kernel(const float2* __restrict__ grad_pix,
                                                                const float* __restrict__ points,
                                                                const float* __restrict__ P, int hw, float eps, float gw,
                                                                float gh, float* __restrict__ grad_points,
                                                                float* __restrict__ partials) {
  const int b = blockIdx.y;
  const Proj m = load_P(P, b);
  const float* x = points + (size_t)b * 4 * hw;
  const float2* g = grad_pix + (size_t)b * hw;
  float* gp = grad_points ? grad_points + (size_t)b * 4 * hw : nullptr;
  float acc[kProjSums];
#pragma unroll
  for (int i = 0; i < kProjSums; ++i) acc[i] = 0.0f;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < hw; p += gridDim.x * kThreads) {
    const float px = x[p], py = x[hw + p], pz = x[2 * (size_t)hw + p], pw = x[3 * (size_t)hw + p];
    const float c0 = m.p[0] * px + m.p[1] * py + m.p[2] * pz + m.p[3] * pw;
    const float c1 = m.p[4] * px + m.p[5] * py + m.p[6] * pz + m.p[7] * pw;
    const float c2 = m.p[8] * px + m.p[9] * py + m.p[10] * pz + m.p[11] * pw;
    const float inv = 1.0f / (c2 + eps);
    const float2 gg = g[p];
    const float d0 = gg.x * gw * inv;              // d loss / d c0;  gw = 2/(W-1)
    const float d1 = gg.y * gh * inv;
    const float d2 = -(d0 * c0 + d1 * c1) * inv;
    if (gp) {
      gp[p] = m.p[0] * d0 + m.p[4] * d1 + m.p[8] * d2;
      gp[hw + p] = m.p[1] * d0 + m.p[5] * d1 + m.p[9] * d2;
      gp[2 * (size_t)hw + p] = m.p[2] * d0 + m.p[6] * d1 + m.p[10] * d2;
      gp[3 * (size_t)hw + p] = m.p[3] * d0 + m.p[7] * d1 + m.p[11] * d2;
    }
    acc[0] += d0 * px, acc[1] += d0 * py, acc[2] += d0 * pz, acc[3] += d0 * pw;
    acc[4] += d1 * px, acc[5] += d1 * py, acc[6] += d1 * pz, acc[7] += d1 * pw;
    acc[8] += d2 * px, acc[9] += d2 * py, acc[10] += d2 * pz, acc[11] += d2 * pw;
  }
  if (!partials) return;
  __shared__ float red[kThreads / 64][kProjSums];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kProjSums; ++i) {
    float v = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kProjSums) {
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) v += red[w][threadIdx.x];
    partials[((size_t)b * gridDim.x + blockIdx.x) * kProjSums + threadIdx.x] = v;
  }
}

// second stage: one wave per image adds the workgroup partials in a fixed order (fp64 accumulator)
__global__ __launch_bounds__(64) void project3d_grad_finish_kernel(const float* __restrict__ partials, int blocks,
                                                                 float* __restrict__ grad_P) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (lane >= kProjSums) return;
  const float* q = partials + (size_t)b * blocks * kProjSums + lane;
  double s = 0.0;
  for (int k = 0; k < blocks; ++k) s += (double)q[(size_t)k * kProjSums];
  grad_P[(size_t)b * kProjSums + lane] = (float)s;
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

int r3d_project3d_f32(r3d_ctx* ctx, const float* d_points, const float* d_P, int batch, int height, int width, float eps,
                      float* d_pix) {
  int rc = check(ctx, d_points, d_P, d_pix, batch, height, width);
  if (rc || batch == 0) return rc;
  R3D_REQUIRE(height > 1 && width > 1, "Project3D normalises by (W-1) and (H-1): need at least 2 x 2");
  const int hw = height * width;
  int bx = (hw + kThreads - 1) / kThreads;
  const int cap = ctx->num_cus * 16;
  if (bx > cap) bx = cap;
  hipLaunchKernelGGL(project3d_kernel, dim3(bx, batch), dim3(kThreads), 0, ctx->stream, d_points, d_P, hw, eps,
                     (float)(width - 1), (float)(height - 1), reinterpret_cast<float2*>(d_pix));
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_project3d_grad_f32(r3d_ctx* ctx, const float* d_grad_pix, const float* d_points, const float* d_P, int batch,
                           int height, int width, float eps, float* d_grad_points, float* d_grad_P) {
  int rc = check(ctx, d_grad_pix, d_points, d_P, batch, height, width);
  if (rc || batch == 0) return rc;
  R3D_REQUIRE(height > 1 && width > 1, "Project3D normalises by (W-1) and (H-1): need at least 2 x 2");
  if (!d_grad_points && !d_grad_P) return R3D_OK;
  const int hw = height * width;
  int bx = (hw + kThreads - 1) / kThreads;
  const int cap = ctx->num_cus * 4;   // several pixels per thread: fewer partials to add up
  if (bx > cap) bx = cap;
  void* part = nullptr;
  if (d_grad_P && (rc = r3d_scratch(ctx, 4, (size_t)batch * bx * kProjSums * sizeof(float), &part))) return rc;
  hipLaunchKernelGGL(project3d_grad_kernel, dim3(bx, batch), dim3(kThreads), 0, ctx->stream,
                     reinterpret_cast<const float2*>(d_grad_pix), d_points, d_P, hw, eps, 2.0f / (float)(width - 1),
                     2.0f / (float)(height - 1), d_grad_points, static_cast<float*>(part));
  R3D_HIP(hipGetLastError());
  if (d_grad_P) {
    hipLaunchKernelGGL(project3d_grad_finish_kernel, dim3(batch), dim3(64), 0, ctx->stream, static_cast<const float*>(part),
                       bx, d_grad_P);
    R3D_HIP(hipGetLastError());
  }
  return R3D_OK;
}

}  // extern "C"
