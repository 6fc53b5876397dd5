// Point-to-plane rigid ICP step: the 29 fp64 sums of the linearised 6x6 normal equations, their deterministic
// workgroup reduction, and the solve (Cholesky + exponential map) as host+device code -- the ICP loop runs without a
// host round trip per iteration and the very same arithmetic is callable on a CPU-only box (r3d_plane_step_from_sums).
//
// NOT IN THE REFERENCE (ICP estimation is build-defined).  What it serves IS the reference's use of
// ICP: "match the point clouds corresponding to two images" (readme.md:25) -- the relative pose of two single-view
// camera clouds, ./point/0.txt and ./point/24.txt (other_tools/transfer_T_icp.py:107-108), which the reference
// obtained by hand in CloudCompare and stored as T_data.txt (icp:99).
//
// One matched pair (p = moved source point, q = its nearest target point, n = unit normal of the target surface at q):
//   residual r = n . (p - q);  a small rigid motion (omega, v) changes it by  (p x n) . omega + n . v,  so with
//   J = [p x n ; n] (6 numbers) the step minimises sum w (r + J . x)^2:   (sum w J J^T) x = - sum w J r.
// Sums: [0] sum w, [1] sum w r^2, [2..7] sum w J r, [8..28] upper triangle of sum w J J^T, row-major.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace r3d_plane {

constexpr int kSums = 29;
constexpr int kThreads = 256;

#if defined(__HIPCC__)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// r = n . (p - q), evaluated left to right in fp64 (the differences of fp32 inputs are exact there)
__device__ __forceinline__ double plane_residual(const double p[3], const double q[3], const double n[3]) {
  return n[0] * (p[0] - q[0]) + n[1] * (p[1] - q[1]) + n[2] * (p[2] - q[2]);
}

__device__ __forceinline__ void pair_accumulate(double acc[kSums], double w, const double p[3], const double n[3], double r) {
  const double J[6] = {p[1] * n[2] - p[2] * n[1], p[2] * n[0] - p[0] * n[2], p[0] * n[1] - p[1] * n[0], n[0], n[1], n[2]};
  acc[0] += w;
  acc[1] += w * r * r;
  int k = 8;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double wj = w * J[a];
    acc[2 + a] += wj * r;
#pragma unroll
    for (int b = a; b < 6; ++b) acc[k++] += wj * J[b];
  }
}

// Workgroup (256 threads) reduction of per-lane accumulators into one row of partials: shuffle tree over the 64 lanes,
// LDS across the 4 waves, fixed order -> bitwise repeatable.  `red` is __shared__ [4][kSums].
__device__ __forceinline__ void block_reduce_store(double acc[kSums], double (*red)[kSums], double* __restrict__ row) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kSums; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kSums) {
    double v = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) v += red[w][threadIdx.x];
    row[threadIdx.x] = v;
  }
}
#define R3D_PLANE_HD __host__ __device__
#else
#define R3D_PLANE_HD
#endif

// T (row-major 4x4, rigid) from the 29 sums.  Returns 0, or 1 when the step is undefined (fewer than 6 pairs, or the
// normal equations are singular to working precision: the matched normals do not constrain all six freedoms -- one
// plane, two parallel walls, ...): T is then the identity.  *rms_out (optional) = sqrt(sum w r^2 / sum w).
R3D_PLANE_HD inline int step_from_sums(const double s[kSums], double T[16], double* rms_out) {
  for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
  const double n = s[0];
  if (rms_out) *rms_out = n > 0.0 ? sqrt((s[1] > 0.0 ? s[1] : 0.0) / n) : 0.0;
  if (!(n >= 6.0)) return 1;
  double A[6][6], x[6];
  int k = 8;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) {
      A[a][b] = s[k];
      A[b][a] = s[k];
      ++k;
    }
  // the rotation block is in (length)^2, the translation block dimensionless: scale the unknowns so that the pivot
  // test means the same for a cloud in millimetres and one in kilometres
  double tr_rot = A[0][0] + A[1][1] + A[2][2], tr_tra = A[3][3] + A[4][4] + A[5][5];
  if (!(tr_rot > 0.0) || !(tr_tra > 0.0)) return 1;
  const double len = sqrt(tr_rot / tr_tra);   // a typical lever arm
  double sc[6] = {1.0 / len, 1.0 / len, 1.0 / len, 1.0, 1.0, 1.0};
  double g[6];
  for (int a = 0; a < 6; ++a) {
    g[a] = -s[2 + a] * sc[a];
    for (int b = 0; b < 6; ++b) A[a][b] *= sc[a] * sc[b];
  }
  double trace = 0.0;
  for (int a = 0; a < 6; ++a) trace += A[a][a];
  // Cholesky A = L L^T in place (lower triangle)
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
    for (int m = 0; m < j; ++m) d -= A[j][m] * A[j][m];
    if (!(d > 1e-10 * trace)) return 1;
    d = sqrt(d);
    A[j][j] = d;
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i][j];
      for (int m = 0; m < j; ++m) v -= A[i][m] * A[j][m];
      A[i][j] = v / d;
    }
  }
  for (int i = 0; i < 6; ++i) {   // L y = g
    double v = g[i];
    for (int m = 0; m < i; ++m) v -= A[i][m] * x[m];
    x[i] = v / A[i][i];
  }
  for (int i = 5; i >= 0; --i) {  // L^T x = y
    double v = x[i];
    for (int m = i + 1; m < 6; ++m) v -= A[m][i] * x[m];
    x[i] = v / A[i][i];
  }
  for (int a = 0; a < 6; ++a) {
    x[a] *= sc[a];
    if (!(fabs(x[a]) < 1e300)) return 1;   // also catches NaN
  }
  // rotation = exp([omega]x) (Rodrigues): exactly orthogonal whatever the step size
  const double wx = x[0], wy = x[1], wz = x[2];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  double a_, b_;                       // R = I + a [w]x + b [w]x^2
  if (th < 1e-6) {
    a_ = 1.0 - th2 / 6.0;
    b_ = 0.5 - th2 / 24.0;
  } else {
    a_ = sin(th) / th;
    b_ = (1.0 - cos(th)) / th2;
  }
  const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double K2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) K2[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[4 * r + c] = (r == c ? 1.0 : 0.0) + a_ * K[3 * r + c] + b_ * K2[3 * r + c];
    T[4 * r + 3] = x[3 + r];
  }
  return 0;
}

}  // namespace r3d_plane
