// Shared pieces of the ICP estimation path (r3d_icp.hip, r3d_nnindex.hip): the 18 fp64 pair sums, their
// deterministic workgroup reduction, and the closed-form similarity (Umeyama 1991) solved from them -- the
// latter as host+device code so that the ICP loop can run without a host round trip per iteration and the
// very same arithmetic is testable on a CPU-only box (r3d_umeyama_from_sums).
//
// NOT IN THE REFERENCE: other_tools/transfer_T_icp.py:33-43,99-108 only consumes the T such a fit produces.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace r3d_icp {

constexpr int kSums = 18;     // n, sum p (3), sum q (3), sum p_a q_b (9, a major), sum |p|^2, sum |q|^2
constexpr int kThreads = 256;

// ---- device-resident ICP state (R3D_ICP_STATE_DOUBLES doubles, see include/r3d.h) ----
constexpr int kStateTTotal = 0;    // [16] row-major 4x4: product of all steps so far (maps the ORIGINAL source)
constexpr int kStateTStep = 16;    // [16] the last step
constexpr int kStateIters = 32;    // iterations solved so far
constexpr int kStateStatus = 33;   // 0 ok; 1 = a step was degenerate (fewer than 3 pairs / zero variance) and was skipped
constexpr int kStateRms = 34;      // weighted RMS match distance seen by the last step (before it was applied)
constexpr int kStatePairs = 35;    // weight sum (= pair count when unweighted) of the last step
constexpr int kStateHistory = 48;  // rms of step k at [48 + k] while it fits
constexpr int kStateDoubles = 512;

#if defined(__HIPCC__)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// One matched pair's weight.  dead_zone <= 0: w = 1.  Otherwise the IRLS weight of the cost
// max(0, d - dead_zone)^2, d = sqrt(d2): w = max(0, 1 - dead_zone/d).  It treats a densely sampled cloud as the
// solid it samples: a distance below the sampling resolution says nothing about the transform.
__device__ __forceinline__ double pair_weight(float d2, float dead_zone) {
  if (!(dead_zone > 0.f)) return 1.0;
  const double d = sqrt((double)d2);
  return d > (double)dead_zone ? 1.0 - (double)dead_zone / d : 0.0;
}

// A pair in which p or q carries a NaN / inf coordinate is no pair: it is left out (one such row would turn all 18 sums
// into NaN).  The six values come from fp32, so their fp64 sum is finite exactly when all of them are.
__device__ __forceinline__ void pair_accumulate(double acc[kSums], double w, const double p[3], const double q[3]) {
  if (!isfinite(((p[0] + p[1]) + p[2]) + ((q[0] + q[1]) + q[2]))) return;
  acc[0] += w;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double wp = w * p[a];
    acc[1 + a] += wp;
    acc[4 + a] += w * q[a];
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[7 + 3 * a + b] += wp * q[b];
    acc[16] += wp * p[a];
    acc[17] += w * q[a] * q[a];
  }
}

// Workgroup (256 threads) reduction of per-lane accumulators into one row of 18 partials: shuffle tree over the
// 64 lanes, LDS across the 4 waves, fixed order -> bitwise repeatable.  `red` is __shared__ [4][kSums].
__device__ __forceinline__ void block_reduce_store(double acc[kSums], double (*red)[kSums],
                                                   double* __restrict__ partial_row) {
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
    partial_row[threadIdx.x] = v;
  }
}
#define R3D_HD __host__ __device__
#else
#define R3D_HD
#endif

// ---- 3x3 SVD by one-sided Jacobi (Hestenes): A V = U diag(sig), sig descending, U and V orthogonal.
// Rank-deficient A: the missing left vectors are completed to a right-handed orthonormal frame.
R3D_HD inline void svd3(const double A_in[9], double U[9], double sig[3], double V[9]) {
  double A[9];
  for (int k = 0; k < 9; ++k) {
    A[k] = A_in[k];
    V[k] = (k % 4 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < 2; ++i)
      for (int j = i + 1; j < 3; ++j) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += A[3 * r + i] * A[3 * r + i];
          beta += A[3 * r + j] * A[3 * r + j];
          gamma += A[3 * r + i] * A[3 * r + j];
        }
        const double lim = 4e-16 * sqrt(alpha * beta);
        if (!(fabs(gamma) > lim) || gamma == 0.0) continue;
        off += fabs(gamma);
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          const double ai = A[3 * r + i], aj = A[3 * r + j];
          A[3 * r + i] = c * ai - s * aj;
          A[3 * r + j] = s * ai + c * aj;
          const double vi = V[3 * r + i], vj = V[3 * r + j];
          V[3 * r + i] = c * vi - s * vj;
          V[3 * r + j] = s * vi + c * vj;
        }
      }
    if (off == 0.0) break;
  }
  double n2[3];
  for (int c = 0; c < 3; ++c) n2[c] = A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c];
  int ord[3] = {0, 1, 2};  // descending singular values
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (n2[ord[b]] > n2[ord[a]]) {
        const int t = ord[a];
        ord[a] = ord[b];
        ord[b] = t;
      }
  double Vs[9];
  for (int c = 0; c < 3; ++c) {
    sig[c] = sqrt(n2[ord[c]]);
    for (int r = 0; r < 3; ++r) {
      Vs[3 * r + c] = V[3 * r + ord[c]];
      U[3 * r + c] = A[3 * r + ord[c]];
    }
  }
  for (int k = 0; k < 9; ++k) V[k] = Vs[k];
  const double tiny = sig[0] * 1e-14;
  int rank = 0;
  for (int c = 0; c < 3; ++c) {
    if (sig[c] > tiny && sig[c] > 0.0) {
      for (int r = 0; r < 3; ++r) U[3 * r + c] /= sig[c];
      rank = c + 1;
    } else {
      break;
    }
  }
  if (rank == 0) {
    for (int k = 0; k < 9; ++k) U[k] = (k % 4 == 0) ? 1.0 : 0.0;
  } else if (rank == 1) {
    // any unit vector orthogonal to u0, then their cross product
    const double u0[3] = {U[0], U[3], U[6]};
    int m = 0;
    if (fabs(u0[1]) < fabs(u0[m])) m = 1;
    if (fabs(u0[2]) < fabs(u0[m])) m = 2;
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d = u0[m];
    double u1[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
    const double nn = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (int r = 0; r < 3; ++r) u1[r] /= nn;
    U[1] = u1[0]; U[4] = u1[1]; U[7] = u1[2];
    U[2] = u0[1] * u1[2] - u0[2] * u1[1];
    U[5] = u0[2] * u1[0] - u0[0] * u1[2];
    U[8] = u0[0] * u1[1] - u0[1] * u1[0];
  } else if (rank == 2) {
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

R3D_HD inline double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// T (row-major 4x4) = [sR t; 0 1] minimising sum w |q - (s R p + t)|^2 from the 18 sums (Umeyama, PAMI 13(4) 1991).
// Returns 0, or 1 when the fit is undefined (weight sum < 3 or no spread in p): T is then the identity.
// *rms_out (optional) = sqrt(sum w |p-q|^2 / sum w) of the pairs as they stand.
R3D_HD inline int umeyama_from_sums(const double s[kSums], int with_scale, double T[16], double* rms_out) {
  for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
  const double n = s[0];
  if (rms_out) {
    const double e = s[16] + s[17] - 2.0 * (s[7] + s[11] + s[15]);
    *rms_out = n > 0.0 ? sqrt((e > 0.0 ? e : 0.0) / n) : 0.0;
  }
  if (!(n >= 3.0)) return 1;
  double mp[3], mq[3];
  for (int a = 0; a < 3; ++a) {
    mp[a] = s[1 + a] / n;
    mq[a] = s[4 + a] / n;
  }
  const double var_p = s[16] / n - (mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2]);
  if (!(var_p > 0.0)) return 1;
  // Sigma_qp[b][a] = E[(q-mq)_b (p-mp)_a] = U D V^T
  double M[9], U[9], D[3], V[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) M[3 * b + a] = s[7 + 3 * a + b] / n - mp[a] * mq[b];
  svd3(M, U, D, V);
  const double sgn = det3(U) * det3(V) < 0.0 ? -1.0 : 1.0;
  double R[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      R[3 * r + c] = U[3 * r + 0] * V[3 * c + 0] + U[3 * r + 1] * V[3 * c + 1] + sgn * U[3 * r + 2] * V[3 * c + 2];
  const double scale = with_scale ? (D[0] + D[1] + sgn * D[2]) / var_p : 1.0;
  if (!(scale > 0.0) || !(scale < 1e300)) return 1;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[4 * r + c] = scale * R[3 * r + c];
    T[4 * r + 3] = mq[r] - scale * (R[3 * r + 0] * mp[0] + R[3 * r + 1] * mp[1] + R[3 * r + 2] * mp[2]);
  }
  return 0;
}

}  // namespace r3d_icp
