// ICP estimation kernels for gfx950 (MI355X): brute-force nearest neighbour + the 18 fp64
// cross-covariance sums of the Umeyama similarity fit.
//
// NOT IN THE REFERENCE: other_tools/transfer_T_icp.py only *applies* a T_data.txt produced by an
// external ICP tool (transfer_T_icp.py:33-43, 99-108).  These kernels produce that matrix
// on-device; the definition is the build's own, parity unpinned.
//
// r3d_icp_nn -- roofline: fp32 VALU (N*M pair evaluations), not HBM.
//   d2(s,t) = fma(dz,dz, fma(dy,dy, dx*dx)) in fp32, argmin over t, lowest index wins ties.
//   * a 256-thread workgroup owns 256*S source points (S per lane, in registers) and sweeps the
//     whole target cloud in LDS tiles of 1024 points stored SoA (x[1024] y[1024] z[1024]).
//   * every lane reads the SAME 4 targets per step (3 broadcast ds_read_b128), so LDS traffic per
//     pair falls with S; the inner loop is 6 VALU per pair + min3 over groups of 32 targets.
//   * no per-pair index bookkeeping: the loop only tracks, per source, the minimum distance and
//     the 32-target GROUP in which it was first reached (strict <, so the earliest group wins);
//     a short epilogue re-evaluates that one group with the identical expression and picks the
//     lowest index that reproduces the minimum bit for bit.
//
// r3d_icp_accumulate -- roofline: HBM/L2 gather, 28 B/pair (12 src + 4 idx + 12 gathered tgt).
//   fp64 accumulators in registers, wavefront shuffle tree (width 64) -> LDS across the 4 waves
//   -> per-workgroup partials -> one fixed-order final pass.  No float atomics: bitwise
//   reproducible run to run.
#include <cmath>

#include "r3d_icp_sums.h"
#include "r3d_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTgtTile = 1024;  // targets per LDS tile
constexpr int kGroup = 32;      // targets per min-tracking group

// grid = (source blocks, target segments).  Segment y sweeps target tiles [y*tiles_per_seg, (y+1)*tiles_per_seg)
// and writes its winners to idx_out/d2_out + y*n_src; with more than one segment nn_merge_kernel picks
// the overall winner (earliest segment on ties = lowest index).  Splitting the targets is what fills
// 256 CUs when the source cloud alone gives too few workgroups.
template <int S>
__global__ __launch_bounds__(kThreads) void nn_kernel(const float* __restrict__ src, int64_t n_src,
                                                      const float* __restrict__ tgt, int64_t n_tgt,
                                                      int64_t tiles_per_seg, uint32_t* __restrict__ idx_out,
                                                      float* __restrict__ d2_out) {
  __shared__ __attribute__((aligned(16))) float tx[kTgtTile];
  __shared__ __attribute__((aligned(16))) float ty[kTgtTile];
  __shared__ __attribute__((aligned(16))) float tz[kTgtTile];

  const uint32_t tid = threadIdx.x;
  const int64_t s_base = (int64_t)blockIdx.x * (kThreads * S);
  float sx[S], sy[S], sz[S], best[S];
  uint32_t best_group[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t i = s_base + (int64_t)s * kThreads + tid;
    const bool ok = i < n_src;
    sx[s] = ok ? src[i * 3 + 0] : 0.f;
    sy[s] = ok ? src[i * 3 + 1] : 0.f;
    sz[s] = ok ? src[i * 3 + 2] : 0.f;
    best[s] = INFINITY;
    best_group[s] = 0;
  }

  const int64_t n_tiles = (n_tgt + kTgtTile - 1) / kTgtTile;
  const int64_t tile_lo = (int64_t)blockIdx.y * tiles_per_seg;
  const int64_t tile_hi = min(n_tiles, tile_lo + tiles_per_seg);
  idx_out += (int64_t)blockIdx.y * n_src;
  if (d2_out) d2_out += (int64_t)blockIdx.y * n_src;
  for (int64_t tile = tile_lo; tile < tile_hi; ++tile) {
    const int64_t t_base = tile * kTgtTile;
    const int64_t n_here = min((int64_t)kTgtTile, n_tgt - t_base);
    __syncthreads();  // previous tile fully consumed
    // AoS global -> SoA LDS; slots past the cloud's end get x = +inf so they can never win
    for (uint32_t e = tid; e < kTgtTile * 3; e += kThreads) {
      const uint32_t p = e / 3, c = e - p * 3;
      float v = (c == 0) ? INFINITY : 0.f;
      if ((int64_t)p < n_here) v = tgt[(t_base + p) * 3 + c];
      (c == 0 ? tx : c == 1 ? ty : tz)[p] = v;
    }
    __syncthreads();

    const uint32_t group0 = (uint32_t)(t_base / kGroup);
    for (int g = 0; g < kTgtTile / kGroup; ++g) {
      float gmin[S];
#pragma unroll
      for (int s = 0; s < S; ++s) gmin[s] = INFINITY;
#pragma unroll
      for (int q = 0; q < kGroup / 4; ++q) {
        const float4 X = reinterpret_cast<const float4*>(tx)[g * (kGroup / 4) + q];
        const float4 Y = reinterpret_cast<const float4*>(ty)[g * (kGroup / 4) + q];
        const float4 Z = reinterpret_cast<const float4*>(tz)[g * (kGroup / 4) + q];
#pragma unroll
        for (int s = 0; s < S; ++s) {
          // two targets per instruction (v_pk_add / v_pk_mul / v_pk_fma_f32): the same IEEE operations as one at a time
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 px = {sx[s], sx[s]}, py = {sy[s], sy[s]}, pz = {sz[s], sz[s]};
          const f32x2 ax = px - f32x2{X.x, X.y}, ay = py - f32x2{Y.x, Y.y}, az = pz - f32x2{Z.x, Z.y};
          const f32x2 bx = px - f32x2{X.z, X.w}, by = py - f32x2{Y.z, Y.w}, bz = pz - f32x2{Z.z, Z.w};
          const f32x2 da = __builtin_elementwise_fma(az, az, __builtin_elementwise_fma(ay, ay, ax * ax));
          const f32x2 db = __builtin_elementwise_fma(bz, bz, __builtin_elementwise_fma(by, by, bx * bx));
          gmin[s] = fminf(fminf(gmin[s], da.x), da.y);  // v_min3_f32
          gmin[s] = fminf(fminf(gmin[s], db.x), db.y);
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (gmin[s] < best[s]) {  // strict: the earliest group that reaches the minimum keeps it
          best[s] = gmin[s];
          best_group[s] = group0 + g;
        }
      }
    }
  }

  // epilogue: lowest index inside the winning group that reproduces the minimum exactly
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t i = s_base + (int64_t)s * kThreads + tid;
    if (i >= n_src) continue;
    const int64_t t0 = best[s] < INFINITY ? (int64_t)best_group[s] * kGroup : tile_lo * kTgtTile;
    uint32_t found = (uint32_t)t0;
    bool have = false;
    for (int k = 0; k < kGroup; ++k) {
      const int64_t t = t0 + k;
      if (t < n_tgt && !have) {
        const float dx = sx[s] - tgt[t * 3 + 0], dy = sy[s] - tgt[t * 3 + 1], dz = sz[s] - tgt[t * 3 + 2];
        const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        if (d == best[s]) {
          found = (uint32_t)t;
          have = true;
        }
      }
    }
    idx_out[i] = found;
    if (d2_out) d2_out[i] = best[s];
  }
}

__global__ __launch_bounds__(kThreads) void nn_merge_kernel(const uint32_t* __restrict__ part_idx,
                                                            const float* __restrict__ part_d2, int n_seg,
                                                            int64_t n_src, uint32_t* __restrict__ idx_out,
                                                            float* __restrict__ d2_out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_src) return;
  float best = part_d2[i];
  uint32_t bi = part_idx[i];
  for (int s = 1; s < n_seg; ++s) {
    const float d = part_d2[(int64_t)s * n_src + i];
    if (d < best) {  // strict: earlier segment (lower indices) keeps ties
      best = d;
      bi = part_idx[(int64_t)s * n_src + i];
    }
  }
  idx_out[i] = bi;
  if (d2_out) d2_out[i] = best;
}

// ---------------------------------------------------------------------------------------------
using r3d_icp::block_reduce_store;
using r3d_icp::kSums;
using r3d_icp::pair_accumulate;
using r3d_icp::pair_weight;

__global__ __launch_bounds__(kThreads) void accumulate_kernel(const float* __restrict__ src, int64_t n_src,
                                                              const float* __restrict__ tgt, int64_t n_tgt,
                                                              const uint32_t* __restrict__ idx,
                                                              const float* __restrict__ d2, float max_d2, float dead_zone,
                                                              double* __restrict__ partials) {
  __shared__ double red[kThreads / 64][kSums];
  double acc[kSums];
#pragma unroll
  for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
  {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_src; i += (int64_t)gridDim.x * kThreads) {
      if (max_d2 >= 0.f && !(d2[i] <= max_d2)) continue;
      const int64_t j = idx ? (int64_t)idx[i] : i;  // no index array: pair k with k (per-cloud moments)
      if (j >= n_tgt) continue;                      // a caller's index array is data: never read past the target cloud
      const double p[3] = {(double)src[i * 3 + 0], (double)src[i * 3 + 1], (double)src[i * 3 + 2]};
      const double q[3] = {(double)tgt[j * 3 + 0], (double)tgt[j * 3 + 1], (double)tgt[j * 3 + 2]};
      pair_accumulate(acc, d2 ? pair_weight(d2[i], dead_zone) : 1.0, p, q);
    }
  }
  block_reduce_store(acc, red, partials + (int64_t)blockIdx.x * kSums);
}

// ---- device-side similarity solve: the ICP loop never waits for the host ---------------------------------
__global__ void icp_state_reset_kernel(double* __restrict__ st) {
  for (int k = threadIdx.x; k < r3d_icp::kStateDoubles; k += blockDim.x) {
    double v = 0.0;
    if (k < 32 && (k % 16) % 5 == 0) v = 1.0;  // T_total = T_step = identity
    st[k] = v;
  }
}

// One thread: step = umeyama(sums); T_total <- step . T_total; history.  A degenerate fit leaves the identity step.
__device__ void icp_solve_step(const double* s, int with_scale, double* __restrict__ st) {
  double T[16], rms = 0.0;
  const int bad = r3d_icp::umeyama_from_sums(s, with_scale, T, &rms);
  double tot[16], nt[16];
  for (int k = 0; k < 16; ++k) tot[k] = st[r3d_icp::kStateTTotal + k];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double v = 0.0;
      for (int m = 0; m < 4; ++m) v += T[4 * r + m] * tot[4 * m + c];
      nt[4 * r + c] = v;
    }
  for (int k = 0; k < 16; ++k) {
    st[r3d_icp::kStateTStep + k] = T[k];
    st[r3d_icp::kStateTTotal + k] = nt[k];
  }
  const int it = (int)st[r3d_icp::kStateIters];
  if (r3d_icp::kStateHistory + it < r3d_icp::kStateDoubles) st[r3d_icp::kStateHistory + it] = rms;
  st[r3d_icp::kStateIters] = (double)(it + 1);
  if (bad) st[r3d_icp::kStateStatus] = 1.0;
  st[r3d_icp::kStateRms] = rms;
  st[r3d_icp::kStatePairs] = s[0];
}

__global__ void icp_solve_kernel(const double* __restrict__ sums, int with_scale, double* __restrict__ st) {
  double s[kSums];
  for (int k = 0; k < kSums; ++k) s[k] = sums[k];
  icp_solve_step(s, with_scale, st);
}

// ONE workgroup finishes a sums pass: (1) the per-workgroup partial rows, lane t taking rows t, t+256, ... in order;
// (2) fixed shuffle / LDS tree -> the 18 sums; (3) optionally the similarity solve and the ICP state update by lane 0.
// Every order is fixed: bitwise repeatable.  One launch instead of (final reduction, solve).
__global__ __launch_bounds__(kThreads) void sums_finish_kernel(const double* __restrict__ partials, int n_rows,
                                                               double* __restrict__ sums_out, int with_scale,
                                                               double* __restrict__ state) {
  __shared__ double red[kThreads / 64][kSums];
  __shared__ double total[kSums];
  double acc[kSums];
#pragma unroll
  for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < n_rows; b += kThreads) {
#pragma unroll
    for (int k = 0; k < kSums; ++k) acc[k] += partials[(int64_t)b * kSums + k];
  }
  block_reduce_store(acc, red, total);
  __syncthreads();
  if (threadIdx.x < kSums) sums_out[threadIdx.x] = total[threadIdx.x];
  if (threadIdx.x == 0 && state != nullptr) {
    double s[kSums];
    for (int k = 0; k < kSums; ++k) s[k] = total[k];
    icp_solve_step(s, with_scale, state);
  }
}

}  // namespace

// Second half of every sums pass (also of the fused NN + sums path of r3d_nnindex.hip): see sums_finish_kernel.
// d_state != NULL additionally solves the similarity step on the GPU and updates the ICP state.
int r3d_icp_sums_finish(r3d_ctx* ctx, const double* d_partials, int n_rows, double* d_sums_out, int with_scale, double* d_state) {
  hipLaunchKernelGGL(sums_finish_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, d_partials, n_rows, d_sums_out, with_scale, d_state);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

extern "C" {

int r3d_icp_nn(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
               uint32_t* d_idx_out, float* d_d2_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_src >= 0 && n_tgt >= 0, "negative cloud size");
  if (n_src == 0) return R3D_OK;
  R3D_REQUIRE(n_tgt >= 1, "target cloud is empty");
  R3D_REQUIRE(n_tgt < ((int64_t)1 << 32), "target cloud too large for uint32 indices");
  R3D_REQUIRE(d_src && d_tgt && d_idx_out, "NULL device pointer");
  // S sources per lane cut the LDS reads per pair; target segments supply the workgroups that S takes away
  int S = ctx->nn_variant;
  if (S != 1 && S != 2 && S != 4) S = 4;
  const int64_t per_block = (int64_t)kThreads * S;
  const int64_t src_blocks = (n_src + per_block - 1) / per_block;
  R3D_REQUIRE(src_blocks < ((int64_t)1 << 31), "source cloud too large");
  const int64_t n_tiles = (n_tgt + kTgtTile - 1) / kTgtTile;
  int64_t want_blocks = (int64_t)ctx->num_cus * 32;
  int64_t n_seg = (want_blocks + src_blocks - 1) / src_blocks;
  if (n_seg > n_tiles) n_seg = n_tiles;
  if (n_seg > 65535) n_seg = 65535;
  if (n_seg < 1) n_seg = 1;
  const int64_t tiles_per_seg = (n_tiles + n_seg - 1) / n_seg;
  n_seg = (n_tiles + tiles_per_seg - 1) / tiles_per_seg;
  uint32_t* k_idx = d_idx_out;
  float* k_d2 = d_d2_out;
  if (n_seg > 1) {
    void* part = nullptr;
    if ((rc = r3d_scratch(ctx, 5, (size_t)n_seg * n_src * 8, &part))) return rc;
    k_idx = static_cast<uint32_t*>(part);
    k_d2 = reinterpret_cast<float*>(k_idx + (size_t)n_seg * n_src);
  }
  const dim3 grid((unsigned)src_blocks, (unsigned)n_seg);
  switch (S) {
    case 1:
      hipLaunchKernelGGL((nn_kernel<1>), grid, dim3(kThreads), 0, ctx->stream, d_src, n_src, d_tgt, n_tgt,
                         tiles_per_seg, k_idx, k_d2);
      break;
    case 2:
      hipLaunchKernelGGL((nn_kernel<2>), grid, dim3(kThreads), 0, ctx->stream, d_src, n_src, d_tgt, n_tgt,
                         tiles_per_seg, k_idx, k_d2);
      break;
    default:
      hipLaunchKernelGGL((nn_kernel<4>), grid, dim3(kThreads), 0, ctx->stream, d_src, n_src, d_tgt, n_tgt,
                         tiles_per_seg, k_idx, k_d2);
      break;
  }
  if (n_seg > 1)
    hipLaunchKernelGGL(nn_merge_kernel, dim3((unsigned)((n_src + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       ctx->stream, k_idx, k_d2, (int)n_seg, n_src, d_idx_out, d_d2_out);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_icp_nn_host(r3d_ctx* ctx, const float* h_src, int64_t n_src, const float* h_tgt, int64_t n_tgt,
                    uint32_t* h_idx_out, float* h_d2_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_src >= 0 && n_tgt >= 0, "negative cloud size");
  if (n_src == 0) return R3D_OK;
  R3D_REQUIRE(n_tgt >= 1, "target cloud is empty");
  R3D_REQUIRE(h_src && h_tgt && h_idx_out, "NULL host pointer");
  void *d_src = nullptr, *d_tgt = nullptr, *d_idx = nullptr, *d_d2 = nullptr;
  if ((rc = r3d_scratch(ctx, 0, (size_t)n_src * 12, &d_src))) return rc;
  if ((rc = r3d_scratch(ctx, 1, (size_t)n_tgt * 12, &d_tgt))) return rc;
  if ((rc = r3d_scratch(ctx, 2, (size_t)n_src * 4, &d_idx))) return rc;
  if ((rc = r3d_scratch(ctx, 3, (size_t)n_src * 4, &d_d2))) return rc;
  R3D_HIP(hipMemcpyAsync(d_src, h_src, (size_t)n_src * 12, hipMemcpyHostToDevice, ctx->stream));
  R3D_HIP(hipMemcpyAsync(d_tgt, h_tgt, (size_t)n_tgt * 12, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = r3d_icp_nn(ctx, (const float*)d_src, n_src, (const float*)d_tgt, n_tgt, (uint32_t*)d_idx, (float*)d_d2)))
    return rc;
  R3D_HIP(hipMemcpyAsync(h_idx_out, d_idx, (size_t)n_src * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (h_d2_out) R3D_HIP(hipMemcpyAsync(h_d2_out, d_d2, (size_t)n_src * 4, hipMemcpyDeviceToHost, ctx->stream));
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  return R3D_OK;
}

// sums over the matched pairs into a DEVICE array of 18 doubles; asynchronous on the ctx stream
static int accumulate_impl(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
                           const uint32_t* d_idx, const float* d_d2, float max_d2, float dead_zone, double* d_sums_out,
                           int with_scale, double* d_state) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_src >= 0 && n_tgt >= 0, "negative cloud size");
  R3D_REQUIRE(d_sums_out != nullptr, "d_sums_out is NULL");
  if (n_src == 0) {
    R3D_HIP(hipMemsetAsync(d_sums_out, 0, kSums * sizeof(double), ctx->stream));
    return R3D_OK;
  }
  R3D_REQUIRE(d_src && d_tgt, "NULL device pointer");
  R3D_REQUIRE(d_idx != nullptr || n_tgt >= n_src, "identity pairing (d_idx == NULL) needs n_tgt >= n_src");
  const bool gated = max_d2 >= 0.f;
  R3D_REQUIRE((!gated && !(dead_zone > 0.f)) || d_d2 != nullptr, "max_d2 >= 0 or dead_zone > 0 needs the d2 array");
  int blocks = (int)((n_src + kThreads - 1) / kThreads);
  if (blocks > ctx->num_cus * 4) blocks = ctx->num_cus * 4;
  void* d_part_v = nullptr;
  if ((rc = r3d_scratch(ctx, 4, ((size_t)blocks + 1) * kSums * sizeof(double), &d_part_v))) return rc;
  double* d_part = static_cast<double*>(d_part_v);
  hipLaunchKernelGGL(accumulate_kernel, dim3(blocks), dim3(kThreads), 0, ctx->stream, d_src, n_src, d_tgt, n_tgt, d_idx,
                     (gated || dead_zone > 0.f) ? d_d2 : nullptr, gated ? max_d2 : -1.f, dead_zone, d_part);
  R3D_HIP(hipGetLastError());
  return r3d_icp_sums_finish(ctx, d_part, blocks, d_sums_out, with_scale, d_state);
}

int r3d_icp_accumulate_dev(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
                           const uint32_t* d_idx, const float* d_d2, float max_d2, float dead_zone,
                           double* d_sums_out) {
  return accumulate_impl(ctx, d_src, n_src, d_tgt, n_tgt, d_idx, d_d2, max_d2, dead_zone, d_sums_out, 0, nullptr);
}

int r3d_icp_accumulate(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
                       const uint32_t* d_idx, const float* d_d2, float max_d2, double* h_sums) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(h_sums != nullptr, "h_sums is NULL");
  for (int k = 0; k < kSums; ++k) h_sums[k] = 0.0;
  R3D_REQUIRE(n_src >= 0 && n_tgt >= 0, "negative cloud size");
  if (n_src == 0) return R3D_OK;
  void* d_sums = nullptr;
  if ((rc = r3d_scratch(ctx, 3, kSums * sizeof(double), &d_sums))) return rc;
  if ((rc = r3d_icp_accumulate_dev(ctx, d_src, n_src, d_tgt, n_tgt, d_idx, d_d2, max_d2, 0.f, (double*)d_sums)))
    return rc;
  R3D_HIP(hipMemcpyAsync(h_sums, d_sums, kSums * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  return R3D_OK;
}

int r3d_umeyama_from_sums(const double* h_sums, int with_scale, double* h_T, double* h_rms_out) {
  R3D_REQUIRE(h_sums && h_T, "NULL argument");
  const int bad = r3d_icp::umeyama_from_sums(h_sums, with_scale, h_T, h_rms_out);
  if (bad) {
    r3d_set_error("similarity fit undefined: weight sum %g (need >= 3) or no spread in the source points", h_sums[0]);
    return R3D_ERR_INVALID;
  }
  return R3D_OK;
}

int r3d_icp_state_reset(r3d_ctx* ctx, double* d_state) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(d_state != nullptr, "d_state is NULL");
  if (ctx->loop_state == d_state) ctx->loop_state = nullptr;   // a new loop starts here
  hipLaunchKernelGGL(icp_state_reset_kernel, dim3(1), dim3(64), 0, ctx->stream, d_state);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_icp_solve_dev(r3d_ctx* ctx, const double* d_sums, int with_scale, double* d_state) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(d_sums && d_state, "NULL device pointer");
  hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(1), 0, ctx->stream, d_sums, with_scale, d_state);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_icp_iterate(r3d_ctx* ctx, r3d_nn_index* index, float* d_src, int64_t n_src, const float* d_tgt, int64_t n_tgt,
                    uint32_t* d_idx, float* d_d2, int n_iters, int with_scale, float max_d2, double* d_state) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_iters >= 0, "n_iters must be >= 0");
  R3D_REQUIRE(n_src >= 3, "need at least 3 source points");
  R3D_REQUIRE(d_src && d_idx && d_d2 && d_state, "NULL device pointer");
  R3D_REQUIRE(index != nullptr || d_tgt != nullptr, "need an index or a target cloud");
  void* d_sums_v = nullptr;
  if ((rc = r3d_scratch(ctx, 3, kSums * sizeof(double), &d_sums_v))) return rc;
  double* d_sums = static_cast<double*>(d_sums_v);
  const bool going_on = index && ctx->loop_state == d_state && ctx->loop_src == d_src && ctx->loop_idx == d_idx && ctx->loop_index == index;
  for (int it = 0; it < n_iters; ++it) {
    if (index) {
      // sources are kept in the index's Morton order by the caller (r3d_nn_index_sort_cloud): no sort, sums fused
      // (the last kernel of the sums pass also solves the step and updates d_state)
      if ((rc = r3d_nn_index_query_solve(index, d_src, n_src, d_idx, d_d2, max_d2, d_sums, with_scale, d_state, it > 0 || going_on)))
        return rc;
    } else {
      if ((rc = r3d_icp_nn(ctx, d_src, n_src, d_tgt, n_tgt, d_idx, d_d2))) return rc;
      if ((rc = accumulate_impl(ctx, d_src, n_src, d_tgt, n_tgt, d_idx, d_d2, max_d2, 0.f, d_sums, with_scale, d_state)))
        return rc;
    }
    if ((rc = r3d_apply_T_dev(ctx, d_src, R3D_F32, n_src, d_state + r3d_icp::kStateTStep, d_src, R3D_F32))) return rc;
  }
  if (index && n_iters > 0) {
    ctx->loop_state = d_state;
    ctx->loop_src = d_src;
    ctx->loop_src_bytes = (size_t)n_src * 12;
    ctx->loop_idx = d_idx;
    ctx->loop_index = index;
  }
  return R3D_OK;
}

}  // extern "C"
