// Apply a general 4x4 transform to an AoS xyz cloud on gfx950 (MI355X).
//
// Replaces local_world(flag=True) / point_camera of transfer_T_icp.py:71-97, 10-12:
//     p' = (T . [x, y, z, 1]^T)[0:3]        (scale lives in T's 3x3 block)
// and serves the standalone SE(3) apply of camera_to_world.py:57-59 on an existing cloud
// (T = [Rinv | -Rinv t]).
//
// Roofline: HBM, 24 B/point (12 read + 12 written) for f32 clouds.
// Default (apply_variant 0): lane-per-point rounds.  In each round the 64 lanes of a wave hold 64
// CONSECUTIVE points: one 12-byte load and one 12-byte nontemporal store per lane at a 12-byte lane
// stride = 768 contiguous bytes per wave instruction, no LDS, no barrier.  The fused kernel's A/B
// (profiles/variants_r01.md) showed this shape beating LDS-transposed 16-byte stores by 1.4x.
// apply_variant 1 keeps that LDS design for comparison: a 256-thread workgroup owns a tile of 1024
// points = 12 KiB, read and written as tile-linear 16-B pieces staged through LDS.
// In-place operation is safe in both: a point is read before it is written, by the same lane.
#include <type_traits>

#include "r3d_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPts = 4;
constexpr int kTile = kThreads * kPts;

struct ApplyArgs {
  const void* in;
  void* out;
  double T[12];  // affine: rows 0..2 of the 4x4, row-major.  SE3: Rinv row-major (9) then t (3)
  const double* d_T;  // non-NULL: the 12 numbers are read from HBM instead (r3d_apply_T_dev: the matrix was made on the GPU)
  uint64_t n;    // points
};

// the 12 coefficients: kernel arguments, or 12 scalar loads from a wave-uniform device address
__device__ __forceinline__ void load_T(const ApplyArgs& a, double T[12]) {
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = a.d_T ? a.d_T[k] : a.T[k];
}

template <typename IT, typename OT, bool VEC, bool SE3>
__global__ __launch_bounds__(kThreads) void apply_kernel(const ApplyArgs a) {
  // one buffer, sized for the wider of the two element types
  constexpr size_t kElt = sizeof(IT) > sizeof(OT) ? sizeof(IT) : sizeof(OT);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kTile * 3 * kElt];
  IT* lin = reinterpret_cast<IT*>(lds_raw);
  OT* lout = reinterpret_cast<OT*>(lds_raw);
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = (a.n + kTile - 1) / kTile;
  double T[12];
  load_T(a, T);

  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t p_base = tile * kTile;
    const uint32_t n_pts = (uint32_t)min((uint64_t)kTile, a.n - p_base);
    const uint32_t n_elts = n_pts * 3;
    const IT* src = static_cast<const IT*>(a.in) + p_base * 3;
    OT* dst = static_cast<OT*>(a.out) + p_base * 3;

    // ---- tile -> LDS ----
    if (VEC) {
      constexpr uint32_t kPerPiece = 16 / sizeof(IT);
      const uint32_t n_pieces = n_elts / kPerPiece;
      using V = typename std::conditional<sizeof(IT) == 4, float4, double2>::type;
      for (uint32_t q = tid; q < n_pieces; q += kThreads)
        reinterpret_cast<V*>(lin)[q] = reinterpret_cast<const V*>(src)[q];
      for (uint32_t e = n_pieces * kPerPiece + tid; e < n_elts; e += kThreads) lin[e] = src[e];
    } else {
      for (uint32_t e = tid; e < n_elts; e += kThreads) lin[e] = src[e];
    }
    __syncthreads();

    // ---- transform this lane's 4 points (fp64, reference order: row . [x y z 1]) ----
    double w[kPts * 3];
    const uint32_t first = tid * kPts;
#pragma unroll
    for (int k = 0; k < kPts; ++k) {
      if (first + k < n_pts) {
        const double x = (double)lin[(first + k) * 3 + 0];
        const double y = (double)lin[(first + k) * 3 + 1];
        const double z = (double)lin[(first + k) * 3 + 2];
        if (SE3) {  // Rinv . (p - t), the order of point_camera (camera_to_world.py:57-59) and of the fused kernel
          const double dx = x - T[9], dy = y - T[10], dz = z - T[11];
#pragma unroll
          for (int r = 0; r < 3; ++r)
            w[3 * k + r] = fma(T[3 * r + 2], dz, fma(T[3 * r + 1], dy, T[3 * r + 0] * dx));
        } else {  // row . [x y z 1]  (transfer_T_icp.py:10-12)
#pragma unroll
          for (int r = 0; r < 3; ++r)
            w[3 * k + r] = fma(T[4 * r + 2], z, fma(T[4 * r + 1], y, T[4 * r + 0] * x)) + T[4 * r + 3];
        }
      }
    }
    __syncthreads();  // everyone has read its inputs before the buffer is reused for outputs
#pragma unroll
    for (int k = 0; k < kPts; ++k) {
      if (first + k < n_pts) {
#pragma unroll
        for (int r = 0; r < 3; ++r) lout[(first + k) * 3 + r] = (OT)w[3 * k + r];
      }
    }
    __syncthreads();

    // ---- LDS -> tile ----
    if (VEC) {
      constexpr uint32_t kPerPiece = 16 / sizeof(OT);
      const uint32_t n_pieces = n_elts / kPerPiece;
      using V = typename std::conditional<sizeof(OT) == 4, float4, double2>::type;
      for (uint32_t q = tid; q < n_pieces; q += kThreads)
        reinterpret_cast<V*>(dst)[q] = reinterpret_cast<const V*>(lout)[q];
      for (uint32_t e = n_pieces * kPerPiece + tid; e < n_elts; e += kThreads) dst[e] = lout[e];
    } else {
      for (uint32_t e = tid; e < n_elts; e += kThreads) dst[e] = lout[e];
    }
    __syncthreads();
  }
}

template <typename T>
struct __attribute__((packed, aligned(4))) Packed3 {
  T x, y, z;
};

template <typename OT>
__device__ __forceinline__ void store3_nt(OT* dst, const double w[3]);
template <>
__device__ __forceinline__ void store3_nt<float>(float* dst, const double w[3]) {
  typedef float v3 __attribute__((ext_vector_type(3)));
  asm volatile("global_store_dwordx3 %0, %1, off nt" ::"v"(dst), "v"(v3{(float)w[0], (float)w[1], (float)w[2]}) : "memory");
}
template <>
__device__ __forceinline__ void store3_nt<double>(double* dst, const double w[3]) {
  __builtin_nontemporal_store(w[0], dst);
  __builtin_nontemporal_store(w[1], dst + 1);
  __builtin_nontemporal_store(w[2], dst + 2);
}

template <typename IT, typename OT, bool SE3>
__global__ __launch_bounds__(kThreads) void apply_lane_kernel(const ApplyArgs a) {
  const IT* in = static_cast<const IT*>(a.in);
  OT* out = static_cast<OT*>(a.out);
  const uint64_t n_tiles = (a.n + kTile - 1) / kTile;
  double T[12];
  load_T(a, T);
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t base = tile * kTile + threadIdx.x;
    Packed3<IT> p[kPts];
#pragma unroll
    for (int r = 0; r < kPts; ++r) {
      const uint64_t i = base + (uint64_t)r * kThreads;
      if (i < a.n) p[r] = reinterpret_cast<const Packed3<IT>*>(in)[i];
    }
#pragma unroll
    for (int r = 0; r < kPts; ++r) {
      const uint64_t i = base + (uint64_t)r * kThreads;
      if (i < a.n) {
        const double x = (double)p[r].x, y = (double)p[r].y, z = (double)p[r].z;
        double w[3];
        if (SE3) {
          const double dx = x - T[9], dy = y - T[10], dz = z - T[11];
#pragma unroll
          for (int c = 0; c < 3; ++c) w[c] = fma(T[3 * c + 2], dz, fma(T[3 * c + 1], dy, T[3 * c + 0] * dx));
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c)
            w[c] = fma(T[4 * c + 2], z, fma(T[4 * c + 1], y, T[4 * c + 0] * x)) + T[4 * c + 3];
        }
        store3_nt<OT>(out + i * 3, w);
      }
    }
  }
}

template <typename IT, typename OT, bool SE3>
void launch(const ApplyArgs& a, int variant, bool vec, int blocks, hipStream_t s) {
  if (variant == 0)
    hipLaunchKernelGGL((apply_lane_kernel<IT, OT, SE3>), dim3(blocks), dim3(kThreads), 0, s, a);
  else if (vec)
    hipLaunchKernelGGL((apply_kernel<IT, OT, true, SE3>), dim3(blocks), dim3(kThreads), 0, s, a);
  else
    hipLaunchKernelGGL((apply_kernel<IT, OT, false, SE3>), dim3(blocks), dim3(kThreads), 0, s, a);
}

template <bool SE3>
int apply_common(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* h_M,
                 void* d_xyz_out, int out_dtype, const double* d_M = nullptr) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(in_dtype == R3D_F32 || in_dtype == R3D_F64, "unknown input dtype %d", in_dtype);
  R3D_REQUIRE(out_dtype == R3D_F32 || out_dtype == R3D_F64, "unknown output dtype %d", out_dtype);
  R3D_REQUIRE(n_points >= 0, "n_points must be >= 0");
  R3D_REQUIRE(h_M != nullptr || d_M != nullptr, "transform is NULL");
  if (n_points == 0) return R3D_OK;
  R3D_REQUIRE(d_xyz_in && d_xyz_out, "NULL device pointer");
  R3D_REQUIRE(d_xyz_in == d_xyz_out ? in_dtype == out_dtype : true, "in-place apply needs equal dtypes");
  ApplyArgs a;
  a.in = d_xyz_in;
  a.out = d_xyz_out;
  for (int k = 0; k < 12; ++k) a.T[k] = h_M ? h_M[k] : 0.0;
  a.d_T = d_M;
  a.n = (uint64_t)n_points;
  // tile bases are multiples of 1024 points = 12 KiB (f32) / 24 KiB (f64): 16-B alignment of
  // every tile follows from the alignment of the two base pointers
  const bool vec = ((uintptr_t)d_xyz_in % 16 == 0) && ((uintptr_t)d_xyz_out % 16 == 0);
  const uint64_t n_tiles = (a.n + kTile - 1) / kTile;
  // one tile per workgroup measured best for this 1:1 read/write stream (5.96 vs 5.67 TB/s at 8 workgroups per CU)
  uint64_t blocks64 = ctx->apply_blocks > 0 ? (uint64_t)ctx->apply_blocks : n_tiles;
  if (blocks64 > n_tiles) blocks64 = n_tiles;
  if (blocks64 > 0x7fffffffull) blocks64 = 0x7fffffffull;
  const int blocks = (int)blocks64;
  if (in_dtype == R3D_F32 && out_dtype == R3D_F32)
    launch<float, float, SE3>(a, ctx->apply_variant, vec, blocks, ctx->stream);
  else if (in_dtype == R3D_F32)
    launch<float, double, SE3>(a, ctx->apply_variant, vec, blocks, ctx->stream);
  else if (out_dtype == R3D_F32)
    launch<double, float, SE3>(a, ctx->apply_variant, vec, blocks, ctx->stream);
  else
    launch<double, double, SE3>(a, ctx->apply_variant, vec, blocks, ctx->stream);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

template <bool SE3>
int apply_host_common(r3d_ctx* ctx, const void* h_xyz_in, int in_dtype, int64_t n_points, const double* h_M,
                      void* h_xyz_out, int out_dtype) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(in_dtype == R3D_F32 || in_dtype == R3D_F64, "unknown input dtype %d", in_dtype);
  R3D_REQUIRE(out_dtype == R3D_F32 || out_dtype == R3D_F64, "unknown output dtype %d", out_dtype);
  R3D_REQUIRE(n_points >= 0, "n_points must be >= 0");
  if (n_points == 0) return R3D_OK;
  R3D_REQUIRE(h_xyz_in && h_xyz_out && h_M, "NULL host pointer");
  const size_t in_bytes = (size_t)n_points * 3 * r3d_xyz_size(in_dtype);
  const size_t out_bytes = (size_t)n_points * 3 * r3d_xyz_size(out_dtype);
  void *d_in = nullptr, *d_out = nullptr;
  if ((rc = r3d_scratch(ctx, 0, in_bytes, &d_in))) return rc;
  if ((rc = r3d_scratch(ctx, 1, out_bytes, &d_out))) return rc;
  // 64 Ki points per pipeline item keeps every chunk boundary on a tile (and 16-byte) boundary
  const int64_t item_pts = 65536;
  const size_t isz = 3 * r3d_xyz_size(in_dtype), osz = 3 * r3d_xyz_size(out_dtype);
  // whole items stream through the pinned pipeline; a ragged tail (< 64 Ki points) goes directly
  const int64_t whole = (n_points / item_pts) * item_pts;
  if (whole > 0) {
    auto launch = [&](int64_t lo, int64_t n) -> int {
      return apply_common<SE3>(ctx, static_cast<char*>(d_in) + (size_t)lo * item_pts * isz, in_dtype, n * item_pts, h_M,
                               static_cast<char*>(d_out) + (size_t)lo * item_pts * osz, out_dtype);
    };
    if ((rc = r3d_host_pipeline(ctx, whole / item_pts, item_pts * isz, item_pts * osz, h_xyz_in, h_xyz_out, d_in, d_out,
                                launch)))
      return rc;
  }
  const int64_t tail = n_points - whole;
  if (tail > 0) {
    const char* hi = static_cast<const char*>(h_xyz_in) + (size_t)whole * isz;
    char* ho = static_cast<char*>(h_xyz_out) + (size_t)whole * osz;
    char* di = static_cast<char*>(d_in) + (size_t)whole * isz;
    char* dout = static_cast<char*>(d_out) + (size_t)whole * osz;
    R3D_HIP(hipMemcpyAsync(di, hi, (size_t)tail * isz, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = apply_common<SE3>(ctx, di, in_dtype, tail, h_M, dout, out_dtype))) return rc;
    R3D_HIP(hipMemcpyAsync(ho, dout, (size_t)tail * osz, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(hipStreamSynchronize(ctx->stream));
  }
  return R3D_OK;
}

}  // namespace

extern "C" {

int r3d_apply_T(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* h_T,
                void* d_xyz_out, int out_dtype) {
  return apply_common<false>(ctx, d_xyz_in, in_dtype, n_points, h_T, d_xyz_out, out_dtype);
}

int r3d_apply_T_host(r3d_ctx* ctx, const void* h_xyz_in, int in_dtype, int64_t n_points, const double* h_T,
                     void* h_xyz_out, int out_dtype) {
  return apply_host_common<false>(ctx, h_xyz_in, in_dtype, n_points, h_T, h_xyz_out, out_dtype);
}

int r3d_apply_T_dev(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* d_T,
                    void* d_xyz_out, int out_dtype) {
  R3D_REQUIRE(d_T != nullptr, "d_T is NULL");
  return apply_common<false>(ctx, d_xyz_in, in_dtype, n_points, nullptr, d_xyz_out, out_dtype, d_T);
}

int r3d_se3_apply(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* h_pose,
                  void* d_xyz_out, int out_dtype) {
  return apply_common<true>(ctx, d_xyz_in, in_dtype, n_points, h_pose, d_xyz_out, out_dtype);
}

int r3d_se3_apply_host(r3d_ctx* ctx, const void* h_xyz_in, int in_dtype, int64_t n_points, const double* h_pose,
                       void* h_xyz_out, int out_dtype) {
  return apply_host_common<true>(ctx, h_xyz_in, in_dtype, n_points, h_pose, h_xyz_out, out_dtype);
}

}  // extern "C"
