// Apply a general 4x4 transform to an AoS xyz cloud on gfx950 (MI355X).
//
// Replaces local_world(flag=True) / point_camera of transfer_T_icp.py:71-97, 10-12:
//     p' = (T . [x, y, z, 1]^T)[0:3]        (scale lives in T's 3x3 block)
// and serves the standalone SE(3) apply of camera_to_world.py:57-59 on an existing cloud (Rinv . (p - t)).
//
// Roofline: HBM, 24 B/point (12 read + 12 written) for f32 clouds.
// Shape (A/B: profiles/r02_ab_kernels.log): a workgroup owns one tile of 1024 consecutive points; all of a lane's
// loads are issued up front with the NONTEMPORAL hint (a once-read stream: +6 % over cached loads on this 1:1
// read/write kernel), and every wave store instruction writes one contiguous run, 12 B per lane at a 12-B lane stride
// (`global_store_dwordx3 ... nt`):
//   * f32 out (apply_lane_kernel): lane-per-point rounds, one x3 store per point.  6.4 TB/s = 0.80 of peak.
//   * f64 out (apply_pair_kernel): two lanes per point -- the even lane computes output rows 0,1 and stores
//     (x_lo x_hi y_lo), the odd lane rows 1,2 and stores (y_hi z_lo z_hi) -- the same trick as fuse_pair_kernel.
// In-place operation is safe: every load of a wave is complete (s_waitcnt) before its first store, and the two lanes
// that share a point are neighbours in one wave.
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

typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// A lane's N points, loaded with nontemporal loads (a once-read stream: +6 % over cached loads on this 1:1 read/write
// kernel).  The compiler does not track vmcnt for inline-asm loads, so ALL of a lane's loads AND the s_waitcnt that
// completes them sit in ONE asm statement: its outputs are defined only after the wait, and no register copy or spill the
// compiler may insert can read them early.  Addresses are clamped by the caller (unconditional loads).
template <typename T>
struct RawPoint;
template <>
struct RawPoint<float> {
  f32x3 v;
  __device__ __forceinline__ void get(double p[3]) const {
    p[0] = (double)v.x;
    p[1] = (double)v.y;
    p[2] = (double)v.z;
  }
};
template <>
struct RawPoint<double> {
  f64x2 xy;
  double z;
  __device__ __forceinline__ void get(double p[3]) const {
    p[0] = xy.x;
    p[1] = xy.y;
    p[2] = z;
  }
};

#define R3D_LD3(o, a) "global_load_dwordx3 %" #o ", %" #a ", off nt\n\t"
#define R3D_LD6(o, o2, a) "global_load_dwordx4 %" #o ", %" #a ", off nt\n\tglobal_load_dwordx2 %" #o2 ", %" #a ", off offset:16 nt\n\t"

__device__ __forceinline__ void load_points(RawPoint<float> (&r)[4], const float* const (&a)[4]) {
  asm volatile(R3D_LD3(0, 4) R3D_LD3(1, 5) R3D_LD3(2, 6) R3D_LD3(3, 7) "s_waitcnt vmcnt(0)"
               : "=&v"(r[0].v), "=&v"(r[1].v), "=&v"(r[2].v), "=&v"(r[3].v)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
               : "memory");
}
__device__ __forceinline__ void load_points(RawPoint<float> (&r)[8], const float* const (&a)[8]) {
  asm volatile(R3D_LD3(0, 8) R3D_LD3(1, 9) R3D_LD3(2, 10) R3D_LD3(3, 11) R3D_LD3(4, 12) R3D_LD3(5, 13) R3D_LD3(6, 14) R3D_LD3(7, 15)
               "s_waitcnt vmcnt(0)"
               : "=&v"(r[0].v), "=&v"(r[1].v), "=&v"(r[2].v), "=&v"(r[3].v), "=&v"(r[4].v), "=&v"(r[5].v), "=&v"(r[6].v), "=&v"(r[7].v)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
               : "memory");
}
__device__ __forceinline__ void load_points(RawPoint<double> (&r)[4], const double* const (&a)[4]) {
  asm volatile(R3D_LD6(0, 1, 8) R3D_LD6(2, 3, 9) R3D_LD6(4, 5, 10) R3D_LD6(6, 7, 11) "s_waitcnt vmcnt(0)"
               : "=&v"(r[0].xy), "=&v"(r[0].z), "=&v"(r[1].xy), "=&v"(r[1].z), "=&v"(r[2].xy), "=&v"(r[2].z), "=&v"(r[3].xy),
                 "=&v"(r[3].z)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
               : "memory");
}
__device__ __forceinline__ void load_points(RawPoint<double> (&r)[8], const double* const (&a)[8]) {
  asm volatile(R3D_LD6(0, 1, 16) R3D_LD6(2, 3, 17) R3D_LD6(4, 5, 18) R3D_LD6(6, 7, 19) R3D_LD6(8, 9, 20) R3D_LD6(10, 11, 21)
               R3D_LD6(12, 13, 22) R3D_LD6(14, 15, 23) "s_waitcnt vmcnt(0)"
               : "=&v"(r[0].xy), "=&v"(r[0].z), "=&v"(r[1].xy), "=&v"(r[1].z), "=&v"(r[2].xy), "=&v"(r[2].z), "=&v"(r[3].xy),
                 "=&v"(r[3].z), "=&v"(r[4].xy), "=&v"(r[4].z), "=&v"(r[5].xy), "=&v"(r[5].z), "=&v"(r[6].xy), "=&v"(r[6].z),
                 "=&v"(r[7].xy), "=&v"(r[7].z)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
               : "memory");
}
#undef R3D_LD3
#undef R3D_LD6

// output row `r` of the transform for the point p (fp64, the reference's order)
template <bool SE3>
__device__ __forceinline__ double out_row(const double T[12], int r, const double p[3], const double d[3]) {
  if (SE3)  // Rinv . (p - t), the order of point_camera (camera_to_world.py:57-59) and of the fused kernel
    return fma(T[3 * r + 2], d[2], fma(T[3 * r + 1], d[1], T[3 * r + 0] * d[0]));
  // row . [x y z 1]  (transfer_T_icp.py:10-12)
  return fma(T[4 * r + 2], p[2], fma(T[4 * r + 1], p[1], T[4 * r + 0] * p[0])) + T[4 * r + 3];
}

template <typename IT, bool SE3>
__global__ __launch_bounds__(kThreads) void apply_lane_kernel(const ApplyArgs a) {
  const IT* in = static_cast<const IT*>(a.in);
  float* out = static_cast<float*>(a.out);
  const uint64_t n_tiles = (a.n + kTile - 1) / kTile;
  double T[12];
  load_T(a, T);
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t base = tile * kTile + threadIdx.x;
    RawPoint<IT> raw[kPts];
    const IT* addr[kPts];
#pragma unroll
    for (int r = 0; r < kPts; ++r) {
      const uint64_t i = base + (uint64_t)r * kThreads;
      addr[r] = in + (i < a.n ? i : a.n - 1) * 3;   // clamped: the loads are unconditional
    }
    load_points(raw, addr);
#pragma unroll
    for (int r = 0; r < kPts; ++r) {
      const uint64_t i = base + (uint64_t)r * kThreads;
      if (i < a.n) {
        double p[3], d[3];
        raw[r].get(p);
        if (SE3) {
          d[0] = p[0] - T[9];
          d[1] = p[1] - T[10];
          d[2] = p[2] - T[11];
        }
        const f32x3 w = {(float)out_row<SE3>(T, 0, p, d), (float)out_row<SE3>(T, 1, p, d), (float)out_row<SE3>(T, 2, p, d)};
        asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(out + i * 3), "v"(w) : "memory");
      }
    }
  }
}

// f64 out: item q of a tile (q = r*256 + tid, r = 0..7) is half (q & 1) of point (q >> 1)
template <typename IT, bool SE3>
__global__ __launch_bounds__(kThreads) void apply_pair_kernel(const ApplyArgs a) {
  constexpr int kItems = 2 * kPts;
  const IT* in = static_cast<const IT*>(a.in);
  uint32_t* out = static_cast<uint32_t*>(a.out);
  const uint64_t n_tiles = (a.n + kTile - 1) / kTile;
  const bool odd = threadIdx.x & 1u;
  double T[12];
  load_T(a, T);
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t p0 = tile * kTile;
    RawPoint<IT> raw[kItems];
    const IT* addr[kItems];
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const uint64_t i = p0 + ((r * kThreads + threadIdx.x) >> 1);
      addr[r] = in + (i < a.n ? i : a.n - 1) * 3;   // clamped: the loads are unconditional
    }
    load_points(raw, addr);
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const uint32_t q = r * kThreads + threadIdx.x;
      const uint64_t i = p0 + (q >> 1);
      if (i < a.n) {
        double p[3], d[3];
        raw[r].get(p);
        if (SE3) {
          d[0] = p[0] - T[9];
          d[1] = p[1] - T[10];
          d[2] = p[2] - T[11];
        }
        // both lanes need row 1; the even lane adds row 0, the odd lane row 2 (same instructions, selected operands)
        const double w1 = out_row<SE3>(T, 1, p, d);
        double Tsel[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Tsel[k] = T[k];
        const int lo = SE3 ? 3 : 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < lo) Tsel[k] = odd ? T[2 * lo + k] : T[k];
        const double w02 = out_row<SE3>(Tsel, 0, p, d);
        const double a_ = odd ? w1 : w02, b_ = odd ? w02 : w1;
        const uint32_t alo = (uint32_t)__double2loint(a_), ahi = (uint32_t)__double2hiint(a_);
        const uint32_t blo = (uint32_t)__double2loint(b_), bhi = (uint32_t)__double2hiint(b_);
        const u32x3 val = odd ? u32x3{ahi, blo, bhi} : u32x3{alo, ahi, blo};
        asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(out + p0 * 6 + (uint64_t)q * 3), "v"(val) : "memory");
      }
    }
  }
}

// k copies of one small f32 cloud, copy blockIdx.y moved by the blockIdx.y-th 4x4 of a table in HBM (the candidate poses of a
// multi-start): the same arithmetic as apply_lane_kernel (out_row), one launch instead of k
__global__ __launch_bounds__(kThreads) void apply_many_kernel(const float* __restrict__ in, uint64_t n, const double* __restrict__ Ts,
                                                              float* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const double* M = Ts + (uint64_t)blockIdx.y * 16;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = M[k];
  const double p[3] = {(double)in[i * 3], (double)in[i * 3 + 1], (double)in[i * 3 + 2]};
  float* o = out + ((uint64_t)blockIdx.y * n + i) * 3;
  o[0] = (float)out_row<false>(T, 0, p, p);
  o[1] = (float)out_row<false>(T, 1, p, p);
  o[2] = (float)out_row<false>(T, 2, p, p);
}

template <typename IT, bool SE3>
void launch(const ApplyArgs& a, int out_dtype, int blocks, hipStream_t s) {
  if (out_dtype == R3D_F32)
    hipLaunchKernelGGL((apply_lane_kernel<IT, SE3>), dim3(blocks), dim3(kThreads), 0, s, a);
  else
    hipLaunchKernelGGL((apply_pair_kernel<IT, SE3>), dim3(blocks), dim3(kThreads), 0, s, a);
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
  r3d_wrote(ctx, d_xyz_out, (size_t)n_points * 3 * r3d_xyz_size(out_dtype));   // an ICP loop working on these points is over
  ApplyArgs a;
  a.in = d_xyz_in;
  a.out = d_xyz_out;
  for (int k = 0; k < 12; ++k) a.T[k] = h_M ? h_M[k] : 0.0;
  a.d_T = d_M;
  a.n = (uint64_t)n_points;
  const uint64_t n_tiles = (a.n + kTile - 1) / kTile;
  // one tile per workgroup measured best for this 1:1 read/write stream
  uint64_t blocks64 = ctx->apply_blocks > 0 ? (uint64_t)ctx->apply_blocks : n_tiles;
  if (blocks64 > n_tiles) blocks64 = n_tiles;
  if (blocks64 > 0x7fffffffull) blocks64 = 0x7fffffffull;
  const int blocks = (int)blocks64;
  if (in_dtype == R3D_F32)
    launch<float, SE3>(a, out_dtype, blocks, ctx->stream);
  else
    launch<double, SE3>(a, out_dtype, blocks, ctx->stream);
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

int r3d_apply_T_many(r3d_ctx* ctx, const void* d_xyz_in, int in_dtype, int64_t n_points, const double* h_Ts, int n_transforms,
                     void* d_xyz_out, int out_dtype) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_transforms >= 0 && (n_transforms == 0 || h_Ts != nullptr), "bad transform list");
  R3D_REQUIRE(in_dtype == R3D_F32 || in_dtype == R3D_F64, "unknown input dtype %d", in_dtype);
  R3D_REQUIRE(out_dtype == R3D_F32 || out_dtype == R3D_F64, "unknown output dtype %d", out_dtype);
  R3D_REQUIRE(n_points >= 0, "n_points must be >= 0");
  if (n_transforms == 0 || n_points == 0) return R3D_OK;
  R3D_REQUIRE(d_xyz_in && d_xyz_out && d_xyz_in != d_xyz_out, "NULL device pointer, or the copies would overwrite the cloud they are made from");
  if (in_dtype == R3D_F32 && out_dtype == R3D_F32 && n_transforms <= 65535) {
    // the table of matrices travels through a scratch slot; the copy is enqueued from a staging copy that outlives the call
    void* d_T = nullptr;
    if ((rc = r3d_scratch(ctx, 4, (size_t)n_transforms * 16 * sizeof(double), &d_T))) return rc;
    R3D_HIP(hipMemcpyAsync(d_T, h_Ts, (size_t)n_transforms * 16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    R3D_HIP(hipStreamSynchronize(ctx->stream));   // h_Ts is the caller's (pageable) memory
    hipLaunchKernelGGL(apply_many_kernel, dim3((unsigned)((n_points + kThreads - 1) / kThreads), (unsigned)n_transforms), dim3(kThreads),
                       0, ctx->stream, static_cast<const float*>(d_xyz_in), (uint64_t)n_points, static_cast<const double*>(d_T),
                       static_cast<float*>(d_xyz_out));
    R3D_HIP(hipGetLastError());
    return R3D_OK;
  }
  const size_t block = (size_t)n_points * 3 * r3d_xyz_size(out_dtype);
  for (int k = 0; k < n_transforms; ++k) {
    rc = apply_common<false>(ctx, d_xyz_in, in_dtype, n_points, h_Ts + 16 * (size_t)k, static_cast<char*>(d_xyz_out) + block * (size_t)k,
                             out_dtype);
    if (rc) return rc;
  }
  return R3D_OK;
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
