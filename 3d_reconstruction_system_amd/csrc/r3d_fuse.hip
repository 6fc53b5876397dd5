// Fused depth-raster -> world-frame point kernel for gfx950 (MI355X).
//
// Replaces, in ONE launch over a whole batch of frames, the reference's two per-point
// Python loops with a text-file round trip between them:
//   gentxtcord      camera_to_world.py:67-83   Z=depth[j,i]; X=(i-cx)/fx*Z; Y=(j-cy)/fy*Z
//   get_pointdata   camera_to_world.py:86-105  p_world = Rinv . (p_cam - t)   (point_camera, :57-59)
//
// Roofline: HBM.  Algorithmic traffic 13 B/point for u8 depth + f32 xyz (1 read, 12 written);
// 14 / 16 B for u16 / f32 depth; +12 B/point with f64 output.
//
// Layout and mapping
//   * depth is [F][H][W] contiguous, output is [F*H*W][3] AoS (12 B/point, not a power of two).
//   * a TILE is 1024 consecutive pixels of one frame = one 256-thread workgroup, 4 pixels per
//     lane: one 4-byte (u8) / 8-byte (u16) / 16-byte (f32) load per lane, 256 B..1 KiB per
//     wave instruction, fully coalesced.
//   * the lane's 4 points (48 B) are staged through LDS (12 KiB per workgroup) and leave as
//     3 x 16-B stores per lane at consecutive 16-B slots: every wave store instruction writes
//     1 KiB contiguous instead of 64 16-B pieces 48 B apart.  (variant 3; variant 2 stores the
//     48 B directly, variant 1 is the scalar any-width path.)
//   * the grid is capped (persistent-style) and strides over tiles; tile -> (frame, tile in
//     frame) is advanced incrementally on the scalar unit, the per-frame pose (96 B) comes in
//     through scalar loads, and the per-lane row/column split is one mulhi (magic division).
//   * arithmetic: fp64 registers, the reference's evaluation order, -ffp-contract=off, one
//     rounding to f32 on store.  fp64 VALU is ~4 cycles per wave instruction; ~19 of them per
//     pixel keep the vector unit ~25 % busy at the HBM rate, so the kernel stays bandwidth-bound.
#include <type_traits>

#include "r3d_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPx = 4;                      // pixels per lane
constexpr int kTile = kThreads * kPx;       // pixels per workgroup tile

struct FuseArgs {
  const void* depth;
  void* out;
  const double* u;     // [W] padded to x4
  const double* v;     // [H]
  const double* pose;  // [F][12] or nullptr (unproject only)
  double scale;
  uint32_t hw;              // H*W
  uint32_t width;
  uint32_t tiles_per_frame; // ceil(hw / kTile)
  uint32_t n_frames;
  uint32_t w_magic;         // floor(x / width) = (x * w_magic) >> w_shift for x < 2^31 (make_magic)
  uint32_t w_shift;
};

// ---- depth loads: 4 consecutive rasters elements -> 4 doubles ----
template <typename DT>
struct Depth4;
template <>
struct Depth4<uint8_t> {
  static __device__ __forceinline__ void load(const void* base, uint64_t idx, double z[4]) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(base) + idx);
    z[0] = (double)(w & 0xffu);
    z[1] = (double)((w >> 8) & 0xffu);
    z[2] = (double)((w >> 16) & 0xffu);
    z[3] = (double)(w >> 24);
  }
};
template <>
struct Depth4<uint16_t> {
  static __device__ __forceinline__ void load(const void* base, uint64_t idx, double z[4]) {
    const uint2 w = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(base) + idx);
    z[0] = (double)(w.x & 0xffffu);
    z[1] = (double)(w.x >> 16);
    z[2] = (double)(w.y & 0xffffu);
    z[3] = (double)(w.y >> 16);
  }
};
template <>
struct Depth4<float> {
  static __device__ __forceinline__ void load(const void* base, uint64_t idx, double z[4]) {
    const float4 w = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx);
    z[0] = (double)w.x;
    z[1] = (double)w.y;
    z[2] = (double)w.z;
    z[3] = (double)w.w;
  }
};

struct Pose {
  double r[9];
  double t[3];
};

// The reference's arithmetic for one pixel, in its evaluation order, fp64.
template <bool POSE>
__device__ __forceinline__ void point(double z, double u, double v, const Pose& p, double o[3]) {
  const double x = u * z;  // (i-cx)/fx * Z      c2w:78
  const double y = v * z;  // (j-cy)/fy * Z      c2w:79
  if (POSE) {
    const double dx = x - p.t[0], dy = y - p.t[1], dz = z - p.t[2];  // p1 - t        c2w:58
    o[0] = fma(p.r[2], dz, fma(p.r[1], dy, p.r[0] * dx));            // Rinv . (p1-t) c2w:58
    o[1] = fma(p.r[5], dz, fma(p.r[4], dy, p.r[3] * dx));
    o[2] = fma(p.r[8], dz, fma(p.r[7], dy, p.r[6] * dx));
  } else {
    o[0] = x;
    o[1] = y;
    o[2] = z;
  }
}

// 16-byte pieces as native clang vectors (the nontemporal builtin takes these, not HIP's structs)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 piece(const float* o, int k) {
  return f32x4{o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]};
}
__device__ __forceinline__ f64x2 piece(const double* o, int k) { return f64x2{o[2 * k], o[2 * k + 1]}; }

template <typename T>
__device__ __forceinline__ void store16(void* dst, const T& v, bool nt) {
  if (nt)
    __builtin_nontemporal_store(v, reinterpret_cast<T*>(dst));
  else
    *reinterpret_cast<T*>(dst) = v;
}

// VARIANT 1: scalar any-width; 2: vec4 loads + direct 48-B stores; 3: vec4 loads + LDS-transposed stores
template <typename DT, typename OT, bool POSE, int VARIANT, bool NT>
__global__ __launch_bounds__(kThreads) void fuse_kernel(const FuseArgs a) {
  constexpr int kVecPerLane = (int)(kPx * 3 * sizeof(OT) / 16);  // 16-B pieces per lane: 3 (f32) or 6 (f64)
  __shared__ __attribute__((aligned(16))) OT lds[VARIANT == 3 ? kTile * 3 : 4];

  const uint32_t tid = threadIdx.x;
  // tile walk: frame / tile-in-frame advance incrementally, all wave-uniform (scalar unit)
  uint32_t tf = blockIdx.x;
  uint32_t frame = 0;
  while (tf >= a.tiles_per_frame) {
    tf -= a.tiles_per_frame;
    ++frame;
  }
  while (frame < a.n_frames) {
    Pose P;
    if (POSE) {
      const double* pp = a.pose + (uint64_t)frame * 12;
#pragma unroll
      for (int k = 0; k < 9; ++k) P.r[k] = pp[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) P.t[k] = pp[9 + k];
    }
    const uint32_t p0 = tf * kTile + tid * kPx;              // first pixel of this lane within the frame
    const uint64_t g0 = (uint64_t)frame * a.hw + p0;         // ... within the batch
    OT o[kPx * 3];

    if (VARIANT == 1) {
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
        const uint32_t p = p0 + k;
        if (p < a.hw) {
          const uint32_t j = (uint32_t)(((uint64_t)p * a.w_magic) >> a.w_shift);
          const uint32_t i = p - j * a.width;
          const double z = (double)static_cast<const DT*>(a.depth)[g0 + k] * a.scale;
          double w[3];
          point<POSE>(z, a.u[i], a.v[j], P, w);
          OT* dst = static_cast<OT*>(a.out) + (g0 + k) * 3;
          dst[0] = (OT)w[0];
          dst[1] = (OT)w[1];
          dst[2] = (OT)w[2];
        }
      }
    } else {
      const bool live = p0 < a.hw;  // hw % 4 == 0 on this path: a lane is wholly in or out
      if (live) {
        const uint32_t j = (uint32_t)(((uint64_t)p0 * a.w_magic) >> a.w_shift);
        const uint32_t i = p0 - j * a.width;  // width % 4 == 0: the 4 pixels share row j
        double z[4];
        Depth4<DT>::load(a.depth, g0, z);
        const double2 u01 = *reinterpret_cast<const double2*>(a.u + i);
        const double2 u23 = *reinterpret_cast<const double2*>(a.u + i + 2);
        const double vj = a.v[j];
        const double uu[4] = {u01.x, u01.y, u23.x, u23.y};
#pragma unroll
        for (int k = 0; k < kPx; ++k) {
          double w[3];
          point<POSE>(z[k] * a.scale, uu[k], vj, P, w);
          o[3 * k + 0] = (OT)w[0];
          o[3 * k + 1] = (OT)w[1];
          o[3 * k + 2] = (OT)w[2];
        }
      }
      using V16 = typename std::conditional<sizeof(OT) == 4, f32x4, f64x2>::type;
      if (VARIANT == 2) {
        if (live) {
          char* dst = static_cast<char*>(a.out) + g0 * (3 * sizeof(OT));
#pragma unroll
          for (int k = 0; k < kVecPerLane; ++k)
            store16<V16>(dst + 16 * k, piece(o, k), NT);
        }
      } else {  // VARIANT 3
        if (live) {
          V16* mine = reinterpret_cast<V16*>(lds) + tid * kVecPerLane;
#pragma unroll
          for (int k = 0; k < kVecPerLane; ++k) mine[k] = piece(o, k);
        }
        __syncthreads();
        // pieces of 16 B, tile-linear: piece q holds bytes [16q, 16q+16) of the tile's output
        const uint32_t px_in_tile = min((uint32_t)kTile, a.hw - tf * kTile);
        const uint32_t n_pieces = px_in_tile * (uint32_t)(3 * sizeof(OT) / 4) / 4;  // px*3*sizeof/16
        char* tile_out = static_cast<char*>(a.out) + ((uint64_t)frame * a.hw + (uint64_t)tf * kTile) * (3 * sizeof(OT));
#pragma unroll
        for (int k = 0; k < kVecPerLane; ++k) {
          const uint32_t q = k * kThreads + tid;
          if (q < n_pieces) store16<V16>(tile_out + (uint64_t)q * 16, reinterpret_cast<const V16*>(lds)[q], NT);
        }
        __syncthreads();
      }
    }
    // next tile of this workgroup
    tf += gridDim.x;
    while (tf >= a.tiles_per_frame) {
      tf -= a.tiles_per_frame;
      ++frame;
    }
  }
}

template <typename DT, typename OT, bool POSE, int VARIANT>
void launch_nt(const FuseArgs& a, int blocks, bool nt, hipStream_t s) {
  if (nt)
    hipLaunchKernelGGL((fuse_kernel<DT, OT, POSE, VARIANT, true>), dim3(blocks), dim3(kThreads), 0, s, a);
  else
    hipLaunchKernelGGL((fuse_kernel<DT, OT, POSE, VARIANT, false>), dim3(blocks), dim3(kThreads), 0, s, a);
}

template <typename DT, typename OT, bool POSE>
void launch_variant(const FuseArgs& a, int variant, int blocks, bool nt, hipStream_t s) {
  switch (variant) {
    case 1: launch_nt<DT, OT, POSE, 1>(a, blocks, false, s); break;
    case 2: launch_nt<DT, OT, POSE, 2>(a, blocks, nt, s); break;
    default: launch_nt<DT, OT, POSE, 3>(a, blocks, nt, s); break;
  }
}

template <typename DT, bool POSE>
void launch_out(const FuseArgs& a, int out_dtype, int variant, int blocks, bool nt, hipStream_t s) {
  if (out_dtype == R3D_F32)
    launch_variant<DT, float, POSE>(a, variant, blocks, nt, s);
  else
    launch_variant<DT, double, POSE>(a, variant, blocks, nt, s);
}

template <bool POSE>
void launch_depth(const FuseArgs& a, int depth_dtype, int out_dtype, int variant, int blocks, bool nt,
                  hipStream_t s) {
  switch (depth_dtype) {
    case R3D_DEPTH_U8: launch_out<uint8_t, POSE>(a, out_dtype, variant, blocks, nt, s); break;
    case R3D_DEPTH_U16: launch_out<uint16_t, POSE>(a, out_dtype, variant, blocks, nt, s); break;
    default: launch_out<float, POSE>(a, out_dtype, variant, blocks, nt, s); break;
  }
}

// Magic number for floor(x / d), exact for every x < 2^31 and d >= 1 (round-up method):
//   s = ceil(log2 d), m = floor(2^(31+s) / d) + 1 (< 2^32), x / d = (x * m) >> (31 + s).
// m*d - 2^(31+s) lies in (0, d] <= 2^s, which is the exactness condition for 31-bit x.
void make_magic(uint32_t d, uint32_t* magic, uint32_t* shift) {
  uint32_t s = 0;
  while (((uint64_t)1 << s) < d) ++s;
  *magic = (uint32_t)((((uint64_t)1 << (31 + s)) / d) + 1);
  *shift = 31 + s;
}

int fuse_common(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                double depth_scale, const double* d_pose, bool with_pose, void* d_out, int out_dtype) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(cam != nullptr, "camera is NULL");
  R3D_REQUIRE(cam->ctx == ctx, "camera belongs to a different ctx");
  R3D_REQUIRE(depth_dtype >= R3D_DEPTH_U8 && depth_dtype <= R3D_DEPTH_F32, "unknown depth dtype %d", depth_dtype);
  R3D_REQUIRE(out_dtype == R3D_F32 || out_dtype == R3D_F64, "unknown output dtype %d", out_dtype);
  R3D_REQUIRE(n_frames >= 0, "n_frames must be >= 0");
  if (n_frames == 0) return R3D_OK;
  R3D_REQUIRE(d_depth && d_out, "NULL device pointer");
  R3D_REQUIRE(!with_pose || d_pose, "pose table is NULL");
  const uint64_t hw = (uint64_t)cam->height * cam->width;
  FuseArgs a;
  a.depth = d_depth;
  a.out = d_out;
  a.u = cam->d_u;
  a.v = cam->d_v;
  a.pose = with_pose ? d_pose : nullptr;
  a.scale = depth_scale;
  a.hw = (uint32_t)hw;
  a.width = (uint32_t)cam->width;
  a.tiles_per_frame = (uint32_t)((hw + kTile - 1) / kTile);
  a.n_frames = (uint32_t)n_frames;
  make_magic(a.width, &a.w_magic, &a.w_shift);
  const uint64_t total_tiles = (uint64_t)a.tiles_per_frame * n_frames;

  // vector paths need the 4 pixels of a lane in one row and 16-B aligned output pieces
  const size_t dsz = r3d_depth_size(depth_dtype);
  const bool vec_ok = (cam->width % 4 == 0) && (((uintptr_t)d_depth % (4 * dsz)) == 0) && (((uintptr_t)d_out % 16) == 0);
  int variant = ctx->fuse_variant;
  if (variant == 0) variant = 3;
  if (!vec_ok) variant = 1;

  int blocks = ctx->fuse_blocks > 0 ? ctx->fuse_blocks : ctx->num_cus * 8;
  if ((uint64_t)blocks > total_tiles) blocks = (int)total_tiles;
  if (with_pose)
    launch_depth<true>(a, depth_dtype, out_dtype, variant, blocks, ctx->nontemporal != 0, ctx->stream);
  else
    launch_depth<false>(a, depth_dtype, out_dtype, variant, blocks, ctx->nontemporal != 0, ctx->stream);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int fuse_host_common(r3d_ctx* ctx, const r3d_camera* cam, const void* h_depth, int depth_dtype, int n_frames,
                     double depth_scale, const double* h_pose, bool with_pose, void* h_out, int out_dtype) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(cam != nullptr, "camera is NULL");
  R3D_REQUIRE(depth_dtype >= R3D_DEPTH_U8 && depth_dtype <= R3D_DEPTH_F32, "unknown depth dtype %d", depth_dtype);
  R3D_REQUIRE(out_dtype == R3D_F32 || out_dtype == R3D_F64, "unknown output dtype %d", out_dtype);
  R3D_REQUIRE(n_frames >= 0, "n_frames must be >= 0");
  if (n_frames == 0) return R3D_OK;
  R3D_REQUIRE(h_depth && h_out, "NULL host pointer");
  R3D_REQUIRE(!with_pose || h_pose, "pose table is NULL");
  const size_t n = (size_t)cam->height * cam->width * n_frames;
  const size_t in_bytes = n * r3d_depth_size(depth_dtype);
  const size_t out_bytes = n * 3 * r3d_xyz_size(out_dtype);
  void *d_in = nullptr, *d_out = nullptr, *d_pose = nullptr;
  if ((rc = r3d_scratch(ctx, 0, in_bytes, &d_in))) return rc;
  if ((rc = r3d_scratch(ctx, 1, out_bytes, &d_out))) return rc;
  R3D_HIP(hipMemcpyAsync(d_in, h_depth, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (with_pose) {
    if ((rc = r3d_scratch(ctx, 2, (size_t)n_frames * 12 * sizeof(double), &d_pose))) return rc;
    R3D_HIP(hipMemcpyAsync(d_pose, h_pose, (size_t)n_frames * 12 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  rc = fuse_common(ctx, cam, d_in, depth_dtype, n_frames, depth_scale, (const double*)d_pose, with_pose, d_out,
                   out_dtype);
  if (rc) return rc;
  R3D_HIP(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  return R3D_OK;
}

}  // namespace

extern "C" {

int r3d_unproject(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                  double depth_scale, void* d_xyz_out, int out_dtype) {
  return fuse_common(ctx, cam, d_depth, depth_dtype, n_frames, depth_scale, nullptr, false, d_xyz_out, out_dtype);
}

int r3d_unproject_host(r3d_ctx* ctx, const r3d_camera* cam, const void* h_depth, int depth_dtype, int n_frames,
                       double depth_scale, void* h_xyz_out, int out_dtype) {
  return fuse_host_common(ctx, cam, h_depth, depth_dtype, n_frames, depth_scale, nullptr, false, h_xyz_out,
                          out_dtype);
}

int r3d_fuse_frames(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                    double depth_scale, const double* d_pose, void* d_xyz_out, int out_dtype) {
  return fuse_common(ctx, cam, d_depth, depth_dtype, n_frames, depth_scale, d_pose, true, d_xyz_out, out_dtype);
}

int r3d_fuse_frames_host(r3d_ctx* ctx, const r3d_camera* cam, const void* h_depth, int depth_dtype, int n_frames,
                         double depth_scale, const double* h_pose, void* h_xyz_out, int out_dtype) {
  return fuse_host_common(ctx, cam, h_depth, depth_dtype, n_frames, depth_scale, h_pose, true, h_xyz_out,
                          out_dtype);
}

}  // extern "C"
