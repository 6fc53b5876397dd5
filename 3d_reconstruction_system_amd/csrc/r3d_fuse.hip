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
// Layout and mapping (default kernel, "variant 5"; the others are kept selectable for A/B, all bit-identical)
//   * depth is [F][H][W] contiguous, output is [F*H*W][3] AoS (12 B/point, not a power of two).
//   * a TILE is 1024 consecutive pixels of one frame = one 256-thread workgroup.  Lane `tid` takes pixels
//     tid, tid+256, tid+512, tid+768, so in every round a wave holds 64 CONSECUTIVE pixels: the read is one
//     coalesced element per lane, the write ONE 12-byte nontemporal store per lane at a 12-byte lane stride =
//     768 contiguous bytes per wave instruction.  No LDS, no barrier, ~30 VGPRs.
//   * the grid is capped at 8 workgroups per CU and strides over tiles; tile -> (frame, tile in frame) and
//     pixel -> (row, column) are magic-number divisions (host-computed), the per-frame pose (96 B) comes in
//     through scalar loads (wave-uniform address, const __restrict__).
//   * arithmetic: fp64 registers, the reference's evaluation order, -ffp-contract=off, one rounding on store.
//     ~19 fp64 instructions per pixel keep the SIMDs ~40 % busy at the HBM rate; the kernel sits on the store
//     stream (within 5-8 % of hipMemset for the same bytes; profiles/variants_r01.md).
//   * variants: 1 scalar any-width (also the fallback for widths not divisible by 4 in variants 2-4);
//     2 four pixels per lane, direct 48-byte stores; 3 four pixels per lane, LDS-transposed 16-byte stores per
//     workgroup (default for f64 xyz, one tile per workgroup); 4 the same per wave, no barrier; 5 lane-per-pixel
//     (default for f32 xyz); 6 lane-per-pixel with all loads batched; 7 lane-per-pixel with scalar tile bases.
#include <type_traits>

#include "r3d_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPx = 4;                      // pixels per lane
constexpr int kTile = kThreads * kPx;       // pixels per workgroup tile

struct FuseDims {
  double scale;
  uint32_t hw;              // H*W
  uint32_t width;
  uint32_t tiles_per_frame; // ceil(hw / tile), tile = 1024 px (variants 1-3) or 256 px (variant 4)
  uint32_t n_frames;
  uint32_t w_magic;         // floor(x / width) = (x * w_magic) >> w_shift for x < 2^31 (make_magic)
  uint32_t w_shift;
  uint32_t w_shift32;       // w_shift - 32: j = umulhi(p, w_magic) >> w_shift32 when width >= 2
  uint32_t width_is_one;    // width == 1 has w_shift == 31: every pixel is its own row
  uint32_t t_magic;         // floor(tile / tiles_per_frame), same scheme
  uint32_t t_shift;
  uint32_t total_tiles;     // tiles_per_frame * n_frames (< 2^31)
};

__device__ __forceinline__ uint32_t magic_div(uint32_t x, uint32_t magic, uint32_t shift) {
  return (uint32_t)(((uint64_t)x * magic) >> shift);
}

// ---- depth loads: 4 consecutive rasters elements -> 4 doubles ----
template <typename DT>
struct Depth4;
template <>
struct Depth4<uint8_t> {
  using raw_t = uint32_t;
  static __device__ __forceinline__ raw_t fetch(const uint8_t* base, uint64_t idx) {
    return *reinterpret_cast<const uint32_t*>(base + idx);
  }
  static __device__ __forceinline__ void unpack(const raw_t& w, double z[4]) {
    z[0] = (double)(w & 0xffu);
    z[1] = (double)((w >> 8) & 0xffu);
    z[2] = (double)((w >> 16) & 0xffu);
    z[3] = (double)(w >> 24);
  }
};
template <>
struct Depth4<uint16_t> {
  using raw_t = uint2;
  static __device__ __forceinline__ raw_t fetch(const uint16_t* base, uint64_t idx) {
    return *reinterpret_cast<const uint2*>(base + idx);
  }
  static __device__ __forceinline__ void unpack(const raw_t& w, double z[4]) {
    z[0] = (double)(w.x & 0xffffu);
    z[1] = (double)(w.x >> 16);
    z[2] = (double)(w.y & 0xffffu);
    z[3] = (double)(w.y >> 16);
  }
};
template <>
struct Depth4<float> {
  using raw_t = float4;
  static __device__ __forceinline__ raw_t fetch(const float* base, uint64_t idx) {
    return *reinterpret_cast<const float4*>(base + idx);
  }
  static __device__ __forceinline__ void unpack(const raw_t& w, double z[4]) {
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

// Everything the four variants share for one lane's 4 pixels that lie in one row.
template <typename DT, typename OT, bool POSE>
__device__ __forceinline__ void quad(const typename Depth4<DT>::raw_t& raw, uint32_t p0, const double* __restrict__ u,
                                     const double* __restrict__ v, const FuseDims& dm, const Pose& P, OT o[12]) {
  const uint32_t j = magic_div(p0, dm.w_magic, dm.w_shift);
  const uint32_t i = p0 - j * dm.width;  // width % 4 == 0: the 4 pixels share row j
  double z[4];
  Depth4<DT>::unpack(raw, z);
  const double2 u01 = *reinterpret_cast<const double2*>(u + i);
  const double2 u23 = *reinterpret_cast<const double2*>(u + i + 2);
  const double vj = v[j];
  const double uu[4] = {u01.x, u01.y, u23.x, u23.y};
#pragma unroll
  for (int k = 0; k < kPx; ++k) {
    double w[3];
    point<POSE>(z[k] * dm.scale, uu[k], vj, P, w);
    o[3 * k + 0] = (OT)w[0];
    o[3 * k + 1] = (OT)w[1];
    o[3 * k + 2] = (OT)w[2];
  }
}

template <bool POSE>
__device__ __forceinline__ void load_pose(const double* __restrict__ pose, uint32_t frame, Pose& P) {
  if (POSE) {
    const double* pp = pose + (uint64_t)frame * 12;  // wave-uniform address: scalar loads
#pragma unroll
    for (int k = 0; k < 9; ++k) P.r[k] = pp[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) P.t[k] = pp[9 + k];
  }
}

// VARIANT 1: scalar any-width; 2: vec4 loads + direct 48-B stores; 3: vec4 loads + LDS-transposed stores,
// 1024-px workgroup tiles, two barriers per tile.
template <typename DT, typename OT, bool POSE, int VARIANT, bool NT>
__global__ __launch_bounds__(kThreads) void fuse_kernel(const DT* __restrict__ depth, OT* __restrict__ out,
                                                        const double* __restrict__ u, const double* __restrict__ v,
                                                        const double* __restrict__ pose, const FuseDims dm) {
  constexpr int kVecPerLane = (int)(kPx * 3 * sizeof(OT) / 16);  // 16-B pieces per lane: 3 (f32) or 6 (f64)
  using V16 = typename std::conditional<sizeof(OT) == 4, f32x4, f64x2>::type;
  __shared__ __attribute__((aligned(16))) OT lds[VARIANT == 3 ? kTile * 3 : 4];

  const uint32_t tid = threadIdx.x;
  // tile walk: tile -> (frame, tile in frame) by magic division, all wave-uniform (scalar unit)
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose<POSE>(pose, frame, P);
    const uint32_t p0 = tf * kTile + tid * kPx;              // first pixel of this lane within the frame
    const uint64_t g0 = (uint64_t)frame * dm.hw + p0;        // ... within the batch
    OT o[kPx * 3];

    if (VARIANT == 1) {
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
        const uint32_t p = p0 + k;
        if (p < dm.hw) {
          const uint32_t j = magic_div(p, dm.w_magic, dm.w_shift);
          const uint32_t i = p - j * dm.width;
          const double z = (double)depth[g0 + k] * dm.scale;
          double w[3];
          point<POSE>(z, u[i], v[j], P, w);
          OT* dst = out + (g0 + k) * 3;
          dst[0] = (OT)w[0];
          dst[1] = (OT)w[1];
          dst[2] = (OT)w[2];
        }
      }
    } else {
      const bool live = p0 < dm.hw;  // hw % 4 == 0 on this path: a lane is wholly in or out
      if (live) quad<DT, OT, POSE>(Depth4<DT>::fetch(depth, g0), p0, u, v, dm, P, o);
      if (VARIANT == 2) {
        if (live) {
          char* dst = reinterpret_cast<char*>(out) + g0 * (3 * sizeof(OT));
#pragma unroll
          for (int k = 0; k < kVecPerLane; ++k) store16<V16>(dst + 16 * k, piece(o, k), NT);
        }
      } else {  // VARIANT 3
        if (live) {
          V16* mine = reinterpret_cast<V16*>(lds) + tid * kVecPerLane;
#pragma unroll
          for (int k = 0; k < kVecPerLane; ++k) mine[k] = piece(o, k);
        }
        __syncthreads();
        // pieces of 16 B, tile-linear: piece q holds bytes [16q, 16q+16) of the tile's output
        const uint32_t px_in_tile = min((uint32_t)kTile, dm.hw - tf * kTile);
        const uint32_t n_pieces = px_in_tile * (uint32_t)(3 * sizeof(OT) / 4) / 4;  // px*3*sizeof/16
        char* tile_out = reinterpret_cast<char*>(out) + ((uint64_t)frame * dm.hw + (uint64_t)tf * kTile) * (3 * sizeof(OT));
#pragma unroll
        for (int k = 0; k < kVecPerLane; ++k) {
          const uint32_t q = k * kThreads + tid;
          if (q < n_pieces) store16<V16>(tile_out + (uint64_t)q * 16, reinterpret_cast<const V16*>(lds)[q], NT);
        }
        __syncthreads();
      }
    }
  }
}

// VARIANT 4: every WAVE is autonomous.  Its tile is 256 consecutive pixels (4 per lane); the 3 KiB of
// xyz it produces go through the wave's private LDS slice and leave as 3 x 1 KiB contiguous stores.
// No workgroup barrier anywhere (same-wave LDS ops complete in order), and the depth word of the
// wave's NEXT tile is already in flight while the current tile is computed and stored.
constexpr int kWaveTile = 64 * kPx;

template <typename DT, typename OT, bool POSE, bool NT>
__global__ __launch_bounds__(kThreads) void fuse_wave_kernel(const DT* __restrict__ depth, OT* __restrict__ out,
                                                             const double* __restrict__ u, const double* __restrict__ v,
                                                             const double* __restrict__ pose, const FuseDims dm) {
  constexpr int kVecPerLane = (int)(kPx * 3 * sizeof(OT) / 16);
  using V16 = typename std::conditional<sizeof(OT) == 4, f32x4, f64x2>::type;
  using Raw = typename Depth4<DT>::raw_t;
  __shared__ __attribute__((aligned(16))) OT lds_all[kThreads / 64][kWaveTile * 3];

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  OT* lds = lds_all[wave];
  const uint32_t stride = gridDim.x * (kThreads / 64);
  uint32_t tile = blockIdx.x * (kThreads / 64) + wave;
  Raw raw_next = {};
  uint32_t frame = 0, tf = 0;
  if (tile < dm.total_tiles) {
    frame = magic_div(tile, dm.t_magic, dm.t_shift);
    tf = tile - frame * dm.tiles_per_frame;
    const uint32_t p0 = tf * kWaveTile + lane * kPx;
    if (p0 < dm.hw) raw_next = Depth4<DT>::fetch(depth, (uint64_t)frame * dm.hw + p0);
  }
  while (tile < dm.total_tiles) {
    const Raw raw = raw_next;
    const uint32_t cur_tf = tf, cur_frame = frame;
    // advance, and put the next tile's depth load in flight before touching the current one
    tile += stride;
    if (tile < dm.total_tiles) {
      frame = magic_div(tile, dm.t_magic, dm.t_shift);
      tf = tile - frame * dm.tiles_per_frame;
      const uint32_t pn = tf * kWaveTile + lane * kPx;
      if (pn < dm.hw) raw_next = Depth4<DT>::fetch(depth, (uint64_t)frame * dm.hw + pn);
    }
    Pose P;
    load_pose<POSE>(pose, cur_frame, P);
    const uint32_t p0 = cur_tf * kWaveTile + lane * kPx;
    OT o[kPx * 3];
    if (p0 < dm.hw) {
      quad<DT, OT, POSE>(raw, p0, u, v, dm, P, o);
      V16* mine = reinterpret_cast<V16*>(lds) + lane * kVecPerLane;
#pragma unroll
      for (int k = 0; k < kVecPerLane; ++k) mine[k] = piece(o, k);
    }
    // same-wave LDS hand-off: writes above are ordered before the reads below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t px_in_tile = min((uint32_t)kWaveTile, dm.hw - cur_tf * kWaveTile);
    const uint32_t n_pieces = px_in_tile * (uint32_t)(3 * sizeof(OT) / 4) / 4;
    char* tile_out = reinterpret_cast<char*>(out) + ((uint64_t)cur_frame * dm.hw + (uint64_t)cur_tf * kWaveTile) * (3 * sizeof(OT));
#pragma unroll
    for (int k = 0; k < kVecPerLane; ++k) {
      const uint32_t q = k * 64 + lane;
      if (q < n_pieces) store16<V16>(tile_out + (uint64_t)q * 16, reinterpret_cast<const V16*>(lds)[q], NT);
    }
    // the next iteration's LDS writes must not pass this iteration's reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// VARIANT 5: lane-per-pixel rounds, no LDS.  A workgroup tile is still 1024 consecutive pixels, but lane
// `tid` takes pixels tid, tid+256, tid+512, tid+768 of it, so in every round the 64 lanes of a wave hold
// 64 CONSECUTIVE pixels: the depth read is one coalesced element per lane and the xyz write is one
// 12-byte (f32) or 24-byte (f64) store per lane at a 12/24-byte lane stride = 768 / 1536 contiguous
// bytes per wave instruction.  Nothing is shared between lanes, so any raster width works.
// store modes of the lane-per-pixel kernel: 0 one x3 store, 1 three nontemporal scalar stores,
// 2 three plain scalar stores, 3 one nontemporal x3 store
template <typename T>
struct Xyz3;
template <>
struct Xyz3<float> {
  typedef float v3 __attribute__((ext_vector_type(3)));
  template <int MODE>
  static __device__ __forceinline__ void store(float* dst, const double w[3]) {
    const float a = (float)w[0], b = (float)w[1], c = (float)w[2];
    if (MODE == 1) {
      __builtin_nontemporal_store(a, dst);
      __builtin_nontemporal_store(b, dst + 1);
      __builtin_nontemporal_store(c, dst + 2);
    } else if (MODE == 2) {
      dst[0] = a;
      dst[1] = b;
      dst[2] = c;
    } else if (MODE == 3) {
      asm volatile("global_store_dwordx3 %0, %1, off nt" ::"v"(dst), "v"(v3{a, b, c}) : "memory");
    } else {
      // one global_store_dwordx3 (4-byte alignment suffices on gfx950)
      asm volatile("global_store_dwordx3 %0, %1, off" ::"v"(dst), "v"(v3{a, b, c}) : "memory");
    }
  }
};
template <>
struct Xyz3<double> {
  template <int MODE>
  static __device__ __forceinline__ void store(double* dst, const double w[3]) {
    if (MODE == 1 || MODE == 3) {
      __builtin_nontemporal_store(w[0], dst);
      __builtin_nontemporal_store(w[1], dst + 1);
      __builtin_nontemporal_store(w[2], dst + 2);
    } else {
      dst[0] = w[0];
      dst[1] = w[1];
      dst[2] = w[2];
    }
  }
};

// VARIANT 6 (A/B): one tile of the lane-per-pixel kernel with batched loads.  WHOLE = all 1024 pixels exist: straight-line code with every load of
// the tile (4 depth elements, 4 u, 4 v) in flight before the first use.  Otherwise each pixel is predicated.
// SCALE1: depth_scale == 1.0 (the reference's case) skips the multiply -- x*1.0 is exact, so results are identical.
template <typename DT, typename OT, bool POSE, int MODE, bool SCALE1, bool WHOLE>
__device__ __forceinline__ void lane_tile(const DT* __restrict__ dptr, OT* __restrict__ optr, const double* __restrict__ u,
                                          const double* __restrict__ v, const Pose& P, const FuseDims& dm, uint32_t px0,
                                          uint32_t tid) {
  DT raw[kPx];
  double uu[kPx], vv[kPx];
  bool live[kPx];
#pragma unroll
  for (int r = 0; r < kPx; ++r) {
    const uint32_t l = r * kThreads + tid;  // pixel within the tile: a 32-bit lane offset from a scalar base
    const uint32_t p = px0 + l;
    live[r] = WHOLE || p < dm.hw;
    const uint32_t pc = live[r] ? p : px0;  // clamp: dead lanes read a valid element and store nothing
    const uint32_t j = dm.width_is_one ? pc : (__umulhi(pc, dm.w_magic) >> dm.w_shift32);
    const uint32_t i = pc - j * dm.width;
    raw[r] = dptr[pc - px0];
    uu[r] = u[i];
    vv[r] = v[j];
  }
#pragma unroll
  for (int r = 0; r < kPx; ++r) {
    const uint32_t l = r * kThreads + tid;
    double z = (double)raw[r];
    if (!SCALE1) z *= dm.scale;
    double w[3];
    point<POSE>(z, uu[r], vv[r], P, w);
    if (live[r]) Xyz3<OT>::template store<MODE>(optr + l * 3, w);
  }
}

template <typename DT, typename OT, bool POSE, int MODE, bool SCALE1>
__global__ __launch_bounds__(kThreads) void fuse_lane_batched_kernel(const DT* __restrict__ depth, OT* __restrict__ out,
                                                             const double* __restrict__ u, const double* __restrict__ v,
                                                             const double* __restrict__ pose, const FuseDims dm) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    // wave-uniform part (scalar unit): which frame, where the tile starts, its base addresses
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    const uint32_t px0 = tf * kTile;
    const uint64_t gbase = (uint64_t)frame * dm.hw + px0;
    const DT* __restrict__ dptr = depth + gbase;
    OT* __restrict__ optr = out + gbase * 3;
    Pose P;
    load_pose<POSE>(pose, frame, P);
    if (px0 + kTile <= dm.hw)
      lane_tile<DT, OT, POSE, MODE, SCALE1, true>(dptr, optr, u, v, P, dm, px0, tid);
    else
      lane_tile<DT, OT, POSE, MODE, SCALE1, false>(dptr, optr, u, v, P, dm, px0, tid);
  }
}

// VARIANT 5 (default for f32 xyz): lane-per-pixel rounds, per-pixel predicates.  Measured fastest of the three
// lane-per-pixel forms in interleaved same-process rounds (profiles/variants_r01.md): 0.093 ms vs 0.097 (scalar
// bases, variant 7) vs 0.101-0.106 (all 12 loads of a tile batched up front, variant 6) -- fewer instructions
// did NOT help; the store stream's cadence did.
template <typename DT, typename OT, bool POSE, int MODE>
__global__ __launch_bounds__(kThreads) void fuse_lane_kernel(const DT* __restrict__ depth, OT* __restrict__ out,
                                                                const double* __restrict__ u, const double* __restrict__ v,
                                                                const double* __restrict__ pose, const FuseDims dm) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose<POSE>(pose, frame, P);
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    DT raw[kPx];
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t p = tf * kTile + r * kThreads + tid;
      raw[r] = p < dm.hw ? depth[fbase + p] : DT(0);
    }
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t p = tf * kTile + r * kThreads + tid;
      if (p < dm.hw) {
        const uint32_t j = magic_div(p, dm.w_magic, dm.w_shift);
        const uint32_t i = p - j * dm.width;
        double w[3];
        point<POSE>((double)raw[r] * dm.scale, u[i], v[j], P, w);
        Xyz3<OT>::template store<MODE>(out + (fbase + p) * 3, w);
      }
    }
  }
}

// VARIANT 7 (A/B): variant 5's per-pixel structure with the cheap scalar/integer savings only --
// scalar tile bases + 32-bit lane offsets, umulhi row split, no scale multiply when depth_scale == 1.
template <typename DT, typename OT, bool POSE, int MODE, bool SCALE1>
__global__ __launch_bounds__(kThreads) void fuse_lane_sbase_kernel(const DT* __restrict__ depth, OT* __restrict__ out,
                                                                const double* __restrict__ u, const double* __restrict__ v,
                                                                const double* __restrict__ pose, const FuseDims dm) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    const uint32_t px0 = tf * kTile;
    const uint64_t gbase = (uint64_t)frame * dm.hw + px0;
    const DT* __restrict__ dptr = depth + gbase;
    OT* __restrict__ optr = out + gbase * 3;
    Pose P;
    load_pose<POSE>(pose, frame, P);
    DT raw[kPx];
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t l = r * kThreads + tid;
      raw[r] = px0 + l < dm.hw ? dptr[l] : DT(0);
    }
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t l = r * kThreads + tid;
      const uint32_t p = px0 + l;
      if (p < dm.hw) {
        const uint32_t j = dm.width_is_one ? p : (__umulhi(p, dm.w_magic) >> dm.w_shift32);
        const uint32_t i = p - j * dm.width;
        double z = (double)raw[r];
        if (!SCALE1) z *= dm.scale;
        double w[3];
        point<POSE>(z, u[i], v[j], P, w);
        Xyz3<OT>::template store<MODE>(optr + l * 3, w);
      }
    }
  }
}

struct FusePtrs {
  const void* depth;
  void* out;
  const double* u;
  const double* v;
  const double* pose;
};

template <typename DT, typename OT, bool POSE, int VARIANT, bool NT>
void launch_one(const FusePtrs& p, const FuseDims& dm, int blocks, hipStream_t s) {
  if (VARIANT == 4)
    hipLaunchKernelGGL((fuse_wave_kernel<DT, OT, POSE, NT>), dim3(blocks), dim3(kThreads), 0, s,
                       static_cast<const DT*>(p.depth), static_cast<OT*>(p.out), p.u, p.v, p.pose, dm);
  else
    hipLaunchKernelGGL((fuse_kernel<DT, OT, POSE, (VARIANT >= 4 ? 3 : VARIANT), NT>), dim3(blocks), dim3(kThreads), 0, s,
                       static_cast<const DT*>(p.depth), static_cast<OT*>(p.out), p.u, p.v, p.pose, dm);
}

template <typename DT, typename OT, bool POSE, int MODE>
void launch_lane(const FusePtrs& p, const FuseDims& dm, int blocks, hipStream_t s) {
  if (dm.scale == 1.0)
    hipLaunchKernelGGL((fuse_lane_batched_kernel<DT, OT, POSE, MODE, true>), dim3(blocks), dim3(kThreads), 0, s,
                       static_cast<const DT*>(p.depth), static_cast<OT*>(p.out), p.u, p.v, p.pose, dm);
  else
    hipLaunchKernelGGL((fuse_lane_batched_kernel<DT, OT, POSE, MODE, false>), dim3(blocks), dim3(kThreads), 0, s,
                       static_cast<const DT*>(p.depth), static_cast<OT*>(p.out), p.u, p.v, p.pose, dm);
}

template <typename DT, typename OT, bool POSE, int MODE>
void launch_lane_orig(const FusePtrs& p, const FuseDims& dm, int blocks, hipStream_t s) {
  hipLaunchKernelGGL((fuse_lane_kernel<DT, OT, POSE, MODE>), dim3(blocks), dim3(kThreads), 0, s,
                     static_cast<const DT*>(p.depth), static_cast<OT*>(p.out), p.u, p.v, p.pose, dm);
}

template <typename DT, typename OT, bool POSE>
void launch_variant(const FusePtrs& p, const FuseDims& dm, int variant, int blocks, int ntmode, hipStream_t s) {
  const bool nt = ntmode != 0;
  if (variant == 7) {
    if (dm.scale == 1.0)
      hipLaunchKernelGGL((fuse_lane_sbase_kernel<DT, OT, POSE, 3, true>), dim3(blocks), dim3(kThreads), 0, s,
                         static_cast<const DT*>(p.depth), static_cast<OT*>(p.out), p.u, p.v, p.pose, dm);
    else
      hipLaunchKernelGGL((fuse_lane_sbase_kernel<DT, OT, POSE, 3, false>), dim3(blocks), dim3(kThreads), 0, s,
                         static_cast<const DT*>(p.depth), static_cast<OT*>(p.out), p.u, p.v, p.pose, dm);
    return;
  }
  if (variant == 6) {
    launch_lane<DT, OT, POSE, 3>(p, dm, blocks, s);
    return;
  }
  if (variant == 5) {
    switch (ntmode) {
      case 0: launch_lane_orig<DT, OT, POSE, 0>(p, dm, blocks, s); break;
      case 2: launch_lane_orig<DT, OT, POSE, 2>(p, dm, blocks, s); break;
      case 1: launch_lane_orig<DT, OT, POSE, 1>(p, dm, blocks, s); break;
      default: launch_lane_orig<DT, OT, POSE, 3>(p, dm, blocks, s); break;
    }
    return;
  }
  switch (variant) {
    case 1: launch_one<DT, OT, POSE, 1, false>(p, dm, blocks, s); break;
    case 2: nt ? launch_one<DT, OT, POSE, 2, true>(p, dm, blocks, s) : launch_one<DT, OT, POSE, 2, false>(p, dm, blocks, s); break;
    case 3: nt ? launch_one<DT, OT, POSE, 3, true>(p, dm, blocks, s) : launch_one<DT, OT, POSE, 3, false>(p, dm, blocks, s); break;
    default: nt ? launch_one<DT, OT, POSE, 4, true>(p, dm, blocks, s) : launch_one<DT, OT, POSE, 4, false>(p, dm, blocks, s); break;
  }
}

template <typename DT, bool POSE>
void launch_out(const FusePtrs& p, const FuseDims& dm, int out_dtype, int variant, int blocks, int nt, hipStream_t s) {
  if (out_dtype == R3D_F32)
    launch_variant<DT, float, POSE>(p, dm, variant, blocks, nt, s);
  else
    launch_variant<DT, double, POSE>(p, dm, variant, blocks, nt, s);
}

template <bool POSE>
void launch_depth(const FusePtrs& p, const FuseDims& dm, int depth_dtype, int out_dtype, int variant, int blocks,
                  int nt, hipStream_t s) {
  switch (depth_dtype) {
    case R3D_DEPTH_U8: launch_out<uint8_t, POSE>(p, dm, out_dtype, variant, blocks, nt, s); break;
    case R3D_DEPTH_U16: launch_out<uint16_t, POSE>(p, dm, out_dtype, variant, blocks, nt, s); break;
    default: launch_out<float, POSE>(p, dm, out_dtype, variant, blocks, nt, s); break;
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
  // vector paths need the 4 pixels of a lane in one row and 16-B aligned output pieces
  const size_t dsz = r3d_depth_size(depth_dtype);
  const bool vec_ok = (cam->width % 4 == 0) && (((uintptr_t)d_depth % (4 * dsz)) == 0) && (((uintptr_t)d_out % 16) == 0);
  int variant = ctx->fuse_variant;
  int ntmode = ctx->nontemporal;
  bool one_tile_per_block = false;
  if (variant < 1 || variant > 7) {
    // auto, from the A/B in profiles/variants_r01.md: f32 xyz -> lane-per-pixel kernel with one nontemporal
    // 12-byte store per lane (6.6-7.0 TB/s); f64 xyz -> 24-byte lane stride does not combine, the
    // LDS-transposed 16-byte-store kernel at one tile per workgroup wins (5.4 vs 2.7 TB/s)
    if (out_dtype == R3D_F64 && vec_ok) {
      variant = 3;
      ntmode = 0;
      one_tile_per_block = true;
    } else {
      variant = 5;
      ntmode = 3;
    }
  }
  if (!vec_ok && variant < 5) variant = 1;
  const uint32_t tile = variant == 4 ? kWaveTile : kTile;
  FusePtrs p{d_depth, d_out, cam->d_u, cam->d_v, with_pose ? d_pose : nullptr};
  FuseDims dm;
  dm.scale = depth_scale;
  dm.hw = (uint32_t)hw;
  dm.width = (uint32_t)cam->width;
  dm.tiles_per_frame = (uint32_t)((hw + tile - 1) / tile);
  dm.n_frames = (uint32_t)n_frames;
  make_magic(dm.width, &dm.w_magic, &dm.w_shift);
  dm.width_is_one = dm.width == 1;
  dm.w_shift32 = dm.width_is_one ? 0 : dm.w_shift - 32;
  make_magic(dm.tiles_per_frame, &dm.t_magic, &dm.t_shift);
  const uint64_t total_tiles = (uint64_t)dm.tiles_per_frame * n_frames;
  R3D_REQUIRE(total_tiles < ((uint64_t)1 << 31), "batch too large for one launch (%llu tiles); split the frames",
              (unsigned long long)total_tiles);
  dm.total_tiles = (uint32_t)total_tiles;
  const uint64_t tiles_per_block = variant == 4 ? kThreads / 64 : 1;
  uint64_t blocks = ctx->fuse_blocks > 0 ? (uint64_t)ctx->fuse_blocks
                                         : one_tile_per_block ? ~(uint64_t)0 : (uint64_t)ctx->num_cus * 8;
  const uint64_t max_blocks = (total_tiles + tiles_per_block - 1) / tiles_per_block;
  if (blocks > max_blocks) blocks = max_blocks;
  if (with_pose)
    launch_depth<true>(p, dm, depth_dtype, out_dtype, variant, (int)blocks, ntmode, ctx->stream);
  else
    launch_depth<false>(p, dm, depth_dtype, out_dtype, variant, (int)blocks, ntmode, ctx->stream);
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
  if (with_pose) {
    if ((rc = r3d_scratch(ctx, 2, (size_t)n_frames * 12 * sizeof(double), &d_pose))) return rc;
    R3D_HIP(hipMemcpyAsync(d_pose, h_pose, (size_t)n_frames * 12 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  // frames stream through the pinned double-buffered pipeline, a few frames per chunk
  const size_t frame_in = (size_t)cam->height * cam->width * r3d_depth_size(depth_dtype);
  const size_t frame_out = (size_t)cam->height * cam->width * 3 * r3d_xyz_size(out_dtype);
  auto launch = [&](int64_t lo, int64_t n) -> int {
    return fuse_common(ctx, cam, static_cast<char*>(d_in) + (size_t)lo * frame_in, depth_dtype, (int)n, depth_scale,
                       with_pose ? static_cast<const double*>(d_pose) + (size_t)lo * 12 : nullptr, with_pose,
                       static_cast<char*>(d_out) + (size_t)lo * frame_out, out_dtype);
  };
  return r3d_host_pipeline(ctx, n_frames, frame_in, frame_out, h_depth, h_out, d_in, d_out, launch);
}

}  // namespace

extern "C" {

// Self-test hook: floor(x / d) through the same host-made magic number the kernels use (x < 2^31, d >= 1).
int r3d_selftest_magic_div(uint32_t d, uint32_t x, uint32_t* q_out) {
  if (d == 0 || x >= ((uint32_t)1 << 31) || !q_out) {
    r3d_set_error("r3d_selftest_magic_div: d must be >= 1, x < 2^31");
    return R3D_ERR_INVALID;
  }
  uint32_t m = 0, sh = 0;
  make_magic(d, &m, &sh);
  *q_out = (uint32_t)(((uint64_t)x * m) >> sh);
  if (d >= 2) {  // the umulhi form used by variants 6 and 7
    const uint32_t hi = (uint32_t)(((uint64_t)x * m) >> 32);
    if ((hi >> (sh - 32)) != *q_out) {
      r3d_set_error("magic forms disagree for d=%u x=%u", d, x);
      return R3D_ERR_HIP;
    }
  }
  return R3D_OK;
}

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
