// Fused depth-raster -> world-frame point kernels for gfx950 (MI355X).
//
// Replace, in ONE launch over a whole batch of frames, the reference's two per-point Python loops with a text-file
// round trip between them:
//   gentxtcord      camera_to_world.py:67-83   Z=depth[j,i]; X=(i-cx)/fx*Z; Y=(j-cy)/fy*Z
//   get_pointdata   camera_to_world.py:86-105  p_world = Rinv . (p_cam - t)   (point_camera, :57-59)
// and, with a colour plane, the per-point colour attach of genply_noRGB (pixel_to_camera.py:55-91).
//
// Roofline: HBM.  Algorithmic traffic 13 B/point for u8 depth + f32 xyz (1 read, 12 written); 14 / 16 B for u16 / f32
// depth; +12 B/point with f64 xyz; +7 B/point with colour (3 B rgb read, 4 B rgba written).
//
// Layout and mapping
//   * depth is [F][H][W] contiguous, output is [F*H*W][3] AoS (12 or 24 B/point), frame order = pose-file order.
//   * a TILE is 1024 consecutive pixels of one frame = one 256-thread workgroup; tile -> (frame, tile in frame) and
//     pixel -> (row, column) are magic-number divisions (host-computed); the per-frame pose (96 B) comes in through
//     scalar loads (wave-uniform address, const __restrict__).
//   * arithmetic: fp64 registers, the reference's evaluation order, -ffp-contract=off, one rounding on store.
//   * THE STORE SHAPE decides everything (A/B history: profiles/variants_r01.md, profiles/r02_ab_kernels.log): a wave
//     instruction must write one contiguous run of bytes, 12 B per lane at a 12-B lane stride, nontemporal
//     (`global_store_dwordx3 ... nt` = 768 contiguous bytes per wave instruction; partial lines must not allocate in L2).
//       - f32 xyz  (fuse_lane_kernel): lane `tid` takes pixels tid, tid+256, tid+512, tid+768 of the tile; one x3 store
//         per pixel.  7.0 TB/s = 0.87 of the 8 TB/s peak (C2, byte raster).  Grid: one tile per workgroup.
//       - f64 xyz  (fuse_pair_kernel): a 24-B row cannot leave in one instruction, and every split of it by instruction
//         (x4+x2, 3 x x2) leaves gaps inside each wave instruction (2.7 TB/s).  So TWO LANES share a pixel: the even lane
//         computes world x,y and stores (x_lo x_hi y_lo), the odd lane computes y,z and stores (y_hi z_lo z_hi) -- again
//         12 B per lane at a 12-B stride, at ~1.3x the fp64 arithmetic per pixel (still far under the SIMD budget).
//         6.4 TB/s = 0.80 of peak (the LDS-transposed 16-B-store kernel of round 1: 5.4-5.6).  Grid: one tile per
//         workgroup.
//       - f32 xyz + colour (fuse_rgb_kernel): the tile's 3072 rgb bytes come in as 192 16-byte loads through LDS, each
//         pixel leaves one nontemporal dword (r | g<<8 | b<<16, alpha 0).  7.1-7.3 TB/s = 0.89-0.91 of peak at
//         20 B/point when the inputs sit in the 256 MiB Infinity Cache between launches (C2), 5.3-6.0 TB/s when
//         depth + colour really stream from HBM (config 5: 1080p f32 depth, 23 B/point).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "r3d_internal.h"
#include "r3d_voxel_dev.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPx = 4;                      // pixels per lane (f32 kernels)
constexpr int kTile = kThreads * kPx;       // pixels per workgroup tile (all kernels)
constexpr int kStageChunkMB = 96;           // ... in steps of this many input MB (sweep: profiles/r02_cold_inputs.log)

struct FuseDims {
  double scale;
  uint32_t hw;              // H*W
  uint32_t width;
  uint32_t tiles_per_frame; // ceil(hw / 1024)
  uint32_t n_frames;
  uint32_t w_magic;         // floor(x / width) = (x * w_magic) >> w_shift for x < 2^31 (make_magic)
  uint32_t w_shift;
  uint32_t t_magic;         // floor(tile / tiles_per_frame), same scheme
  uint32_t t_shift;
  uint32_t total_tiles;     // tiles_per_frame * n_frames (< 2^31)
  uint32_t rgb_vec_ok;      // colour plane: tile bases are 16-byte aligned (hw % 16 == 0 and aligned base pointer)
  uint32_t depth_vec_ok;    // raster: every whole tile starts 4-byte aligned (hw * sizeof(element) % 4 == 0, aligned base)
};

__device__ __forceinline__ uint32_t magic_div(uint32_t x, uint32_t magic, uint32_t shift) {
  return (uint32_t)(((uint64_t)x * magic) >> shift);
}

struct Pose {
  double r[9];
  double t[3];
};

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

typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

// one global_store_dwordx3 with the nontemporal hint (4-byte alignment suffices on gfx950).
// The s_nop is part of the store: a VMEM store of more than 64 bits reads its data registers up to two cycles after issue
// and a VALU write to them in that window lands in the stored value (gfx940+ hazard; the compiler pads its own stores, it
// cannot see into an asm).  Found in round 3 when fuse_voxel_kernel's key arithmetic reused the registers right behind the
// store: lanes 12-15 of every row stored halves of the next doubles.  Every dwordx3 asm store in the library carries it.
__device__ __forceinline__ void store_x3_nt(void* dst, f32x3 v) {
  asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void store_x3_nt(void* dst, u32x3 v) {
  asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

// ---- f32 xyz: lane-per-pixel rounds ----------------------------------------------------------------------------
// ITEMS loads per lane; item q = r*256 + tid reads pixel q >> SHIFT of the tile (SHIFT 1: two lanes share a pixel).
// CLAMP: UNCONDITIONAL loads from a clamped index.  With a predicated load (`p < hw ? depth[p] : 0`) of a u16 / f32
// element the compiler sinks each element's conversion into the load's own branch and puts an s_waitcnt vmcnt(0)
// behind every load -- four serialised memory round trips per tile, invisible while the raster sits in the Infinity
// Cache, a 1.3-1.5x loss when it really comes from HBM (1080p batches; profiles/r02_c5_probe.log).  For byte rasters the
// loads were batched either way; measured per kernel: the fused f32-xyz kernel is 4 % faster clamped at one tile per
// workgroup (7.0 vs 6.7 TB/s on C2), the lane-pair f64 kernel and the pose-less unprojection are faster predicated
// (6.6 vs 5.5, 7.1 vs 6.4), so each takes its own form.
template <typename DT, int ITEMS, int SHIFT, bool CLAMP>
__device__ __forceinline__ void load_tile(const DT* __restrict__ depth, const FuseDims& dm, uint32_t tile, uint32_t tid,
                                          DT raw[ITEMS]) {
  const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
  const uint32_t tf = tile - frame * dm.tiles_per_frame;
  const uint64_t fbase = (uint64_t)frame * dm.hw;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t p = tf * kTile + ((r * kThreads + tid) >> SHIFT);
    if (CLAMP)
      raw[r] = depth[fbase + min(p, dm.hw - 1)];  // lanes past the frame's end read its last pixel and store nothing
    else
      raw[r] = p < dm.hw ? depth[fbase + p] : DT(0);
  }
}

// ---- vector loads + in-wave redistribution (the WAVE-CONTIGUOUS mapping; byte-raster unprojection only) ---------
// Every wave owns 256 CONSECUTIVE pixels of the tile (lane l, round r -> pixel 256*wave + 64*r + l), brings them in as
// ONE coalesced dword load per lane (256 B per wave instead of four 64-B crumbs) and each lane fetches its bytes from
// the lane that holds them with ds_bpermute_b32 -- no LDS allocation, no barrier; stores keep THE shape (one contiguous
// 768-B run per wave instruction).  Whole, 4-byte-aligned tiles; ragged / unaligned ones take clamped byte loads in the
// same mapping.  A/B over every kernel, raster in cache and not (profiles/r02_cold_inputs.log): it wins where the
// arithmetic is lightest -- the pose-less u8 -> f32 unprojection, 7.35 vs 7.09 TB/s -- and loses or ties elsewhere
// (u8 fused 6.8 vs 7.1, u8 -> f64 5.5 vs 6.6, u16 equal), so only that kernel uses it.
template <int ROUNDS>
__device__ __forceinline__ void load_tile_wave(const uint8_t* __restrict__ depth, const FuseDims& dm, uint64_t fbase, uint32_t tf,
                                               uint32_t tid, uint8_t raw[ROUNDS]) {
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  const bool whole = dm.depth_vec_ok && (tf + 1) * kTile <= dm.hw;  // workgroup-uniform
  if (whole) {
    const uint32_t dw = reinterpret_cast<const uint32_t*>(depth + fbase + (uint64_t)tf * kTile)[tid];  // pixels 4*tid ..
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      // pixel 64*r + lane of the wave's 256 sits in the dword of lane 16*r + lane/4, byte lane%4
      const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((16u * r + (lane >> 2)) << 2), (int)dw);
      raw[r] = (uint8_t)(got >> (8u * (lane & 3u)));
    }
  } else {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const uint32_t p = tf * kTile + 256u * wave + 64u * r + lane;
      raw[r] = depth[fbase + min(p, dm.hw - 1)];
    }
  }
}

template <typename DT, bool POSE, bool WAVE>
__global__ __launch_bounds__(kThreads) void fuse_lane_kernel(const DT* __restrict__ depth, float* __restrict__ out,
                                                             const double* __restrict__ u, const double* __restrict__ v,
                                                             const double* __restrict__ pose, const FuseDims dm) {
  const uint32_t tid = threadIdx.x;
  const uint32_t first = WAVE ? 256u * (tid >> 6) + (tid & 63u) : tid;  // this lane's pixel of round 0 within the tile
  constexpr uint32_t kRound = WAVE ? 64u : (uint32_t)kThreads;           // pixel step between its rounds
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose<POSE>(pose, frame, P);
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    DT raw[kPx];
    if constexpr (WAVE)
      load_tile_wave<kPx>(depth, dm, fbase, tf, tid, raw);
    else
      load_tile<DT, kPx, 0, (POSE || sizeof(DT) > 1)>(depth, dm, tile, tid, raw);  // measured per variant, see load_tile
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t p = tf * kTile + r * kRound + first;
      if (p < dm.hw) {
        const uint32_t j = magic_div(p, dm.w_magic, dm.w_shift);
        const uint32_t i = p - j * dm.width;
        double w[3];
        point<POSE>((double)raw[r] * dm.scale, u[i], v[j], P, w);
        store_x3_nt(out + (fbase + p) * 3, f32x3{(float)w[0], (float)w[1], (float)w[2]});
      }
    }
  }
}

// ---- f64 xyz: two lanes per pixel, each stores its 12-byte half of the 24-byte row ------------------------------
// item q of a tile (q = r*256 + tid, r = 0..7) is half (q & 1) of pixel (q >> 1); the even half is (x, y_lo), the odd
// half (y_hi, z): the even lane evaluates world rows 0,1, the odd lane rows 1,2.
template <typename DT, bool POSE>
__global__ __launch_bounds__(kThreads) void fuse_pair_kernel(const DT* __restrict__ depth, double* __restrict__ out,
                                                             const double* __restrict__ u, const double* __restrict__ v,
                                                             const double* __restrict__ pose, const FuseDims dm) {
  constexpr int kItems = 2 * kPx;
  const uint32_t tid = threadIdx.x;
  const bool odd = tid & 1u;
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose<POSE>(pose, frame, P);
    double ra[3], rb[3];  // this lane's two rows of Rinv: (0,1) or (1,2)
    if (POSE) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ra[c] = odd ? P.r[3 + c] : P.r[c];
        rb[c] = odd ? P.r[6 + c] : P.r[3 + c];
      }
    }
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    DT raw[kItems];
    load_tile<DT, kItems, 1, (sizeof(DT) > 1)>(depth, dm, tile, tid, raw);
    uint32_t* tile_out = reinterpret_cast<uint32_t*>(out) + (fbase + (uint64_t)tf * kTile) * 6;
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const uint32_t q = r * kThreads + tid;
      const uint32_t p = tf * kTile + (q >> 1);
      if (p < dm.hw) {
        const uint32_t j = magic_div(p, dm.w_magic, dm.w_shift);
        const uint32_t i = p - j * dm.width;
        const double z = (double)raw[r] * dm.scale;
        const double x = u[i] * z;  // c2w:78
        const double y = v[j] * z;  // c2w:79
        double a, b;
        if (POSE) {
          const double dx = x - P.t[0], dy = y - P.t[1], dz = z - P.t[2];
          a = fma(ra[2], dz, fma(ra[1], dy, ra[0] * dx));
          b = fma(rb[2], dz, fma(rb[1], dy, rb[0] * dx));
        } else {
          a = odd ? y : x;
          b = odd ? z : y;
        }
        const uint32_t alo = (uint32_t)__double2loint(a), ahi = (uint32_t)__double2hiint(a);
        const uint32_t blo = (uint32_t)__double2loint(b), bhi = (uint32_t)__double2hiint(b);
        store_x3_nt(tile_out + (uint64_t)q * 3, odd ? u32x3{ahi, blo, bhi} : u32x3{alo, ahi, blo});
      }
    }
  }
}

// ---- f32 xyz + colour -------------------------------------------------------------------------------------------
// rgb is [F][H][W][3] uint8 (the image the depth raster belongs to, R,G,B order); rgba_out is [F*H*W] uint32 =
// r | g<<8 | b<<16 (bytes R,G,B,0 in memory: the "R G B 0" of genply_noRGB's rows, pixel_to_camera.py:84).
template <typename DT, bool POSE>
__global__ __launch_bounds__(kThreads) void fuse_rgb_kernel(const DT* __restrict__ depth, const uint8_t* __restrict__ rgb,
                                                            float* __restrict__ out, uint32_t* __restrict__ rgba_out,
                                                            const double* __restrict__ u, const double* __restrict__ v,
                                                            const double* __restrict__ pose, const FuseDims dm) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kTile * 3];
  const uint32_t tid = threadIdx.x;
  // whole, aligned tiles bring their 3072 colour bytes in as 192 16-byte loads (wave-uniform choice per tile)
  auto is_staged = [&](uint32_t tile) {
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    return dm.rgb_vec_ok && (tf + 1) * kTile <= dm.hw;
  };
  auto colour_chunk = [&](uint32_t tile) -> u32x4 {
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    return reinterpret_cast<const u32x4*>(rgb + ((uint64_t)frame * dm.hw + (uint64_t)tf * kTile) * 3)[tid];
  };
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose<POSE>(pose, frame, P);
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    const bool staged = is_staged(tile);
    DT raw[kPx];
    u32x4 ccur = {0, 0, 0, 0};
    load_tile<DT, kPx, 0, true>(depth, dm, tile, tid, raw);
    if (staged && tid < kTile * 3 / 16) ccur = colour_chunk(tile);
    if (staged) {
      if (tid < kTile * 3 / 16) reinterpret_cast<u32x4*>(lds)[tid] = ccur;
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t l = r * kThreads + tid;
      const uint32_t p = tf * kTile + l;
      if (p < dm.hw) {
        const uint32_t j = magic_div(p, dm.w_magic, dm.w_shift);
        const uint32_t i = p - j * dm.width;
        double w[3];
        point<POSE>((double)raw[r] * dm.scale, u[i], v[j], P, w);
        store_x3_nt(out + (fbase + p) * 3, f32x3{(float)w[0], (float)w[1], (float)w[2]});
        uint32_t c;
        if (staged) {
          c = (uint32_t)lds[l * 3] | ((uint32_t)lds[l * 3 + 1] << 8) | ((uint32_t)lds[l * 3 + 2] << 16);
        } else {
          const uint8_t* s = rgb + (fbase + p) * 3;
          c = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
        }
        __builtin_nontemporal_store(c, rgba_out + fbase + p);
      }
    }
    if (staged) __syncthreads();  // the next tile's colour bytes must not land before everyone has read these
  }
}

// ---- f32 xyz (+ colour) AND the occupied-voxel set in one pass ---------------------------------------------------
// Config 5 builds the cloud and the voxel map of the same points: fuse_rgb_kernel writes 16 B/point, voxel_insert_kernel
// reads 12 of them back.  This kernel is fuse_rgb_kernel's body with voxel_insert_kernel<true>'s per-point work appended
// while the world point is still in registers: the key comes from the SAME three floats the store writes (so the set equals
// the one r3d_voxelset_insert builds from the cloud, bit for bit), neighbour-lane filter, claim in the workgroup's LDS set,
// the set flushed to the global table by all lanes when it holds kLdsKeepBelow codes and at the end of the run.  A workgroup
// walks a CONTIGUOUS run of tiles (neighbouring rows hit the same voxels); the cloud is not read back.
constexpr uint64_t kVoxelRun = 64;  // tiles per workgroup of fuse_voxel_kernel on big batches

struct VoxelView {
  uint64_t* table;
  unsigned long long* counters;
  double factor;
  int log2cap;
};

template <typename DT, bool POSE, bool RGB>
__global__ __launch_bounds__(kThreads) void fuse_voxel_kernel(const DT* __restrict__ depth, const uint8_t* __restrict__ rgb,
                                                              float* __restrict__ out, uint32_t* __restrict__ rgba_out,
                                                              const double* __restrict__ u, const double* __restrict__ v,
                                                              const double* __restrict__ pose, const FuseDims dm,
                                                              const VoxelView vv) {
  using namespace r3d_vox;
  __shared__ __attribute__((aligned(16))) uint8_t lds[RGB ? kTile * 3 : 16];
  __shared__ unsigned long long local_set[kLdsSlots];
  __shared__ unsigned local_fill[3];  // codes gained per tile, in rotation (see voxel_insert_kernel)
  const uint32_t tid = threadIdx.x;
  const int lane = tid & 63;
  const uint64_t mask = ((uint64_t)1 << vv.log2cap) - 1;
  unsigned n_new = 0, n_ignored = 0, n_over = 0;
  const uint32_t per_wg = (dm.total_tiles + gridDim.x - 1) / gridDim.x;
  const uint64_t lo64 = (uint64_t)blockIdx.x * per_wg;
  const uint32_t tile_lo = lo64 < dm.total_tiles ? (uint32_t)lo64 : dm.total_tiles;
  const uint32_t tile_hi = dm.total_tiles - tile_lo < per_wg ? dm.total_tiles : tile_lo + per_wg;
  for (int k = tid; k < kLdsSlots; k += kThreads) local_set[k] = kEmpty;
  if (tid < 3) local_fill[tid] = 0;
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < kLdsSlots / kThreads; ++k) {
      const int s = k * kThreads + tid;
      const uint64_t code = local_set[s];
      if (code != kEmpty) {
        local_set[s] = kEmpty;
        const int r = table_insert(vv.table, mask, vv.log2cap, code);
        n_new += r > 0 ? 1u : 0u;
        n_over += r < 0 ? 1u : 0u;
      }
    }
  };
  unsigned total = 0, j = 0;
  for (uint32_t tile = tile_lo; tile < tile_hi; ++tile, ++j) {
    // the previous tile's lookups, its count and its colour reads are done (first tile: the wipe above has landed)
    lds_settle();
    __syncthreads();
    if (j > 0) total += local_fill[(j - 1) % 3];
    if (tid == 0) local_fill[(j + 1) % 3] = 0;
    // No barrier of its own: the one in front of the tile's claims (below) is taken by every wave on every tile, whatever
    // `total` says -- a wave that flushed its slots while another did not leaves every code in the set or in the table.
    if (total >= (unsigned)kLdsKeepBelow) {
      flush();
      total = 0;
    }
    const uint32_t frame = magic_div(tile, dm.t_magic, dm.t_shift);
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose<POSE>(pose, frame, P);
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    const bool staged = RGB && dm.rgb_vec_ok && (tf + 1) * kTile <= dm.hw;
    DT raw[kPx];
    load_tile<DT, kPx, 0, true>(depth, dm, tile, tid, raw);
    if (RGB && staged && tid < kTile * 3 / 16)
      reinterpret_cast<u32x4*>(lds)[tid] = reinterpret_cast<const u32x4*>(rgb + (fbase + (uint64_t)tf * kTile) * 3)[tid];
    lds_settle();
    __syncthreads();  // the tile's colour bytes are in LDS; nobody is flushing any more
    unsigned claimed = 0;
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t l = r * kThreads + tid;
      const uint32_t p = tf * kTile + l;
      uint64_t code = kEmpty;
      bool live = p < dm.hw;
      if (live) {
        const uint32_t jrow = magic_div(p, dm.w_magic, dm.w_shift);
        const uint32_t i = p - jrow * dm.width;
        double w[3];
        point<POSE>((double)raw[r] * dm.scale, u[i], v[jrow], P, w);
        const float fx = (float)w[0], fy = (float)w[1], fz = (float)w[2];
        store_x3_nt(out + (fbase + p) * 3, f32x3{fx, fy, fz});
        if (RGB) {
          uint32_t c;
          if (staged) {
            c = (uint32_t)lds[l * 3] | ((uint32_t)lds[l * 3 + 1] << 8) | ((uint32_t)lds[l * 3 + 2] << 16);
          } else {
            const uint8_t* sp = rgb + (fbase + p) * 3;
            c = (uint32_t)sp[0] | ((uint32_t)sp[1] << 8) | ((uint32_t)sp[2] << 16);
          }
          __builtin_nontemporal_store(c, rgba_out + fbase + p);
        }
        if (!voxel_key(fx, fy, fz, vv.factor, &code)) {
          ++n_ignored;
          live = false;
          code = kEmpty;
        }
      }
      const uint64_t prev = prev_lane_u64(code);
      if (live && lane > 0 && prev == code) live = false;
      if (live) {
        bool mine = false;
        const bool done = lds_set_claim(local_set, code, &mine);
        claimed += mine ? 1u : 0u;
        if (!done) {
          const int r2 = table_insert(vv.table, mask, vv.log2cap, code);
          n_new += r2 > 0 ? 1u : 0u;
          n_over += r2 < 0 ? 1u : 0u;
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) claimed += __shfl_down(claimed, off, 64);
    if (lane == 0 && claimed) atomicAdd(&local_fill[j % 3], claimed);
  }
  __syncthreads();
  flush();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off, 64);
    n_ignored += __shfl_down(n_ignored, off, 64);
    n_over += __shfl_down(n_over, off, 64);
  }
  if (lane == 0) {
    if (n_new) atomicAdd(&vv.counters[0], (unsigned long long)n_new);
    if (n_ignored) atomicAdd(&vv.counters[1], (unsigned long long)n_ignored);
    if (n_over) atomicAdd(&vv.counters[2], (unsigned long long)n_over);
  }
}

// colour only: [n][3] uint8 -> [n] rgba dwords, for clouds whose xyz was made elsewhere (f64 xyz path)
__global__ __launch_bounds__(kThreads) void rgb_expand_kernel(const uint8_t* __restrict__ rgb, uint32_t* __restrict__ rgba_out,
                                                              uint64_t n, uint32_t vec_ok) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kTile * 3];
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = (n + kTile - 1) / kTile;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t base = tile * kTile;
    const bool staged = vec_ok && base + kTile <= n;
    if (staged) {
      if (tid < kTile * 3 / 16) reinterpret_cast<uint4*>(lds)[tid] = reinterpret_cast<const uint4*>(rgb + base * 3)[tid];
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t l = r * kThreads + tid;
      if (base + l < n) {
        uint32_t c;
        if (staged) {
          c = (uint32_t)lds[l * 3] | ((uint32_t)lds[l * 3 + 1] << 8) | ((uint32_t)lds[l * 3 + 2] << 16);
        } else {
          const uint8_t* s = rgb + (base + l) * 3;
          c = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
        }
        __builtin_nontemporal_store(c, rgba_out + base + l);
      }
    }
    if (staged) __syncthreads();
  }
}

// ---- staging the inputs in the Infinity Cache -------------------------------------------------------------------
// Measured (tools/cold_inputs.hip, tools/mall_effect.py): when the raster is NOT already in the 256 MiB Infinity Cache,
// its reads -- 8 % of the bytes -- arrive at the HBM controllers sprinkled among twelve times as many writes and the
// fused kernel drops from 7.0 to 4.0-4.4 TB/s (prefetching tiles ahead inside the kernel, vector loads, other launch
// geometries: no help -- it is not latency, it is read/write mixing at the DRAM).  A read-only sweep of the inputs
// FIRST (5.6 TB/s: 9 us for C2's 49 MB) leaves them in the Infinity Cache, the fused kernel that follows reads them
// from there and sends a pure write stream to HBM: 6.2 TB/s for the pair, cold.  Big batches go chunk by chunk.
// Four independent 16-byte loads per lane and iteration (one load per iteration left the sweep latency-bound: six
// dependent round trips for C2's raster -- 7.4 us even when every line was already in the cache).
__global__ __launch_bounds__(kThreads) void cache_touch_kernel(const uint4* __restrict__ src, uint64_t n16,
                                                               uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * (kThreads * 4);
  for (uint64_t base = (uint64_t)blockIdx.x * (kThreads * 4) + threadIdx.x; base < n16; base += stride) {
    uint4 q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t i = base + (uint64_t)k * kThreads;
      q[k] = src[i < n16 ? i : n16 - 1];   // clamped: unconditional loads, all four in flight together
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= q[k].x ^ q[k].y ^ q[k].z ^ q[k].w;
  }
  if (acc == 0x9e3779b9u && n16 == ~(uint64_t)0) *sink = acc;  // never true: keeps the loads, writes nothing
}

// whole 16-byte pieces inside [p, p + bytes): the ragged ends share their cache lines with the pieces next to them
void cache_touch(r3d_ctx* ctx, const void* p, uint64_t bytes) {
  const uintptr_t lo = ((uintptr_t)p + 15) & ~(uintptr_t)15, hi = ((uintptr_t)p + bytes) & ~(uintptr_t)15;
  if (hi <= lo) return;
  const uint64_t n16 = (hi - lo) / 16;
  uint64_t blocks = (n16 + kThreads * 4 - 1) / (kThreads * 4);
  if (blocks > (uint64_t)ctx->num_cus * 8) blocks = (uint64_t)ctx->num_cus * 8;
  hipLaunchKernelGGL(cache_touch_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, ctx->stream,
                     reinterpret_cast<const uint4*>(lo), n16, static_cast<uint32_t*>(nullptr));
}

struct FusePtrs {
  const void* depth;
  void* out;
  const double* u;
  const double* v;
  const double* pose;
  const uint8_t* rgb;
  uint32_t* rgba;
};

template <typename DT, bool POSE>
void launch_typed(const FusePtrs& p, const FuseDims& dm, int out_dtype, int blocks, bool wave, hipStream_t s) {
  const DT* d = static_cast<const DT*>(p.depth);
  if (out_dtype == R3D_F64) {
    hipLaunchKernelGGL((fuse_pair_kernel<DT, POSE>), dim3(blocks), dim3(kThreads), 0, s, d, static_cast<double*>(p.out), p.u,
                       p.v, p.pose, dm);
  } else if (p.rgb) {
    hipLaunchKernelGGL((fuse_rgb_kernel<DT, POSE>), dim3(blocks), dim3(kThreads), 0, s, d, p.rgb, static_cast<float*>(p.out),
                       p.rgba, p.u, p.v, p.pose, dm);
  } else {
    if constexpr (std::is_same<DT, uint8_t>::value && !POSE) {
      if (wave) {
        hipLaunchKernelGGL((fuse_lane_kernel<DT, POSE, true>), dim3(blocks), dim3(kThreads), 0, s, d, static_cast<float*>(p.out),
                           p.u, p.v, p.pose, dm);
        return;
      }
    }
    hipLaunchKernelGGL((fuse_lane_kernel<DT, POSE, false>), dim3(blocks), dim3(kThreads), 0, s, d, static_cast<float*>(p.out),
                       p.u, p.v, p.pose, dm);
  }
}

template <bool POSE>
void launch_depth(const FusePtrs& p, const FuseDims& dm, int depth_dtype, int out_dtype, int blocks, bool wave, hipStream_t s) {
  switch (depth_dtype) {
    case R3D_DEPTH_U8: launch_typed<uint8_t, POSE>(p, dm, out_dtype, blocks, wave, s); break;
    case R3D_DEPTH_U16: launch_typed<uint16_t, POSE>(p, dm, out_dtype, blocks, wave, s); break;
    default: launch_typed<float, POSE>(p, dm, out_dtype, blocks, wave, s); break;
  }
}

template <typename DT>
void launch_voxel_typed(const FusePtrs& p, const FuseDims& dm, const VoxelView& vv, bool with_pose, int blocks, hipStream_t s) {
  const DT* d = static_cast<const DT*>(p.depth);
  float* o = static_cast<float*>(p.out);
#define R3D_LAUNCH_FV(POSE, RGB)                                                                                         \
  hipLaunchKernelGGL((fuse_voxel_kernel<DT, POSE, RGB>), dim3(blocks), dim3(kThreads), 0, s, d, p.rgb, o, p.rgba, p.u, p.v, \
                     p.pose, dm, vv)
  if (with_pose) {
    if (p.rgb) R3D_LAUNCH_FV(true, true); else R3D_LAUNCH_FV(true, false);
  } else {
    if (p.rgb) R3D_LAUNCH_FV(false, true); else R3D_LAUNCH_FV(false, false);
  }
#undef R3D_LAUNCH_FV
}

void launch_voxel(const FusePtrs& p, const FuseDims& dm, const VoxelView& vv, int depth_dtype, bool with_pose, int blocks,
                  hipStream_t s) {
  switch (depth_dtype) {
    case R3D_DEPTH_U8: launch_voxel_typed<uint8_t>(p, dm, vv, with_pose, blocks, s); break;
    case R3D_DEPTH_U16: launch_voxel_typed<uint16_t>(p, dm, vv, with_pose, blocks, s); break;
    default: launch_voxel_typed<float>(p, dm, vv, with_pose, blocks, s); break;
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
                double depth_scale, const double* d_pose, bool with_pose, void* d_out, int out_dtype,
                const uint8_t* d_rgb = nullptr, uint32_t* d_rgba = nullptr, r3d_voxelset* vs = nullptr) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(cam != nullptr, "camera is NULL");
  VoxelView vv{nullptr, nullptr, 0.0, 0};
  if (vs) {
    r3d_ctx* vctx = nullptr;
    if ((rc = r3d_voxelset_device_view(vs, &vctx, &vv.factor, &vv.table, &vv.log2cap, &vv.counters))) return rc;
    R3D_REQUIRE(vctx == ctx, "voxel set belongs to a different ctx");
    R3D_REQUIRE(out_dtype == R3D_F32, "the voxel keys are taken from the f32 cloud: out_dtype must be R3D_F32");
  }
  R3D_REQUIRE(cam->ctx == ctx, "camera belongs to a different ctx");
  R3D_REQUIRE(depth_dtype >= R3D_DEPTH_U8 && depth_dtype <= R3D_DEPTH_F32, "unknown depth dtype %d", depth_dtype);
  R3D_REQUIRE(out_dtype == R3D_F32 || out_dtype == R3D_F64, "unknown output dtype %d", out_dtype);
  R3D_REQUIRE(n_frames >= 0, "n_frames must be >= 0");
  R3D_REQUIRE((d_rgb == nullptr) == (d_rgba == nullptr), "colour needs both the rgb plane and the rgba output");
  if (n_frames == 0) return R3D_OK;
  R3D_REQUIRE(d_depth && d_out, "NULL device pointer");
  R3D_REQUIRE(!with_pose || d_pose, "pose table is NULL");
  const uint64_t hw = (uint64_t)cam->height * cam->width;
  const uint64_t dsz = r3d_depth_size(depth_dtype), osz = 3 * (uint64_t)r3d_xyz_size(out_dtype);
  {
    const uint64_t all_tiles = ((hw + kTile - 1) / kTile) * (uint64_t)n_frames;
    R3D_REQUIRE(all_tiles < ((uint64_t)1 << 31), "batch too large for one launch (%llu tiles); split the frames",
                (unsigned long long)all_tiles);
  }
  const bool colour_after = d_rgb && out_dtype == R3D_F64;  // f64 xyz: colour goes through its own pass
  // byte-raster unprojection: dword loads + in-wave redistribution (load_tile_wave); knob 1 = element loads (A/B)
  const bool wave = !with_pose && out_dtype == R3D_F32 && depth_dtype == R3D_DEPTH_U8 && !(d_rgb && !colour_after);
  // Inputs staged through the Infinity Cache chunk by chunk (cache_touch_kernel).  auto = when the inputs are a small share of
  // the launch's traffic (the sweep is an extra pass over them; measured at 1000 frames, profiles/r02_cold_inputs.log: u8
  // 4.3 -> 6.4 TB/s, u8 + colour 3.8 -> 5.5, u16 4.8 -> 6.3, f32 depth -> f64 xyz 4.4 -> 5.6, but f32 depth + colour (7 of
  // 23 B/point are inputs) 5.7 -> 5.0: that one stays as it is), big enough for the extra launch to pay, AND NOT PRESUMED TO BE
  // IN THE CACHE ALREADY (r3d_inputs_*, r3d_ctx.hip): a raster this device's launches have just read is; one that an H2D copy,
  // the host pipeline or a collective has just written is not, nor is one the library has never seen, nor any that exceeds
  // what the cache keeps.  Round 3 staged by size alone and paid a 7.8 us sweep per launch on a raster that was cached.
  const bool rgb_in_kernel = d_rgb && !colour_after;
  const uint64_t in_per_frame = hw * (dsz + (rgb_in_kernel ? 3 : 0));
  const uint64_t in_bytes = in_per_frame * (uint64_t)n_frames;
  const uint64_t chunk_bytes = (uint64_t)(ctx->fuse_chunk_mb > 0 ? ctx->fuse_chunk_mb : kStageChunkMB) << 20;
  const uint64_t budget = (uint64_t)ctx->fuse_resident_mb << 20;
  const uint64_t depth_bytes = hw * dsz * (uint64_t)n_frames, rgb_bytes = rgb_in_kernel ? hw * 3 * (uint64_t)n_frames : 0;
  const bool small_inputs = dsz <= 2 || (out_dtype == R3D_F64 && !rgb_in_kernel);
  bool stage_depth = ctx->fuse_prefetch == 2, stage_rgb = ctx->fuse_prefetch == 2 && rgb_in_kernel;
  if (ctx->fuse_prefetch == 0 && small_inputs && in_bytes > ((uint64_t)ctx->fuse_stage_auto_mb << 20)) {
    stage_depth = !r3d_inputs_resident(ctx, d_depth, depth_bytes, budget);
    stage_rgb = rgb_in_kernel && !r3d_inputs_resident(ctx, d_rgb, rgb_bytes, budget);
  }
  const bool stage = stage_depth || stage_rgb;
  r3d_wrote(ctx, d_out, hw * osz * (uint64_t)n_frames);
  // whatever the decision: once this launch has run, it HAS read its inputs
  r3d_inputs_read(ctx, d_depth, depth_bytes, budget);
  if (rgb_in_kernel) r3d_inputs_read(ctx, d_rgb, rgb_bytes, budget);
  int frames_per_step = n_frames;
  if (stage) {
    const uint64_t f = chunk_bytes / (in_per_frame ? in_per_frame : 1);
    frames_per_step = (int)(f < 1 ? 1 : (f > (uint64_t)n_frames ? (uint64_t)n_frames : f));
  }
  for (int f0 = 0; f0 < n_frames; f0 += frames_per_step) {
    const int nf = n_frames - f0 < frames_per_step ? n_frames - f0 : frames_per_step;
    const uint64_t px0 = hw * (uint64_t)f0;
    const char* dd = static_cast<const char*>(d_depth) + px0 * dsz;
    const uint8_t* rr = d_rgb ? d_rgb + px0 * 3 : nullptr;
    FusePtrs p{dd, static_cast<char*>(d_out) + px0 * osz, cam->d_u, cam->d_v, with_pose ? d_pose + (uint64_t)f0 * 12 : nullptr,
               colour_after ? nullptr : rr, d_rgba ? d_rgba + px0 : nullptr};
    FuseDims dm;
    dm.scale = depth_scale;
    dm.hw = (uint32_t)hw;
    dm.width = (uint32_t)cam->width;
    dm.tiles_per_frame = (uint32_t)((hw + kTile - 1) / kTile);
    dm.n_frames = (uint32_t)nf;
    make_magic(dm.width, &dm.w_magic, &dm.w_shift);
    make_magic(dm.tiles_per_frame, &dm.t_magic, &dm.t_shift);
    const uint64_t total_tiles = (uint64_t)dm.tiles_per_frame * nf;
    dm.total_tiles = (uint32_t)total_tiles;
    dm.rgb_vec_ok = rr && ((uintptr_t)rr % 16 == 0) && (hw % 16 == 0);
    dm.depth_vec_ok = ((uintptr_t)dd % 4 == 0) && ((hw * dsz) % 4 == 0);
    if (stage_depth) cache_touch(ctx, dd, hw * dsz * (uint64_t)nf);
    if (stage_rgb) cache_touch(ctx, rr, hw * 3 * (uint64_t)nf);
    ctx->fuse_sweeps += (int)stage_depth + (int)stage_rgb;
    // measured (profiles/r02_c5_probe.log, r02_ab_kernels.log, r02_all_kernels.json): one tile per workgroup for every kernel
    // (the element-load form of the byte-raster unprojection, kept for A/B, likes 8 striding workgroups per CU)
    const bool stride8 = !wave && !with_pose && out_dtype == R3D_F32 && depth_dtype == R3D_DEPTH_U8 && !p.rgb;
    uint64_t blocks = ctx->fuse_blocks > 0 ? (uint64_t)ctx->fuse_blocks : stride8 ? (uint64_t)ctx->num_cus * 8 : total_tiles;
    if (blocks > total_tiles) blocks = total_tiles;
    if (vs) {
      // contiguous runs of tiles per workgroup: fuse_blocks > 0 sets the grid, else runs of kVoxelRun tiles but no fewer than
      // 8 workgroups per CU (measured on 2000 x 1080p frames: 8/CU 36.1 ms, 4096 workgroups 30.9, 16384 28.0, 65536 27.2)
      uint64_t g = ctx->fuse_blocks > 0 ? (uint64_t)ctx->fuse_blocks
                                         : std::max<uint64_t>((uint64_t)ctx->num_cus * 8, (total_tiles + kVoxelRun - 1) / kVoxelRun);
      if (g > total_tiles) g = total_tiles;
      launch_voxel(p, dm, vv, depth_dtype, with_pose, (int)g, ctx->stream);
    } else if (with_pose)
      launch_depth<true>(p, dm, depth_dtype, out_dtype, (int)blocks, wave, ctx->stream);
    else
      launch_depth<false>(p, dm, depth_dtype, out_dtype, (int)blocks, wave, ctx->stream);
  }
  if (colour_after) {
    const uint64_t n = hw * (uint64_t)n_frames;
    uint64_t b = (n + kTile - 1) / kTile;
    if (b > (uint64_t)ctx->num_cus * 16) b = (uint64_t)ctx->num_cus * 16;
    hipLaunchKernelGGL(rgb_expand_kernel, dim3((unsigned)b), dim3(kThreads), 0, ctx->stream, d_rgb, d_rgba, n,
                       (uint32_t)((uintptr_t)d_rgb % 16 == 0));
  }
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

// RGBD batch from host memory: depth + colour stream in, xyz + rgba stream out, chunk by chunk through the pinned pipeline
int fuse_rgb_host(r3d_ctx* ctx, const r3d_camera* cam, const void* h_depth, int depth_dtype, int n_frames, double depth_scale,
                  const double* h_pose, const unsigned char* h_rgb, void* h_xyz_out, int out_dtype, uint32_t* h_rgba_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(cam != nullptr, "camera is NULL");
  R3D_REQUIRE(depth_dtype >= R3D_DEPTH_U8 && depth_dtype <= R3D_DEPTH_F32, "unknown depth dtype %d", depth_dtype);
  R3D_REQUIRE(out_dtype == R3D_F32 || out_dtype == R3D_F64, "unknown output dtype %d", out_dtype);
  R3D_REQUIRE(n_frames >= 0, "n_frames must be >= 0");
  if (n_frames == 0) return R3D_OK;
  R3D_REQUIRE(h_depth && h_rgb && h_xyz_out && h_rgba_out, "NULL host pointer");
  const size_t px = (size_t)cam->height * cam->width, n = px * n_frames;
  const size_t f_depth = px * r3d_depth_size(depth_dtype), f_rgb = px * 3, f_xyz = px * 3 * r3d_xyz_size(out_dtype),
               f_rgba = px * 4;
  void *d_depth = nullptr, *d_xyz = nullptr, *d_pose = nullptr, *d_rgb = nullptr, *d_rgba = nullptr;
  if ((rc = r3d_scratch(ctx, 0, n * r3d_depth_size(depth_dtype), &d_depth))) return rc;
  if ((rc = r3d_scratch(ctx, 1, n * 3 * r3d_xyz_size(out_dtype), &d_xyz))) return rc;
  if ((rc = r3d_scratch(ctx, 3, n * 3, &d_rgb))) return rc;
  if ((rc = r3d_scratch(ctx, 4, n * 4, &d_rgba))) return rc;
  if (h_pose) {
    if ((rc = r3d_scratch(ctx, 2, (size_t)n_frames * 12 * sizeof(double), &d_pose))) return rc;
    R3D_HIP(hipMemcpyAsync(d_pose, h_pose, (size_t)n_frames * 12 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  const r3d_pipe_buf ins[2] = {{const_cast<void*>(h_depth), d_depth, f_depth}, {const_cast<unsigned char*>(h_rgb), d_rgb, f_rgb}};
  const r3d_pipe_buf outs[2] = {{h_xyz_out, d_xyz, f_xyz}, {h_rgba_out, d_rgba, f_rgba}};
  auto launch = [&](int64_t lo, int64_t cnt) -> int {
    return fuse_common(ctx, cam, static_cast<char*>(d_depth) + (size_t)lo * f_depth, depth_dtype, (int)cnt, depth_scale,
                       h_pose ? static_cast<const double*>(d_pose) + (size_t)lo * 12 : nullptr, h_pose != nullptr,
                       static_cast<char*>(d_xyz) + (size_t)lo * f_xyz, out_dtype,
                       static_cast<const uint8_t*>(d_rgb) + (size_t)lo * f_rgb, static_cast<uint32_t*>(d_rgba) + (size_t)lo * px);
  };
  return r3d_host_pipeline_multi(ctx, n_frames, ins, 2, outs, 2, launch);
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
  return R3D_OK;
}

int r3d_cache_prefetch(r3d_ctx* ctx, const void* d_ptr, size_t bytes) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  if (bytes == 0) return R3D_OK;
  R3D_REQUIRE(d_ptr != nullptr, "NULL device pointer");
  cache_touch(ctx, d_ptr, bytes);
  r3d_inputs_read(ctx, d_ptr, bytes, (size_t)ctx->fuse_resident_mb << 20);
  R3D_HIP(hipGetLastError());
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

int r3d_fuse_frames_rgb(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                        double depth_scale, const double* d_pose, const unsigned char* d_rgb, void* d_xyz_out,
                        int out_dtype, uint32_t* d_rgba_out) {
  R3D_REQUIRE(n_frames == 0 || (d_rgb && d_rgba_out), "colour plane / rgba output is NULL");
  return fuse_common(ctx, cam, d_depth, depth_dtype, n_frames, depth_scale, d_pose, d_pose != nullptr, d_xyz_out, out_dtype,
                     d_rgb, d_rgba_out);
}

int r3d_fuse_frames_voxel(r3d_ctx* ctx, const r3d_camera* cam, const void* d_depth, int depth_dtype, int n_frames,
                          double depth_scale, const double* d_pose, const unsigned char* d_rgb, float* d_xyz_out,
                          uint32_t* d_rgba_out, r3d_voxelset* vs) {
  R3D_REQUIRE(vs != nullptr, "voxel set is NULL");
  R3D_REQUIRE((d_rgb == nullptr) == (d_rgba_out == nullptr), "colour needs both the rgb plane and the rgba output");
  R3D_REQUIRE(ctx != nullptr && cam != nullptr, "NULL argument");
  {  // the set's stream must be this ctx's: the probe below samples and inserts on it right behind the fuse
    r3d_ctx* vctx = nullptr;
    VoxelView view;
    int rc = r3d_voxelset_device_view(vs, &vctx, &view.factor, &view.table, &view.log2cap, &view.counters);
    if (rc) return rc;
    R3D_REQUIRE(vctx == ctx, "voxel set belongs to a different ctx");
  }
  const bool with_pose = d_pose != nullptr;
  // Which form is faster depends on the CLOUD: where neighbouring pixels share voxels (scans) the one-launch kernel saves
  // reading the cloud back; where nearly every point has a voxel of its own, its per-point CAS into the table is the whole cost
  // and plain fuse + the sort-merge insert wins 2-3x (r3d_voxel.hip).  So a big batch is probed: the first frames are fused by
  // the plain kernel, a sample of THEIR cloud decides for the rest ("voxel_path" 1 / 2 force a form).  Same cloud, same set.
  const int64_t hw = (int64_t)cam->height * cam->width, total = hw * (int64_t)n_frames;
  const int path = ctx->voxel_path;
  if (path == 1 || n_frames <= 0 || !r3d_voxelset_sort_feasible(vs, total, path == 2)) {
    if (n_frames > 0) ctx->voxel_last_path = 1;
    return fuse_common(ctx, cam, d_depth, depth_dtype, n_frames, depth_scale, d_pose, with_pose, d_xyz_out, R3D_F32, d_rgb,
                       d_rgba_out, vs);
  }
  const int64_t want = (int64_t)1 << 18;
  const int f0 = path == 2 ? n_frames : (int)std::min<int64_t>(n_frames, std::max<int64_t>(1, (want + hw - 1) / hw));
  int rc = fuse_common(ctx, cam, d_depth, depth_dtype, f0, depth_scale, d_pose, with_pose, d_xyz_out, R3D_F32, d_rgb, d_rgba_out);
  if (rc) return rc;
  bool sort = path == 2;
  if (!sort && (rc = r3d_voxelset_sample(vs, d_xyz_out, hw * f0, total, 1.5, &sort))) return rc;
  const size_t dsz = r3d_depth_size(depth_dtype);
  const char* depth_rest = static_cast<const char*>(d_depth) + (size_t)hw * f0 * dsz;
  const double* pose_rest = with_pose ? d_pose + (size_t)f0 * 12 : nullptr;
  const unsigned char* rgb_rest = d_rgb ? d_rgb + (size_t)hw * f0 * 3 : nullptr;
  uint32_t* rgba_rest = d_rgba_out ? d_rgba_out + (size_t)hw * f0 : nullptr;
  float* xyz_rest = d_xyz_out + (size_t)hw * f0 * 3;
  if (sort) {
    if (f0 < n_frames && (rc = fuse_common(ctx, cam, depth_rest, depth_dtype, n_frames - f0, depth_scale, pose_rest, with_pose, xyz_rest,
                                           R3D_F32, rgb_rest, rgba_rest)))
      return rc;
    return r3d_voxelset_insert_path(vs, d_xyz_out, total, 2);
  }
  if ((rc = r3d_voxelset_insert_path(vs, d_xyz_out, hw * f0, 1))) return rc;
  if (f0 == n_frames) return R3D_OK;
  return fuse_common(ctx, cam, depth_rest, depth_dtype, n_frames - f0, depth_scale, pose_rest, with_pose, xyz_rest, R3D_F32, rgb_rest,
                     rgba_rest, vs);
}

int r3d_fuse_frames_rgb_host(r3d_ctx* ctx, const r3d_camera* cam, const void* h_depth, int depth_dtype, int n_frames,
                             double depth_scale, const double* h_pose, const unsigned char* h_rgb, void* h_xyz_out,
                             int out_dtype, uint32_t* h_rgba_out) {
  return fuse_rgb_host(ctx, cam, h_depth, depth_dtype, n_frames, depth_scale, h_pose, h_rgb, h_xyz_out, out_dtype, h_rgba_out);
}

}  // extern "C"
