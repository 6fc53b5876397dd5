// A/B harness for candidate kernels that are NOT (yet) in libr3d_hip.so: built by tools/Makefile into tools/ab_kernels,
// run on the GPU box.  Every candidate is checked bit for bit against the library's kernel of the same op before it is
// timed; timings are interleaved rounds in one process (HIP events, 200 launches per sample).
//   fuse -> f64 xyz (25 B/point): library (LDS-tile) vs lane-per-pixel with dwordx4+dwordx2 nontemporal stores vs
//                                 wave-private LDS transposition
//   apply-T f32 -> f32 (24 B/point): library vs persistent 2-stage pipelined vs nontemporal loads
//   fuse + colour (u8 depth + u8 rgb[3] -> f32 xyz + u8 rgba[4]): byte loads vs LDS-staged dword loads
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#include "r3d.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)
#define RK(x)                                                                \
  do {                                                                       \
    int rc_ = (x);                                                           \
    if (rc_ != 0) {                                                          \
      fprintf(stderr, "r3d error %d: %s at %s:%d\n", rc_, r3d_last_error(), __FILE__, __LINE__); \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

constexpr int kThreads = 256;
constexpr int kPx = 4;
constexpr int kTile = kThreads * kPx;

struct Dims {
  uint32_t hw, width, tiles_per_frame, total_tiles;
};
struct Pose {
  double r[9], t[3];
};

__device__ __forceinline__ void load_pose(const double* __restrict__ pose, uint32_t frame, Pose& P) {
  const double* pp = pose + (uint64_t)frame * 12;
#pragma unroll
  for (int k = 0; k < 9; ++k) P.r[k] = pp[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) P.t[k] = pp[9 + k];
}

__device__ __forceinline__ void point(double z, double u, double v, const Pose& p, double o[3]) {
  const double x = u * z, y = v * z;
  const double dx = x - p.t[0], dy = y - p.t[1], dz = z - p.t[2];
  o[0] = fma(p.r[2], dz, fma(p.r[1], dy, p.r[0] * dx));
  o[1] = fma(p.r[5], dz, fma(p.r[4], dy, p.r[3] * dx));
  o[2] = fma(p.r[8], dz, fma(p.r[7], dy, p.r[6] * dx));
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));

// ---- fuse -> f32 (the headline kernel) with PX pixels per lane, one tile per workgroup, clamped loads
template <int PX>
__global__ __launch_bounds__(kThreads) void fuse32_lane(const uint8_t* __restrict__ depth, float* __restrict__ out,
                                                        const double* __restrict__ u, const double* __restrict__ v,
                                                        const double* __restrict__ pose, const Dims dm, uint32_t tiles_per_frame,
                                                        uint32_t total_tiles) {
  constexpr uint32_t tile_px = kThreads * PX;
  const uint32_t tid = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const uint32_t frame = tile / tiles_per_frame;
    const uint32_t tf = tile - frame * tiles_per_frame;
    Pose P;
    load_pose(pose, frame, P);
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    uint8_t raw[PX];
#pragma unroll
    for (int r = 0; r < PX; ++r) raw[r] = depth[fbase + min(tf * tile_px + r * kThreads + tid, dm.hw - 1)];
#pragma unroll
    for (int r = 0; r < PX; ++r) {
      const uint32_t p = tf * tile_px + r * kThreads + tid;
      if (p < dm.hw) {
        const uint32_t j = p / dm.width, i = p - j * dm.width;
        double w[3];
        point((double)raw[r], u[i], v[j], P, w);
        asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(out + (fbase + p) * 3),
                     "v"(f32x3{(float)w[0], (float)w[1], (float)w[2]})
                     : "memory");
      }
    }
  }
}

// ---- fuse -> f64, candidate B: lane-per-pixel, one 16-byte + one 8-byte nontemporal store per lane (24-B lane stride)
template <int MODE>  // 0: x4 nt + x2 nt, 1: three x2 nt, 2: x4 + x2 plain
__global__ __launch_bounds__(kThreads) void fuse64_lane(const uint8_t* __restrict__ depth, double* __restrict__ out,
                                                        const double* __restrict__ u, const double* __restrict__ v,
                                                        const double* __restrict__ pose, const Dims dm) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = tile / dm.tiles_per_frame;
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose(pose, frame, P);
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    uint8_t raw[kPx];
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t p = tf * kTile + r * kThreads + tid;
      raw[r] = p < dm.hw ? depth[fbase + p] : 0;
    }
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t p = tf * kTile + r * kThreads + tid;
      if (p < dm.hw) {
        const uint32_t j = p / dm.width, i = p - j * dm.width;
        double w[3];
        point((double)raw[r], u[i], v[j], P, w);
        double* dst = out + (fbase + p) * 3;
        if (MODE == 0) {
          // 24-byte rows: 8-byte aligned only, the ISA allows dwordx4 stores at 4-byte alignment
          asm volatile("global_store_dwordx4 %0, %1, off nt\n\tglobal_store_dwordx2 %0, %2, off offset:16 nt" ::"v"(dst),
                       "v"(f64x2{w[0], w[1]}), "v"(w[2])
                       : "memory");
        } else if (MODE == 1) {
          __builtin_nontemporal_store(w[0], dst);
          __builtin_nontemporal_store(w[1], dst + 1);
          __builtin_nontemporal_store(w[2], dst + 2);
        } else {
          asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx2 %0, %2, off offset:16" ::"v"(dst),
                       "v"(f64x2{w[0], w[1]}), "v"(w[2])
                       : "memory");
        }
      }
    }
  }
}

// ---- fuse -> f64, candidate P: TWO lanes per pixel, each computes two of the three world components (even lane rows
// 0,1; odd lane rows 1,2) and stores ITS 12-byte half of the 24-byte row with one global_store_dwordx3 nt: every wave
// instruction writes 768 contiguous bytes -- the shape that makes the f32 kernel fast -- with no LDS and no shuffles,
// at the price of ~1.3x the fp64 arithmetic per pixel.
template <int ITEMS>  // (pixel, half) items per lane per tile; tile = 256*ITEMS/2 pixels
__global__ __launch_bounds__(kThreads) void fuse64_pair(const uint8_t* __restrict__ depth, double* __restrict__ out,
                                                        const double* __restrict__ u, const double* __restrict__ v,
                                                        const double* __restrict__ pose, const Dims dm, uint32_t tiles_per_frame,
                                                        uint32_t total_tiles) {
  constexpr uint32_t tile_px = kThreads * ITEMS / 2;
  const uint32_t tid = threadIdx.x;
  const bool odd = tid & 1u;
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const uint32_t frame = tile / tiles_per_frame;
    const uint32_t tf = tile - frame * tiles_per_frame;
    Pose P;
    load_pose(pose, frame, P);
    // this lane's two rows: (0,1) or (1,2)
    double ra[3], rb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ra[c] = odd ? P.r[3 + c] : P.r[c];
      rb[c] = odd ? P.r[6 + c] : P.r[3 + c];
    }
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    uint8_t raw[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const uint32_t p = tf * tile_px + ((r * kThreads + tid) >> 1);
      raw[r] = p < dm.hw ? depth[fbase + p] : 0;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const uint32_t q = r * kThreads + tid;
      const uint32_t p = tf * tile_px + (q >> 1);
      if (p < dm.hw) {
        const uint32_t j = p / dm.width, i = p - j * dm.width;
        const double z = (double)raw[r];
        const double x = u[i] * z, y = v[j] * z;
        const double dx = x - P.t[0], dy = y - P.t[1], dz = z - P.t[2];
        const double a = fma(ra[2], dz, fma(ra[1], dy, ra[0] * dx));
        const double b = fma(rb[2], dz, fma(rb[1], dy, rb[0] * dx));
        const uint32_t alo = (uint32_t)__double2loint(a), ahi = (uint32_t)__double2hiint(a);
        const uint32_t blo = (uint32_t)__double2loint(b), bhi = (uint32_t)__double2hiint(b);
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        // even: x_lo x_hi y_lo   odd: y_hi z_lo z_hi  (odd lane: a = y, b = z)
        const u32x3 val = odd ? u32x3{ahi, blo, bhi} : u32x3{alo, ahi, blo};
        uint32_t* dst = reinterpret_cast<uint32_t*>(out + fbase * 3) + (uint64_t)(tf * tile_px) * 6 + (uint64_t)q * 3;
        asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(val) : "memory");
      }
    }
  }
}

// ---- fuse -> f64, candidate E: every wave transposes its 64 points x 24 B through a private LDS slice and writes
// 1536 contiguous bytes as 16-byte pieces (one full + one half-wave store), no workgroup barrier
template <bool NT>
__global__ __launch_bounds__(kThreads) void fuse64_wave(const uint8_t* __restrict__ depth, double* __restrict__ out,
                                                        const double* __restrict__ u, const double* __restrict__ v,
                                                        const double* __restrict__ pose, const Dims dm) {
  __shared__ __attribute__((aligned(16))) double lds_all[kThreads / 64][64 * 3];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  double* lds = lds_all[wave];
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = tile / dm.tiles_per_frame;
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose(pose, frame, P);
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    uint8_t raw[kPx];
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t p = tf * kTile + r * kThreads + tid;
      raw[r] = p < dm.hw ? depth[fbase + p] : 0;
    }
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t p0 = tf * kTile + r * kThreads + wave * 64;  // first pixel of this wave's 64 in this round
      const uint32_t p = p0 + lane;
      if (p0 >= dm.hw) continue;  // wave-uniform
      double w[3] = {0, 0, 0};
      if (p < dm.hw) {
        const uint32_t j = p / dm.width, i = p - j * dm.width;
        point((double)raw[r], u[i], v[j], P, w);
      }
      lds[lane * 3 + 0] = w[0];
      lds[lane * 3 + 1] = w[1];
      lds[lane * 3 + 2] = w[2];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t n_px = min(64u, dm.hw - p0);
      const uint32_t n_pieces = n_px * 3 / 2;  // 16-byte pieces (n_px even for every raster used here)
      char* base = reinterpret_cast<char*>(out + (fbase + p0) * 3);
      const f64x2* src = reinterpret_cast<const f64x2*>(lds);
      if (lane < n_pieces) {
        if (NT) __builtin_nontemporal_store(src[lane], reinterpret_cast<f64x2*>(base) + lane);
        else reinterpret_cast<f64x2*>(base)[lane] = src[lane];
      }
      if (lane + 64 < n_pieces) {
        if (NT) __builtin_nontemporal_store(src[lane + 64], reinterpret_cast<f64x2*>(base) + lane + 64);
        else reinterpret_cast<f64x2*>(base)[lane + 64] = src[lane + 64];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// ---- apply-T f32 -> f32 candidates -------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) P3 {
  float x, y, z;
};
struct Aff {
  double T[12];
};
__device__ __forceinline__ void aff(const Aff& a, const P3& p, float* dst) {
  const double x = p.x, y = p.y, z = p.z;
  double w[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) w[c] = fma(a.T[4 * c + 2], z, fma(a.T[4 * c + 1], y, a.T[4 * c + 0] * x)) + a.T[4 * c + 3];
  asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(f32x3{(float)w[0], (float)w[1], (float)w[2]}) : "memory");
}
// persistent: grid-stride over tiles, the NEXT tile's 4 loads are in flight while the current tile is computed/stored
template <int PTS>
__global__ __launch_bounds__(kThreads) void apply_pipe(const float* __restrict__ in, float* __restrict__ out, uint64_t n,
                                                       const Aff a) {
  constexpr uint64_t tile_pts = (uint64_t)kThreads * PTS;
  const uint64_t n_tiles = (n + tile_pts - 1) / tile_pts;
  uint64_t tile = blockIdx.x;
  P3 nxt[PTS];
  if (tile < n_tiles) {
#pragma unroll
    for (int r = 0; r < PTS; ++r) {
      const uint64_t i = tile * tile_pts + (uint64_t)r * kThreads + threadIdx.x;
      if (i < n) nxt[r] = reinterpret_cast<const P3*>(in)[i];
    }
  }
  while (tile < n_tiles) {
    P3 cur[PTS];
#pragma unroll
    for (int r = 0; r < PTS; ++r) cur[r] = nxt[r];
    const uint64_t t_next = tile + gridDim.x;
    if (t_next < n_tiles) {
#pragma unroll
      for (int r = 0; r < PTS; ++r) {
        const uint64_t i = t_next * tile_pts + (uint64_t)r * kThreads + threadIdx.x;
        if (i < n) nxt[r] = reinterpret_cast<const P3*>(in)[i];
      }
    }
#pragma unroll
    for (int r = 0; r < PTS; ++r) {
      const uint64_t i = tile * tile_pts + (uint64_t)r * kThreads + threadIdx.x;
      if (i < n) aff(a, cur[r], out + i * 3);
    }
    tile = t_next;
  }
}

// one tile per workgroup, PTS points per lane, all loads up front (the library's shape with a different PTS)
template <int PTS>
__global__ __launch_bounds__(kThreads) void apply_flat(const float* __restrict__ in, float* __restrict__ out, uint64_t n,
                                                       const Aff a) {
  const uint64_t base = (uint64_t)blockIdx.x * kThreads * PTS + threadIdx.x;
  P3 p[PTS];
#pragma unroll
  for (int r = 0; r < PTS; ++r) {
    const uint64_t i = base + (uint64_t)r * kThreads;
    if (i < n) p[r] = reinterpret_cast<const P3*>(in)[i];
  }
#pragma unroll
  for (int r = 0; r < PTS; ++r) {
    const uint64_t i = base + (uint64_t)r * kThreads;
    if (i < n) aff(a, p[r], out + i * 3);
  }
}

template <int PTS, bool NTL>
__global__ __launch_bounds__(kThreads) void apply_flat_nt(const float* __restrict__ in, float* __restrict__ out, uint64_t n,
                                                          const Aff a) {
  const uint64_t base = (uint64_t)blockIdx.x * kThreads * PTS + threadIdx.x;
  f32x3 p[PTS];
#pragma unroll
  for (int r = 0; r < PTS; ++r) {
    const uint64_t i = base + (uint64_t)r * kThreads;
    if (i < n) {
      if (NTL) asm volatile("global_load_dwordx3 %0, %1, off nt" : "=v"(p[r]) : "v"(in + i * 3) : "memory");
      else asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(p[r]) : "v"(in + i * 3) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < PTS; ++r) {
    const uint64_t i = base + (uint64_t)r * kThreads;
    if (i < n) aff(a, P3{p[r].x, p[r].y, p[r].z}, out + i * 3);
  }
}

// LDS-free 16-byte loads: lane q of a tile reads piece q (16 B) of the tile's 12 KiB; a point's 3 floats straddle
// pieces, so the loads go to LDS and points are read back from there (reads wide, stores dwordx3 nt)
__global__ __launch_bounds__(kThreads) void apply_ldsread(const float* __restrict__ in, float* __restrict__ out, uint64_t n,
                                                          const Aff a) {
  __shared__ __attribute__((aligned(16))) float lds[kTile * 3];
  const uint64_t n_tiles = n / kTile;  // whole tiles only (the harness uses n % 1024 == 0)
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const f32x4* src = reinterpret_cast<const f32x4*>(in + tile * kTile * 3);
#pragma unroll
    for (int k = 0; k < 3; ++k) reinterpret_cast<f32x4*>(lds)[k * kThreads + threadIdx.x] = src[k * kThreads + threadIdx.x];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t l = r * kThreads + threadIdx.x;
      const P3 p = {lds[l * 3], lds[l * 3 + 1], lds[l * 3 + 2]};
      aff(a, p, out + (tile * kTile + l) * 3);
    }
    __syncthreads();
  }
}

// ---- fuse + colour: u8 depth + u8 rgb[3] -> f32 xyz + u8 rgba[4] (alpha 0) -----------------------------------------
template <int MODE>  // 0: three byte loads per pixel; 1: tile's 3072 rgb bytes through LDS as dwordx4 loads
__global__ __launch_bounds__(kThreads) void fuse_rgb(const uint8_t* __restrict__ depth, const uint8_t* __restrict__ rgb,
                                                     float* __restrict__ out, uint32_t* __restrict__ rgba,
                                                     const double* __restrict__ u, const double* __restrict__ v,
                                                     const double* __restrict__ pose, const Dims dm) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[MODE == 1 ? kTile * 3 : 16];
  const uint32_t tid = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < dm.total_tiles; tile += gridDim.x) {
    const uint32_t frame = tile / dm.tiles_per_frame;
    const uint32_t tf = tile - frame * dm.tiles_per_frame;
    Pose P;
    load_pose(pose, frame, P);
    const uint64_t fbase = (uint64_t)frame * dm.hw;
    uint8_t raw[kPx];
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t p = tf * kTile + r * kThreads + tid;
      raw[r] = p < dm.hw ? depth[fbase + p] : 0;
    }
    if (MODE == 1) {
      // whole tiles only on this path (hw % 1024 == 0 in the harness): 192 lanes move 16 B each
      if (tid < kTile * 3 / 16)
        reinterpret_cast<uint4*>(lds)[tid] = reinterpret_cast<const uint4*>(rgb + (fbase + (uint64_t)tf * kTile) * 3)[tid];
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < kPx; ++r) {
      const uint32_t l = r * kThreads + tid;
      const uint32_t p = tf * kTile + l;
      if (p < dm.hw) {
        const uint32_t j = p / dm.width, i = p - j * dm.width;
        double w[3];
        point((double)raw[r], u[i], v[j], P, w);
        asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(out + (fbase + p) * 3),
                     "v"(f32x3{(float)w[0], (float)w[1], (float)w[2]})
                     : "memory");
        uint32_t c;
        if (MODE == 1) {
          c = (uint32_t)lds[l * 3] | ((uint32_t)lds[l * 3 + 1] << 8) | ((uint32_t)lds[l * 3 + 2] << 16);
        } else {
          const uint8_t* s = rgb + (fbase + p) * 3;
          c = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
        }
        __builtin_nontemporal_store(c, rgba + fbase + p);
      }
    }
    if (MODE == 1) __syncthreads();
  }
}

// -------------------------------------------------------------------------------------------------------------------
template <typename F>
float time_ms(hipStream_t st, F&& launch, int iters = 200) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return ms / iters;
}

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 384, W = argc > 2 ? atoi(argv[2]) : 1280, F = argc > 3 ? atoi(argv[3]) : 100;
  const uint64_t hw = (uint64_t)H * W, n = hw * F;
  r3d_ctx* ctx = nullptr;
  RK(r3d_ctx_create(0, nullptr, 0, &ctx));
  void* stv = nullptr;
  RK(r3d_ctx_stream(ctx, &stv));
  hipStream_t st = (hipStream_t)stv;
  r3d_camera* cam = nullptr;
  RK(r3d_camera_create(ctx, H, W, 600.391, 600.079, 320, 240, &cam));
  std::mt19937 rng(1234);
  std::vector<uint8_t> depth(n), rgb(n * 3);
  for (auto& d : depth) d = (uint8_t)(1 + rng() % 255);
  for (auto& c : rgb) c = (uint8_t)(rng() & 255);
  std::vector<double> pose((size_t)F * 12), u(W), v(H);
  std::normal_distribution<double> nd;
  for (int f = 0; f < F; ++f) {
    double q[4], nn = 0;
    for (auto& x : q) { x = nd(rng); nn += x * x; }
    nn = sqrt(nn);
    const double x = q[0] / nn, y = q[1] / nn, z = q[2] / nn, w = q[3] / nn;
    double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                   2 * (y * z - x * w),     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    for (int k = 0; k < 9; ++k) pose[f * 12 + k] = R[(k % 3) * 3 + k / 3];  // transpose = inverse
    for (int k = 0; k < 3; ++k) pose[f * 12 + 9 + k] = nd(rng) * 10;
  }
  for (int i = 0; i < W; ++i) u[i] = ((double)i - 320) / 600.391;
  for (int j = 0; j < H; ++j) v[j] = ((double)j - 240) / 600.079;
  uint8_t *d_depth, *d_rgb;
  double *d_pose, *d_u, *d_v, *d_ref64, *d_out64;
  float *d_ref32, *d_out32, *d_in32;
  uint32_t* d_rgba;
  CK(hipMalloc(&d_depth, n));
  CK(hipMalloc(&d_rgb, n * 3));
  CK(hipMalloc(&d_pose, pose.size() * 8));
  CK(hipMalloc(&d_u, W * 8));
  CK(hipMalloc(&d_v, H * 8));
  CK(hipMalloc(&d_ref64, n * 24));
  CK(hipMalloc(&d_out64, n * 24));
  CK(hipMalloc(&d_ref32, n * 12));
  CK(hipMalloc(&d_out32, n * 12));
  CK(hipMalloc(&d_in32, n * 12));
  CK(hipMalloc(&d_rgba, n * 4));
  CK(hipMemcpy(d_depth, depth.data(), n, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rgb, rgb.data(), n * 3, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_pose, pose.data(), pose.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_u, u.data(), W * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_v, v.data(), H * 8, hipMemcpyHostToDevice));
  Dims dm{(uint32_t)hw, (uint32_t)W, (uint32_t)((hw + kTile - 1) / kTile), 0};
  dm.total_tiles = dm.tiles_per_frame * F;
  int cus = 256;
  const int grid8 = cus * 8;

  auto same = [&](const void* a, const void* b, size_t bytes, const char* what) {
    std::vector<char> ha(bytes), hb(bytes);
    CK(hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost));
    const bool ok = memcmp(ha.data(), hb.data(), bytes) == 0;
    printf("  %-34s %s\n", what, ok ? "bit-identical" : "MISMATCH");
    return ok;
  };

  // ---------------- fuse -> f32: pixels per lane at one tile per workgroup
  printf("== fuse u8 -> f32 xyz (headline), 13 B/point = %.1f MB\n", n * 13 / 1e6);
  RK(r3d_fuse_frames(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, d_pose, d_ref32, R3D_F32));
  CK(hipStreamSynchronize(st));
  {
    struct Cand32 { const char* name; std::function<void()> fn; };
    auto tp = [&](int px) { return (uint32_t)((hw + 256 * px - 1) / (256 * px)); };
    std::vector<Cand32> c32 = {
        {"library", [&] { RK(r3d_fuse_frames(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, d_pose, d_out32, R3D_F32)); }},
        {"1 px/lane, 1 tile/WG", [&] { hipLaunchKernelGGL(fuse32_lane<1>, dim3(tp(1) * F), dim3(kThreads), 0, st, d_depth, d_out32, d_u, d_v, d_pose, dm, tp(1), tp(1) * F); }},
        {"2 px/lane, 1 tile/WG", [&] { hipLaunchKernelGGL(fuse32_lane<2>, dim3(tp(2) * F), dim3(kThreads), 0, st, d_depth, d_out32, d_u, d_v, d_pose, dm, tp(2), tp(2) * F); }},
        {"4 px/lane, 1 tile/WG", [&] { hipLaunchKernelGGL(fuse32_lane<4>, dim3(tp(4) * F), dim3(kThreads), 0, st, d_depth, d_out32, d_u, d_v, d_pose, dm, tp(4), tp(4) * F); }},
        {"8 px/lane, 1 tile/WG", [&] { hipLaunchKernelGGL(fuse32_lane<8>, dim3(tp(8) * F), dim3(kThreads), 0, st, d_depth, d_out32, d_u, d_v, d_pose, dm, tp(8), tp(8) * F); }},
        {"16 px/lane, 1 tile/WG", [&] { hipLaunchKernelGGL(fuse32_lane<16>, dim3(tp(16) * F), dim3(kThreads), 0, st, d_depth, d_out32, d_u, d_v, d_pose, dm, tp(16), tp(16) * F); }},
    };
    for (auto& c : c32) {
      CK(hipMemsetAsync(d_out32, 0xff, n * 12, st));
      c.fn();
      CK(hipStreamSynchronize(st));
      same(d_out32, d_ref32, n * 12, c.name);
    }
    for (int round = 0; round < 3; ++round)
      for (auto& c : c32) {
        const float ms = time_ms(st, c.fn, 400);
        printf("  round %d  %-32s %.4f ms  %.2f TB/s  (%.3f of 8)\n", round, c.name, ms, n * 13 / ms / 1e9, n * 13 / ms / 1e9 / 8);
      }
  }

  // ---------------- fuse -> f64
  printf("== fuse u8 -> f64 xyz, %dx%d x %d frames, 25 B/point = %.1f MB\n", W, H, F, n * 25 / 1e6);
  RK(r3d_fuse_frames(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, d_pose, d_ref64, R3D_F64));
  CK(hipStreamSynchronize(st));
  struct Cand { const char* name; std::function<void()> fn; };
  std::vector<Cand> c64 = {
      {"library (LDS tile, 1 tile/WG)", [&] { RK(r3d_fuse_frames(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, d_pose, d_out64, R3D_F64)); }},
      {"lane x4+x2 nt, 8 WG/CU", [&] { hipLaunchKernelGGL(fuse64_lane<0>, dim3(grid8), dim3(kThreads), 0, st, d_depth, d_out64, d_u, d_v, d_pose, dm); }},
      {"wave-LDS 16B nt, 8 WG/CU", [&] { hipLaunchKernelGGL(fuse64_wave<true>, dim3(grid8), dim3(kThreads), 0, st, d_depth, d_out64, d_u, d_v, d_pose, dm); }},
      {"wave-LDS 16B nt, 1 tile/WG", [&] { hipLaunchKernelGGL(fuse64_wave<true>, dim3(dm.total_tiles), dim3(kThreads), 0, st, d_depth, d_out64, d_u, d_v, d_pose, dm); }},
      {"pair x3 nt, 8 items, 8 WG/CU", [&] { const uint32_t tpf = (uint32_t)((hw + 1023) / 1024); hipLaunchKernelGGL(fuse64_pair<8>, dim3(grid8), dim3(kThreads), 0, st, d_depth, d_out64, d_u, d_v, d_pose, dm, tpf, tpf * F); }},
      {"pair x3 nt, 8 items, 1 tile/WG", [&] { const uint32_t tpf = (uint32_t)((hw + 1023) / 1024); hipLaunchKernelGGL(fuse64_pair<8>, dim3(tpf * F), dim3(kThreads), 0, st, d_depth, d_out64, d_u, d_v, d_pose, dm, tpf, tpf * F); }},
      {"pair x3 nt, 4 items, 8 WG/CU", [&] { const uint32_t tpf = (uint32_t)((hw + 511) / 512); hipLaunchKernelGGL(fuse64_pair<4>, dim3(grid8), dim3(kThreads), 0, st, d_depth, d_out64, d_u, d_v, d_pose, dm, tpf, tpf * F); }},
      {"pair x3 nt, 4 items, 16 WG/CU", [&] { const uint32_t tpf = (uint32_t)((hw + 511) / 512); hipLaunchKernelGGL(fuse64_pair<4>, dim3(grid8 * 2), dim3(kThreads), 0, st, d_depth, d_out64, d_u, d_v, d_pose, dm, tpf, tpf * F); }},
  };
  for (auto& c : c64) {
    CK(hipMemsetAsync(d_out64, 0xff, n * 24, st));
    c.fn();
    CK(hipStreamSynchronize(st));
    same(d_out64, d_ref64, n * 24, c.name);
  }
  for (int round = 0; round < 3; ++round)
    for (auto& c : c64) {
      const float ms = time_ms(st, c.fn);
      printf("  round %d  %-32s %.4f ms  %.2f TB/s  (%.3f of 8)\n", round, c.name, ms, n * 25 / ms / 1e9, n * 25 / ms / 1e9 / 8);
    }

  // ---------------- apply-T
  printf("== apply-T f32 -> f32, %.1f M points, 24 B/point = %.1f MB\n", n / 1e6, n * 24 / 1e6);
  RK(r3d_fuse_frames(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, d_pose, d_in32, R3D_F32));
  double T[16] = {1.7 * 0.98, -1.7 * 0.17, 0.05, 1, 1.7 * 0.17, 1.7 * 0.98, -0.03, 2, -0.04, 0.02, 1.69, 3, 0, 0, 0, 1};
  Aff a;
  for (int k = 0; k < 12; ++k) a.T[k] = T[k];
  RK(r3d_apply_T(ctx, d_in32, R3D_F32, (int64_t)n, T, d_ref32, R3D_F32));
  CK(hipStreamSynchronize(st));
  const unsigned t4 = (unsigned)((n + 1023) / 1024), t8 = (unsigned)((n + 2047) / 2048), t2 = (unsigned)((n + 511) / 512);
  std::vector<Cand> ca = {
      {"library (lane, 1 tile/WG)", [&] { RK(r3d_apply_T(ctx, d_in32, R3D_F32, (int64_t)n, T, d_out32, R3D_F32)); }},
      {"flat 4 pts/lane", [&] { hipLaunchKernelGGL(apply_flat<4>, dim3(t4), dim3(kThreads), 0, st, d_in32, d_out32, n, a); }},
      {"flat 2 pts/lane", [&] { hipLaunchKernelGGL(apply_flat<2>, dim3(t2), dim3(kThreads), 0, st, d_in32, d_out32, n, a); }},
      {"flat 1 pt/lane", [&] { hipLaunchKernelGGL(apply_flat<1>, dim3((unsigned)((n + 255) / 256)), dim3(kThreads), 0, st, d_in32, d_out32, n, a); }},
      {"flat 2 pts, nt loads", [&] { hipLaunchKernelGGL((apply_flat_nt<2, true>), dim3(t2), dim3(kThreads), 0, st, d_in32, d_out32, n, a); }},
      {"flat 4 pts, nt loads", [&] { hipLaunchKernelGGL((apply_flat_nt<4, true>), dim3(t4), dim3(kThreads), 0, st, d_in32, d_out32, n, a); }},
      {"flat 2 pts, asm plain loads", [&] { hipLaunchKernelGGL((apply_flat_nt<2, false>), dim3(t2), dim3(kThreads), 0, st, d_in32, d_out32, n, a); }},
      {"LDS-staged 16B reads, 1 tile/WG", [&] { hipLaunchKernelGGL(apply_ldsread, dim3(t4), dim3(kThreads), 0, st, d_in32, d_out32, n, a); }},
  };
  for (auto& c : ca) {
    CK(hipMemsetAsync(d_out32, 0xff, n * 12, st));
    c.fn();
    CK(hipStreamSynchronize(st));
    same(d_out32, d_ref32, n * 12, c.name);
  }
  for (int round = 0; round < 3; ++round)
    for (auto& c : ca) {
      const float ms = time_ms(st, c.fn);
      printf("  round %d  %-32s %.4f ms  %.2f TB/s  (%.3f of 8)\n", round, c.name, ms, n * 24 / ms / 1e9, n * 24 / ms / 1e9 / 8);
    }

  // ---------------- fuse + colour
  printf("== fuse u8 depth + u8 rgb -> f32 xyz + u8 rgba, 20 B/point = %.1f MB\n", n * 20 / 1e6);
  RK(r3d_fuse_frames(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, d_pose, d_ref32, R3D_F32));
  CK(hipStreamSynchronize(st));
  std::vector<Cand> cc = {
      {"xyz only (library)", [&] { RK(r3d_fuse_frames(ctx, cam, d_depth, R3D_DEPTH_U8, F, 1.0, d_pose, d_out32, R3D_F32)); }},
      {"rgb byte loads, 8 WG/CU", [&] { hipLaunchKernelGGL(fuse_rgb<0>, dim3(grid8), dim3(kThreads), 0, st, d_depth, d_rgb, d_out32, d_rgba, d_u, d_v, d_pose, dm); }},
      {"rgb via LDS 16B, 8 WG/CU", [&] { hipLaunchKernelGGL(fuse_rgb<1>, dim3(grid8), dim3(kThreads), 0, st, d_depth, d_rgb, d_out32, d_rgba, d_u, d_v, d_pose, dm); }},
  };
  std::vector<uint32_t> want_rgba(n);
  for (uint64_t k = 0; k < n; ++k) want_rgba[k] = rgb[3 * k] | (rgb[3 * k + 1] << 8) | (rgb[3 * k + 2] << 16);
  for (size_t k = 1; k < cc.size(); ++k) {
    CK(hipMemsetAsync(d_out32, 0xff, n * 12, st));
    CK(hipMemsetAsync(d_rgba, 0xff, n * 4, st));
    cc[k].fn();
    CK(hipStreamSynchronize(st));
    same(d_out32, d_ref32, n * 12, cc[k].name);
    std::vector<uint32_t> got(n);
    CK(hipMemcpy(got.data(), d_rgba, n * 4, hipMemcpyDeviceToHost));
    printf("  %-34s rgba %s\n", cc[k].name, got == want_rgba ? "identical" : "MISMATCH");
  }
  for (int round = 0; round < 3; ++round)
    for (size_t k = 0; k < cc.size(); ++k) {
      const float ms = time_ms(st, cc[k].fn);
      const double bytes = (double)n * (k == 0 ? 13 : 20);
      printf("  round %d  %-32s %.4f ms  %.2f TB/s  (%.3f of 8)\n", round, cc[k].name, ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8);
    }
  r3d_camera_destroy(cam);
  r3d_ctx_destroy(ctx);
  return 0;
}
