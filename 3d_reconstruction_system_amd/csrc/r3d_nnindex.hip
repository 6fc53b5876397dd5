// Exact nearest neighbour with spatial culling for gfx950 (MI355X): the brute-force LDS-tiled sweep of
// r3d_icp.hip, but over Morton-sorted clouds so that whole 1024-target tiles are skipped when their
// bounding box is provably farther than the best distance found so far.
//
// NOT IN THE REFERENCE (ICP estimation is build-defined).  Results are identical to r3d_icp_nn:
// the same fp32 expression d2 = fma(dz,dz,fma(dy,dy,dx*dx)) decides, lowest ORIGINAL target index wins exact ties.
//
//   build (once per target cloud):  bounding box -> 64-bit keys (Morton code of the quantised point | index) ->
//       GPU radix sort (r3d_sort.hip) -> targets gathered in key order as float4 (x,y,z,original index) ->
//       per-tile bounding boxes (tile = 1024 consecutive sorted targets) and first Morton code of every tile.
//   query:  sources get the same keys and sort, so a workgroup's 256*S sources are spatially compact.  The
//       workgroup starts at the tile whose Morton range contains its first source and walks outward
//       (t0, t0+1, t0-1, ...).  Per tile every lane computes a lower bound of its distance to the tile's box;
//       the tile is loaded into LDS and swept only if some lane of the workgroup could still improve
//       (bound * (1 - 16u) <= best).  The sweep is the r3d_icp.hip inner loop: 6 VALU + 1/2 min3 per pair,
//       minimum tracked per group of 32 sorted targets.
//       Inside a swept tile the wave skips 256-target quarters and then 32-target groups whose boxes (the group boxes
//       ride into LDS with the tile) cannot improve any lane: at 500k x 500k near alignment 15 tiles are swept per
//       workgroup but only ~940 targets evaluated per source (brute force: 500,000), 0.32 ms per query.
//   resolve: the winning group is re-evaluated exactly; equal distances pick the lowest original index.  A source
//       that sees the SAME minimum again in another group (an exact tie across groups) looks through that group on the
//       spot and remembers the lowest original index among its equal targets, so the lowest index overall still wins
//       (round 2 handed such sources to a fallback kernel in which ONE wave rescanned every target: a single tie cost
//       more than a millisecond at 300k targets -- 70 % of a point-to-plane iteration on organised clouds).
#include <cmath>

#include "r3d_icp_sums.h"
#include <cstdlib>

#include "r3d_internal.h"

struct r3d_nn_index {
  r3d_ctx* ctx = nullptr;
  int device = 0;  // kept so that destroy never has to touch a ctx that may already be gone
  int64_t n = 0;
  int64_t capacity = 0;  // target points the allocations can hold (r3d_nn_index_rebuild reuses them)
  int64_t n_tiles = 0;
  int idx_bits = 1, axis_bits = 16;
  float* d_tgt = nullptr;      // [n][3] original order (tie winners of other groups, pair sums, the plane kernels' gathers)
  float4* d_tgt4 = nullptr;    // [n_tiles*1024] sorted, w = original index bits; padding has x = +inf
  float* d_tile_box = nullptr; // [n_tiles][6] lo xyz, hi xyz
  float* d_sub_box = nullptr;  // [n_tiles*4][6] boxes of the 256-target quarters of every tile
  float* d_group_box = nullptr;  // [n_tiles*32][6] boxes of the 32-target groups (staged in LDS with a swept tile)
  float* d_super_box = nullptr; // [ceil(n_tiles/16)][6] boxes of 16 consecutive tiles
  uint64_t* d_tile_code = nullptr;  // [n_tiles] Morton code (without index bits) of the tile's first target
  float* d_frame = nullptr;    // [8]: lo xyz, scale xyz, unused: quantisation frame shared by both clouds
  void* d_slab = nullptr;      // ONE allocation holds every table above (eight hipMalloc / hipFree pairs per index were a
                               // measurable share of a 10 ms estimate)
  // warm start (see nn_cull_kernel): the buffers of the last presorted query against this build of the index
  const float* warm_src = nullptr;
  const uint32_t* warm_idx = nullptr;
  int64_t warm_n = 0;
};

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;
constexpr int kGroup = 32;
constexpr int kSub = 256;  // targets per sub-tile (wave-level culling inside a swept tile)
constexpr int kSuper = 16;  // tiles per super-box (workgroup-level culling of 16 tiles at once)
constexpr float kShrink = 1.0f - 16.0f * 5.9604645e-8f;  // (1 - 16u): makes the box bound a strict lower bound

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct __attribute__((packed, aligned(4))) P3 {
  float x, y, z;
};

// ---- build -------------------------------------------------------------------------------------------------
// Two stages, no atomics: every workgroup leaves its six bounds as one row of `partial`; frame_kernel (one wave) folds the
// rows and derives the quantisation frame.  (Round 2: six same-address atomics per WAVE, 74 us for a 6 MB read.)
__global__ __launch_bounds__(kThreads) void bbox_kernel(const float* __restrict__ xyz, int64_t n, float* __restrict__ partial) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const P3 p = reinterpret_cast<const P3*>(xyz)[i];
    const float v[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a)
      if (isfinite(v[a])) {
        lo[a] = fminf(lo[a], v[a]);
        hi[a] = fmaxf(hi[a], v[a]);
      }
  }
  __shared__ float red[6][kThreads / 64];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      red[a][threadIdx.x >> 6] = lo[a];
      red[3 + a][threadIdx.x >> 6] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[threadIdx.x][0];
    for (int w = 1; w < kThreads / 64; ++w) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
    partial[(int64_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

// one wave: fold the per-workgroup rows (an empty / all-non-finite cloud keeps (+inf, -inf): scale 0), write the frame
__global__ __launch_bounds__(64) void frame_kernel(const float* __restrict__ partial, int n_rows, int axis_bits,
                                                   float* __restrict__ frame) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int r = threadIdx.x; r < n_rows; r += 64) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], partial[r * 6 + a]);
      hi[a] = fmaxf(hi[a], partial[r * 6 + 3 + a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
    }
    if (threadIdx.x == 0) {
      const float ext = hi[a] - lo[a];
      frame[a] = lo[a];
      frame[3 + a] = (ext > 0.f && isfinite(ext)) ? (float)((1 << axis_bits) - 1) / ext : 0.f;
    }
  }
}

__device__ __forceinline__ uint64_t spread3(uint32_t v) {
  uint64_t x = v & 0xffffu;
  x = (x | x << 16) & 0x0000ff0000ffull;
  x = (x | x << 8) & 0x00f00f00f00full;
  x = (x | x << 4) & 0x0c30c30c30c3ull;
  x = (x | x << 2) & 0x249249249249ull;
  return x;
}

__device__ __forceinline__ uint64_t point_code(const P3& p, const float* __restrict__ frame, int axis_bits) {
  const float top = (float)((1 << axis_bits) - 1);
  const float q[3] = {(p.x - frame[0]) * frame[3], (p.y - frame[1]) * frame[4], (p.z - frame[2]) * frame[5]};
  uint32_t k[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) k[a] = (uint32_t)fminf(fmaxf(q[a], 0.f), top);  // NaN -> 0
  return spread3(k[0]) | (spread3(k[1]) << 1) | (spread3(k[2]) << 2);
}

__global__ __launch_bounds__(kThreads) void keys_kernel(const float* __restrict__ xyz, int64_t n, const float* __restrict__ frame,
                                                        int axis_bits, int idx_bits, uint64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const P3 p = reinterpret_cast<const P3*>(xyz)[i];
  keys[i] = (point_code(p, frame, axis_bits) << idx_bits) | (uint64_t)i;
}

// the same keys, with every row that has a NaN / inf coordinate given a code ABOVE all codes (one more key bit): the stable sort
// leaves such rows at the end, in their input order; valid rows are counted (one atomic per workgroup)
__global__ __launch_bounds__(kThreads) void keys_valid_kernel(const float* __restrict__ xyz, int64_t n, const float* __restrict__ frame,
                                                              int axis_bits, int idx_bits, uint64_t* __restrict__ keys,
                                                              unsigned long long* __restrict__ n_valid) {
  __shared__ unsigned wave_cnt[kThreads / 64];
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  bool valid = false;
  if (i < n) {
    const P3 p = reinterpret_cast<const P3*>(xyz)[i];
    valid = (p.x - p.x == 0.f) && (p.y - p.y == 0.f) && (p.z - p.z == 0.f);
    const uint64_t code = valid ? point_code(p, frame, axis_bits) : ((uint64_t)1 << (3 * axis_bits));
    keys[i] = (code << idx_bits) | (uint64_t)i;
  }
  const unsigned c = (unsigned)__popcll(__ballot(valid));
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += wave_cnt[w];
    if (t) atomicAdd(n_valid, (unsigned long long)t);
  }
}

// rows that are exactly (0, 0, 0) -> (NaN, NaN, NaN): the "no depth" pixels of gentxtcord's clouds (p2c:34-44 emits them like any
// other point) stop being points BEFORE a pose moves them somewhere plausible
__global__ __launch_bounds__(kThreads) void zero_rows_to_nan_kernel(float* __restrict__ xyz, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  P3* P = reinterpret_cast<P3*>(xyz);
  const P3 p = P[i];
  if (p.x == 0.f && p.y == 0.f && p.z == 0.f) P[i] = P3{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
}

// sorted keys -> float4 (x, y, z, original index); slots past n are padded with x = +inf so they never win
__global__ __launch_bounds__(kThreads) void gather4_kernel(const float* __restrict__ xyz, const uint64_t* __restrict__ keys,
                                                           int64_t n, int64_t n_padded, int idx_bits, float4* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= n_padded) return;
  if (j < n) {
    const uint32_t i = (uint32_t)(keys[j] & (((uint64_t)1 << idx_bits) - 1));
    const P3 p = reinterpret_cast<const P3*>(xyz)[i];
    out[j] = make_float4(p.x, p.y, p.z, __uint_as_float(i));
  } else {
    out[j] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(0xffffffffu));
  }
}

// one workgroup per tile: bounding box of its (finite) targets and the Morton code of its first target
__global__ __launch_bounds__(kThreads) void tile_box_kernel(const float4* __restrict__ tgt4, const uint64_t* __restrict__ keys,
                                                            int64_t n, int idx_bits, int span, float* __restrict__ tile_box,
                                                            uint64_t* __restrict__ tile_code) {
  __shared__ float red[6][kThreads / 64];
  const int64_t base = (int64_t)blockIdx.x * span;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = threadIdx.x; k < span; k += kThreads) {
    if (base + k < n) {
      const float4 p = tgt4[base + k];
      const float v[3] = {p.x, p.y, p.z};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(lo[a], v[a]);  // fminf/fmaxf drop NaN operands
        hi[a] = fmaxf(hi[a], v[a]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      red[a][threadIdx.x >> 6] = lo[a];
      red[3 + a][threadIdx.x >> 6] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[threadIdx.x][0];
    for (int w = 1; w < kThreads / 64; ++w) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
    tile_box[(int64_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
  if (threadIdx.x == 0 && tile_code) tile_code[blockIdx.x] = keys[base] >> idx_bits;
}

// boxes of the 32-target groups of one tile per workgroup: 8 lanes per group, 4 targets per lane, shuffle-reduced
__global__ __launch_bounds__(kThreads) void group_box_kernel(const float4* __restrict__ tgt4, int64_t n,
                                                             float* __restrict__ group_box) {
  const int64_t base = (int64_t)blockIdx.x * kTile;
  const int g = threadIdx.x >> 3, l = threadIdx.x & 7;  // 32 groups x 8 lanes
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t j = base + g * kGroup + l * 4 + k;
    if (j < n) {
      const float4 p = tgt4[j];
      const float v[3] = {p.x, p.y, p.z};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(lo[a], v[a]);
        hi[a] = fmaxf(hi[a], v[a]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 8));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 8));
    }
  }
  if (l == 0) {
    float* o = group_box + ((int64_t)blockIdx.x * (kTile / kGroup) + g) * 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      o[a] = lo[a];          // an empty group keeps (+inf, -inf): its bound is +inf, it is always skipped
      o[3 + a] = hi[a];
    }
  }
}

// ---- query -------------------------------------------------------------------------------------------------
// SRC4: sources come as float4 (x, y, z, original index) in sorted order (one-off queries sort a copy);
// otherwise plain xyz whose order the caller keeps spatially coherent (r3d_nn_index_sort_cloud) -- results
// then land at the same positions.
template <int S, bool SRC4>
__global__ __launch_bounds__(kThreads) void nn_cull_kernel(const void* __restrict__ src_any, int64_t n_src,
                                                           const float* __restrict__ frame, int axis_bits,
                                                           const float4* __restrict__ tgt4, int64_t n_tgt, int64_t n_tiles,
                                                           const float* __restrict__ tile_box,
                                                           const float* __restrict__ sub_box,
                                                           const float* __restrict__ group_box,
                                                           const float* __restrict__ super_box,
                                                           const uint64_t* __restrict__ tile_code,
                                                           const float* __restrict__ tgt_orig,
                                                           uint32_t* idx_out, float* __restrict__ d2_out,
                                                           unsigned long long* __restrict__ stats,
                                                           double* __restrict__ partials, float max_d2, float dead_zone,
                                                           const uint32_t* idx_warm) {
  __shared__ __attribute__((aligned(16))) float tx[kTile];
  __shared__ __attribute__((aligned(16))) float ty[kTile];
  __shared__ __attribute__((aligned(16))) float tz[kTile];
  __shared__ __attribute__((aligned(16))) float gbox[kTile / kGroup][8];  // lo xyz, hi xyz of every 32-target group (+ pad)
  __shared__ int start_tile;

  const uint32_t tid = threadIdx.x;
  const int64_t s_base = (int64_t)blockIdx.x * (kThreads * S);
  float sx[S], sy[S], sz[S], best[S];
  uint32_t best_group[S], tie_idx[S];   // tie_idx: lowest original index among equal-distance targets of OTHER groups
  bool ok[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t i = s_base + (int64_t)s * kThreads + tid;
    ok[s] = i < n_src;
    if (SRC4) {
      const float4 p = ok[s] ? static_cast<const float4*>(src_any)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      sx[s] = p.x; sy[s] = p.y; sz[s] = p.z;
    } else {
      const P3 p = ok[s] ? static_cast<const P3*>(src_any)[i] : P3{0.f, 0.f, 0.f};
      sx[s] = p.x; sy[s] = p.y; sz[s] = p.z;
    }
    best[s] = INFINITY;
    best_group[s] = 0;
    tie_idx[s] = 0xffffffffu;
  }
  // Warm start (ICP iterations: the same sources, moved a little, against the same index): the distance to the target this
  // source matched LAST time is an upper bound of its nearest-neighbour distance now, whatever the move was -- it is the
  // distance to SOME target.  Starting from a bound STRICTLY above it (so that the true neighbour still wins the `<` below and
  // ties are still resolved by the search itself) every level of culling works from the first tile on instead of waiting for
  // the slowest lane of the workgroup to find something near.  Results are the cold search's, bit for bit; a stale or
  // meaningless idx_warm (it is the previous content of idx_out, clamped to the target count) only makes the bound useless.
  if (!SRC4 && idx_warm) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int64_t i = s_base + (int64_t)s * kThreads + tid;
      if (ok[s]) {
        const uint32_t j = min(idx_warm[i], (uint32_t)(n_tgt - 1));
        const P3 q = reinterpret_cast<const P3*>(tgt_orig)[j];
        const float dx = sx[s] - q.x, dy = sy[s] - q.y, dz = sz[s] - q.z;
        const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        if (d < INFINITY) best[s] = d * 1.000002f + 1.17549435e-38f;   // NaN / inf: no bound
      }
    }
  }
  if (tid == 0) {
    // tile whose Morton range holds this workgroup's first source: last tile with first code <= code
    const uint64_t code = point_code(P3{sx[0], sy[0], sz[0]}, frame, axis_bits);
    int64_t lo = 0, hi = n_tiles;  // answer in [lo, hi)
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (tile_code[mid] <= code) lo = mid; else hi = mid;
    }
    start_tile = (int)lo;
  }
  __syncthreads();
  const int64_t t0 = start_tile;
  unsigned swept = 0, groups_done = 0;

  // two-level outward walk: super-boxes of 16 tiles s0, s0+1, s0-1, ...; a super-box nobody can improve in is
  // skipped with one test + one vote instead of 16; inside a kept one the tiles are visited starting at t0's slot
  const int64_t n_super = (n_tiles + kSuper - 1) / kSuper;
  const int64_t s0 = t0 / kSuper;
  for (int64_t sstep = 0; sstep < 2 * n_super; ++sstep) {
   const int64_t sup = (sstep & 1) ? s0 + ((sstep + 1) >> 1) : s0 - (sstep >> 1);
   if (sup < 0 || sup >= n_super) continue;  // uniform
   {
    const float* sb = super_box + sup * 6;
    bool want = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float ex = fmaxf(fmaxf(sb[0] - sx[s], sx[s] - sb[3]), 0.f);
      const float ey = fmaxf(fmaxf(sb[1] - sy[s], sy[s] - sb[4]), 0.f);
      const float ez = fmaxf(fmaxf(sb[2] - sz[s], sz[s] - sb[5]), 0.f);
      want |= ok[s] && !(fmaf(ez, ez, fmaf(ey, ey, ex * ex)) * kShrink > best[s]);
    }
    if (!__syncthreads_or(want)) continue;
   }
   for (int k = 0; k < kSuper; ++k) {
    const int64_t tile = sup * kSuper + ((t0 + k) & (kSuper - 1));
    if (tile >= n_tiles) continue;  // uniform
    const float* box = tile_box + tile * 6;
    const float blo[3] = {box[0], box[1], box[2]}, bhi[3] = {box[3], box[4], box[5]};
    bool need = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float ex = fmaxf(fmaxf(blo[0] - sx[s], sx[s] - bhi[0]), 0.f);
      const float ey = fmaxf(fmaxf(blo[1] - sy[s], sy[s] - bhi[1]), 0.f);
      const float ez = fmaxf(fmaxf(blo[2] - sz[s], sz[s] - bhi[2]), 0.f);
      const float lb = fmaf(ez, ez, fmaf(ey, ey, ex * ex)) * kShrink;
      need |= ok[s] && !(lb > best[s]);  // NaN bounds never allow a skip
    }
    if (!__syncthreads_or(need)) continue;  // nobody in the workgroup can improve inside this tile
    ++swept;
    const int64_t t_base = tile * kTile;
    for (uint32_t k = tid; k < kTile; k += kThreads) {
      const float4 p = tgt4[t_base + k];
      tx[k] = p.x; ty[k] = p.y; tz[k] = p.z;
    }
    if (tid < (kTile / kGroup) * 6) gbox[tid / 6][tid % 6] = group_box[tile * ((kTile / kGroup) * 6) + tid];
    __syncthreads();
    if (__any(need)) {
      const uint32_t group0 = (uint32_t)(t_base / kGroup);
      for (int g = 0; g < kTile / kGroup; ++g) {
        if ((g & (kSub / kGroup - 1)) == 0) {
          // entering a 256-target quarter: the whole WAVE skips it when no lane can improve inside its box
          const float* sb = sub_box + (tile * (kTile / kSub) + g / (kSub / kGroup)) * 6;
          bool want = false;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const float ex = fmaxf(fmaxf(sb[0] - sx[s], sx[s] - sb[3]), 0.f);
            const float ey = fmaxf(fmaxf(sb[1] - sy[s], sy[s] - sb[4]), 0.f);
            const float ez = fmaxf(fmaxf(sb[2] - sz[s], sz[s] - sb[5]), 0.f);
            want |= ok[s] && !(fmaf(ez, ez, fmaf(ey, ey, ex * ex)) * kShrink > best[s]);
          }
          if (!__any(want)) {
            g += kSub / kGroup - 1;
            continue;
          }
        }
        {
          // third level inside a swept tile: the wave skips a 32-target group no lane can improve in (box from LDS,
          // ~20 instructions against the ~210 of evaluating the group)
          const float4 glo = *reinterpret_cast<const float4*>(gbox[g]);
          const float4 ghi = *reinterpret_cast<const float4*>(gbox[g] + 4);
          // layout per group: [lo.x lo.y lo.z hi.x | hi.y hi.z pad pad]
          bool want = false;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const float ex = fmaxf(fmaxf(glo.x - sx[s], sx[s] - glo.w), 0.f);
            const float ey = fmaxf(fmaxf(glo.y - sy[s], sy[s] - ghi.x), 0.f);
            const float ez = fmaxf(fmaxf(glo.z - sz[s], sz[s] - ghi.y), 0.f);
            want |= ok[s] && !(fmaf(ez, ez, fmaf(ey, ey, ex * ex)) * kShrink > best[s]);
          }
          if (!__any(want)) continue;
        }
        ++groups_done;
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
            // two targets per instruction (v_pk_add / v_pk_mul / v_pk_fma_f32): the same IEEE operations, the same bits
            const f32x2 px = {sx[s], sx[s]}, py = {sy[s], sy[s]}, pz = {sz[s], sz[s]};
            const f32x2 ax = px - f32x2{X.x, X.y}, ay = py - f32x2{Y.x, Y.y}, az = pz - f32x2{Z.x, Z.y};
            const f32x2 bx = px - f32x2{X.z, X.w}, by = py - f32x2{Y.z, Y.w}, bz = pz - f32x2{Z.z, Z.w};
            const f32x2 da = __builtin_elementwise_fma(az, az, __builtin_elementwise_fma(ay, ay, ax * ax));
            const f32x2 db = __builtin_elementwise_fma(bz, bz, __builtin_elementwise_fma(by, by, bx * bx));
            gmin[s] = fminf(fminf(gmin[s], da.x), da.y);
            gmin[s] = fminf(fminf(gmin[s], db.x), db.y);
          }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
          if (gmin[s] < best[s]) {
            best[s] = gmin[s];
            best_group[s] = group0 + g;
            tie_idx[s] = 0xffffffffu;
          } else if (gmin[s] == best[s] && best_group[s] != group0 + g && gmin[s] < INFINITY) {
            // the same minimum in another group (rare): original-index order must decide -- note the lowest original
            // index among this group's equal targets now
            const float4* gp = tgt4 + (int64_t)(group0 + g) * kGroup;
            for (int k = 0; k < kGroup; ++k) {
              const float4 p = gp[k];
              const float dx = sx[s] - p.x, dy = sy[s] - p.y, dz = sz[s] - p.z;
              if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) == best[s] && __float_as_uint(p.w) < tie_idx[s]) tie_idx[s] = __float_as_uint(p.w);
            }
          }
        }
      }
    }
    __syncthreads();
   }
  }

  // resolve inside the winning group: exact distance, lowest original index among equals.  With `partials` the
  // 18 fp64 pair sums of the Umeyama fit are taken right here (p = this source, q = its winner): the ICP loop
  // needs no separate gather pass over (src, idx, tgt).
  double acc[r3d_icp::kSums];
  if (partials) {
#pragma unroll
    for (int k = 0; k < r3d_icp::kSums; ++k) acc[k] = 0.0;
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t i = s_base + (int64_t)s * kThreads + tid;
    if (!ok[s]) continue;
    const uint32_t orig = SRC4 ? __float_as_uint(static_cast<const float4*>(src_any)[i].w) : (uint32_t)i;
    const int64_t g0 = (int64_t)best_group[s] * kGroup;
    uint32_t found = 0xffffffffu;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    for (int k = 0; k < kGroup; ++k) {
      const float4 p = tgt4[g0 + k];
      const float dx = sx[s] - p.x, dy = sy[s] - p.y, dz = sz[s] - p.z;
      const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      if (d == best[s] && __float_as_uint(p.w) < found) {
        found = __float_as_uint(p.w);
        qx = p.x; qy = p.y; qz = p.z;
      }
    }
    if (tie_idx[s] < found) {   // an equal-distance target of another group has the lower original index
      found = tie_idx[s];
      const P3 q = reinterpret_cast<const P3*>(tgt_orig)[found];
      qx = q.x; qy = q.y; qz = q.z;
    }
    if (!(best[s] < INFINITY) || found == 0xffffffffu) {
      // no finite distance at all (the source, or every target, has a NaN / inf coordinate): index 0, d2 = +inf, no pair
      idx_out[orig] = 0u;
      if (d2_out) d2_out[orig] = INFINITY;
    } else {
      idx_out[orig] = found;
      if (d2_out) d2_out[orig] = best[s];
      if (partials && !(max_d2 >= 0.f && !(best[s] <= max_d2))) {
        const double p3[3] = {(double)sx[s], (double)sy[s], (double)sz[s]};
        const double q3[3] = {(double)qx, (double)qy, (double)qz};
        r3d_icp::pair_accumulate(acc, r3d_icp::pair_weight(best[s], dead_zone), p3, q3);
      }
    }
  }
  if (partials) {
    __shared__ double red[kThreads / 64][r3d_icp::kSums];
    r3d_icp::block_reduce_store(acc, red, partials + (int64_t)blockIdx.x * r3d_icp::kSums);
  }
  if (tid == 0 && stats) atomicAdd(&stats[0], (unsigned long long)swept);
  if ((tid & 63) == 0 && stats) atomicAdd(&stats[1], (unsigned long long)groups_done);  // 32-target groups evaluated, per wave
}

// ---- warm query: one wave, no LDS, no barrier --------------------------------------------------------------------
// When every source starts with a tight bound (the warm start above: an ICP iteration after the first) only a handful of
// 32-target groups can still improve any lane -- staging whole 1024-target tiles in LDS behind workgroup barriers, as
// nn_cull_kernel does for a search that starts from nothing, is then nearly all overhead (measured on the two-view pair:
// 6.5 tiles staged per workgroup whether warm or cold).  Here a WAVE is on its own: the boxes of a level are tested in
// PARALLEL, one per lane, against the wave's own source box inflated by its largest bound (a conservative filter); the
// survivors are visited one by one with the exact per-lane test of nn_cull_kernel (their box comes from the holding lane by
// v_readlane, no reload); a group that passes is fetched with ONE coalesced load (lane t holds target t) and its 32 targets
// are broadcast by v_readlane.  Same candidates-or-better than the cold walk, same `<` / tie rule, same epilogue: the
// results are the cold search's bit for bit (tests/test_gpu_icp.py::test_warm_*).
// Measured (tools/nn_probe.py, sources unmoved since the previous query): two 480x640 views 349 us cold, 179 with the bounds in
// front of nn_cull_kernel, 103 here; C3's 500k x 500k 468 / 432 / 230.  Requesting the surviving groups' targets four at a time
// and the next tile's group boxes ahead (to shorten a lone wave's chain of memory round trips) made it SLOWER (114 / 243): the
// kernel is bound by the ~350 vector instructions of a group evaluation, not by those waits.
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// squared distance between two axis-aligned boxes (0 when they overlap); an empty box (+inf, -inf) is infinitely far
__device__ __forceinline__ float box_box_d2(const float blo[3], const float bhi[3], const float slo[3], const float shi[3]) {
  const float ex = fmaxf(fmaxf(blo[0] - shi[0], slo[0] - bhi[0]), 0.f);
  const float ey = fmaxf(fmaxf(blo[1] - shi[1], slo[1] - bhi[1]), 0.f);
  const float ez = fmaxf(fmaxf(blo[2] - shi[2], slo[2] - bhi[2]), 0.f);
  return fmaf(ez, ez, fmaf(ey, ey, ex * ex));
}
__device__ __forceinline__ float box_point_d2(float lx, float ly, float lz, float hx, float hy, float hz, float px, float py, float pz) {
  const float ex = fmaxf(fmaxf(lx - px, px - hx), 0.f);
  const float ey = fmaxf(fmaxf(ly - py, py - hy), 0.f);
  const float ez = fmaxf(fmaxf(lz - pz, pz - hz), 0.f);
  return fmaf(ez, ez, fmaf(ey, ey, ex * ex));
}

__global__ __launch_bounds__(kThreads) void nn_warm_kernel(const float* __restrict__ src, int64_t n_src,
                                                           const float4* __restrict__ tgt4, int64_t n_tgt, int64_t n_tiles,
                                                           const float* __restrict__ tile_box, const float* __restrict__ group_box,
                                                           const float* __restrict__ super_box, const float* __restrict__ tgt_orig,
                                                           uint32_t* idx_out, float* __restrict__ d2_out,
                                                           unsigned long long* __restrict__ stats, double* __restrict__ partials,
                                                           float max_d2, float dead_zone, const uint32_t* idx_warm,
                                                           const float* __restrict__ src_orig, const double* __restrict__ d_T,
                                                           float* src_moved) {
  const uint32_t tid = threadIdx.x;
  const int lane = tid & 63;
  const int64_t i = (int64_t)blockIdx.x * kThreads + tid;
  const bool ok = i < n_src;
  P3 sp = {0.f, 0.f, 0.f};
  if (src_orig) {
    // the move of an ICP iteration folded into its search: this lane's source = T . (its point of the ORIGINAL cloud), in
    // apply_lane_kernel's arithmetic (fp64 row . [x y z 1], rounded to f32 once), left in src_moved for the kernels that follow
    if (ok) {
      const P3 o = reinterpret_cast<const P3*>(src_orig)[i];
      const double x = o.x, y = o.y, z = o.z;
      sp.x = (float)(fma(d_T[2], z, fma(d_T[1], y, d_T[0] * x)) + d_T[3]);
      sp.y = (float)(fma(d_T[6], z, fma(d_T[5], y, d_T[4] * x)) + d_T[7]);
      sp.z = (float)(fma(d_T[10], z, fma(d_T[9], y, d_T[8] * x)) + d_T[11]);
      reinterpret_cast<P3*>(src_moved)[i] = sp;
    }
  } else if (ok) {
    sp = reinterpret_cast<const P3*>(src)[i];
  }
  const float sx = sp.x, sy = sp.y, sz = sp.z;
  // a source with a NaN / inf coordinate has no finite distance to anything (its answer is "index 0, +inf", as in the cold
  // search): it takes no part in the culling votes instead of holding every box open for its wave
  const bool act = ok && (sx - sx == 0.f) && (sy - sy == 0.f) && (sz - sz == 0.f);
  float best = INFINITY;
  uint32_t best_group = 0, tie_idx = 0xffffffffu;
  if (act) {   // the bound: see nn_cull_kernel
    const uint32_t j = min(idx_warm[i], (uint32_t)(n_tgt - 1));
    const P3 q = reinterpret_cast<const P3*>(tgt_orig)[j];
    const float dx = sx - q.x, dy = sy - q.y, dz = sz - q.z;
    const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    if (d < INFINITY) best = d * 1.000002f + 1.17549435e-38f;
  }
  // the wave's source box and its largest bound (+inf as soon as one active lane has none)
  const float slo[3] = {wave_min(act ? sx : INFINITY), wave_min(act ? sy : INFINITY), wave_min(act ? sz : INFINITY)};
  const float shi[3] = {wave_max(act ? sx : -INFINITY), wave_max(act ? sy : -INFINITY), wave_max(act ? sz : -INFINITY)};
  float R = wave_max(act ? best : 0.f);
  unsigned swept = 0, groups_done = 0;
  const int64_t n_super = (n_tiles + kSuper - 1) / kSuper;
  for (int64_t s0 = 0; s0 < n_super; s0 += 64) {
    // 64 super-boxes at a time, one per lane
    const int64_t sk = s0 + lane;
    float sb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (sk < n_super) {
#pragma unroll
      for (int a = 0; a < 6; ++a) sb[a] = super_box[sk * 6 + a];
    }
    unsigned long long smask = __ballot(!(box_box_d2(sb, sb + 3, slo, shi) * kShrink > R));
    while (smask) {
      const int sl = __ffsll((long long)smask) - 1;
      smask &= smask - 1;
      {
        const float lb = box_point_d2(lane_value(sb[0], sl), lane_value(sb[1], sl), lane_value(sb[2], sl), lane_value(sb[3], sl),
                                      lane_value(sb[4], sl), lane_value(sb[5], sl), sx, sy, sz) * kShrink;
        if (!__any(act && !(lb > best))) continue;
      }
      // its 16 tiles, one per lane
      const int64_t tile_l = (s0 + sl) * kSuper + (lane & (kSuper - 1));
      float tb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (lane < kSuper && tile_l < n_tiles) {
#pragma unroll
        for (int a = 0; a < 6; ++a) tb[a] = tile_box[tile_l * 6 + a];
      }
      unsigned long long tmask = __ballot(lane < kSuper && !(box_box_d2(tb, tb + 3, slo, shi) * kShrink > R));
      while (tmask) {
        const int tl = __ffsll((long long)tmask) - 1;
        tmask &= tmask - 1;
        {
          const float lb = box_point_d2(lane_value(tb[0], tl), lane_value(tb[1], tl), lane_value(tb[2], tl), lane_value(tb[3], tl),
                                        lane_value(tb[4], tl), lane_value(tb[5], tl), sx, sy, sz) * kShrink;
          if (!__any(act && !(lb > best))) continue;
        }
        ++swept;
        const int64_t tile = (s0 + sl) * kSuper + tl;
        // its 32 groups, one per lane (the upper half of the wave mirrors the lower: harmless)
        const int gl = lane & (kTile / kGroup - 1);
        float gb[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) gb[a] = group_box[(tile * (kTile / kGroup) + gl) * 6 + a];
        unsigned long long gmask = __ballot(lane < kTile / kGroup && !(box_box_d2(gb, gb + 3, slo, shi) * kShrink > R));
        while (gmask) {
          const int g = __ffsll((long long)gmask) - 1;
          gmask &= gmask - 1;
          {
            const float lb = box_point_d2(lane_value(gb[0], g), lane_value(gb[1], g), lane_value(gb[2], g), lane_value(gb[3], g),
                                          lane_value(gb[4], g), lane_value(gb[5], g), sx, sy, sz) * kShrink;
            if (!__any(act && !(lb > best))) continue;
          }
          ++groups_done;
          const uint32_t group = (uint32_t)(tile * (kTile / kGroup) + g);
          const float4 mine = tgt4[(int64_t)group * kGroup + (lane & (kGroup - 1))];   // lane t (and t + 32) holds target t
          // two targets per instruction (v_pk_add / v_pk_mul / v_pk_fma_f32: the same IEEE operations as the scalar form, so the
          // distances are the cold kernel's bits)
          float gmin = INFINITY;
          const f32x2 sx2 = {sx, sx}, sy2 = {sy, sy}, sz2 = {sz, sz};
#pragma unroll
          for (int t = 0; t < kGroup; t += 2) {
            const f32x2 X = {lane_value(mine.x, t), lane_value(mine.x, t + 1)}, Y = {lane_value(mine.y, t), lane_value(mine.y, t + 1)},
                        Z = {lane_value(mine.z, t), lane_value(mine.z, t + 1)};
            const f32x2 dx = sx2 - X, dy = sy2 - Y, dz = sz2 - Z;
            const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            gmin = fminf(fminf(gmin, d.x), d.y);
          }
          if (gmin < best) {
            best = gmin;
            best_group = group;
            tie_idx = 0xffffffffu;
          } else if (gmin == best && best_group != group && gmin < INFINITY) {
            // the same minimum in another group (rare): original-index order decides -- the lowest original index among this
            // group's equal targets is noted now (as in nn_cull_kernel)
            const float4* gp = tgt4 + (int64_t)group * kGroup;
            for (int k = 0; k < kGroup; ++k) {
              const float4 p = gp[k];
              const float dx = sx - p.x, dy = sy - p.y, dz = sz - p.z;
              if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) == best && __float_as_uint(p.w) < tie_idx) tie_idx = __float_as_uint(p.w);
            }
          }
          R = wave_max(act ? best : 0.f);   // bounds only shrink: the filters above stay valid, the later ones get tighter
        }
      }
    }
  }
  // epilogue: exactly nn_cull_kernel's (S = 1, plain xyz sources)
  double acc[r3d_icp::kSums];
  if (partials) {
#pragma unroll
    for (int k = 0; k < r3d_icp::kSums; ++k) acc[k] = 0.0;
  }
  if (ok) {
    const int64_t g0 = (int64_t)best_group * kGroup;
    uint32_t found = 0xffffffffu;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    for (int k = 0; k < kGroup; ++k) {
      const float4 p = tgt4[g0 + k];
      const float dx = sx - p.x, dy = sy - p.y, dz = sz - p.z;
      const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      if (d == best && __float_as_uint(p.w) < found) {
        found = __float_as_uint(p.w);
        qx = p.x; qy = p.y; qz = p.z;
      }
    }
    if (tie_idx < found) {
      found = tie_idx;
      const P3 q = reinterpret_cast<const P3*>(tgt_orig)[found];
      qx = q.x; qy = q.y; qz = q.z;
    }
    if (!(best < INFINITY) || found == 0xffffffffu) {
      idx_out[i] = 0u;
      if (d2_out) d2_out[i] = INFINITY;
    } else {
      idx_out[i] = found;
      if (d2_out) d2_out[i] = best;
      if (partials && !(max_d2 >= 0.f && !(best <= max_d2))) {
        const double p3[3] = {(double)sx, (double)sy, (double)sz};
        const double q3[3] = {(double)qx, (double)qy, (double)qz};
        r3d_icp::pair_accumulate(acc, r3d_icp::pair_weight(best, dead_zone), p3, q3);
      }
    }
  }
  if (partials) {
    __shared__ double red[kThreads / 64][r3d_icp::kSums];
    r3d_icp::block_reduce_store(acc, red, partials + (int64_t)blockIdx.x * r3d_icp::kSums);
  }
  if (lane == 0 && stats) {
    atomicAdd(&stats[0], (unsigned long long)swept);   // tiles opened, per wave here (per workgroup in nn_cull_kernel)
    atomicAdd(&stats[1], (unsigned long long)groups_done);
  }
}

int bits_for(int64_t n) {
  int b = 1;
  while (((int64_t)1 << b) < n) ++b;
  return b;
}

// keys + sort for a cloud in the index's quantisation frame; result in d_keys (sorted)
int sorted_keys(r3d_ctx* ctx, const float* d_xyz, int64_t n, const float* d_frame, int axis_bits, int idx_bits,
                uint64_t* d_keys, uint64_t* d_tmp) {
  hipLaunchKernelGGL(keys_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_xyz, n,
                     d_frame, axis_bits, idx_bits, d_keys);
  R3D_HIP(hipGetLastError());
  // the low idx_bits hold the row number, which already ascends in the input: the stable sort skips those digits
  return r3d_radix_sort_u64(ctx, d_keys, d_tmp, n, 3 * axis_bits + idx_bits, idx_bits);
}

}  // namespace

extern "C" {

int r3d_nn_index_destroy(r3d_nn_index* ix) {
  if (!ix) return R3D_OK;
  (void)hipSetDevice(ix->device);
  (void)hipDeviceSynchronize();
  if (ix->d_slab) (void)hipFree(ix->d_slab);
  delete ix;
  return R3D_OK;
}

// (re)builds every table of the index for a target cloud of n_tgt <= capacity points; asynchronous on the ctx stream
static int nn_index_build(r3d_nn_index* ix, const float* d_tgt, int64_t n_tgt) {
  r3d_ctx* ctx = ix->ctx;
  ix->n = n_tgt;
  ix->warm_src = nullptr;   // matches against the previous target mean nothing for this one
  ix->warm_idx = nullptr;
  ix->warm_n = 0;
  ix->n_tiles = (n_tgt + kTile - 1) / kTile;
  ix->idx_bits = bits_for(n_tgt);
  ix->axis_bits = 10;  // 2^30 cells order any cloud finely enough for tile coherence and leave 34 bits for indices
  const int64_t n_pad = ix->n_tiles * kTile;
  int rc;
  void *keys = nullptr, *tmp = nullptr;
  if ((rc = r3d_scratch(ctx, 0, (size_t)n_tgt * 8, &keys)) || (rc = r3d_scratch(ctx, 2, (size_t)n_tgt * 8, &tmp))) return rc;
  hipStream_t st = ctx->stream;
  if (d_tgt != ix->d_tgt) R3D_HIP(hipMemcpyAsync(ix->d_tgt, d_tgt, (size_t)n_tgt * 12, hipMemcpyDeviceToDevice, st));
  // 4 points per thread in flight; the rows of bounds go to a scratch slot
  const int blocks = (int)std::min<int64_t>((n_tgt + 4 * kThreads - 1) / (4 * kThreads), (int64_t)ctx->num_cus * 2);
  void* rows = nullptr;
  if ((rc = r3d_scratch(ctx, 4, (size_t)blocks * 6 * sizeof(float), &rows))) return rc;
  hipLaunchKernelGGL(bbox_kernel, dim3(blocks), dim3(kThreads), 0, st, ix->d_tgt, n_tgt, (float*)rows);
  hipLaunchKernelGGL(frame_kernel, dim3(1), dim3(64), 0, st, (const float*)rows, blocks, ix->axis_bits, ix->d_frame);
  if ((rc = sorted_keys(ctx, ix->d_tgt, n_tgt, ix->d_frame, ix->axis_bits, ix->idx_bits, (uint64_t*)keys, (uint64_t*)tmp))) return rc;
  hipLaunchKernelGGL(gather4_kernel, dim3((unsigned)((n_pad + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, ix->d_tgt,
                     (const uint64_t*)keys, n_tgt, n_pad, ix->idx_bits, ix->d_tgt4);
  hipLaunchKernelGGL(tile_box_kernel, dim3((unsigned)ix->n_tiles), dim3(kThreads), 0, st, ix->d_tgt4, (const uint64_t*)keys,
                     n_tgt, ix->idx_bits, kTile, ix->d_tile_box, ix->d_tile_code);
  hipLaunchKernelGGL(tile_box_kernel, dim3((unsigned)(ix->n_tiles * (kTile / kSub))), dim3(kThreads), 0, st, ix->d_tgt4,
                     (const uint64_t*)keys, n_tgt, ix->idx_bits, kSub, ix->d_sub_box, (uint64_t*)nullptr);
  hipLaunchKernelGGL(group_box_kernel, dim3((unsigned)ix->n_tiles), dim3(kThreads), 0, st, ix->d_tgt4, n_tgt, ix->d_group_box);
  hipLaunchKernelGGL(tile_box_kernel, dim3((unsigned)((ix->n_tiles + kSuper - 1) / kSuper)), dim3(kThreads), 0, st, ix->d_tgt4,
                     (const uint64_t*)keys, n_tgt, ix->idx_bits, kSuper * kTile, ix->d_super_box, (uint64_t*)nullptr);
  // asynchronous: the scratch key buffers are only ever reused by later work on this same stream
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_nn_index_create(r3d_ctx* ctx, const float* d_tgt, int64_t n_tgt, r3d_nn_index** ix_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(ix_out != nullptr, "ix_out is NULL");
  *ix_out = nullptr;
  R3D_REQUIRE(n_tgt >= 1, "target cloud is empty");
  R3D_REQUIRE(n_tgt < ((int64_t)1 << 32), "target cloud too large for uint32 indices");
  R3D_REQUIRE(d_tgt != nullptr, "NULL device pointer");
  r3d_nn_index* ix = new (std::nothrow) r3d_nn_index();
  if (!ix) {
    r3d_set_error("host allocation failed");
    return R3D_ERR_NOMEM;
  }
  ix->ctx = ctx;
  ix->device = ctx->device;
  ix->capacity = n_tgt;
  const int64_t n_tiles = (n_tgt + kTile - 1) / kTile, n_pad = n_tiles * kTile;
  // one slab, every table at a 256-byte boundary
  size_t off = 0;
  auto take = [&off](size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  };
  const size_t o_tgt4 = take((size_t)n_pad * sizeof(float4)), o_tgt = take((size_t)n_tgt * 12),
               o_tile = take((size_t)n_tiles * 6 * sizeof(float)), o_sub = take((size_t)n_tiles * (kTile / kSub) * 6 * sizeof(float)),
               o_group = take((size_t)n_tiles * (kTile / kGroup) * 6 * sizeof(float)),
               o_super = take((size_t)((n_tiles + kSuper - 1) / kSuper) * 6 * sizeof(float)),
               o_code = take((size_t)n_tiles * sizeof(uint64_t)), o_frame = take(16 * sizeof(float));
  hipError_t e = hipMalloc(&ix->d_slab, off);
  if (e != hipSuccess) {
    r3d_nn_index_destroy(ix);
    return r3d_fail_hip(e, "nn index allocation", __FILE__, __LINE__);
  }
  char* slab = static_cast<char*>(ix->d_slab);
  ix->d_tgt4 = reinterpret_cast<float4*>(slab + o_tgt4);
  ix->d_tgt = reinterpret_cast<float*>(slab + o_tgt);
  ix->d_tile_box = reinterpret_cast<float*>(slab + o_tile);
  ix->d_sub_box = reinterpret_cast<float*>(slab + o_sub);
  ix->d_group_box = reinterpret_cast<float*>(slab + o_group);
  ix->d_super_box = reinterpret_cast<float*>(slab + o_super);
  ix->d_tile_code = reinterpret_cast<uint64_t*>(slab + o_code);
  ix->d_frame = reinterpret_cast<float*>(slab + o_frame);
  if ((rc = nn_index_build(ix, d_tgt, n_tgt))) {
    r3d_nn_index_destroy(ix);
    return rc;
  }
  *ix_out = ix;
  return R3D_OK;
}

int r3d_nn_index_rebuild(r3d_nn_index* ix, const float* d_tgt, int64_t n_tgt) {
  R3D_REQUIRE(ix != nullptr, "nn index is NULL");
  int rc = r3d_ctx_enter(ix->ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_tgt >= 1, "target cloud is empty");
  R3D_REQUIRE(n_tgt <= ix->capacity, "index was created for %lld points, cannot hold %lld", (long long)ix->capacity,
              (long long)n_tgt);
  R3D_REQUIRE(d_tgt != nullptr, "NULL device pointer");
  return nn_index_build(ix, d_tgt, n_tgt);
}

static int nn_index_query_impl(r3d_nn_index* ix, const float* d_src, int64_t n_src, uint32_t* d_idx_out, float* d_d2_out,
                               int presorted, int64_t* h_tiles_swept, bool want_sums, float max_d2, float dead_zone,
                               double* d_sums_out, int with_scale = 0, double* d_state = nullptr, int small_motion = 0,
                               const float* d_src_orig = nullptr, const double* d_T_move = nullptr, int* moved_out = nullptr) {
  R3D_REQUIRE(ix != nullptr, "nn index is NULL");
  r3d_ctx* ctx = ix->ctx;
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_src >= 0, "negative cloud size");
  if (h_tiles_swept) *h_tiles_swept = 0;
  hipStream_t st = ctx->stream;
  if (want_sums) R3D_REQUIRE(d_sums_out != nullptr, "d_sums_out is NULL");
  if (n_src == 0) {
    if (want_sums) R3D_HIP(hipMemsetAsync(d_sums_out, 0, r3d_icp::kSums * sizeof(double), st));
    return R3D_OK;
  }
  R3D_REQUIRE(n_src < ((int64_t)1 << 32), "source cloud too large");
  R3D_REQUIRE(d_src && d_idx_out, "NULL device pointer");
  R3D_REQUIRE(!want_sums || d_d2_out != nullptr, "the fused pair sums need the d2 output array");
  void *src4 = nullptr, *misc = nullptr;
  if (!presorted) {
    // one-off query: sort a float4 copy of the sources into index order (results are scattered back by index)
    const int src_idx_bits = bits_for(n_src);
    void *keys = nullptr, *tmp = nullptr;
    if ((rc = r3d_scratch(ctx, 0, (size_t)n_src * 8, &keys))) return rc;
    if ((rc = r3d_scratch(ctx, 2, (size_t)n_src * 8, &tmp))) return rc;
    if ((rc = r3d_scratch(ctx, 1, (size_t)n_src * sizeof(float4), &src4))) return rc;
    if ((rc = sorted_keys(ctx, d_src, n_src, ix->d_frame, ix->axis_bits, src_idx_bits, (uint64_t*)keys, (uint64_t*)tmp))) return rc;
    hipLaunchKernelGGL(gather4_kernel, dim3((unsigned)((n_src + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, d_src,
                       (const uint64_t*)keys, n_src, n_src, src_idx_bits, (float4*)src4);
  }
  if ((rc = r3d_scratch(ctx, 5, 64, &misc))) return rc;
  unsigned long long* stats = static_cast<unsigned long long*>(misc);  // [0] tile sweeps, [1] groups evaluated
  if (h_tiles_swept) R3D_HIP(hipMemsetAsync(misc, 0, 32, st));
  int S = ctx->nn_variant;
  if (S != 1 && S != 2 && S != 4) S = 1;
  const int64_t per_block = (int64_t)kThreads * S;
  const unsigned blocks = (unsigned)((n_src + per_block - 1) / per_block);
  double* partials = nullptr;
  int n_rows = (int)blocks;
  if (want_sums) {
    void* pv = nullptr;
    if ((rc = r3d_scratch(ctx, 4, (size_t)blocks * r3d_icp::kSums * sizeof(double), &pv))) return rc;
    partials = static_cast<double*>(pv);
  }
  // same sources buffer, same count, same output buffer as the last presorted query against this build of the index: the
  // output buffer still holds that query's matches ("nn_warm" = 1 switches the warm start off: A/B, tests)
  const int warm_mode = ctx->nn_warm;
  const uint32_t* warm = nullptr;
  if (presorted && warm_mode != 1 && ix->warm_src == d_src && ix->warm_idx == d_idx_out && ix->warm_n == n_src) warm = d_idx_out;
  if (presorted) {
    ix->warm_src = d_src;
    ix->warm_idx = d_idx_out;
    ix->warm_n = n_src;
  }
#define R3D_LAUNCH_CULL(SS, FMT, PTR)                                                                                  \
  hipLaunchKernelGGL((nn_cull_kernel<SS, FMT>), dim3(blocks), dim3(kThreads), 0, st, (const void*)(PTR), n_src,        \
                     (const float*)ix->d_frame, ix->axis_bits, ix->d_tgt4, ix->n, ix->n_tiles, ix->d_tile_box,         \
                     ix->d_sub_box, ix->d_group_box, ix->d_super_box, ix->d_tile_code, (const float*)ix->d_tgt, d_idx_out, d_d2_out,     \
                     h_tiles_swept ? stats : (unsigned long long*)nullptr, partials, max_d2, dead_zone, warm)
  // Which kernel: the bounds always help nn_cull_kernel (never slower than the cold walk); the wave-local kernel wins big when
  // the bounds are TIGHT (the sources moved by one ICP step since the matches were made: the loops say so with small_motion)
  // and loses when they are not (a jump to another start pose: each wave would scan group after group on its own).
  if (warm && warm_mode != 2 && (small_motion || warm_mode == 3)) {
    // (the partial rows are per 256 sources there as here: S is 1)
    const unsigned wblocks = (unsigned)((n_src + kThreads - 1) / kThreads);
    if (want_sums) {
      void* pv = nullptr;
      if ((rc = r3d_scratch(ctx, 4, (size_t)wblocks * r3d_icp::kSums * sizeof(double), &pv))) return rc;
      partials = static_cast<double*>(pv);
    }
    hipLaunchKernelGGL(nn_warm_kernel, dim3(wblocks), dim3(kThreads), 0, st, d_src, n_src, ix->d_tgt4, ix->n, ix->n_tiles,
                       ix->d_tile_box, ix->d_group_box, ix->d_super_box, (const float*)ix->d_tgt, d_idx_out, d_d2_out,
                       h_tiles_swept ? stats : (unsigned long long*)nullptr, partials, max_d2, dead_zone, warm, d_src_orig, d_T_move,
                       const_cast<float*>(d_src));
    if (moved_out) *moved_out = d_src_orig != nullptr;
    n_rows = (int)wblocks;
  } else if (presorted) {
    if (S == 1) R3D_LAUNCH_CULL(1, false, d_src);
    else if (S == 2) R3D_LAUNCH_CULL(2, false, d_src);
    else R3D_LAUNCH_CULL(4, false, d_src);
  } else {
    if (S == 1) R3D_LAUNCH_CULL(1, true, src4);
    else if (S == 2) R3D_LAUNCH_CULL(2, true, src4);
    else R3D_LAUNCH_CULL(4, true, src4);
  }
#undef R3D_LAUNCH_CULL
  R3D_HIP(hipGetLastError());
  if (want_sums &&
      (rc = r3d_icp_sums_finish(ctx, partials, n_rows, d_sums_out, with_scale, d_state)))
    return rc;
  if (h_tiles_swept) {
    unsigned long long v[2] = {0, 0};
    R3D_HIP(hipMemcpyAsync(v, stats, sizeof(v), hipMemcpyDeviceToHost, st));
    R3D_HIP(hipStreamSynchronize(st));
    *h_tiles_swept = (int64_t)v[0];
  }
  return R3D_OK;
}

int r3d_nn_index_query(r3d_nn_index* ix, const float* d_src, int64_t n_src, uint32_t* d_idx_out, float* d_d2_out,
                       int presorted, int64_t* h_tiles_swept) {
  return nn_index_query_impl(ix, d_src, n_src, d_idx_out, d_d2_out, presorted, h_tiles_swept, false, -1.f, 0.f, nullptr);
}

int r3d_nn_index_query_sums(r3d_nn_index* ix, const float* d_src, int64_t n_src, uint32_t* d_idx_out, float* d_d2_out,
                            int presorted, float max_d2, float dead_zone, double* d_sums_out) {
  return nn_index_query_impl(ix, d_src, n_src, d_idx_out, d_d2_out, presorted, nullptr, true, max_d2, dead_zone,
                             d_sums_out);
}

}  // extern "C"

int r3d_nn_index_target(r3d_nn_index* ix, const float** d_tgt, int64_t* n_tgt, r3d_ctx** ctx) {
  R3D_REQUIRE(ix != nullptr, "nn index is NULL");
  if (d_tgt) *d_tgt = ix->d_tgt;
  if (n_tgt) *n_tgt = ix->n;
  if (ctx) *ctx = ix->ctx;
  return R3D_OK;
}

int r3d_nn_index_query_solve(r3d_nn_index* ix, const float* d_src, int64_t n_src, uint32_t* d_idx_out, float* d_d2_out,
                             float max_d2, double* d_sums_out, int with_scale, double* d_state, int small_motion) {
  return nn_index_query_impl(ix, d_src, n_src, d_idx_out, d_d2_out, 1, nullptr, true, max_d2, 0.f, d_sums_out, with_scale,
                             d_state, small_motion);
}

int r3d_nn_index_query_step(r3d_nn_index* ix, const float* d_src, int64_t n_src, uint32_t* d_idx_out, float* d_d2_out,
                            int small_motion, const float* d_src_orig, const double* d_T_move, int* moved_out) {
  if (moved_out) *moved_out = 0;
  return nn_index_query_impl(ix, d_src, n_src, d_idx_out, d_d2_out, 1, nullptr, false, -1.f, 0.f, nullptr, 0, nullptr,
                             small_motion, d_src_orig, d_T_move, moved_out);
}

extern "C" {

// gathers xyz rows by the index part of sorted keys
__global__ __launch_bounds__(kThreads) void gather3_kernel(const float* __restrict__ xyz, const uint64_t* __restrict__ keys,
                                                           int64_t n, int idx_bits, float* __restrict__ out,
                                                           uint32_t* __restrict__ perm) {
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= n) return;
  const uint32_t i = (uint32_t)(keys[j] & (((uint64_t)1 << idx_bits) - 1));
  reinterpret_cast<P3*>(out)[j] = reinterpret_cast<const P3*>(xyz)[i];
  if (perm) perm[j] = i;
}

int r3d_cloud_zero_rows_to_nan(r3d_ctx* ctx, float* d_xyz, int64_t n) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n >= 0, "negative cloud size");
  if (n == 0) return R3D_OK;
  R3D_REQUIRE(d_xyz != nullptr, "NULL device pointer");
  r3d_wrote(ctx, d_xyz, (size_t)n * 12);
  hipLaunchKernelGGL(zero_rows_to_nan_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_xyz, n);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

static int sort_cloud_impl(r3d_nn_index* ix, float* d_xyz, int64_t n, uint32_t* d_perm_out, int64_t* h_n_valid) {
  R3D_REQUIRE(ix != nullptr, "nn index is NULL");
  r3d_ctx* ctx = ix->ctx;
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n >= 0 && n < ((int64_t)1 << 32), "bad cloud size");
  if (h_n_valid) *h_n_valid = 0;
  if (n == 0) return R3D_OK;
  R3D_REQUIRE(d_xyz != nullptr, "NULL device pointer");
  r3d_wrote(ctx, d_xyz, (size_t)n * 12);   // reordered in place
  const int idx_bits = bits_for(n);
  void *keys = nullptr, *tmp = nullptr, *copy = nullptr, *misc = nullptr;
  if ((rc = r3d_scratch(ctx, 0, (size_t)n * 8, &keys))) return rc;
  if ((rc = r3d_scratch(ctx, 2, (size_t)n * 8, &tmp))) return rc;
  if ((rc = r3d_scratch(ctx, 1, (size_t)n * 12, &copy))) return rc;
  if (h_n_valid) {
    if ((rc = r3d_scratch(ctx, 5, 64, &misc))) return rc;
    R3D_HIP(hipMemsetAsync(misc, 0, 8, ctx->stream));
    hipLaunchKernelGGL(keys_valid_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                       (const float*)d_xyz, n, (const float*)ix->d_frame, ix->axis_bits, idx_bits, (uint64_t*)keys,
                       (unsigned long long*)misc);
    R3D_HIP(hipGetLastError());
    // one more key bit than sorted_keys(): the "not a point" code
    if ((rc = r3d_radix_sort_u64(ctx, (uint64_t*)keys, (uint64_t*)tmp, n, 3 * ix->axis_bits + 1 + idx_bits, idx_bits))) return rc;
  } else if ((rc = sorted_keys(ctx, d_xyz, n, ix->d_frame, ix->axis_bits, idx_bits, (uint64_t*)keys, (uint64_t*)tmp))) {
    return rc;
  }
  R3D_HIP(hipMemcpyAsync(copy, d_xyz, (size_t)n * 12, hipMemcpyDeviceToDevice, ctx->stream));
  hipLaunchKernelGGL(gather3_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                     (const float*)copy, (const uint64_t*)keys, n, idx_bits, d_xyz, d_perm_out);
  R3D_HIP(hipGetLastError());
  if (h_n_valid) {
    unsigned long long v = 0;
    R3D_HIP(hipMemcpyAsync(&v, misc, 8, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(hipStreamSynchronize(ctx->stream));
    *h_n_valid = (int64_t)v;
  }
  return R3D_OK;
}

int r3d_nn_index_sort_cloud(r3d_nn_index* ix, float* d_xyz, int64_t n, uint32_t* d_perm_out) {
  return sort_cloud_impl(ix, d_xyz, n, d_perm_out, nullptr);
}

int r3d_nn_index_sort_cloud_valid(r3d_nn_index* ix, float* d_xyz, int64_t n, uint32_t* d_perm_out, int64_t* n_valid_out) {
  R3D_REQUIRE(n_valid_out != nullptr, "n_valid_out is NULL");
  return sort_cloud_impl(ix, d_xyz, n, d_perm_out, n_valid_out);
}

// rows first, first + step, ... of an xyz cloud
__global__ __launch_bounds__(kThreads) void rows_strided_kernel(const float* __restrict__ xyz, int64_t first, int64_t step,
                                                                int64_t n_out, float* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= n_out) return;
  reinterpret_cast<P3*>(out)[j] = reinterpret_cast<const P3*>(xyz)[first + j * step];
}

// inverse of a permutation: inv[perm[j]] = j
__global__ __launch_bounds__(kThreads) void perm_invert_kernel(const uint32_t* __restrict__ perm, int64_t n, uint32_t* __restrict__ inv) {
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= n) return;
  const uint32_t i = perm[j];
  if ((int64_t)i < n) inv[i] = (uint32_t)j;
}

// values[k] <- table[values[k]] (0xffffffff where values[k] is outside the table)
__global__ __launch_bounds__(kThreads) void remap_kernel(uint32_t* __restrict__ values, int64_t n, const uint32_t* __restrict__ table,
                                                         int64_t n_table) {
  const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  const uint32_t v = values[k];
  values[k] = (int64_t)v < n_table ? table[v] : 0xffffffffu;
}

int r3d_permutation_invert(r3d_ctx* ctx, const uint32_t* d_perm, int64_t n, uint32_t* d_inverse_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n >= 0, "negative size");
  if (n == 0) return R3D_OK;
  R3D_REQUIRE(d_perm && d_inverse_out && d_perm != d_inverse_out, "NULL or aliased device pointer");
  hipLaunchKernelGGL(perm_invert_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_perm, n,
                     d_inverse_out);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_remap_u32(r3d_ctx* ctx, uint32_t* d_values, int64_t n, const uint32_t* d_table, int64_t n_table) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n >= 0 && n_table >= 0, "negative size");
  if (n == 0) return R3D_OK;
  R3D_REQUIRE(d_values && d_table, "NULL device pointer");
  hipLaunchKernelGGL(remap_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_values, n, d_table,
                     n_table);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

// rows perm[0], perm[1], ... of an xyz cloud
__global__ __launch_bounds__(kThreads) void rows_by_index_kernel(const float* __restrict__ xyz, int64_t n_points,
                                                                 const uint32_t* __restrict__ perm, int64_t n_out,
                                                                 float* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= n_out) return;
  const int64_t i = perm[j];
  reinterpret_cast<P3*>(out)[j] = i < n_points ? reinterpret_cast<const P3*>(xyz)[i] : P3{NAN, NAN, NAN};
}

int r3d_gather_rows(r3d_ctx* ctx, const float* d_xyz, int64_t n_points, const uint32_t* d_rows, int64_t n_out, float* d_xyz_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_points >= 0 && n_out >= 0, "negative size");
  if (n_out == 0) return R3D_OK;
  R3D_REQUIRE(d_xyz && d_rows && d_xyz_out && d_xyz != d_xyz_out, "NULL or aliased device pointer");
  r3d_wrote(ctx, d_xyz_out, (size_t)n_out * 12);
  hipLaunchKernelGGL(rows_by_index_kernel, dim3((unsigned)((n_out + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_xyz,
                     n_points, d_rows, n_out, d_xyz_out);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_gather_rows_strided(r3d_ctx* ctx, const float* d_xyz, int64_t n_points, int64_t first, int64_t step, int64_t n_out,
                            float* d_xyz_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_points >= 0 && first >= 0 && step >= 1 && n_out >= 0, "bad row selection");
  if (n_out == 0) return R3D_OK;
  R3D_REQUIRE(first + (n_out - 1) * step < n_points, "row selection runs past the cloud (%lld rows)", (long long)n_points);
  R3D_REQUIRE(d_xyz && d_xyz_out && d_xyz != d_xyz_out, "NULL or aliased device pointer");
  r3d_wrote(ctx, d_xyz_out, (size_t)n_out * 12);
  hipLaunchKernelGGL(rows_strided_kernel, dim3((unsigned)((n_out + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_xyz,
                     first, step, n_out, d_xyz_out);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

}  // extern "C"
