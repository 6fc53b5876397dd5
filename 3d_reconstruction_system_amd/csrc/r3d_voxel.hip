// Occupied-voxel set of a world cloud on gfx950 (MI355X) + OctoMap binary (.bt) export.
//
// Replaces the per-point `tree.updateNode(xyz, True)` loop, `updateInnerOccupancy()` and
// `writeBinary()` of octomap/txt_transfer_octomap.py:16-36 (== octomap/ply_transfer_octomap.py:16-48).
// The arithmetic of that path lives in the third-party OctoMap library (not vendored, not pinned by the
// reference): restated from its published semantics; parity unpinned.
//
// A hits-only tree written with writeBinary() depends only on the SET of voxels that received a point
// (toMaxLikelihood makes every hit leaf "occupied"), so the GPU's job is a set insert:
//   * voxel_insert_kernel (12 B/point read): lane-per-point 12-byte loads, key per axis
//     = (int)floor((1/res) * (double)x) + 32768 in fp64 like OcTreeBaseImpl::coordToKey, the three 16-bit keys
//     packed into one word, lanes whose predecessor lane holds the same word drop out, the rest go into an
//     open-addressing hash set in HBM (64-bit atomicCAS, multiplicative hash, linear probing).
//   * voxel_compact_kernel: table -> dense list of 48-bit Morton codes (x lowest, as OctoMap's child index): the
//     interleave is paid per distinct voxel here, not per point in the insert (r3d_voxel_dev.h).
// The distinct codes are radix-sorted on the GPU (r3d_sort.hip); the host emits the pruned octree depth-first:
// a child subtree is a pruned leaf exactly when its code range holds 8^(levels below) codes.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "r3d_hostpool.h"
#include "r3d_internal.h"
#include "r3d_sort_dev.h"
#include "r3d_voxel_dev.h"

struct r3d_voxelset {
  r3d_ctx* ctx = nullptr;
  int device = 0;  // kept so that destroy never has to touch a ctx that may already be gone
  double res = 0.1;
  double factor = 10.0;
  uint64_t* d_table = nullptr;
  uint64_t capacity = 0;  // power of two
  int log2cap = 0;
  unsigned long long* d_counters = nullptr;  // [0] voxels, [1] ignored points, [2] overflow, [3] compaction cursor
  bool pristine = true;   // nothing has gone into the table since it was created / cleared (the merge then need not read it)
};

namespace {

constexpr int kThreads = 256;
using r3d_vox::kEmpty;
using r3d_vox::kLdsKeepBelow;
using r3d_vox::kLdsSlots;
using r3d_vox::lds_set_claim;
using r3d_vox::prev_lane_u64;
using r3d_vox::table_insert;

struct __attribute__((packed, aligned(4))) P3 {
  float x, y, z;
};

// DEDUPE: a workgroup funnels its codes through a small LDS hash set and walks a CONTIGUOUS run of tiles (neighbouring
// image rows fall into the SAME voxels), keeping the set from tile to tile.  Round 2 sent every newly claimed code to the
// global table on the spot: a handful of returning global atomics per tile, each a ~2 us round trip that the whole wave sat
// out (PMC: waves waiting 85 % of their cycles, 109 VALU instructions per 64 points; in-kernel clocks: 46 % in those atomics).
// Round 3: while it walks its tiles a workgroup touches LDS only -- a claimed code simply STAYS in the set -- and the set is
// flushed to the global table as a whole, all 256 lanes inserting in parallel, when it has collected kLdsKeepBelow codes and
// at the end of the run: the round trips are paid once per ~512 codes instead of once per tile.  A code that finds the set full
// (cannot happen below 75 % load) goes to the global table directly.
template <bool DEDUPE, bool COND_BARRIER = false>
__global__ __launch_bounds__(kThreads) void voxel_insert_kernel(const float* __restrict__ xyz, int64_t n, double factor,
                                                                uint64_t* __restrict__ table, int log2cap,
                                                                unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long local_set[DEDUPE ? kLdsSlots : 1];
  // codes gained per tile, three counters in rotation: tile j adds into [j % 3], everyone reads it after the next barrier,
  // thread 0 zeroes [(j + 1) % 3] there -- whose last readers all passed that barrier -- so the running total every thread
  // keeps in a register is the same in all of them and the decision to flush is workgroup-uniform
  __shared__ unsigned local_fill[3];
  const uint64_t mask = ((uint64_t)1 << log2cap) - 1;
  const int lane = threadIdx.x & 63;
  // statistics stay in registers and reach the three global counters once per wave: a per-insert atomicAdd on one word would
  // cap the kernel at that word's ~0.09 G atomics/s.  (Once per WORKGROUP -- flush_counts, r3d_voxel_dev.h -- was tried in round
  // 4 after the merge kernel's lesson: here the 8192 adds trickle in over milliseconds and never queue, and the extra
  // barriers + LDS hop changed the main loop's code generation: 216-222 -> 182 Gpoints/s on scans, 3.26 -> 3.65 ms on the worst
  // case.  Reverted.)
  unsigned n_new = 0, n_ignored = 0, n_over = 0;
  const int64_t n_tiles = (n + kThreads * 4 - 1) / (kThreads * 4);
  const int64_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;  // a contiguous run of tiles per workgroup
  const int64_t tile_lo = (int64_t)blockIdx.x * per_wg, tile_hi = tile_lo + per_wg < n_tiles ? tile_lo + per_wg : n_tiles;
  if (DEDUPE) {
    for (int k = threadIdx.x; k < kLdsSlots; k += kThreads) local_set[k] = kEmpty;
    if (threadIdx.x < 3) local_fill[threadIdx.x] = 0;
  }
  // every code of the set into the global table, the set emptied: all lanes at once, kLdsSlots / kThreads slots each
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < kLdsSlots / kThreads; ++k) {
      const int s = k * kThreads + threadIdx.x;
      const uint64_t code = local_set[s];
      if (code != kEmpty) {
        local_set[s] = kEmpty;
        const int r = table_insert(table, mask, log2cap, code);
        n_new += r > 0 ? 1u : 0u;
        n_over += r < 0 ? 1u : 0u;
      }
    }
  };
  unsigned total = 0, j = 0;  // codes in the set (same value in every thread), tiles done by this workgroup
  // the next tile's points, in flight.  (Round 5: as nontemporal loads -- a 3-float vector type of 4-byte alignment, so that the
  // compiler still tracks them -- 223 vs 223 and 231 vs 214 Gpoints/s on scans, same process: within the noise, not kept.  Inline-asm
  // loads whose wait is a second asm statement further down are not an option at all: the compiler copies and re-uses their
  // destination registers in between, and a load that lands in what has become an address register is a memory fault.)
  P3 pn[4];
  if (tile_lo < tile_hi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = tile_lo * (kThreads * 4) + threadIdx.x + (int64_t)r * kThreads;
      pn[r] = reinterpret_cast<const P3*>(xyz)[i < n ? i : n - 1];
    }
  }
  for (int64_t tile = tile_lo; tile < tile_hi; ++tile, ++j) {
    if (DEDUPE) {
      r3d_vox::lds_settle();
      __syncthreads();  // the previous tile's lookups and its count are done (first tile: the wipe above has landed)
      if (j > 0) total += local_fill[(j - 1) % 3];
      if (threadIdx.x == 0) local_fill[(j + 1) % 3] = 0;
      if (COND_BARRIER) {
        if (total >= (unsigned)kLdsKeepBelow) {  // workgroup-uniform
          flush();
          total = 0;
          __syncthreads();
        }
      } else {
        if (total >= (unsigned)kLdsKeepBelow) {
          flush();
          total = 0;
        }
        r3d_vox::lds_settle();
        __syncthreads();
      }
    }
    const int64_t base = tile * (kThreads * 4) + threadIdx.x;
    // the NEXT tile's points are requested before this tile's are worked on (clamped addresses: unconditional loads), so
    // that the HBM round trip of tile k+1 runs under the key arithmetic and the LDS lookups of tile k
    P3 p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = pn[r];
    if (tile + 1 < tile_hi) {
      const int64_t nbase = base + kThreads * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = nbase + (int64_t)r * kThreads;
        pn[r] = reinterpret_cast<const P3*>(xyz)[i < n ? i : n - 1];
      }
    }
    unsigned claimed = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = base + (int64_t)r * kThreads;
      uint64_t code = kEmpty;
      bool live = i < n;
      if (live && !r3d_vox::voxel_key(p[r].x, p[r].y, p[r].z, factor, &code)) {
        ++n_ignored;
        live = false;
        code = kEmpty;
      }
      // neighbouring pixels mostly fall into the same voxel: a lane whose predecessor carries the same
      // code leaves the insert to it
      const uint64_t prev = prev_lane_u64(code);
      if (live && lane > 0 && prev == code) live = false;
      if (DEDUPE && live) {
        bool mine = false;
        const bool done = lds_set_claim(local_set, code, &mine);
        claimed += mine ? 1u : 0u;
        live = !done;  // a full set (cannot happen: < 512 + 1024 codes in 2048 slots) sends the code on directly
      }
      if (live) {
        const int r2 = table_insert(table, mask, log2cap, code);
        n_new += r2 > 0 ? 1u : 0u;
        n_over += r2 < 0 ? 1u : 0u;
      }
    }
    if (DEDUPE) {  // one LDS add per wave: how many codes the set gained in this tile
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) claimed += __shfl_down(claimed, off, 64);
      if (lane == 0 && claimed) atomicAdd(&local_fill[j % 3], claimed);
    }
  }
  if (DEDUPE) {
    __syncthreads();   // every lane's last lookups have landed
    flush();
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off, 64);
    n_ignored += __shfl_down(n_ignored, off, 64);
    n_over += __shfl_down(n_over, off, 64);
  }
  if (lane == 0) {
    if (n_new) atomicAdd(&counters[0], (unsigned long long)n_new);
    if (n_ignored) atomicAdd(&counters[1], (unsigned long long)n_ignored);
    if (n_over) atomicAdd(&counters[2], (unsigned long long)n_over);
  }
}


// ---- sort-merge insert: the path for clouds whose points mostly fall into DIFFERENT voxels -----------------------------------
// The LDS-set kernel above wins when neighbouring pixels share voxels (scans: tens of points per voxel).  On a cloud without
// surfaces (BASELINE C2's random depth: 49.2 M points -> 48.4 M voxels) nothing dedupes and every point ends as a 64-bit CAS
// at a random place of a 1 GB table: 3.3 ms, ~80 B written per 8-byte key (profiles/r03_all_kernels.json) -- every CAS drags
// a whole line through HBM and back.  Random access is the cost, so this path has none.
//
// Round 5 form.  A key's place in the table is the top bits of h48 = key * G mod 2^48, a BIJECTION of the 48-bit keys
// (r3d_voxel_dev.h).  The top 16 bits of h48 name one of 65536 PIECES of the table (hi | lo, a byte each); within a piece a key
// is the low 32 bits of h48.  So the sort moves 4-byte remainders (+ one digit byte while it is still needed), not 8-byte words:
//   voxel_bin_kernel        12 B/point in; keys, ranked by lo, straight into per-XCD bin segments: rem (4 B) + hi (1 B) out -- no
//                           histogram in front, no round trip of the elements (see below);
//   segment_histogram_kernel  of the segments' hi bytes, per 4096-element chunk: 1 B/point;
//   segment_scatter_kernel  by hi: 5 B in, rem out (4 B) in piece order -- and the RUN STARTS of all 65536 pieces for free: a chunk
//                           lies inside one lo bin, so starts[hi | lo] is the offset in bin hi of lo's first chunk;
//   voxel_merge32_kernel / voxel_merge_kernel  persistent workgroups walk the table's 2048..8192-slot regions: the region is
//                           initialised in LDS (or comes in, when the table is not fresh), the remainders of its piece(s) are
//                           inserted THERE (LDS compare-and-swap, linear probing from the home slot: the same placement rule as
//                           table_insert), keys are rebuilt (key = (piece << 32 | rem) * G^-1 mod 2^48) on the way out in 16-byte
//                           stores.  A probe that runs off the region's end is deferred to a spill list (voxel_spill_kernel,
//                           ordinary CAS, ~0.1 % of the keys at load 0.4).
// HBM sees streams only.  Per point: 17 (keys + first pass) + 1 + 9 (second pass) + 4 + 8 x slots per point (merge); the first form
// of this round (a key kernel, a dense first pass behind its histogram) moved 43 + 8 x slots, rounds 2-4 76 + 8 x slots.
constexpr int kRegionMinLog2 = 11;   // slots per LDS region: 2048 (16 KB of LDS) ... 8192 (64 KB)
constexpr int kRegionMaxLog2 = 13;
constexpr int kPieceBits = 16;       // the table is sorted into 2^16 pieces (the top 16 bits of h48)
constexpr uint32_t kPieces = 1u << kPieceBits;
using r3d_vox::hash48;
using r3d_vox::kMask48;
using r3d_vox::unhash48;

// ---- the sort's front half without a histogram in front of the first pass ------------------------------------------------------
// A dense radix pass needs every (tile, bin) offset before it can write: a histogram in front of it, i.e. a round trip of the
// elements through HBM -- whoever makes the keys writes them (6 B/point) for the scatter to read back (6 B/point) once the scan
// is done (this round's first form: voxel_keys_kernel + piece_scatter_kernel<1>, 372 us of the insert's 930).  The first pass does
// not have to be dense, though, nor in any order.  Here every bin has one SEGMENT per XCD, with room for 1.125 x what
// a hash spreads into it, and a cursor: a tile's workgroup turns its points into keys, ranks them by lo, takes room for each of
// its 256 runs with one returning add on the cursor of (lo, its XCD) and writes -- 12 B/point in, 5 out, nothing in between.  The
// workgroups that share a cursor run on one XCD: runs taken one after the other lie side by side and the lines they share are
// completed in that XCD's L2 (private segments per workgroup, tried first, were not: a run's neighbour came a tile later, the
// line had left the L2 half written -- 437 MB written for 241, and a partly written line is slow at the memory:
// tools/scatter_runs.hip).  The second pass walks the segments in chunks of 4096 -- a tile lies inside ONE lo bin, so the run
// starts of the 65536 pieces are simply its offsets -- and is dense as before.  Points without a key and the previous lane's
// duplicates are dropped here instead of travelling on as markers.
// A segment that is full (keys that crowd into one bin: every pixel without depth of a frame is the same point) sends what it
// cannot take to the list of deferred keys, which voxel_spill_kernel inserts the ordinary way.
constexpr int kXcds = 8;
constexpr int kSegments = 256 * kXcds;
constexpr int kCursorStride = 32;   // words between two cursors: a line each
struct SegPlan {
  int cap = kSortTile;       // elements per segment (a multiple of 4096)
  int chunks = 1;            // second-pass tiles per segment
  int n_tiles2 = kSegments;
};

static SegPlan seg_plan(int64_t n_points) {
  SegPlan p;
  const int64_t mean = (n_points + kSegments - 1) / kSegments;
  p.chunks = (int)((mean + mean / 8 + 1024 + kSortTile - 1) / kSortTile);   // (C2: 24 000 +- 155 elements per segment, room for 28 672)
  p.cap = p.chunks * kSortTile;
  p.n_tiles2 = kSegments * p.chunks;
  return p;
}

constexpr int kBinThreads = 512;
constexpr int kBinRounds = kSortTile / kBinThreads;
// voxel_bin_kernel's rare way out, kept out of line so that it costs the kernel no registers: a segment is full (keys that crowd
// into one bin), the tile's elements that found no room go to the deferred list -- with ONE add on the list's counter per tile (a
// point that occurs millions of times would otherwise queue a hundred thousand adds at that one address, 11 ns each: C2's cloud
// with a fifth of its pixels without depth spent 0.9 ms there).  Called by all threads of the workgroup.
__device__ __attribute__((noinline)) void defer_full_segments(const uint2* s_el, const uint32_t* s_base, uint32_t* s_defer, int n_live, int cap,
                                                              uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_count,
                                                              unsigned long long spill_cap) {
  if (threadIdx.x == 0) s_defer[0] = 0;
  __syncthreads();
  uint32_t n_mine = 0;
  for (int j = threadIdx.x; j < n_live; j += kBinThreads) n_mine += s_base[s_el[j].y & 0xff] + (uint32_t)j >= (uint32_t)cap ? 1u : 0u;
  uint32_t at = n_mine ? atomicAdd(&s_defer[0], n_mine) : 0u;   // this thread's places among the tile's deferred keys
  r3d_vox::lds_settle();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long first = atomicAdd(spill_count, (unsigned long long)s_defer[0]);
    s_defer[1] = (uint32_t)first;
    s_defer[2] = (uint32_t)(first >> 32);
  }
  __syncthreads();
  const unsigned long long first = (unsigned long long)s_defer[1] | ((unsigned long long)s_defer[2] << 32);
  for (int j = threadIdx.x; j < n_live; j += kBinThreads) {
    const uint2 el = s_el[j];
    if (s_base[el.y & 0xff] + (uint32_t)j < (uint32_t)cap) continue;
    if (first + at < spill_cap) spill[first + at] = unhash48(((uint64_t)(el.y & 0xffffu) << 32) | el.x);   // el.y = lo | hi << 8: h48's top 16 bits
    ++at;
  }
}

// flags[2..3]: points without a key (64 bits; added to the set's counter by voxel_spill_kernel).  cursors[(lo * 8 + xcd) * kCursorStride]: elements in the segment (may exceed cap: clamp).
// 512 threads, eight points each: the kernel waits for latencies in turn (points, LDS adds, the cursor, the stores), so it wants
// waves -- four workgroups of eight per CU fill it (256 threads x 16 points: five of four, 20 of 32 wave slots: 272 -> 252 us).
__global__ __launch_bounds__(kBinThreads, 8) void voxel_bin_kernel(const float* __restrict__ xyz, int64_t n, double factor, float safe_abs,
                                                                   int n_tiles, int cap, uint32_t* __restrict__ seg_rem,
                                                                   uint8_t* __restrict__ seg_hi, uint32_t* __restrict__ cursors,
                                                                   uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_count,
                                                                   unsigned long long spill_cap, uint32_t* __restrict__ flags) {
  constexpr int kBins = r3d_sort::kBins;
  __shared__ uint2 s_el[kSortTile];   // the tile in bin order: rem, lo | hi << 8 (one LDS write and one read per element)
  __shared__ uint32_t s_base[kBins];
  __shared__ uint32_t bin_count[kBins], bin_start[kBins], wave_sum[4], s_defer[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & (kXcds - 1);   // (workgroups go round the XCDs; nothing but locality depends on it)
  unsigned n_ignored = 0;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {   // (gridDim.x is a multiple of 8: a workgroup stays with its cursors)
    const int64_t t0 = (int64_t)tile * kSortTile;
    const uint32_t n_tile = n - t0 < (int64_t)kSortTile ? (uint32_t)(n - t0) : (uint32_t)kSortTile;
    const bool full = n_tile == (uint32_t)kSortTile;
    const P3* __restrict__ tile_xyz = reinterpret_cast<const P3*>(xyz) + t0;
    uint32_t rem[kBinRounds], dc[kBinRounds], live_mask = 0;   // dc: lo | hi << 8, later | the place in the tile's bin << 16
    // the lane's eight points: nontemporal 12-byte loads, all in flight together, in ONE asm statement with the wait that
    // completes them (the compiler does not count inline-asm loads: r3d_apply.hip).  Plain loads four at a time read the cloud
    // at 3.9 TB/s (153 us with everything else switched off), this form at 6.9 (86 us).  One lane offset and eight scalar
    // bases (a full tile's addresses are affine in the round) instead of eight 64-bit lane addresses: no spills at 64 registers.
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    f32x3 raw[kBinRounds];
    static_assert(kBinRounds == 8, "eight loads are written out below");
    if (full) {
      const uint32_t voff = threadIdx.x * 12u;
      const char* b0 = reinterpret_cast<const char*>(tile_xyz);
      constexpr int kStep = kBinThreads * 12;
#define R3D_LD3(o, b) "global_load_dwordx3 %" #o ", %8, %" #b " nt\n\t"
      asm volatile(R3D_LD3(0, 9) R3D_LD3(1, 10) R3D_LD3(2, 11) R3D_LD3(3, 12) R3D_LD3(4, 13) R3D_LD3(5, 14) R3D_LD3(6, 15) R3D_LD3(7, 16)
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(raw[0]), "=&v"(raw[1]), "=&v"(raw[2]), "=&v"(raw[3]), "=&v"(raw[4]), "=&v"(raw[5]), "=&v"(raw[6]), "=&v"(raw[7])
                   : "v"(voff), "s"(b0), "s"(b0 + kStep), "s"(b0 + 2 * kStep), "s"(b0 + 3 * kStep), "s"(b0 + 4 * kStep), "s"(b0 + 5 * kStep),
                     "s"(b0 + 6 * kStep), "s"(b0 + 7 * kStep)
                   : "memory");
#undef R3D_LD3
    } else {   // the cloud's last tile: clamped addresses, ordinary loads
#pragma unroll
      for (int r = 0; r < kBinRounds; ++r) {
        const uint32_t e = (uint32_t)r * kBinThreads + threadIdx.x;
        const P3 v = tile_xyz[e < n_tile ? e : n_tile - 1];
        raw[r] = f32x3{v.x, v.y, v.z};
      }
    }
#pragma unroll
    for (int q = 0; q < kBinRounds / 4; ++q) {
      P3 p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = P3{raw[q * 4 + r].x, raw[q * 4 + r].y, raw[q * 4 + r].z};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t e = (uint32_t)(q * 4 + r) * kBinThreads + threadIdx.x;
        uint64_t key = kEmpty;
        bool live = full || e < n_tile;
        if (fabsf(p[r].x) < safe_abs && fabsf(p[r].y) < safe_abs && fabsf(p[r].z) < safe_abs) {   // in range for sure (a NaN fails)
          const uint32_t ix = (uint32_t)((int)floor(factor * (double)p[r].x) + r3d_vox::kTreeMaxVal);
          const uint32_t iy = (uint32_t)((int)floor(factor * (double)p[r].y) + r3d_vox::kTreeMaxVal);
          const uint32_t iz = (uint32_t)((int)floor(factor * (double)p[r].z) + r3d_vox::kTreeMaxVal);
          key = (uint64_t)(ix | (iy << 16)) | ((uint64_t)iz << 32);
        } else if (live && !r3d_vox::voxel_key(p[r].x, p[r].y, p[r].z, factor, &key)) {
          ++n_ignored;
          live = false;
        }
        if (!live) key = kEmpty;
        const uint64_t prev = prev_lane_u64(key);
        if (lane > 0 && prev == key) live = false;
        const uint64_t h = hash48(key & kMask48);
        if (live && h == kMask48) {   // the one key whose h48 reads as "no key" in the merge: it takes the deferred way in
          const unsigned long long at = atomicAdd(spill_count, 1ull);
          if (at < spill_cap) spill[at] = key;
          live = false;
        }
        rem[q * 4 + r] = (uint32_t)h;
        dc[q * 4 + r] = (uint32_t)(h >> 32) & 0xffffu;
        live_mask |= (live ? 1u : 0u) << (q * 4 + r);
      }
    }
    if (threadIdx.x < kBins) bin_count[threadIdx.x] = 0;
    __syncthreads();   // (also: the previous tile's readers of the staging arrays are through)
#pragma unroll
    for (int r = 0; r < kBinRounds; ++r)   // any order inside a bin: the arrival number, kept beside the digits
      if ((live_mask >> r) & 1u) dc[r] |= atomicAdd(&bin_count[dc[r] & 0xff], 1u) << 16;
    r3d_vox::lds_settle();
    __syncthreads();
    uint32_t mine = 0, inc = 0, base = 0;
    if (threadIdx.x < kBins) {   // thread = bin (waves 0..3)
      mine = bin_count[threadIdx.x];
      inc = r3d_sort::wave_inclusive_scan(mine, lane);
      if (lane == 63) wave_sum[wave] = inc;
      // room for this tile's run of the bin: the add is on its way while the tile is staged
      if (mine) base = atomicAdd(&cursors[(threadIdx.x * kXcds + xcd) * kCursorStride], mine);
    }
    __syncthreads();
    if (threadIdx.x < kBins) {
      uint32_t start = inc - mine;
      for (int w = 0; w < wave; ++w) start += wave_sum[w];
      bin_start[threadIdx.x] = start;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kBinRounds; ++r) {
      if ((live_mask >> r) & 1u) {
        const uint32_t at = bin_start[dc[r] & 0xff] + (dc[r] >> 16);
        s_el[at] = uint2{rem[r], dc[r] & 0xffffu};
      }
    }
    if (threadIdx.x < kBins) {
      s_base[threadIdx.x] = base - bin_start[threadIdx.x];   // (modulo 2^32: element j of the bin order goes to s_base[its bin] + j)
    }
    __syncthreads();
    const int n_live = (int)(bin_start[kBins - 1] + bin_count[kBins - 1]);
    bool full_seg = false;
#pragma unroll 4
    for (int j = threadIdx.x; j < n_live; j += kBinThreads) {
      const uint2 el = s_el[j];
      const uint32_t d = el.y & 0xff;
      const uint32_t at = s_base[d] + (uint32_t)j;
      if (at < (uint32_t)cap) {
        const uint64_t to = (uint64_t)(d * kXcds + xcd) * (uint64_t)cap + at;
        seg_rem[to] = el.x;
        seg_hi[to] = (uint8_t)(el.y >> 8);
      } else {
        full_seg = true;
      }
    }
    if (__syncthreads_or(full_seg)) defer_full_segments(s_el, s_base, s_defer, n_live, cap, spill, spill_count, spill_cap);   // (uniform)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n_ignored += __shfl_down(n_ignored, off, 64);
  if (lane == 0 && n_ignored) atomicAdd(reinterpret_cast<unsigned long long*>(flags + 2), (unsigned long long)n_ignored);
}

// hist[hi][tile] for the second pass's tiles: tile T = chunk T % chunks of segment T / chunks.  Eight tiles per workgroup, two per
// wave (whole-sector stores, as byte_histogram_kernel).
__global__ __launch_bounds__(kThreads) void segment_histogram_kernel(const uint8_t* __restrict__ seg_hi, const uint32_t* __restrict__ cursors,
                                                                     int cap, int chunks, int n_tiles2, uint32_t* __restrict__ hist, int stride) {
  __shared__ uint32_t bins[8][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < 8; ++w) bins[w][threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int slot = wave * 2 + half;
    const int tile = blockIdx.x * 8 + slot;
    if (tile >= n_tiles2) continue;
    const int seg = tile / chunks, chunk = tile % chunks;
    uint32_t count = cursors[seg * kCursorStride];
    if (count > (uint32_t)cap) count = cap;
    const uint32_t c0 = (uint32_t)chunk * kSortTile;
    if (count <= c0) continue;
    const uint32_t n_tile = count - c0 < (uint32_t)kSortTile ? count - c0 : (uint32_t)kSortTile;
    const uint8_t* __restrict__ bytes = seg_hi + (uint64_t)seg * cap + c0;
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t at = (uint32_t)(k * 64 + lane) * 16;
      v[k] = at < n_tile ? *reinterpret_cast<const uint4*>(bytes + at) : uint4{0, 0, 0, 0};   // (inside the segment: cap is a multiple of 4096)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t at = (uint32_t)(k * 64 + lane) * 16;
      if (at >= n_tile) continue;
      const uint32_t w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      const uint32_t valid = n_tile - at < 16u ? n_tile - at : 16u;
      if (valid == 16u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          atomicAdd(&bins[slot][w4[c] & 0xff], 1u);
          atomicAdd(&bins[slot][(w4[c] >> 8) & 0xff], 1u);
          atomicAdd(&bins[slot][(w4[c] >> 16) & 0xff], 1u);
          atomicAdd(&bins[slot][w4[c] >> 24], 1u);
        }
      } else {
        for (uint32_t c = 0; c < valid; ++c) atomicAdd(&bins[slot][(w4[c >> 2] >> (8 * (c & 3))) & 0xff], 1u);
      }
    }
  }
  r3d_vox::lds_settle();
  __syncthreads();
  uint32_t* row = hist + (int64_t)threadIdx.x * stride + blockIdx.x * 8;
  if (blockIdx.x * 8 + 8 <= n_tiles2) {   // (rows are 32-byte aligned: r3d_sort_stride)
    reinterpret_cast<uint4*>(row)[0] = uint4{bins[0][threadIdx.x], bins[1][threadIdx.x], bins[2][threadIdx.x], bins[3][threadIdx.x]};
    reinterpret_cast<uint4*>(row)[1] = uint4{bins[4][threadIdx.x], bins[5][threadIdx.x], bins[6][threadIdx.x], bins[7][threadIdx.x]};
  } else {
    for (int w = 0; w < 8 && blockIdx.x * 8 + w < n_tiles2; ++w) row[w] = bins[w][threadIdx.x];
  }
}

// (The first pass's trick a level down -- a segment and a cursor per PIECE, no histogram and no scan in front of the second pass --
// was built and measured: 95 MB less traffic, but the 256 returning adds per tile cost the second pass what the two small kernels
// had (150 us against 102 + 24 + 12) and C2 came out at 0.665 ms against 0.65.  Not kept.)
// The second pass over segments: one workgroup per tile (a chunk of a segment: one lo), digit = hi, any order inside a bin, the
// remainders out in piece order -- and starts[hi * 256 + lo] from the tile that comes first in its lo bin: its own offset in bin hi.
// 512 threads; a thread takes EIGHT CONSECUTIVE elements (any assignment will do for an any-order ranking): two 16-byte loads of
// remainders and one 8-byte load of hi bytes instead of sixteen 4-byte and 1-byte ones.
__global__ __launch_bounds__(kBinThreads, 8) void segment_scatter_kernel(const uint32_t* __restrict__ seg_rem, const uint8_t* __restrict__ seg_hi,
                                                                         const uint32_t* __restrict__ cursors, int cap, int chunks,
                                                                         const uint32_t* __restrict__ hist, int stride,
                                                                         const uint32_t* __restrict__ totals, uint32_t* __restrict__ rem_out,
                                                                         uint32_t* __restrict__ starts) {
  constexpr int kBins = r3d_sort::kBins;
  __shared__ uint2 s_el[kSortTile];   // the tile in bin order: rem, hi
  __shared__ uint32_t s_to[kBins];    // where bin d's first element goes, minus its place in the tile
  __shared__ uint32_t bin_count[kBins], bin_start[kBins], wave_sum[4], total_of_wave[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = r3d_sort::xcd_contiguous(blockIdx.x, gridDim.x);
  const int seg = tile / chunks, chunk = tile % chunks;
  uint32_t count = cursors[seg * kCursorStride];
  if (count > (uint32_t)cap) count = cap;
  const uint32_t c0 = (uint32_t)chunk * kSortTile;
  const bool first_of_lo = tile % (chunks * kXcds) == 0;
  if (count <= c0 && !first_of_lo && tile != 0) return;   // (uniform) nothing in this chunk, nothing to announce
  const uint32_t n_tile = count <= c0 ? 0u : (count - c0 < (uint32_t)kSortTile ? count - c0 : (uint32_t)kSortTile);
  // the elements first: they are on their way while the bins' bases are worked out
  const uint64_t base = (uint64_t)seg * cap + c0;
  const uint32_t e0 = threadIdx.x * kBinRounds;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x4 ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
  u32x2 hb = {0, 0};
  if (e0 < n_tile) {   // (whole loads: the segment's capacity is a multiple of the tile)
    ra = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(seg_rem + base + e0));
    rb = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(seg_rem + base + e0) + 1);
    hb = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(seg_hi + base + e0));
  }
  uint32_t my_base = 0;
  if (threadIdx.x < kBins) {   // thread = hi: bin base = the totals below it, + this tile's offset in the bin
    bin_count[threadIdx.x] = 0;
    const uint32_t tot = totals[threadIdx.x];
    const uint32_t inc = r3d_sort::wave_inclusive_scan(tot, lane);
    if (lane == 63) total_of_wave[wave] = inc;
    my_base = inc - tot + hist[(int64_t)threadIdx.x * stride + tile];
  }
  __syncthreads();
  if (threadIdx.x < kBins) {
    for (int w = 0; w < wave; ++w) my_base += total_of_wave[w];
    if (first_of_lo) starts[threadIdx.x * kBins + seg / kXcds] = my_base;
    if (tile == 0 && threadIdx.x == kBins - 1) {
      const uint32_t total = my_base - hist[(int64_t)threadIdx.x * stride + tile] + totals[kBins - 1];
      starts[kPieces] = total;
      starts[kPieces + 1] = total;
    }
  }
  if (n_tile == 0) return;   // (uniform)
  const uint32_t rem[kBinRounds] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
  uint32_t dg[kBinRounds];
#pragma unroll
  for (int r = 0; r < kBinRounds; ++r) dg[r] = ((r < 4 ? hb.x : hb.y) >> (8 * (r & 3))) & 0xff;
#pragma unroll
  for (int r = 0; r < kBinRounds; ++r)
    if (e0 + r < n_tile) dg[r] |= atomicAdd(&bin_count[dg[r]], 1u) << 16;
  r3d_vox::lds_settle();
  __syncthreads();
  uint32_t mine = 0, inc = 0;
  if (threadIdx.x < kBins) {
    mine = bin_count[threadIdx.x];
    inc = r3d_sort::wave_inclusive_scan(mine, lane);
    if (lane == 63) wave_sum[wave] = inc;
  }
  __syncthreads();
  if (threadIdx.x < kBins) {
    uint32_t start = inc - mine;
    for (int w = 0; w < wave; ++w) start += wave_sum[w];
    bin_start[threadIdx.x] = start;
    s_to[threadIdx.x] = my_base - start;   // (modulo 2^32)
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kBinRounds; ++r)
    if (e0 + r < n_tile) s_el[bin_start[dg[r] & 0xff] + (dg[r] >> 16)] = uint2{rem[r], dg[r] & 0xff};
  __syncthreads();
#pragma unroll 4
  for (uint32_t j = threadIdx.x; j < n_tile; j += kBinThreads) {
    const uint2 el = s_el[j];
    rem_out[s_to[el.y] + j] = el.x;
  }
}

// REGION_LOG2: slots per LDS region.  SUB: several pieces per region (tables below 2^27 slots, whose pieces have fewer than
// 2048 slots: a region then takes 2^sub_log2 consecutive ones); otherwise a piece IS a region and sub_log2 is 0.
//
// What this kernel waits for is not only HBM (round 5, one stage switched off at a time on one box: everything 350 us; without
// the compare-and-swaps 294, without the element loads 267, without the table stores 248, the loop's skeleton alone -- bounds,
// barriers, LDS initialisation -- 96).  Two things are therefore done differently from round 4, each measured in the SAME process
// against the old form (BATCHED / PIPED = false; medians of 7 launches, twice): bounds alone 412 -> 397 us, attempts alone
// 412 -> 401, both 412 -> 385.  (1) BATCHED: a workgroup fetches the bounds of 128 of its regions at once into LDS instead of
// two scalar loads per region that the next barrier waits for (their lines are evicted from the L2 by the table stream all the
// time: a memory round trip per region, exposed); (2) PIPED: a thread's four elements make their first compare-and-swap attempt
// back to back, four LDS round trips in flight, before the (rare) re-probes are walked one by one.  (The kernel's duration
// differs by up to 1.4x between boxes of the pool -- 283 us and 412 us for the same binary -- while the streaming kernels
// around it agree within 2 %: compare variants inside one process only.)
template <int REGION_LOG2, bool SUB, bool BATCHED = true, bool PIPED = true>
__global__ __launch_bounds__(kThreads) void voxel_merge_kernel(const uint32_t* __restrict__ rems, const uint32_t* __restrict__ starts,
                                                               uint32_t n_regions, int sub_log2_arg, uint64_t* __restrict__ table, int log2cap,
                                                               uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_count,
                                                               unsigned long long spill_cap, int pristine,
                                                               unsigned long long* __restrict__ partials) {
  const int sub_log2 = SUB ? sub_log2_arg : 0;
  constexpr int kSlots = 1 << REGION_LOG2;
  constexpr int kAhead = 4;    // elements per thread fetched one region ahead (1024 per region: a 2048-slot region holds ~750 at load 0.36)
  constexpr int kBatch = 128;  // regions whose bounds a workgroup holds in LDS at a time
  __shared__ unsigned wg_count[2];
  if (threadIdx.x < 2) wg_count[threadIdx.x] = 0;   // (ordered before its first use by the barrier in front of the adds at the end)
  __shared__ __attribute__((aligned(16))) unsigned long long region[kSlots];
  __shared__ uint32_t s_lo[kBatch + 1], s_hi[kBatch + 1];
  __shared__ unsigned changed;
  const int lane = threadIdx.x & 63;
  unsigned n_new = 0, n_over = 0;
  bool mine_changed = false;
  // element i of region r: which piece it belongs to (its position says so), hence its h48, its key and its home slot
  auto hash_of = [&](uint32_t r, uint32_t i, uint32_t rem) -> uint64_t {
    uint32_t piece = r << sub_log2;
    if (SUB) {   // the last piece of the region whose run starts at or before i
      uint32_t a = 0, b = (1u << sub_log2) - 1;
      while (a < b) {
        const uint32_t mid = (a + b + 1) >> 1;
        if (starts[piece + mid] <= i) a = mid; else b = mid - 1;
      }
      piece += a;
    }
    return ((uint64_t)piece << 32) | rem;
  };
  // the probe from slot s on (the first attempt at the home slot may already have been made: `old` is what it found)
  auto probe_on = [&](uint64_t key, uint32_t s, unsigned long long old) {
    bool done = false;
    for (;;) {
      if (old == kEmpty) {
        ++n_new;
        mine_changed = true;
        done = true;
        break;
      }
      if (old == key) {
        done = true;
        break;
      }
      if (++s >= (uint32_t)kSlots) break;
      old = atomicCAS(&region[s], (unsigned long long)kEmpty, (unsigned long long)key);
    }
    if (!done) {   // every slot from home to the region's end is taken by others: the probe goes on in the next region -- later
      const unsigned long long at = atomicAdd(spill_count, 1ull);
      if (at < spill_cap) spill[at] = key; else ++n_over;
    }
  };
  auto insert_one = [&](uint32_t r, uint32_t i, uint32_t rem) {
    const uint64_t h = hash_of(r, i, rem);
    if (h == kMask48) return;   // an element that carries no key
    const uint64_t key = unhash48(h);
    const uint32_t s = (uint32_t)(h >> (48 - log2cap)) & (kSlots - 1);
    probe_on(key, s, atomicCAS(&region[s], (unsigned long long)kEmpty, (unsigned long long)key));
  };
  auto fetch = [&](uint32_t lo_, uint32_t hi_, uint32_t (&dst)[kAhead]) {
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const uint32_t i = lo_ + (uint32_t)k * kThreads + threadIdx.x;
      dst[k] = i < hi_ ? rems[i] : 0;
    }
  };
  // the bounds of the regions of batch `it0`: iteration j of the batch works on region blockIdx.x + (it0 + j) * gridDim.x
  auto load_bounds = [&](uint32_t it0) {
    if (threadIdx.x <= (unsigned)kBatch) {
      const uint64_t rr = (uint64_t)blockIdx.x + (uint64_t)(it0 + threadIdx.x) * gridDim.x;
      uint32_t a = 0, b = 0;
      if (rr < n_regions) {
        a = starts[(uint32_t)rr << sub_log2];
        b = starts[((uint32_t)rr + 1) << sub_log2];
      }
      s_lo[threadIdx.x] = a;
      s_hi[threadIdx.x] = b;
    }
  };
  // The loop is a pipeline: a region's first elements are fetched while the region before it is being worked on.
  uint32_t it = 0, r = blockIdx.x;
  if (BATCHED) load_bounds(0);
  __syncthreads();
  uint32_t lo = 0, hi = 0;
  if (BATCHED) {
    lo = s_lo[0];
    hi = s_hi[0];
  } else if (r < n_regions) {
    lo = starts[r << sub_log2];
    hi = starts[(r + 1) << sub_log2];
  }
  uint32_t cur[kAhead];
  fetch(lo, hi, cur);
  while (r < n_regions) {   // workgroup-uniform trip count
    const uint32_t j = it % kBatch;
    const uint32_t rn = r + gridDim.x;
    uint32_t lon = 0, hin = 0;
    if (BATCHED) {
      lon = s_lo[j + 1];   // (entry kBatch of a batch = entry 0 of the next one)
      hin = s_hi[j + 1];
    } else if (rn < n_regions) {
      lon = starts[rn << sub_log2];
      hin = starts[(rn + 1) << sub_log2];
    }
    uint32_t nxt[kAhead];
    if (lo != hi) {   // (a region that received nothing is not even read)
      ulonglong2* g = reinterpret_cast<ulonglong2*>(table + ((uint64_t)r << REGION_LOG2));
      r3d_vox::lds_barrier();   // the previous region's write-back has read the LDS copy (its stores may still be in flight)
      if (threadIdx.x == 0) changed = 0;
      if (pristine) {   // nothing has been inserted since the set was cleared: the region is known to be empty, half the stream saved
#pragma unroll
        for (int k = 0; k < kSlots / 2 / kThreads; ++k) reinterpret_cast<ulonglong2*>(region)[k * kThreads + threadIdx.x] = ulonglong2{kEmpty, kEmpty};
      } else {
#pragma unroll
        for (int k = 0; k < kSlots / 2 / kThreads; ++k) reinterpret_cast<ulonglong2*>(region)[k * kThreads + threadIdx.x] = g[k * kThreads + threadIdx.x];
      }
      r3d_vox::lds_barrier();
      fetch(lon, hin, nxt);   // in flight while this region's keys go in (and across the barriers: they order LDS only)
      mine_changed = false;
      if (!PIPED) {
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
          const uint32_t i = lo + (uint32_t)k * kThreads + threadIdx.x;
          if (i < hi) insert_one(r, i, cur[k]);
        }
      } else {
        // first attempts of the thread's (up to) four elements back to back, then the re-probes
        uint64_t key[kAhead];
        uint32_t slot[kAhead];
        unsigned long long old[kAhead];
        bool has[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
          const uint32_t i = lo + (uint32_t)k * kThreads + threadIdx.x;
          const uint64_t h = i < hi ? hash_of(r, i, cur[k]) : kMask48;
          has[k] = h != kMask48;
          key[k] = unhash48(h);
          slot[k] = (uint32_t)(h >> (48 - log2cap)) & (kSlots - 1);
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k)
          old[k] = has[k] ? atomicCAS(&region[slot[k]], (unsigned long long)kEmpty, (unsigned long long)key[k]) : 0ull;
#pragma unroll
        for (int k = 0; k < kAhead; ++k)
          if (has[k]) probe_on(key[k], slot[k], old[k]);
      }
      for (uint32_t i = lo + kAhead * kThreads + threadIdx.x; i < hi; i += kThreads) insert_one(r, i, rems[i]);   // a longer run than usual
      if (mine_changed) changed = 1;   // (benign race: everybody writes the same value)
      r3d_vox::lds_barrier();
      if (changed) {   // (nontemporal stores changed nothing here in round 4; in the 32-bit form they are worth 8 %)
#pragma unroll
        for (int k = 0; k < kSlots / 2 / kThreads; ++k) g[k * kThreads + threadIdx.x] = reinterpret_cast<const ulonglong2*>(region)[k * kThreads + threadIdx.x];
      }
    } else {
      fetch(lon, hin, nxt);
    }
#pragma unroll
    for (int k = 0; k < kAhead; ++k) cur[k] = nxt[k];
    r = rn;
    lo = lon;
    hi = hin;
    ++it;
    if (BATCHED && it % kBatch == 0) {   // the next batch of bounds (workgroup-uniform)
      __syncthreads();        // everybody has read entry kBatch
      load_bounds(it);
      __syncthreads();
    }
  }
  // The counts leave as ONE pair of words per workgroup, each in a slot of its own, summed by voxel_spill_kernel.  (One
  // atomicAdd per wave on the set's counters -- the first form -- was what the whole kernel waited for: adds to ONE address
  // complete at ~0.09 G/s on this chip, 32768 of them = 360 of the launch's 455 us; with one workgroup per region, 2.9 ms.)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off, 64);
    n_over += __shfl_down(n_over, off, 64);
  }
  r3d_vox::lds_barrier();
  if (lane == 0) {
    if (n_new) atomicAdd(&wg_count[0], n_new);
    if (n_over) atomicAdd(&wg_count[1], n_over);
  }
  r3d_vox::lds_barrier();
  if (threadIdx.x < 2) partials[2 * (uint64_t)blockIdx.x + threadIdx.x] = wg_count[threadIdx.x];
}

// The merge with 32-BIT slots in LDS, for tables whose regions ARE pieces (2^27 slots and more): inside a region every key
// shares the top 16 bits of h48, so a slot only needs the 32-bit remainder -- half the LDS traffic of the 64-bit form, 32-bit
// compare-and-swaps (the 64-bit returning ones are what this kernel spends its LDS time on), and the home slot is simply the
// remainder's top bits.  0xffffffff marks a free slot, 0xfffffffe (a table that is not fresh only) a slot that holds a key of
// ANOTHER piece -- put there by the CAS path's probing across a region's end -- which must stay as it is; the two keys per piece
// whose remainders are those values take the deferred way in.
// Tried on top of it, each in a same-process A/B, and not kept: two / four pieces per iteration (one set of barriers for 4096 /
// 8192 slots: 345 -> 385 / 520 us) and 128-thread workgroups (4096 of them: 344, no gain) -- the kernel wants what it has, many
// small independent regions; two LDS copies of a region so that region k is written back
// while region k + 1 is filled (one barrier fewer per region: 363 -> 393 us, slower); a read-only sweep of the remainders into
// the Infinity Cache in front of the launch (359 vs 362 us, + 29 us for the sweep); the deferred keys through an LDS list and
// one counter add per workgroup (379-391 either way).
template <int REGION_LOG2, bool PRISTINE>
__global__ __launch_bounds__(kThreads) void voxel_merge32_kernel(const uint32_t* __restrict__ rems, const uint32_t* __restrict__ starts,
                                                                 uint32_t n_regions, uint64_t* __restrict__ table,
                                                                 uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_count,
                                                                 unsigned long long spill_cap, unsigned long long* __restrict__ partials) {
  constexpr int kSlots = 1 << REGION_LOG2;
  constexpr int kAhead = 4;
  constexpr int kBatch = 128;
  constexpr int kPairs = kSlots / 2 / kThreads;   // slot pairs per thread
  constexpr uint32_t kFree = 0xffffffffu, kForeign = 0xfffffffeu;
  __shared__ unsigned wg_count[2];
  if (threadIdx.x < 2) wg_count[threadIdx.x] = 0;
  __shared__ __attribute__((aligned(16))) uint32_t region[kSlots];
  __shared__ uint32_t s_lo[kBatch + 1], s_hi[kBatch + 1];
  __shared__ unsigned changed;
  const int lane = threadIdx.x & 63;
  unsigned n_new = 0, n_over = 0;
  bool mine_changed = false;
  auto defer = [&](uint64_t key) {
    const unsigned long long at = atomicAdd(spill_count, 1ull);
    if (at < spill_cap) spill[at] = key; else ++n_over;
  };
  auto probe_on = [&](uint32_t piece, uint32_t rem, uint32_t s, uint32_t old) {
    for (;;) {
      if (old == kFree) {
        ++n_new;
        mine_changed = true;
        return;
      }
      if (old == rem) return;
      if (++s >= (uint32_t)kSlots) break;
      old = atomicCAS(&region[s], kFree, rem);
    }
    defer(unhash48(((uint64_t)piece << 32) | rem));   // every slot from home to the region's end is taken: the probe goes on later
  };
  auto fetch = [&](uint32_t lo_, uint32_t hi_, uint32_t (&dst)[kAhead]) {
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const uint32_t i = lo_ + (uint32_t)k * kThreads + threadIdx.x;
      dst[k] = i < hi_ ? rems[i] : kFree;   // (plain loads: nontemporal ones 305 -> 322 us)
    }
  };
  auto load_bounds = [&](uint32_t it0) {
    if (threadIdx.x <= (unsigned)kBatch) {
      const uint64_t rr = (uint64_t)blockIdx.x + (uint64_t)(it0 + threadIdx.x) * gridDim.x;
      uint32_t a = 0, b = 0;
      if (rr < n_regions) {
        a = starts[(uint32_t)rr];
        b = starts[(uint32_t)rr + 1];
      }
      s_lo[threadIdx.x] = a;
      s_hi[threadIdx.x] = b;
    }
  };
  // an element: free marker on piece 65535 = no key; the two reserved remainders go the deferred way; else slot + first attempt
  auto usable = [&](uint32_t r, uint32_t rem) -> bool {
    if (rem < kForeign) return true;
    if (!(r == kPieces - 1 && rem == kFree)) defer(unhash48(((uint64_t)r << 32) | rem));
    return false;
  };
  uint32_t it = 0, r = blockIdx.x;
  load_bounds(0);
  __syncthreads();
  uint32_t lo = s_lo[0], hi = s_hi[0];
  uint32_t cur[kAhead];
  fetch(lo, hi, cur);
  while (r < n_regions) {   // workgroup-uniform trip count
    const uint32_t j = it % kBatch;
    const uint32_t rn = r + gridDim.x;
    const uint32_t lon = s_lo[j + 1], hin = s_hi[j + 1];
    uint32_t nxt[kAhead];
    if (lo != hi) {
      ulonglong2* g = reinterpret_cast<ulonglong2*>(table + ((uint64_t)r << REGION_LOG2));
      ulonglong2 orig[PRISTINE ? 1 : kPairs];
      r3d_vox::lds_barrier();
      if (threadIdx.x == 0) changed = 0;
      if (PRISTINE) {
#pragma unroll
        for (int k = 0; k < kSlots / 4 / kThreads; ++k) reinterpret_cast<uint4*>(region)[k * kThreads + threadIdx.x] = uint4{kFree, kFree, kFree, kFree};
      } else {
#pragma unroll
        for (int k = 0; k < kPairs; ++k) orig[k] = g[k * kThreads + threadIdx.x];
#pragma unroll
        for (int k = 0; k < kPairs; ++k) {
          uint32_t v[2];
          const uint64_t key[2] = {orig[k].x, orig[k].y};
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint64_t h = hash48(key[q]);
            v[q] = key[q] == kEmpty ? kFree : (((uint32_t)(h >> 32) == r && (uint32_t)h < kForeign) ? (uint32_t)h : kForeign);
          }
          reinterpret_cast<uint2*>(region)[k * kThreads + threadIdx.x] = uint2{v[0], v[1]};
        }
      }
      r3d_vox::lds_barrier();
      fetch(lon, hin, nxt);
      mine_changed = false;
      {
        uint32_t slot[kAhead], old[kAhead];
        bool has[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
          const uint32_t i = lo + (uint32_t)k * kThreads + threadIdx.x;
          has[k] = i < hi && usable(r, cur[k]);
          slot[k] = cur[k] >> (32 - REGION_LOG2);
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k) old[k] = has[k] ? atomicCAS(&region[slot[k]], kFree, cur[k]) : 0u;
#pragma unroll
        for (int k = 0; k < kAhead; ++k)
          if (has[k]) probe_on(r, cur[k], slot[k], old[k]);
      }
      for (uint32_t i = lo + kAhead * kThreads + threadIdx.x; i < hi; i += kThreads) {   // a longer run than usual
        const uint32_t rem = rems[i];
        if (usable(r, rem)) {
          const uint32_t s = rem >> (32 - REGION_LOG2);
          probe_on(r, rem, s, atomicCAS(&region[s], kFree, rem));
        }
      }
      if (mine_changed) changed = 1;
      r3d_vox::lds_barrier();
      if (changed) {
#pragma unroll
        for (int k = 0; k < kPairs; ++k) {
          const uint2 v = reinterpret_cast<const uint2*>(region)[k * kThreads + threadIdx.x];
          ulonglong2 out;
          // (rebuilding the keys with the piece's share of the product taken out of the loop and only the partial products that
          // reach the low 48 bits -- two 32-bit multiplies and a 24-bit one -- changed nothing: same-process A/B)
          out.x = v.x == kFree ? kEmpty : unhash48(((uint64_t)r << 32) | v.x);
          out.y = v.y == kFree ? kEmpty : unhash48(((uint64_t)r << 32) | v.y);
          if (!PRISTINE) {
            if (v.x == kForeign) out.x = orig[k].x;
            if (v.y == kForeign) out.y = orig[k].y;
          }
          // (Every pair is written, free or not.  A fresh table already says "empty" everywhere, but leaving out the free 16-byte
          // pairs -- 4 in 10 at load 0.36 -- or only whole free 32-byte sectors -- 1 in 6 -- took 234 MB off the write traffic and
          // ADDED 200 / 150 us: lines written in part are slow at the memory.  Same-process A/B, round 5.)
          typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
          // (nontemporal: 325 -> 296-310 us against the plain store, same process)
          __builtin_nontemporal_store(u64x2{out.x, out.y}, reinterpret_cast<u64x2*>(g) + k * kThreads + threadIdx.x);
        }
      }
    } else {
      fetch(lon, hin, nxt);
    }
#pragma unroll
    for (int k = 0; k < kAhead; ++k) cur[k] = nxt[k];
    r = rn;
    lo = lon;
    hi = hin;
    ++it;
    if (it % kBatch == 0) {
      __syncthreads();
      load_bounds(it);
      __syncthreads();
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off, 64);
    n_over += __shfl_down(n_over, off, 64);
  }
  r3d_vox::lds_barrier();
  if (lane == 0) {
    if (n_new) atomicAdd(&wg_count[0], n_new);
    if (n_over) atomicAdd(&wg_count[1], n_over);
  }
  r3d_vox::lds_barrier();
  if (threadIdx.x < 2) partials[2 * (uint64_t)blockIdx.x + threadIdx.x] = wg_count[threadIdx.x];
}

// The merge into a table of 2^27 slots with a piece per WAVE instead of per workgroup: no barrier anywhere in the loop (the LDS
// serves a wave's operations in the order it issued them), four independent pieces in flight per workgroup and twenty per CU, the
// next piece's bounds and remainders requested a piece ahead.  The workgroup form above spends a third of its time on the loop's
// skeleton -- three barriers per piece, each waiting for the slowest of four waves' probe chains: 303-330 -> 250-275 us, same process.
// (One loop for a group's four re-probe chains instead of four loops, four swaps in flight per step: 257 -> 300 us.  Not kept.)
// PRISTINE = false: the piece's slots come in from the table first; a slot that holds a key of ANOTHER piece (kForeign) is read
// again on the way out -- rare, so nothing is kept in registers for it.
template <int REGION_LOG2, bool PRISTINE>
__global__ __launch_bounds__(kThreads) void voxel_merge32w_kernel(const uint32_t* __restrict__ rems, const uint32_t* __restrict__ starts,
                                                                  uint32_t n_regions, uint64_t* __restrict__ table,
                                                                  uint64_t* __restrict__ spill, unsigned long long* __restrict__ spill_count,
                                                                  unsigned long long spill_cap, unsigned long long* __restrict__ partials) {
  constexpr int kSlots = 1 << REGION_LOG2;
  constexpr int kWavesPerWg = kThreads / 64;
  constexpr int kAhead = 12;                    // remainders per lane requested ahead: 768 per piece (mean 737 at 2.73 slots per point)
  constexpr int kPairs = kSlots / 2 / 64;       // slot pairs per lane
  constexpr uint32_t kFree = 0xffffffffu, kForeign = 0xfffffffeu;
  __shared__ __attribute__((aligned(16))) uint32_t region_all[kWavesPerWg][kSlots];   // 32 KB exactly: five workgroups per CU
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* region = region_all[wave];
  const uint32_t n_waves = gridDim.x * kWavesPerWg;
  unsigned n_new = 0, n_over = 0;
  auto defer = [&](uint64_t key) {
    const unsigned long long at = atomicAdd(spill_count, 1ull);
    if (at < spill_cap) spill[at] = key; else ++n_over;
  };
  auto probe_on = [&](uint32_t piece, uint32_t rem, uint32_t s, uint32_t old) {
    for (;;) {
      if (old == kFree) {
        ++n_new;
        return;
      }
      if (old == rem) return;
      if (++s >= (uint32_t)kSlots) break;
      old = atomicCAS(&region[s], kFree, rem);
    }
    defer(unhash48(((uint64_t)piece << 32) | rem));   // every slot from home to the piece's end is taken: the probe goes on later
  };
  auto usable = [&](uint32_t r, uint32_t rem) -> bool {
    if (rem < kForeign) return true;
    if (!(r == kPieces - 1 && rem == kFree)) defer(unhash48(((uint64_t)r << 32) | rem));
    return false;
  };
  auto fetch = [&](uint32_t lo_, uint32_t hi_, uint32_t (&dst)[kAhead]) {
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const uint32_t i = lo_ + (uint32_t)k * 64 + lane;
      dst[k] = i < hi_ ? rems[i] : kFree;
    }
  };
  uint32_t r = blockIdx.x * kWavesPerWg + wave;
  uint32_t lo = 0, hi = 0;
  if (r < n_regions) {
    lo = starts[r];
    hi = starts[r + 1];
  }
  uint32_t cur[kAhead];
  fetch(lo, hi, cur);
  while (r < n_regions) {   // (wave-uniform)
    const uint32_t rn = r + n_waves;
    uint32_t lon = 0, hin = 0;
    if (rn < n_regions) {
      lon = starts[rn];
      hin = starts[rn + 1];
    }
    uint32_t nxt[kAhead];
    if (lo != hi) {
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      u64x2* g = reinterpret_cast<u64x2*>(table + ((uint64_t)r << REGION_LOG2));
      const unsigned n_before = n_new;
      if (PRISTINE) {
#pragma unroll
        for (int k = 0; k < kSlots / 4 / 64; ++k) reinterpret_cast<uint4*>(region)[k * 64 + lane] = uint4{kFree, kFree, kFree, kFree};
      } else {
#pragma unroll 4
        for (int k = 0; k < kPairs; ++k) {
          const u64x2 key = g[k * 64 + lane];
          uint32_t v[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint64_t h = hash48(key[q]);
            v[q] = key[q] == kEmpty ? kFree : (((uint32_t)(h >> 32) == r && (uint32_t)h < kForeign) ? (uint32_t)h : kForeign);
          }
          reinterpret_cast<uint2*>(region)[k * 64 + lane] = uint2{v[0], v[1]};
        }
      }
      __builtin_amdgcn_wave_barrier();   // (the compiler keeps the order; the LDS keeps a wave's operations in order by itself)
      fetch(lon, hin, nxt);
#pragma unroll
      for (int g = 0; g < kAhead; g += 4) {
        uint32_t slot[4], old[4];
        bool has[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t i = lo + (uint32_t)(g + k) * 64 + lane;
          has[k] = i < hi && usable(r, cur[g + k]);
          slot[k] = cur[g + k] >> (32 - REGION_LOG2);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) old[k] = has[k] ? atomicCAS(&region[slot[k]], kFree, cur[g + k]) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (has[k]) probe_on(r, cur[g + k], slot[k], old[k]);
      }
      // a longer run than usual -- possibly MUCH longer (a point that occurs a hundred thousand times: pixels without depth): eight
      // loads at a time, and a look at the home slot before the swap (the same key again is then a broadcast read, not 64 swaps
      // of one word in a row)
      for (uint32_t i0 = lo + kAhead * 64 + lane; i0 < hi + lane; i0 += 8 * 64) {   // (wave-uniform trip count)
        uint32_t more[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) more[k] = i0 + k * 64 < hi ? rems[i0 + k * 64] : kFree;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (i0 + k * 64 < hi && usable(r, more[k])) {
            const uint32_t s = more[k] >> (32 - REGION_LOG2);
            if (region[s] != more[k]) probe_on(r, more[k], s, atomicCAS(&region[s], kFree, more[k]));
          }
        }
      }
      r3d_vox::lds_settle();
      __builtin_amdgcn_wave_barrier();
      if (PRISTINE || __any(n_new != n_before)) {   // (a table that is not fresh: only a piece that gained a key goes back)
#pragma unroll
        for (int k = 0; k < kPairs; ++k) {
          const uint2 v = reinterpret_cast<const uint2*>(region)[k * 64 + lane];
          u64x2 out;
          out.x = v.x == kFree ? kEmpty : unhash48(((uint64_t)r << 32) | v.x);
          out.y = v.y == kFree ? kEmpty : unhash48(((uint64_t)r << 32) | v.y);
          if (!PRISTINE && (v.x == kForeign || v.y == kForeign)) {
            const u64x2 was = g[k * 64 + lane];
            if (v.x == kForeign) out.x = was.x;
            if (v.y == kForeign) out.y = was.y;
          }
          __builtin_nontemporal_store(out, g + k * 64 + lane);
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      fetch(lon, hin, nxt);
    }
#pragma unroll
    for (int k = 0; k < kAhead; ++k) cur[k] = nxt[k];
    r = rn;
    lo = lon;
    hi = hin;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off, 64);
    n_over += __shfl_down(n_over, off, 64);
  }
  __syncthreads();   // the pieces are done with: two of their words take the workgroup's counts
  unsigned* wg_count = region_all[0];
  if (threadIdx.x < 2) wg_count[threadIdx.x] = 0;
  __syncthreads();
  if (lane == 0) {
    if (n_new) atomicAdd(&wg_count[0], n_new);
    if (n_over) atomicAdd(&wg_count[1], n_over);
  }
  __syncthreads();
  if (threadIdx.x < 2) partials[2 * (uint64_t)blockIdx.x + threadIdx.x] = wg_count[threadIdx.x];
}

// the deferred keys, by the ordinary CAS (their count is known on the device only: fixed grid, device-side bound).  The list may
// hold one key very many times (the points a first-pass segment had no room for: e.g. every pixel without depth of a frame is
// the same point): a probe LOOKS before it swaps, so that those end as reads of a cached line instead of queueing at one address.
__global__ __launch_bounds__(kThreads) void voxel_spill_kernel(const uint64_t* __restrict__ spill, const unsigned long long* __restrict__ spill_count,
                                                               unsigned long long spill_cap, uint64_t* __restrict__ table, int log2cap,
                                                               unsigned long long* __restrict__ counters,
                                                               const unsigned long long* __restrict__ partials, int n_partials,
                                                               const uint32_t* __restrict__ flags) {
  const uint64_t mask = ((uint64_t)1 << log2cap) - 1;
  unsigned long long n_new = 0, n_over = 0;
  unsigned long long n = *spill_count;
  if (n > spill_cap) n = spill_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // the points the first pass found no key for
    const unsigned long long ign = *reinterpret_cast<const unsigned long long*>(flags + 2);
    if (ign) atomicAdd(&counters[1], ign);
  }
  for (unsigned long long i = (unsigned long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * kThreads) {
    const uint64_t key = spill[i];
    const uint64_t prev = prev_lane_u64(key);   // (by every lane of the iteration: a lane switched off would hand its neighbour that neighbour's own key)
    if ((threadIdx.x & 63) > 0 && prev == key) continue;   // a full segment defers the same key wave after wave: the previous lane inserts it
    uint64_t slot = r3d_vox::home_slot(key, log2cap);
    int r = -1;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
      uint64_t old = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == kEmpty) old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[slot]), (unsigned long long)kEmpty, (unsigned long long)key);
      if (old == kEmpty) { r = 1; break; }
      if (old == key) { r = 0; break; }
      slot = (slot + 1) & mask;
    }
    n_new += r > 0 ? 1u : 0u;
    n_over += r < 0 ? 1u : 0u;
  }
  // ... and the merge launch's per-workgroup counts (pairs: new, no slot), a pair per thread over the whole grid (one block
  // walking 2048 pairs was sixteen dependent round trips: 10 of this kernel's 15 us)
  for (int k = blockIdx.x * kThreads + threadIdx.x; k < n_partials; k += gridDim.x * kThreads) {
    const ulonglong2 pr = reinterpret_cast<const ulonglong2*>(partials)[k];
    n_new += pr.x;
    n_over += pr.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off, 64);
    n_over += __shfl_down(n_over, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_new) atomicAdd(&counters[0], n_new);
    if (n_over) atomicAdd(&counters[2], n_over);
  }
}

// How alike are neighbouring points?  Each sampling workgroup takes 4 consecutive tiles (4096 points: a few image rows) and
// counts the distinct voxels among them in an LDS set; sums[0] += points that have a key, sums[1] += distinct keys.
constexpr int kSampleSlots = 8192;
__global__ __launch_bounds__(kThreads) void voxel_sample_kernel(const float* __restrict__ xyz, int64_t n, double factor, int64_t stride_tiles,
                                                                unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long set[kSampleSlots];
  for (int k = threadIdx.x; k < kSampleSlots; k += kThreads) set[k] = kEmpty;
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * stride_tiles * (kThreads * 4);
  unsigned valid = 0, distinct = 0;
  for (int r = 0; r < 16; ++r) {
    const int64_t i = first + (int64_t)r * kThreads + threadIdx.x;
    if (i >= n) break;   // (every thread still reaches the barriers of flush_counts below)
    const P3 p = reinterpret_cast<const P3*>(xyz)[i];
    uint64_t key;
    if (!r3d_vox::voxel_key(p.x, p.y, p.z, factor, &key)) continue;
    ++valid;
    uint32_t s = (((uint32_t)key * 0x9E3779B1u) ^ ((uint32_t)(key >> 32) * 0x85EBCA77u)) >> 19;   // 13 bits
    for (int probe = 0; probe < kSampleSlots; ++probe) {   // <= 4096 keys in 8192 slots: always ends
      const unsigned long long old = atomicCAS(&set[s], (unsigned long long)kEmpty, (unsigned long long)key);
      if (old == kEmpty) {
        ++distinct;
        break;
      }
      if (old == key) break;
      s = (s + 1) & (kSampleSlots - 1);
    }
  }
  __shared__ unsigned wg_counts[3];
  r3d_vox::flush_counts(valid, distinct, 0u, wg_counts, sums);   // sums[0] += valid, sums[1] += distinct: one add per workgroup
}

// Insert ready-made 48-bit Morton codes (another rank's occupied voxels: the union step of a sharded map).
__global__ __launch_bounds__(kThreads) void voxel_insert_codes_kernel(const uint64_t* __restrict__ codes, int64_t n,
                                                                      uint64_t* __restrict__ table, int log2cap,
                                                                      unsigned long long* __restrict__ counters) {
  const uint64_t mask = ((uint64_t)1 << log2cap) - 1;
  const int lane = threadIdx.x & 63;
  unsigned n_new = 0, n_ignored = 0, n_over = 0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n + lane; i += (int64_t)gridDim.x * kThreads) {
    // (the loop bound keeps whole waves together for the shuffle below)
    uint64_t code = i < n ? codes[i] : kEmpty;
    bool live = i < n;
    if (live && (code >> 48) != 0) {  // not a depth-16 octree key
      ++n_ignored;
      live = false;
      code = kEmpty;
    }
    const uint64_t prev = __shfl_up(code, 1, 64);
    if (live && lane > 0 && prev == code) live = false;  // sorted inputs repeat a code in neighbouring lanes
    if (live) {
      code = r3d_vox::key_of_morton(code);  // the table holds packed keys (r3d_voxel_dev.h)
      uint64_t slot = r3d_vox::home_slot(code, log2cap);
      bool done = false;
      for (uint64_t probe = 0; probe <= mask && !done; ++probe) {
        const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[slot]), (unsigned long long)kEmpty,
                                       (unsigned long long)code);
        if (old == kEmpty) {
          ++n_new;
          done = true;
        } else if (old == code) {
          done = true;
        } else {
          slot = (slot + 1) & mask;
        }
      }
      if (!done) ++n_over;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off, 64);
    n_ignored += __shfl_down(n_ignored, off, 64);
    n_over += __shfl_down(n_over, off, 64);
  }
  if (lane == 0) {
    if (n_new) atomicAdd(&counters[0], (unsigned long long)n_new);
    if (n_ignored) atomicAdd(&counters[1], (unsigned long long)n_ignored);
    if (n_over) atomicAdd(&counters[2], (unsigned long long)n_over);
  }
}

// table (packed keys) -> dense list of Morton codes (order irrelevant: the radix sort follows).  One cursor bump per WORKGROUP-STEP of
// kCompactSlots table words held in registers (round 2 bumped the one global cursor once per wave per 64 words: ~2 M
// same-address returning atomics for a 1 GB table = 25 ms = 0.8 % of HBM; a same-address returning atomic completes at
// ~0.09 G/s, so the count per bump decides everything).  Steps that hold no code skip the atomic.
constexpr int kCompactPerThread = 32;                            // table words per thread per step: 16 x 16-byte loads
constexpr int kCompactSlots = kThreads * kCompactPerThread;      // 8192 words = 64 KB per workgroup-step

__global__ __launch_bounds__(kThreads) void voxel_compact_kernel(const uint64_t* __restrict__ table, uint64_t capacity,
                                                                 uint64_t* __restrict__ out,
                                                                 unsigned long long* __restrict__ counters) {
  __shared__ unsigned wave_total[kThreads / 64];
  __shared__ unsigned long long step_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t n_steps = (capacity + kCompactSlots - 1) / kCompactSlots;
  for (uint64_t step = blockIdx.x; step < n_steps; step += gridDim.x) {   // workgroup-uniform trip count
    // thread t holds words (2t, 2t+1) + k * 512 of the step: every wave instruction reads 1 KB contiguous
    const uint64_t lo = step * kCompactSlots + 2 * (uint64_t)threadIdx.x;
    uint64_t v[kCompactPerThread];
#pragma unroll
    for (int k = 0; k < kCompactPerThread / 2; ++k) {
      const uint64_t i = lo + (uint64_t)k * (2 * kThreads);
      if (i + 1 < capacity) {   // capacity is a power of two >= 1024: pairs never straddle the end
        const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(table + i);
        v[2 * k] = w.x;
        v[2 * k + 1] = w.y;
      } else {
        v[2 * k] = kEmpty;
        v[2 * k + 1] = kEmpty;
      }
    }
    // per word slot k the wave's hits leave as ONE contiguous run (ballot-ranked): the counts are wave-uniform scalars
    unsigned wave_cnt = 0;
#pragma unroll
    for (int k = 0; k < kCompactPerThread; ++k) wave_cnt += (unsigned)__popcll(__ballot(v[k] != kEmpty));
    if (lane == 0) wave_total[wave] = wave_cnt;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
      const unsigned t = wave_total[w];
      if (w < wave) before += t;
      total += t;
    }
    if (threadIdx.x == 0 && total) step_base = atomicAdd(&counters[3], (unsigned long long)total);
    __syncthreads();
    if (total) {
      unsigned long long at = step_base + before;
      const unsigned long long below = (1ull << lane) - 1;
#pragma unroll
      for (int k = 0; k < kCompactPerThread; ++k) {
        const bool hit = v[k] != kEmpty;
        const unsigned long long ballot = __ballot(hit);
        if (hit) out[at + __popcll(ballot & below)] = r3d_vox::morton_of_key(v[k]);  // packed key -> Morton code on the way out
        at += __popcll(ballot);
      }
    }
    // (the next step's first barrier separates this step's reads of wave_total / step_base from their next writes)
  }
}

// ---- octree serialisation (host) ----
constexpr int kDepth = 16;

struct BtWriter {
  const uint64_t* codes;
  std::string body;
  int64_t n_nodes = 0;

  static bool full(int64_t count, int child_depth) {
    const int levels = kDepth - child_depth;  // 8^levels leaves below a node at child_depth
    return levels <= 20 && count == ((int64_t)1 << (3 * levels));
  }

  // one node record: child boundaries, the two mask bytes, and which children are inner nodes
  int record(int64_t lo, int64_t hi, int depth, int64_t inner[8][2]) {
    ++n_nodes;
    const int shift = 3 * (kDepth - 1 - depth);
    int64_t bounds[9];
    bounds[0] = lo;
    for (int c = 1; c <= 8; ++c) {
      // first index whose child id at this level is >= c
      const uint64_t* first = std::lower_bound(codes + bounds[c - 1], codes + hi, (uint64_t)c,
                                               [shift](uint64_t v, uint64_t cc) { return ((v >> shift) & 7u) < cc; });
      bounds[c] = first - codes;
    }
    unsigned char b[2] = {0, 0};
    int n_inner = 0;
    for (int c = 0; c < 8; ++c) {
      const int64_t clo = bounds[c], chi = bounds[c + 1];
      if (chi == clo) continue;
      if (depth + 1 == kDepth || full(chi - clo, depth + 1)) {
        b[c / 4] |= (unsigned char)(2u << (2 * (c % 4)));  // occupied leaf (possibly a pruned subtree)
        ++n_nodes;
      } else {
        b[c / 4] |= (unsigned char)(3u << (2 * (c % 4)));
        inner[n_inner][0] = clo;
        inner[n_inner][1] = chi;
        ++n_inner;
      }
    }
    body.push_back((char)b[0]);
    body.push_back((char)b[1]);
    return n_inner;
  }

  void node(int64_t lo, int64_t hi, int depth) {
    int64_t inner[8][2];
    const int n_inner = record(lo, hi, depth, inner);
    for (int k = 0; k < n_inner; ++k) node(inner[k][0], inner[k][1], depth + 1);
  }
};

// Depth-first order means a subtree's bytes are one contiguous run: the subtrees hanging below `split_depth`
// are serialised by worker threads and spliced in order.
void build_parallel(const uint64_t* codes, int64_t n, std::string* body, int64_t* n_nodes) {
  constexpr int kSplitDepth = 3;  // up to 512 independent subtrees
  struct Piece {
    bool is_task;
    int64_t lo, hi;
    std::string bytes;
    int64_t nodes = 0;
  };
  std::vector<Piece> pieces;
  BtWriter top;
  top.codes = codes;
  // walk the top levels sequentially; every inner child at kSplitDepth becomes a task
  struct Frame {
    int64_t lo, hi;
    int depth;
  };
  std::vector<Frame> stack;
  stack.push_back({0, n, 0});
  while (!stack.empty()) {
    const Frame f = stack.back();
    stack.pop_back();
    if (f.depth >= kSplitDepth || f.hi - f.lo < 4096) {
      if (!top.body.empty()) {
        pieces.push_back({false, 0, 0, std::move(top.body), 0});
        top.body.clear();
      }
      pieces.push_back({true, f.lo, f.hi, std::string(), 0});
      pieces.back().nodes = f.depth;  // stash the depth until the worker overwrites it
      continue;
    }
    int64_t inner[8][2];
    const int n_inner = top.record(f.lo, f.hi, f.depth, inner);
    for (int k = n_inner - 1; k >= 0; --k) stack.push_back({inner[k][0], inner[k][1], f.depth + 1});  // DFS order
  }
  if (!top.body.empty()) pieces.push_back({false, 0, 0, std::move(top.body), 0});
  unsigned hw = r3d_host::cpu_budget();
  const unsigned n_workers = std::max(1u, std::min(hw == 0 ? 1u : hw, 32u));
  std::vector<std::thread> pool;
  std::atomic<size_t> next{0};
  const r3d_host::Spread spread;
  for (unsigned w = 0; w < n_workers; ++w)
    pool.emplace_back([&, w]() {
      spread.place(w);
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= pieces.size()) return;
        Piece& p = pieces[i];
        if (!p.is_task) continue;
        BtWriter sub;
        sub.codes = codes;
        sub.node(p.lo, p.hi, (int)p.nodes);
        p.bytes = std::move(sub.body);
        p.nodes = sub.n_nodes;
      }
    });
  for (auto& t : pool) t.join();
  size_t total = 0;
  int64_t nodes = top.n_nodes;
  for (const auto& p : pieces) {
    total += p.bytes.size();
    if (p.is_task) nodes += p.nodes;
  }
  body->clear();
  body->reserve(total);
  for (const auto& p : pieces) body->append(p.bytes);
  *n_nodes = nodes;
}

int build_bt(const uint64_t* codes, int64_t n, double res, std::string* out, int64_t* n_nodes) {
  for (int64_t i = 1; i < n; ++i)
    if (codes[i] <= codes[i - 1]) {
      r3d_set_error("octree export needs strictly ascending Morton codes (violated at index %lld)", (long long)i);
      return R3D_ERR_INVALID;
    }
  if (n > 0 && (codes[n - 1] >> 48) != 0) {
    r3d_set_error("Morton code above 48 bits");
    return R3D_ERR_INVALID;
  }
  struct {
    std::string body;
    int64_t n_nodes = 0;
  } w;
  if (n > 0) {
    if (BtWriter::full(n, 0)) {
      w.n_nodes = 1;
      w.body.assign(2, '\0');
    } else {
      build_parallel(codes, n, &w.body, &w.n_nodes);
    }
  }
  char head[256];
  snprintf(head, sizeof(head),
           "# Octomap OcTree binary file\n# (feel free to add / change comments, but leave the first line as it is!)\n#\n"
           "id OcTree\nsize %lld\nres %g\ndata\n",
           (long long)w.n_nodes, res);
  *out = std::string(head) + w.body;
  *n_nodes = w.n_nodes;
  return R3D_OK;
}

}  // namespace

int r3d_voxelset_device_view(r3d_voxelset* vs, r3d_ctx** ctx, double* factor, uint64_t** d_table, int* log2cap,
                             unsigned long long** d_counters) {
  R3D_REQUIRE(vs != nullptr, "voxel set is NULL");
  *ctx = vs->ctx;
  *factor = vs->factor;
  *d_table = vs->d_table;
  *log2cap = vs->log2cap;
  *d_counters = vs->d_counters;
  vs->pristine = false;   // whoever asks for the table is about to write it
  return R3D_OK;
}


extern "C" {

int r3d_voxelset_create(r3d_ctx* ctx, double resolution, int64_t capacity, r3d_voxelset** vs_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(vs_out != nullptr, "vs_out is NULL");
  *vs_out = nullptr;
  R3D_REQUIRE(resolution > 0.0 && std::isfinite(resolution), "resolution must be positive");
  R3D_REQUIRE(capacity >= 0, "capacity must be >= 0");
  r3d_voxelset* vs = new (std::nothrow) r3d_voxelset();
  if (!vs) {
    r3d_set_error("host allocation failed");
    return R3D_ERR_NOMEM;
  }
  vs->ctx = ctx;
  vs->device = ctx->device;
  vs->res = resolution;
  vs->factor = 1.0 / resolution;  // OcTreeBaseImpl::resolution_factor
  vs->log2cap = 10;
  while (((int64_t)1 << vs->log2cap) < capacity && vs->log2cap < 40) ++vs->log2cap;
  vs->capacity = (uint64_t)1 << vs->log2cap;
  hipError_t e = hipMalloc((void**)&vs->d_table, vs->capacity * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMalloc((void**)&vs->d_counters, 4 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemsetAsync(vs->d_table, 0xff, vs->capacity * sizeof(uint64_t), ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(vs->d_counters, 0, 4 * sizeof(unsigned long long), ctx->stream);
  if (e != hipSuccess) {
    r3d_voxelset_destroy(vs);
    return r3d_fail_hip(e, "voxel set allocation", __FILE__, __LINE__);
  }
  *vs_out = vs;
  return R3D_OK;
}

int r3d_voxelset_destroy(r3d_voxelset* vs) {
  if (!vs) return R3D_OK;
  (void)hipSetDevice(vs->device);
  (void)hipDeviceSynchronize();
  if (vs->d_table) (void)hipFree(vs->d_table);
  if (vs->d_counters) (void)hipFree(vs->d_counters);
  delete vs;
  return R3D_OK;
}

int r3d_voxelset_clear(r3d_voxelset* vs) {
  R3D_REQUIRE(vs != nullptr, "voxel set is NULL");
  int rc = r3d_ctx_enter(vs->ctx);
  if (rc) return rc;
  R3D_HIP(hipMemsetAsync(vs->d_table, 0xff, vs->capacity * sizeof(uint64_t), vs->ctx->stream));
  R3D_HIP(hipMemsetAsync(vs->d_counters, 0, 4 * sizeof(unsigned long long), vs->ctx->stream));
  vs->pristine = true;
  return R3D_OK;
}

int r3d_voxelset_insert(r3d_voxelset* vs, const float* d_xyz, int64_t n_points) {
  R3D_REQUIRE(vs != nullptr, "voxel set is NULL");
  int rc = r3d_ctx_enter(vs->ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_points >= 0, "n_points must be >= 0");
  if (n_points == 0) return R3D_OK;
  R3D_REQUIRE(d_xyz != nullptr, "NULL device pointer");
  // big inserts: a sample of the cloud decides between the two paths ("voxel_path": 1 / 2 force one)
  int path = 1;
  if (vs->ctx->voxel_path == 2 && r3d_voxelset_sort_feasible(vs, n_points, true)) {
    path = 2;
  } else if (vs->ctx->voxel_path == 0 && r3d_voxelset_sort_feasible(vs, n_points, false)) {
    bool sort = false;
    if ((rc = r3d_voxelset_sample(vs, d_xyz, n_points, n_points, 4.5, &sort))) return rc;
    path = sort ? 2 : 1;
  }
  return r3d_voxelset_insert_path(vs, d_xyz, n_points, path);
}

}  // extern "C"

// Is the sort-merge path possible for this set (region sizes that fit LDS, region ids that fit above the key) and, unless
// `forced`, worth considering for this many points (its fixed costs -- a dozen launches, the table streamed once -- want a
// big insert and a table that is not vastly larger than it)?
bool r3d_voxelset_sort_feasible(const r3d_voxelset* vs, int64_t n_points, bool forced) {
  if (vs->log2cap < kPieceBits || vs->log2cap > kPieceBits + kRegionMaxLog2) return false;
  if (forced) return n_points >= 1;
  return n_points >= ((int64_t)1 << 22) && vs->capacity <= (uint64_t)n_points * 16;
}

// *sort_out = by the sample, the sort-merge insert of `n_insert` points into this set's table will be the faster one.  The sample
// gives r = distinct voxels per point among neighbours (256 groups of 4096 consecutive points); the two paths' costs on this chip,
// from tools/voxel_path_crossover.py and the stage profiles (round 5):
//   sort-merge   7.5 ps per point (both passes) + 2 ps per table slot (the merge streams the whole table) + 40 us of launches;
//   LDS set + CAS   (4.5 + 62 r) ps per point -- 66 ps for a voxel per point (3.26 ms for C2), 6.7 ps at 28 points per voxel;
//   `cas_base_ps`: the 4.5 (r3d_fuse_frames_voxel's one-launch kernel does not read the cloud back: 1.5).
// Rounds 2-5 asked for >= 1 distinct voxel per 2 points, which left clouds of 2.7 / 5 / 10 points per voxel with the CAS path
// at 1.55 / 0.97 / 0.62 ms where the sort-merge path takes 0.52 / 0.50 / 0.49.  Synchronises the stream (16 bytes come back).
int r3d_voxelset_sample(r3d_voxelset* vs, const float* d_xyz, int64_t n_points, int64_t n_insert, double cas_base_ps, bool* sort_out) {
  *sort_out = false;
  r3d_ctx* ctx = vs->ctx;
  const int64_t n_tiles = (n_points + kThreads * 4 - 1) / (kThreads * 4);
  const int64_t samples = std::max<int64_t>(1, std::min<int64_t>(256, n_tiles / 4));
  void* ws = nullptr;
  int rc = r3d_scratch(ctx, 5, 64, &ws);
  if (rc) return rc;
  unsigned long long* d_sums = static_cast<unsigned long long*>(ws);
  R3D_HIP(hipMemsetAsync(d_sums, 0, 16, ctx->stream));
  hipLaunchKernelGGL(voxel_sample_kernel, dim3((unsigned)samples), dim3(kThreads), 0, ctx->stream, d_xyz, n_points, vs->factor,
                     n_tiles / samples, d_sums);
  R3D_HIP(hipGetLastError());
  unsigned long long h[2] = {0, 0};
  R3D_HIP(hipMemcpyAsync(h, d_sums, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  if (h[0] > 0) {
    const double r = (double)h[1] / (double)h[0];
    const double us_sort = (double)n_insert * 7.5e-6 + (double)vs->capacity * 2.0e-6 + 40.0;
    const double us_cas = (double)n_insert * (cas_base_ps + 62.0 * r) * 1e-6;
    *sort_out = us_sort < us_cas;
  }
  return R3D_OK;
}

static int insert_sorted(r3d_voxelset* vs, const float* d_xyz, int64_t n_points) {
  r3d_ctx* ctx = vs->ctx;
  const int region_log2 = std::max(kRegionMinLog2, vs->log2cap - kPieceBits);   // slots per LDS region
  const int sub_log2 = kPieceBits - (vs->log2cap - region_log2);                // pieces per region (log2)
  const uint32_t n_regions = (uint32_t)1 << (vs->log2cap - region_log2);
  const int64_t chunk = (int64_t)1 << 27;   // points per round: 0.5 GB of sorted remainders + 1.5 GB of segments
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  int rc;
  for (int64_t off = 0; off < n_points; off += chunk) {
    const int64_t m = std::min(chunk, n_points - off);
    const float* src = d_xyz + off * 3;
    const uint64_t spill_cap = (uint64_t)m;   // every key may be deferred (a nearly full table): the list can take them all
    const SegPlan plan = seg_plan(m);
    const int64_t n_tiles1 = (m + kSortTile - 1) / kSortTile;   // the first pass's tiles
    const int n_tiles = plan.n_tiles2, stride = r3d_sort_stride(n_tiles);   // the second pass's tiles
    const size_t seg_elems = (size_t)kSegments * plan.cap;
    void *a_v = nullptr, *b_v = nullptr, *ws = nullptr;
    if ((rc = r3d_scratch(ctx, 1, up((size_t)m * 4), &a_v))) return rc;   // the remainders in piece order
    if ((rc = r3d_scratch(ctx, 2, up(seg_elems * 4) + up(seg_elems), &b_v))) return rc;   // the first pass's segments: rem | hi
    uint32_t* rem_a = static_cast<uint32_t*>(a_v);
    uint32_t* rem_b = static_cast<uint32_t*>(b_v);
    uint8_t* hi_b = reinterpret_cast<uint8_t*>(static_cast<char*>(b_v) + up(seg_elems * 4));
    const unsigned merge_grid = (unsigned)ctx->num_cus * 8;   // (1536 .. 4096 workgroups measured within 3 % of each other)
    const size_t partial_bytes = up((size_t)merge_grid * 2 * 2 * sizeof(unsigned long long));   // (the wave form's grid is twice merge_grid)
    const size_t starts_bytes = up(((size_t)kPieces + 2) * sizeof(uint32_t));
    const size_t hist_bytes = up((size_t)256 * stride * sizeof(uint32_t));
    const size_t count_bytes = up((size_t)kSegments * kCursorStride * sizeof(uint32_t));   // the segments' cursors, a line each
    if ((rc = r3d_scratch(ctx, 5, partial_bytes + starts_bytes + hist_bytes + 1024 + 256 + count_bytes + spill_cap * 8, &ws))) return rc;
    char* w = static_cast<char*>(ws);
    unsigned long long* d_partials = reinterpret_cast<unsigned long long*>(w);
    uint32_t* d_starts = reinterpret_cast<uint32_t*>(w + partial_bytes);
    uint32_t* hist_hi = reinterpret_cast<uint32_t*>(w + partial_bytes + starts_bytes);
    uint32_t* totals_hi = reinterpret_cast<uint32_t*>(w + partial_bytes + starts_bytes + hist_bytes);
    char* zeroed = w + partial_bytes + starts_bytes + hist_bytes + 1024;   // one memset: the deferred keys' count, the flags, the cursors
    unsigned long long* d_spill_count = reinterpret_cast<unsigned long long*>(zeroed);
    uint32_t* d_flags = reinterpret_cast<uint32_t*>(zeroed + 64);   // [2..3] points without a key
    uint32_t* d_cursors = reinterpret_cast<uint32_t*>(zeroed + 256);
    uint64_t* d_spill = reinterpret_cast<uint64_t*>(zeroed + 256 + count_bytes);
    R3D_HIP(hipMemsetAsync(zeroed, 0, 256 + count_bytes, ctx->stream));
    // |x| < safe_abs  =>  |factor x| < 32767: every key in range whatever the rounding of the fp64 product (a bound strictly
    // inside the map's edge 32768 / factor, rounded towards zero and shrunk by 2^-20 on top)
    const float safe_abs = nextafterf((float)((32767.0 / vs->factor) * (1.0 - 1.0 / 1048576.0)), 0.0f);
    const unsigned bin_grid = (unsigned)((n_tiles1 + 7) / 8 * 8);   // a tile each (a multiple of 8: see the kernel)
    hipLaunchKernelGGL(voxel_bin_kernel, dim3(bin_grid), dim3(kBinThreads), 0, ctx->stream, src, m, vs->factor, safe_abs, (int)n_tiles1,
                       plan.cap, rem_b, hi_b, d_cursors, d_spill, d_spill_count, (unsigned long long)spill_cap, d_flags);
    hipLaunchKernelGGL(segment_histogram_kernel, dim3((unsigned)((n_tiles + 7) / 8)), dim3(kThreads), 0, ctx->stream, (const uint8_t*)hi_b,
                       (const uint32_t*)d_cursors, plan.cap, plan.chunks, n_tiles, hist_hi, stride);
    r3d_sort_launch_scan(ctx, hist_hi, n_tiles, stride, totals_hi);
    hipLaunchKernelGGL(segment_scatter_kernel, dim3((unsigned)n_tiles), dim3(kBinThreads), 0, ctx->stream, (const uint32_t*)rem_b,
                       (const uint8_t*)hi_b, (const uint32_t*)d_cursors, plan.cap, plan.chunks, (const uint32_t*)hist_hi, stride,
                       (const uint32_t*)totals_hi, rem_a, d_starts);
    const int pristine = vs->pristine ? 1 : 0;
    vs->pristine = false;
    const unsigned merge_blocks = std::min<uint32_t>(n_regions, merge_grid);   // persistent workgroups: the loop inside is a pipeline
#define R3D_LAUNCH_MERGE(L2, SUB)                                                                                                       \
  hipLaunchKernelGGL((voxel_merge_kernel<L2, SUB>), dim3(merge_blocks), dim3(kThreads), 0, ctx->stream, (const uint32_t*)rem_a,         \
                     (const uint32_t*)d_starts, n_regions, sub_log2, vs->d_table, vs->log2cap, d_spill, d_spill_count,                  \
                     (unsigned long long)spill_cap, pristine, d_partials)
#define R3D_LAUNCH_MERGE32(L2, PR)                                                                                                      \
  hipLaunchKernelGGL((voxel_merge32_kernel<L2, PR>), dim3(merge_blocks), dim3(kThreads), 0, ctx->stream, (const uint32_t*)rem_a,        \
                     (const uint32_t*)d_starts, n_regions, vs->d_table, d_spill, d_spill_count, (unsigned long long)spill_cap, d_partials)
    const bool narrow = sub_log2 == 0;   // regions are pieces: 32-bit slots in LDS (same-process A/B against the 64-bit form: 385-397 -> 359-362 us)
    unsigned wave_grid = 0;   // tables of 2^27 slots: a piece per wave (voxel_merge32w_kernel)
    if (narrow && region_log2 == 11) {
      wave_grid = (unsigned)ctx->num_cus * 16;
      if (pristine)
        hipLaunchKernelGGL((voxel_merge32w_kernel<11, true>), dim3(wave_grid), dim3(kThreads), 0, ctx->stream, (const uint32_t*)rem_a,
                           (const uint32_t*)d_starts, n_regions, vs->d_table, d_spill, d_spill_count, (unsigned long long)spill_cap, d_partials);
      else
        hipLaunchKernelGGL((voxel_merge32w_kernel<11, false>), dim3(wave_grid), dim3(kThreads), 0, ctx->stream, (const uint32_t*)rem_a,
                           (const uint32_t*)d_starts, n_regions, vs->d_table, d_spill, d_spill_count, (unsigned long long)spill_cap, d_partials);
    }
    else if (narrow && region_log2 == 12) { if (pristine) R3D_LAUNCH_MERGE32(12, true); else R3D_LAUNCH_MERGE32(12, false); }
    else if (narrow) { if (pristine) R3D_LAUNCH_MERGE32(13, true); else R3D_LAUNCH_MERGE32(13, false); }
    else if (sub_log2 > 0) R3D_LAUNCH_MERGE(11, true);       // tables below 2^27 slots: several pieces per 2048-slot region
    else if (region_log2 == 11) R3D_LAUNCH_MERGE(11, false);
    else if (region_log2 == 12) R3D_LAUNCH_MERGE(12, false);
    else R3D_LAUNCH_MERGE(13, false);
#undef R3D_LAUNCH_MERGE
#undef R3D_LAUNCH_MERGE32
    hipLaunchKernelGGL(voxel_spill_kernel, dim3((unsigned)ctx->num_cus), dim3(kThreads), 0, ctx->stream, (const uint64_t*)d_spill,
                       (const unsigned long long*)d_spill_count, (unsigned long long)spill_cap, vs->d_table, vs->log2cap, vs->d_counters,
                       (const unsigned long long*)d_partials, (int)(wave_grid ? wave_grid : merge_blocks), (const uint32_t*)d_flags);
    R3D_HIP(hipGetLastError());
  }
  return R3D_OK;
}

// path 1: the LDS-set + CAS kernel; path 2: sort-merge (the caller has checked r3d_voxelset_sort_feasible)
int r3d_voxelset_insert_path(r3d_voxelset* vs, const float* d_xyz, int64_t n_points, int path) {
  if (n_points <= 0) return R3D_OK;
  vs->ctx->voxel_last_path = path;
  if (path == 2) return insert_sorted(vs, d_xyz, n_points);
  vs->pristine = false;
  const int64_t n_tiles = (n_points + kThreads * 4 - 1) / (kThreads * 4);
  int blocks = vs->ctx->num_cus * 8;
  if ((int64_t)blocks > n_tiles) blocks = (int)n_tiles;
  if (vs->ctx->voxel_dedupe == 3)
    hipLaunchKernelGGL((voxel_insert_kernel<true, true>), dim3(blocks), dim3(kThreads), 0, vs->ctx->stream, d_xyz, n_points,
                       vs->factor, vs->d_table, vs->log2cap, vs->d_counters);
  else if (vs->ctx->voxel_dedupe != 1)  // 0 auto / 2 on: LDS dedupe; 1: off
    hipLaunchKernelGGL(voxel_insert_kernel<true>, dim3(blocks), dim3(kThreads), 0, vs->ctx->stream, d_xyz, n_points,
                       vs->factor, vs->d_table, vs->log2cap, vs->d_counters);
  else
    hipLaunchKernelGGL(voxel_insert_kernel<false>, dim3(blocks), dim3(kThreads), 0, vs->ctx->stream, d_xyz, n_points,
                       vs->factor, vs->d_table, vs->log2cap, vs->d_counters);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

extern "C" {

int r3d_voxelset_insert_host(r3d_voxelset* vs, const float* h_xyz, int64_t n_points) {
  R3D_REQUIRE(vs != nullptr, "voxel set is NULL");
  int rc = r3d_ctx_enter(vs->ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_points >= 0, "n_points must be >= 0");
  if (n_points == 0) return R3D_OK;
  R3D_REQUIRE(h_xyz != nullptr, "NULL host pointer");
  void* d = nullptr;
  if ((rc = r3d_scratch(vs->ctx, 0, (size_t)n_points * 12, &d))) return rc;
  R3D_HIP(hipMemcpyAsync(d, h_xyz, (size_t)n_points * 12, hipMemcpyHostToDevice, vs->ctx->stream));
  if ((rc = r3d_voxelset_insert(vs, static_cast<const float*>(d), n_points))) return rc;
  R3D_HIP(hipStreamSynchronize(vs->ctx->stream));
  return R3D_OK;
}

int r3d_voxelset_stats(r3d_voxelset* vs, int64_t* n_voxels, int64_t* n_ignored, int64_t* n_overflow) {
  R3D_REQUIRE(vs != nullptr, "voxel set is NULL");
  int rc = r3d_ctx_enter(vs->ctx);
  if (rc) return rc;
  unsigned long long c[4];
  R3D_HIP(hipMemcpyAsync(c, vs->d_counters, sizeof(c), hipMemcpyDeviceToHost, vs->ctx->stream));
  R3D_HIP(hipStreamSynchronize(vs->ctx->stream));
  if (n_voxels) *n_voxels = (int64_t)c[0];
  if (n_ignored) *n_ignored = (int64_t)c[1];
  if (n_overflow) *n_overflow = (int64_t)c[2];
  return R3D_OK;
}

int r3d_voxelset_insert_codes(r3d_voxelset* vs, const uint64_t* d_codes, int64_t n_codes) {
  R3D_REQUIRE(vs != nullptr, "voxel set is NULL");
  int rc = r3d_ctx_enter(vs->ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_codes >= 0, "n_codes must be >= 0");
  if (n_codes == 0) return R3D_OK;
  R3D_REQUIRE(d_codes != nullptr, "NULL device pointer");
  vs->pristine = false;
  int64_t blocks = (n_codes + kThreads - 1) / kThreads;
  if (blocks > (int64_t)vs->ctx->num_cus * 16) blocks = (int64_t)vs->ctx->num_cus * 16;
  hipLaunchKernelGGL(voxel_insert_codes_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, vs->ctx->stream, d_codes, n_codes,
                     vs->d_table, vs->log2cap, vs->d_counters);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

// distinct codes of the set as a sorted list in HBM (scratch slot 1); *n_out = how many
static int codes_to_device_list(r3d_voxelset* vs, uint64_t** d_list_out, int64_t* n_out) {
  int64_t n = 0, ign = 0, over = 0;
  int rc = r3d_voxelset_stats(vs, &n, &ign, &over);
  if (rc) return rc;
  *n_out = n;
  *d_list_out = nullptr;
  if (over > 0) {
    r3d_set_error("voxel set overflowed (%lld points found no slot): create it with a larger capacity", (long long)over);
    return R3D_ERR_NOMEM;
  }
  if (n == 0) return R3D_OK;
  void *d_list = nullptr, *d_tmp = nullptr;
  if ((rc = r3d_scratch(vs->ctx, 1, (size_t)n * sizeof(uint64_t), &d_list))) return rc;
  if ((rc = r3d_scratch(vs->ctx, 2, (size_t)n * sizeof(uint64_t), &d_tmp))) return rc;
  R3D_HIP(hipMemsetAsync(vs->d_counters + 3, 0, sizeof(unsigned long long), vs->ctx->stream));
  int blocks = vs->ctx->num_cus * 8;
  const uint64_t need = (vs->capacity + kCompactSlots - 1) / kCompactSlots;
  if ((uint64_t)blocks > need) blocks = (int)need;
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(blocks), dim3(kThreads), 0, vs->ctx->stream, vs->d_table, vs->capacity,
                     static_cast<uint64_t*>(d_list), vs->d_counters);
  R3D_HIP(hipGetLastError());
  if ((rc = r3d_radix_sort_u64(vs->ctx, static_cast<uint64_t*>(d_list), static_cast<uint64_t*>(d_tmp), n, 48))) return rc;
  *d_list_out = static_cast<uint64_t*>(d_list);
  return R3D_OK;
}

// Config 5 (frames sharded, ONE map): every rank voxelises its own shard of the world cloud into its own HBM hash set --
// 12 B/point never leave the GPU -- then the ranks exchange only their DISTINCT codes (8 B/voxel, unequal shards) and
// each folds the others' into its set.  Afterwards every rank holds the union.
int r3d_voxelset_union(r3d_voxelset* vs, r3d_comm* comm) {
  R3D_REQUIRE(vs != nullptr && comm != nullptr, "NULL argument");
  int rank = 0, world = 1;
  int rc = r3d_comm_info(comm, &rank, &world, nullptr);
  if (rc) return rc;
  // the exchange runs on the communicator's stream, the set's kernels on the set's: one context orders them
  R3D_REQUIRE(r3d_comm_context(comm) == vs->ctx, "create the communicator on the voxel set's context");
  uint64_t* d_mine = nullptr;
  int64_t n_mine = 0;
  const int rc_mine = codes_to_device_list(vs, &d_mine, &n_mine);
  if (world == 1) return rc_mine;
  // how many codes every rank brings: an all-gather of one int64 each.  A rank whose set cannot be listed (overflow)
  // still takes part and says -1, so that EVERY rank returns the error instead of waiting for it forever.
  if (rc_mine) n_mine = -1;
  void* d_cnt = nullptr;
  if ((rc = r3d_scratch(vs->ctx, 3, (size_t)(world + 1) * sizeof(int64_t), &d_cnt))) return rc;
  int64_t* d_counts = static_cast<int64_t*>(d_cnt);
  R3D_HIP(hipMemcpyAsync(d_counts + world, &n_mine, sizeof(int64_t), hipMemcpyHostToDevice, vs->ctx->stream));
  std::vector<int64_t> eight((size_t)world, (int64_t)sizeof(int64_t)), counts((size_t)world), bytes((size_t)world);
  if ((rc = r3d_comm_allgather(comm, d_counts + world, eight.data(), d_counts, R3D_GATHER_AUTO))) return rc;
  R3D_HIP(hipMemcpyAsync(counts.data(), d_counts, (size_t)world * sizeof(int64_t), hipMemcpyDeviceToHost, vs->ctx->stream));
  R3D_HIP(hipStreamSynchronize(vs->ctx->stream));
  if (rc_mine) return rc_mine;  // this rank's own error message stands
  int64_t total = 0;
  for (int r = 0; r < world; ++r) {
    if (counts[r] < 0) {
      r3d_set_error("rank %d could not list its voxel set (overflow): no union was formed", r);
      return R3D_ERR_NOMEM;
    }
    bytes[r] = counts[r] * (int64_t)sizeof(uint64_t);
    total += counts[r];
  }
  if (total == 0) return R3D_OK;
  void* d_all = nullptr;
  if ((rc = r3d_scratch(vs->ctx, 0, (size_t)total * sizeof(uint64_t), &d_all))) return rc;
  if ((rc = r3d_comm_allgather(comm, d_mine, bytes.data(), d_all, R3D_GATHER_AUTO))) return rc;
  return r3d_voxelset_insert_codes(vs, static_cast<const uint64_t*>(d_all), total);
}

int r3d_voxelset_codes(r3d_voxelset* vs, uint64_t* h_codes_sorted, int64_t cap, int64_t* n_out) {
  R3D_REQUIRE(vs != nullptr && n_out != nullptr, "NULL argument");
  if (!h_codes_sorted) {   // count only
    int64_t n = 0, ign = 0, over = 0;
    int rc = r3d_voxelset_stats(vs, &n, &ign, &over);
    if (rc) return rc;
    *n_out = n;
    if (over > 0) {
      r3d_set_error("voxel set overflowed (%lld points found no slot): create it with a larger capacity", (long long)over);
      return R3D_ERR_NOMEM;
    }
    return R3D_OK;
  }
  uint64_t* d_list = nullptr;
  int64_t n = 0;
  int rc = codes_to_device_list(vs, &d_list, &n);
  *n_out = n;
  if (rc) return rc;
  R3D_REQUIRE(cap >= n, "buffer holds %lld codes, set has %lld", (long long)cap, (long long)n);
  if (n == 0) return R3D_OK;
  return r3d_download_pageable(vs->ctx, h_codes_sorted, d_list, (size_t)n * sizeof(uint64_t));
}

int r3d_octree_format_bt(const uint64_t* h_codes_sorted, int64_t n_codes, double resolution, char* h_buf,
                         size_t buf_cap, size_t* n_bytes_out, int64_t* n_nodes_out) {
  if (n_codes < 0 || (n_codes > 0 && !h_codes_sorted) || !n_bytes_out || !(resolution > 0.0)) {
    r3d_set_error("r3d_octree_format_bt: bad argument");
    return R3D_ERR_INVALID;
  }
  std::string out;
  int64_t nodes = 0;
  int rc = build_bt(h_codes_sorted, n_codes, resolution, &out, &nodes);
  if (rc) return rc;
  *n_bytes_out = out.size();
  if (n_nodes_out) *n_nodes_out = nodes;
  if (!h_buf) return R3D_OK;
  if (buf_cap < out.size()) {
    r3d_set_error("r3d_octree_format_bt: buffer of %zu bytes is too small for %zu", buf_cap, out.size());
    return R3D_ERR_NOMEM;
  }
  memcpy(h_buf, out.data(), out.size());
  return R3D_OK;
}

int r3d_octree_write_bt(const char* path, const uint64_t* h_codes_sorted, int64_t n_codes, double resolution,
                        int64_t* n_nodes_out) {
  if (!path || n_codes < 0 || (n_codes > 0 && !h_codes_sorted) || !(resolution > 0.0)) {
    r3d_set_error("r3d_octree_write_bt: bad argument");
    return R3D_ERR_INVALID;
  }
  std::string out;
  int64_t nodes = 0;
  int rc = build_bt(h_codes_sorted, n_codes, resolution, &out, &nodes);
  if (rc) return rc;
  FILE* f = fopen(path, "wb");
  if (!f) {
    r3d_set_error("r3d_octree_write_bt: cannot open '%s'", path);
    return R3D_ERR_INVALID;
  }
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  if (fclose(f) != 0 || !ok) {
    r3d_set_error("r3d_octree_write_bt: short write to '%s'", path);
    return R3D_ERR_INVALID;
  }
  if (n_nodes_out) *n_nodes_out = nodes;
  return R3D_OK;
}

}  // extern "C"
