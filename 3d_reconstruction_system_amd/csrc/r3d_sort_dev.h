// Device-side pieces of the 8-bit-digit radix passes: how a workgroup ranks its tile of 4096 elements by one digit, stably and
// without a barrier per round, and where each bin of the tile then sits (r3d_sort.hip, 64-bit keys); the blockIdx -> tile map and
// the wave scan are shared with the sort-merge insert's passes (r3d_voxel.hip), whose order inside a bin is free -- they rank with
// one returning LDS add per element instead.  See digit_scatter_kernel (r3d_sort.hip) for the measurements behind the shape.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace r3d_sort {

constexpr int kThreads = 256;
constexpr int kRounds = 16;
constexpr int kTile = kThreads * kRounds;   // elements per workgroup
constexpr int kBins = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kPerWave = kTile / kWaves;

// LDS of the ranking.  Element e of a tile belongs to lane (e & 63) of wave (e >> 10), round ((e >> 6) & 15).
struct RankShared {
  uint32_t wave_cnt[kWaves][kBins];   // running counts while ranking, then the waves' offsets inside each bin
  uint32_t bin_start[kBins];          // where bin b starts in the tile's bin order
  uint32_t bin_count[kBins];
  uint32_t wave_sum[kWaves];
};

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// workgroup b of a grid of g -> item x * (g / 8) + min(x, g % 8) + b / 8 with x = b % 8: the workgroups of one XCD walk a
// contiguous eighth of the items, in order (neighbouring tiles write adjacent runs in every bin: their partial lines then meet
// in ONE L2 instead of reaching HBM as two read-modify-write halves)
__device__ __forceinline__ int xcd_contiguous(unsigned b, unsigned g) {
  const unsigned x = b & 7u, j = b >> 3, q = g >> 3, r = g & 7u;
  return (int)(x * q + (x < r ? x : r) + j);
}

// Every thread zeroes its column of the counters; a barrier must follow before rank_rounds.
__device__ __forceinline__ void rank_reset(RankShared& sh) {
#pragma unroll
  for (int w = 0; w < kWaves; ++w) sh.wave_cnt[w][threadIdx.x] = 0;
  sh.bin_count[threadIdx.x] = 0;
}

// place[r] = rank of the lane's round-r element among the elements OF ITS WAVE'S QUARTER that have the same digit (input
// order).  live_mask bit r: the element exists.  Four rounds at a time, in three sweeps, so that a round does not wait for the
// LDS round trips of the one before it: (1) who shares my digit -- 8 ballots, pure ALU; (2) the four counter bumps back to back
// (the lowest peer adds the peer count to the wave's own LDS counter: the returned value is the running count); (3) the four
// hand-overs from the lowest peer.  No other wave is involved: no barrier.
__device__ __forceinline__ void rank_rounds(const uint32_t (&digit)[kRounds], uint32_t live_mask, uint32_t (&place)[kRounds], RankShared& sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kGroup = 4;
#pragma unroll
  for (int g = 0; g < kRounds; g += kGroup) {
    uint32_t info[kGroup];   // rank among the peers | their number << 8 | the lowest peer's lane << 16 | live << 24
#pragma unroll
    for (int q = 0; q < kGroup; ++q) {
      const int r = g + q;
      const bool live = (live_mask >> r) & 1u;
      unsigned long long peers = __ballot(live);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long m = __ballot((digit[r] >> b) & 1);
        peers &= ((digit[r] >> b) & 1) ? m : ~m;
      }
      const uint32_t rank = __popcll(peers & ((1ull << lane) - 1));
      info[q] = rank | ((uint32_t)__popcll(peers) << 8) | ((peers ? (uint32_t)__ffsll((long long)peers) - 1u : 0u) << 16) | ((uint32_t)live << 24);
    }
#pragma unroll
    for (int q = 0; q < kGroup; ++q) {
      const int r = g + q;
      place[r] = 0;
      if ((info[q] >> 24) && (info[q] & 0xff) == 0)   // live, and the lowest of its peers
        place[r] = atomicAdd(&sh.wave_cnt[wave][digit[r] & 0xff], (info[q] >> 8) & 0xff);
    }
#pragma unroll
    for (int q = 0; q < kGroup; ++q) {
      const int r = g + q;
      place[r] = (uint32_t)__shfl((int)place[r], (int)((info[q] >> 16) & 0xff), 64) + (info[q] & 0xff);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the groups apart: merged, their ballot masks do not fit the scalar registers
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// After a barrier behind rank_rounds: thread b turns bin b's four wave counts into the bin's place in the tile (bin_start,
// bin_count) and the waves' offsets inside the bin (wave_cnt).  Contains one barrier; a barrier must follow before the
// results are read.  The element (wave, round r) of digit d then goes to bin_start[d] + wave_cnt[wave][d] + place[r].
__device__ __forceinline__ void rank_place_bins(RankShared& sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t c[kWaves], tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    c[w] = sh.wave_cnt[w][threadIdx.x];
    tot += c[w];
  }
  const uint32_t inc = wave_inclusive_scan(tot, lane);
  if (lane == 63) sh.wave_sum[wave] = inc;
  __syncthreads();   // everybody has read the running counts; the wave sums are there
  uint32_t start = inc - tot;
  for (int w = 0; w < wave; ++w) start += sh.wave_sum[w];
  sh.bin_start[threadIdx.x] = start;
  sh.bin_count[threadIdx.x] = tot;
  uint32_t off = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    sh.wave_cnt[w][threadIdx.x] = off;
    off += c[w];
  }
}

// exclusive prefix over the 256 values the threads hold (thread b: bin b), e.g. bin totals -> bin bases; `wave_total` holds
// kWaves words and is free again after the barrier that must follow the caller's next LDS write to it
__device__ __forceinline__ uint64_t block_exclusive_scan_256(uint64_t mine, uint64_t* wave_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_total[wave] = inc;
  __syncthreads();
  uint64_t base = inc - mine;
  for (int w = 0; w < wave; ++w) base += wave_total[w];
  return base;
}

}  // namespace r3d_sort
