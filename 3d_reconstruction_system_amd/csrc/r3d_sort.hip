// LSD radix sort of 64-bit keys in HBM for gfx950 (MI355X): 8-bit digits, stable, three launches per pass.
//
// Used by the voxel path (csrc/r3d_voxel.hip) to put the distinct 48-bit Morton codes in octree order on the
// GPU instead of on the host.  HBM-bound: per pass 8 B/key read for the histogram, 8 B read + 8 B written by the
// scatter = 24 B/key/pass; 6 passes cover 48 bits.
//
//   digit_histogram_kernel  tile of 4096 keys per workgroup -> 256-bin LDS histogram -> hist[bin][workgroup]
//   digit_scan_kernel       one workgroup per bin: exclusive prefix over workgroups (contiguous chunk per thread, wave
//                           shuffle scan, LDS across waves), bin totals (the scatter turns them into bin bases itself)
//   digit_scatter_kernel    re-reads the tile; every wave ranks its quarter on its own (8 ballots per round find the lanes
//                           with the same digit, a per-wave LDS counter carries the earlier rounds), the tile is put into
//                           bin order in LDS and written out by consecutive lanes; destination = bin base + workgroup
//                           prefix + place inside the tile's bin.  Order of equal digits is preserved (stable).
#include "r3d_internal.h"
#include "r3d_sort_dev.h"

namespace {

using r3d_sort::kBins;
using r3d_sort::kRounds;
using r3d_sort::kThreads;
using r3d_sort::kTile;
using r3d_sort::wave_inclusive_scan;
using r3d_sort::xcd_contiguous;

__global__ __launch_bounds__(kThreads) void digit_histogram_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                                   uint32_t* __restrict__ hist, int stride) {
  __shared__ uint32_t bins[kBins];
  bins[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kTile;
  // all sixteen loads in flight, then the sixteen LDS adds (four at a time, each waiting for its load, left the read stream at
  // 3.3 TB/s: 120 us for 49 M keys)
  uint64_t k[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = base + r * kThreads + threadIdx.x;
    k[r] = keys[i < n ? i : n - 1];
  }
#pragma unroll
  for (int r = 0; r < kRounds; ++r)
    if (base + r * kThreads + threadIdx.x < n) atomicAdd(&bins[(k[r] >> shift) & 0xff], 1u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  hist[(int64_t)threadIdx.x * stride + blockIdx.x] = bins[threadIdx.x];
}

// grid = kBins workgroups of 256 threads: bin b's counts over the workgroups become exclusive prefixes in place.  The row
// (stride = the workgroup count rounded up to 4 counters: rows start 16-byte aligned) is walked in segments of 1024 counters:
// one COALESCED 16-byte load per thread -- four consecutive counters -- a wave shuffle scan of the threads' sums, LDS across
// the four waves, a carry from segment to segment; the next segment's load is in flight meanwhile.  (Rounds 2-4 gave every
// thread a contiguous chunk of the row: 47 loads per thread on a 48 M-key sort, each wave instruction touching 64 different
// cache lines -- 34 us per pass for 12 MB.)
__global__ __launch_bounds__(kThreads) void digit_scan_kernel(uint32_t* __restrict__ hist, int n_blocks, int stride, uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_sum[2][kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4* row = reinterpret_cast<uint4*>(hist + (int64_t)blockIdx.x * stride);
  const int n4 = stride / 4;
  auto fetch = [&](int i4) -> uint4 {
    uint4 v = i4 < n4 ? row[i4] : uint4{0, 0, 0, 0};
    const int e = 4 * i4;   // counters beyond the last workgroup are padding: whatever they hold counts as zero
    if (e + 0 >= n_blocks) v.x = 0;
    if (e + 1 >= n_blocks) v.y = 0;
    if (e + 2 >= n_blocks) v.z = 0;
    if (e + 3 >= n_blocks) v.w = 0;
    return v;
  };
  uint32_t carry = 0;
  uint4 nxt = fetch((int)threadIdx.x);
  int parity = 0;
  for (int seg = 0; seg < n4; seg += kThreads, parity ^= 1) {
    const int i4 = seg + (int)threadIdx.x;
    const uint4 v = nxt;
    nxt = fetch(i4 + kThreads);
    const uint32_t mine = v.x + v.y + v.z + v.w;
    const uint32_t inc = wave_inclusive_scan(mine, lane);
    if (lane == 63) wave_sum[parity][wave] = inc;
    __syncthreads();   // (the two copies of wave_sum alternate: one barrier per segment is enough)
    uint32_t before = carry + inc - mine, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
      const uint32_t t = wave_sum[parity][w];
      if (w < wave) before += t;
      total += t;
    }
    if (i4 < n4) row[i4] = uint4{before, before + v.x, before + v.x + v.y, before + v.x + v.y + v.z};
    carry += total;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// The tile's 4096 keys leave in BIN ORDER: ranked without a workgroup barrier per round, staged in LDS, written out by
// consecutive lanes.  (Until round 4 every round of 256 keys took four barriers and every lane stored its key where its
// rank said -- a wave instruction of 64 eight-byte stores to ~56 different places: 490 us per pass on 49 M keys = 1.6 TB/s.)
//   * wave w owns keys [1024 w, 1024 (w + 1)) of the tile, 16 rounds of 64: inside a round the lanes with the same digit find
//     each other by 8 ballots, the lowest of them bumps the wave's OWN digit counter in LDS (returning add: the wave's earlier
//     rounds are in it) and hands the old value to its peers -- rank within the wave's quarter, no other wave involved;
//   * one barrier; thread b turns bin b's four wave counts into the bin's place in the tile (exclusive scan over the bins),
//     the waves' offsets inside the bin, and the bin's global position for this tile (scan kernel's prefix + bin base);
//   * every key goes to sorted[bin start + waves before + rank] in LDS (input order inside a bin: the sort stays stable);
//   * the tile is written out front to back: lanes that follow each other write addresses that follow each other as long
//     as the bin does not change -- runs of ~16 keys = 128 B on uniform digits, the whole tile in one piece on sorted input.
// 220-250 us per pass on 49 M uniformly distributed keys = 3.2-3.6 TB/s of its 16 B/key.  What bounds it is the memory side,
// not the ranking: an any-order variant for first passes (place = a returning LDS add on the bin's cursor: 687 instructions per
// thread and tile instead of 2627) ran in 222 us against 229 (round 4; not kept), nontemporal stores in 431.
__global__ __launch_bounds__(kThreads) void digit_scatter_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                                 const uint32_t* __restrict__ hist, int stride,
                                                                 const uint32_t* __restrict__ totals,
                                                                 uint64_t* __restrict__ out) {
  using r3d_sort::kPerWave;
  using r3d_sort::kWaves;
  __shared__ uint64_t sorted[kTile];
  __shared__ r3d_sort::RankShared rk;
  __shared__ uint64_t g_base[kBins];
  __shared__ uint64_t wave_total[kWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // bin bases = exclusive prefix of the 256 bin totals, computed here by every workgroup (a launch of its own for one wave's
  // work cost more in launch latency than all workgroups repeating it: sorts of 0.3 - 0.5 M keys are launch-bound)
  r3d_sort::rank_reset(rk);
  const uint64_t bin_base = r3d_sort::block_exclusive_scan_256(totals[threadIdx.x], wave_total);   // (its barrier covers rank_reset)
  // Which tile: workgroups are dealt to the chip's 8 XCDs round-robin (workgroup b -> XCD b % 8), each XCD behind an L2 of its
  // own.  Tiles t and t + 1 write ADJACENT runs in every bin, and a run begins and ends inside a 128-byte line: dealt round-robin
  // the two halves of such a line are written through two different L2s and reach HBM as two partial (read-modify-write)
  // lines.  So an XCD takes a CONTIGUOUS eighth of the tiles: the neighbour of a partial line arrives in the same L2.
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
  const int64_t base = (int64_t)tile * kTile;
  const int64_t first = base + (int64_t)wave * kPerWave + lane;
  uint64_t key[kRounds];
  uint32_t digit[kRounds], place[kRounds], live_mask = 0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = first + r * 64;
    key[r] = i < n ? keys[i] : 0;
    digit[r] = (uint32_t)(key[r] >> shift) & 0xff;
    live_mask |= (i < n ? 1u : 0u) << r;
  }
  r3d_sort::rank_rounds(digit, live_mask, place, rk);
  __syncthreads();
  r3d_sort::rank_place_bins(rk);
  g_base[threadIdx.x] = bin_base + hist[(int64_t)threadIdx.x * stride + tile];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    if ((live_mask >> r) & 1u) sorted[rk.bin_start[digit[r]] + rk.wave_cnt[wave][digit[r]] + place[r]] = key[r];
  }
  __syncthreads();
  const int n_tile = (int)(n - base < (int64_t)kTile ? n - base : (int64_t)kTile);
#pragma unroll 4
  for (int j = threadIdx.x; j < n_tile; j += kThreads) {
    const uint64_t k = sorted[j];
    const uint32_t d = (uint32_t)(k >> shift) & 0xff;
    out[g_base[d] + (uint32_t)(j - (int)rk.bin_start[d])] = k;   // (plain stores: nontemporal ones -- partial lines past the L2 -- took 431 us instead of 239)
  }
}

}  // namespace

// Sorts d_keys[0..n) ascending by their bits [first_bit, bits) (the span rounded up to whole 8-bit digits, <= 64), STABLY:
// keys that agree on those bits keep their input order -- so a caller whose low bits already ascend (an index packed under
// a Morton code) skips the passes over them.
// d_tmp: scratch of n keys.  The result is in d_keys when the number of passes is even, else it is copied back -- unless the
// caller asks where it ended up (d_result != NULL: *d_result = d_keys or d_tmp, no copy).
// The workspace of a sort of n keys (scratch slot 3): hist[bin][r3d_sort_stride(workgroups)] + the 256 bin totals.  A producer that writes the
// keys tile by tile (kSortTile keys per workgroup, same tiling as the sort) can fill `hist` for the FIRST digit itself and
// save the sort its first histogram pass (r3d_voxel.hip's key kernel does).
int r3d_radix_sort_workspace(r3d_ctx* ctx, int64_t n, uint32_t** hist_out, int* n_blocks_out) {
  const int64_t n_blocks64 = (n + kTile - 1) / kTile;
  R3D_REQUIRE(n_blocks64 < ((int64_t)1 << 31), "too many keys for one sort");
  void* ws = nullptr;
  const size_t hist_bytes = (size_t)kBins * r3d_sort_stride((int)n_blocks64) * sizeof(uint32_t);
  int rc = r3d_scratch(ctx, 3, hist_bytes + kBins * sizeof(uint32_t) + 64, &ws);
  if (rc) return rc;
  *hist_out = static_cast<uint32_t*>(ws);
  *n_blocks_out = (int)n_blocks64;
  return R3D_OK;
}

// hist[bin][tile] -> exclusive prefixes over the tiles in place + the 256 bin totals (for radix passes built outside this file)
void r3d_sort_launch_scan(r3d_ctx* ctx, uint32_t* hist, int n_blocks, int stride, uint32_t* totals) {
  hipLaunchKernelGGL(digit_scan_kernel, dim3(kBins), dim3(kThreads), 0, ctx->stream, hist, n_blocks, stride, totals);
}

int r3d_radix_sort_u64(r3d_ctx* ctx, uint64_t* d_keys, uint64_t* d_tmp, int64_t n, int bits, int first_bit, uint64_t** d_result,
                       bool first_hist_done) {
  static_assert(kTile == kSortTile, "r3d_internal.h announces the sort's tile size");
  if (d_result) *d_result = d_keys;
  if (n <= 1) return R3D_OK;
  if (first_bit < 0 || first_bit >= bits) first_bit = 0;
  const int passes = (bits - first_bit + 7) / 8;
  uint32_t* hist = nullptr;
  int n_blocks = 0;
  int rc = r3d_radix_sort_workspace(ctx, n, &hist, &n_blocks);
  if (rc) return rc;
  const int stride = r3d_sort_stride(n_blocks);
  const size_t hist_bytes = (size_t)kBins * stride * sizeof(uint32_t);
  uint32_t* totals = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(hist) + ((hist_bytes + 15) & ~(size_t)15));
  uint64_t* src = d_keys;
  uint64_t* dst = d_tmp;
  for (int p = 0; p < passes; ++p) {
    const int shift = first_bit + 8 * p;
    if (p > 0 || !first_hist_done)
      hipLaunchKernelGGL(digit_histogram_kernel, dim3(n_blocks), dim3(kThreads), 0, ctx->stream, src, n, shift, hist, stride);
    hipLaunchKernelGGL(digit_scan_kernel, dim3(kBins), dim3(kThreads), 0, ctx->stream, hist, n_blocks, stride, totals);
    hipLaunchKernelGGL(digit_scatter_kernel, dim3(n_blocks), dim3(kThreads), 0, ctx->stream, src, n, shift, hist, stride,
                       (const uint32_t*)totals, dst);
    uint64_t* t = src;
    src = dst;
    dst = t;
  }
  R3D_HIP(hipGetLastError());
  if (d_result)
    *d_result = src;
  else if (src != d_keys)
    R3D_HIP(hipMemcpyAsync(d_keys, src, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
  return R3D_OK;
}

extern "C" {

int r3d_sort_u64(r3d_ctx* ctx, uint64_t* d_keys, int64_t n_keys, int key_bits) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_keys >= 0, "n_keys must be >= 0");
  R3D_REQUIRE(key_bits >= 1 && key_bits <= 64, "key_bits must be in [1,64]");
  if (n_keys <= 1) return R3D_OK;
  R3D_REQUIRE(d_keys != nullptr, "NULL device pointer");
  void* tmp = nullptr;
  if ((rc = r3d_scratch(ctx, 2, (size_t)n_keys * sizeof(uint64_t), &tmp))) return rc;
  return r3d_radix_sort_u64(ctx, d_keys, static_cast<uint64_t*>(tmp), n_keys, key_bits, 0);
}

}  // extern "C"
