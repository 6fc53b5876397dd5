// LSD radix sort of 64-bit keys in HBM for gfx950 (MI355X): 8-bit digits, stable, three launches per pass.
//
// Used by the voxel path (csrc/r3d_voxel.hip) to put the distinct 48-bit Morton codes in octree order on the
// GPU instead of on the host.  HBM-bound: per pass 8 B/key read for the histogram, 8 B read + 8 B written by the
// scatter = 24 B/key/pass; 6 passes cover 48 bits.
//
//   digit_histogram_kernel  tile of 4096 keys per workgroup -> 256-bin LDS histogram -> hist[bin][workgroup]
//   digit_scan_kernel       one workgroup per bin: exclusive prefix over workgroups (contiguous chunk per thread, wave
//                           shuffle scan, LDS across waves), bin totals (the scatter turns them into bin bases itself)
//   digit_scatter_kernel    re-reads the tile in 16 rounds of 256 keys; inside a round every lane finds the lanes of
//                           its wave with the same digit by 8 ballots, the rank among them by popcount, waves are
//                           ordered through a [4][256] LDS count table; destination = bin base + workgroup prefix +
//                           running count of earlier rounds + rank.  Order of equal digits is preserved (stable).
#include "r3d_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRounds = 16;
constexpr int kTile = kThreads * kRounds;  // keys per workgroup
constexpr int kBins = 256;

__global__ __launch_bounds__(kThreads) void digit_histogram_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                                   uint32_t* __restrict__ hist, int n_blocks) {
  __shared__ uint32_t bins[kBins];
  bins[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kTile;
#pragma unroll 4
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = base + r * kThreads + threadIdx.x;
    if (i < n) atomicAdd(&bins[(keys[i] >> shift) & 0xff], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * n_blocks + blockIdx.x] = bins[threadIdx.x];
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// grid = kBins workgroups of 256 threads: bin b's counts over the workgroups become exclusive prefixes in place.  Every
// thread owns a CONTIGUOUS chunk of the row (local sum -> wave shuffle scan -> LDS across the 4 waves -> local rescan):
// two passes over the row instead of a chain of 64-wide scans (one wave per bin walked 11.8 k counters of a 48 M-key sort
// in 185 dependent steps: 95 us per pass).
__global__ __launch_bounds__(kThreads) void digit_scan_kernel(uint32_t* __restrict__ hist, int n_blocks, uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_sum[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* row = hist + (int64_t)blockIdx.x * n_blocks;
  const int per = (n_blocks + kThreads - 1) / kThreads;
  const int lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
  uint32_t mine = 0;
  for (int b = lo; b < hi; ++b) mine += row[b];
  const uint32_t inc = wave_inclusive_scan(mine, lane);
  if (lane == 63) wave_sum[wave] = inc;
  __syncthreads();
  uint32_t before = inc - mine;
  for (int w = 0; w < wave; ++w) before += wave_sum[w];
  for (int b = lo; b < hi; ++b) {
    const uint32_t v = row[b];
    row[b] = before;
    before += v;
  }
  if (threadIdx.x == kThreads - 1) totals[blockIdx.x] = before;   // the last chunk ends at the row's total (empty chunks pass it on)
}

__global__ __launch_bounds__(kThreads) void digit_scatter_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                                 const uint32_t* __restrict__ hist, int n_blocks,
                                                                 const uint32_t* __restrict__ totals,
                                                                 uint64_t* __restrict__ out) {
  __shared__ uint64_t dest[kBins];                // next free output slot of every bin for this workgroup
  __shared__ uint32_t cnt[kThreads / 64][kBins];  // per-wave digit counts of the current round
  __shared__ uint64_t wave_total[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // bin bases = exclusive prefix of the 256 bin totals, computed here by every workgroup (a launch of its own for one wave's
  // work cost more in launch latency than all workgroups repeating it: sorts of 0.3 - 0.5 M keys are launch-bound)
  uint64_t bin_base;
  {
    const uint64_t mine = totals[threadIdx.x];
    uint64_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint64_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wave_total[wave] = inc;
    __syncthreads();
    bin_base = inc - mine;
    for (int w = 0; w < wave; ++w) bin_base += wave_total[w];
  }
  dest[threadIdx.x] = bin_base + hist[(int64_t)threadIdx.x * n_blocks + blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kTile;
  for (int r = 0; r < kRounds; ++r) {
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = base + r * kThreads + threadIdx.x;
    const bool live = i < n;
    const uint64_t key = live ? keys[i] : 0;
    const uint32_t digit = (uint32_t)(key >> shift) & 0xff;
    // lanes of this wave that hold the same digit (dead lanes match nobody)
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((digit >> b) & 1);
      peers &= ((digit >> b) & 1) ? m : ~m;
    }
    const uint32_t rank = __popcll(peers & ((1ull << lane) - 1));
    if (live && rank == 0) cnt[wave][digit] = __popcll(peers);  // the lowest peer lane publishes the wave's count
    __syncthreads();
    if (live) {
      uint32_t before = 0;
#pragma unroll
      for (int w = 0; w < kThreads / 64; ++w) before += (w < wave) ? cnt[w][digit] : 0;
      out[dest[digit] + before + rank] = key;
    }
    __syncthreads();
    {
      uint32_t total = 0;
#pragma unroll
      for (int w = 0; w < kThreads / 64; ++w) total += cnt[w][threadIdx.x];
      dest[threadIdx.x] += total;
    }
    // the next round's zeroing of cnt is ordered behind this update by the barrier at its top
    __syncthreads();
  }
}

}  // namespace

// Sorts d_keys[0..n) ascending by their bits [first_bit, bits) (the span rounded up to whole 8-bit digits, <= 64), STABLY:
// keys that agree on those bits keep their input order -- so a caller whose low bits already ascend (an index packed under
// a Morton code) skips the passes over them.
// d_tmp: scratch of n keys.  The result is in d_keys when the number of passes is even, else it is copied back -- unless the
// caller asks where it ended up (d_result != NULL: *d_result = d_keys or d_tmp, no copy).
int r3d_radix_sort_u64(r3d_ctx* ctx, uint64_t* d_keys, uint64_t* d_tmp, int64_t n, int bits, int first_bit, uint64_t** d_result) {
  if (d_result) *d_result = d_keys;
  if (n <= 1) return R3D_OK;
  if (first_bit < 0 || first_bit >= bits) first_bit = 0;
  const int passes = (bits - first_bit + 7) / 8;
  const int64_t n_blocks64 = (n + kTile - 1) / kTile;
  R3D_REQUIRE(n_blocks64 < ((int64_t)1 << 31), "too many keys for one sort");
  const int n_blocks = (int)n_blocks64;
  void* ws = nullptr;
  const size_t hist_bytes = (size_t)kBins * n_blocks * sizeof(uint32_t);
  int rc = r3d_scratch(ctx, 3, hist_bytes + kBins * sizeof(uint32_t) + 64, &ws);
  if (rc) return rc;
  uint32_t* hist = static_cast<uint32_t*>(ws);
  uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + ((hist_bytes + 15) & ~(size_t)15));
  uint64_t* src = d_keys;
  uint64_t* dst = d_tmp;
  for (int p = 0; p < passes; ++p) {
    const int shift = first_bit + 8 * p;
    hipLaunchKernelGGL(digit_histogram_kernel, dim3(n_blocks), dim3(kThreads), 0, ctx->stream, src, n, shift, hist, n_blocks);
    hipLaunchKernelGGL(digit_scan_kernel, dim3(kBins), dim3(kThreads), 0, ctx->stream, hist, n_blocks, totals);
    hipLaunchKernelGGL(digit_scatter_kernel, dim3(n_blocks), dim3(kThreads), 0, ctx->stream, src, n, shift, hist, n_blocks,
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
