// Device-side pieces of the occupied-voxel set shared by r3d_voxel.hip (insert from a cloud in HBM) and r3d_fuse.hip (insert
// straight from the fused launch's registers): OctoMap's key arithmetic, the global open-addressing table, the neighbour-lane
// test, the 48-bit Morton code.  See r3d_voxel.hip for the semantics and their sources.
//
// What the sets hold: the three 16-bit OctoMap keys PACKED (x | y << 16 | z << 32), not their Morton code.  The Morton
// interleave costs ~45 vector instructions and was paid per POINT (of 123 per 64 points in voxel_insert_kernel, which PMC shows
// two-thirds VALU-busy); a set only needs an injective code, so the interleave moved to where codes leave the table
// (voxel_compact_kernel: per distinct VOXEL, tens of times fewer) and its inverse to where ready-made codes enter it.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace r3d_vox {

constexpr uint64_t kEmpty = ~0ull;
constexpr int kTreeMaxVal = 32768;
constexpr int kLdsSlots = 2048;      // per-workgroup dedupe table (16 KB)
constexpr int kLdsKeepBelow = 512;   // it is flushed to the global table once it holds this many codes (then <= 75 % full)

// 8 bits -> every third bit of 24, in 32-bit registers (the 64-bit spread costs two instructions per step)
__device__ __forceinline__ uint32_t spread3_byte(uint32_t x) {
  x = (x | (x << 8)) & 0x0000f00fu;
  x = (x | (x << 4)) & 0x000c30c3u;
  x = (x | (x << 2)) & 0x00249249u;
  return x;
}

// 48-bit Morton code of three 16-bit keys (x lowest): low bytes -> bits 0..23, high bytes -> bits 24..47
__device__ __forceinline__ uint64_t morton48(uint32_t ix, uint32_t iy, uint32_t iz) {
  const uint32_t lo = spread3_byte(ix & 0xffu) | (spread3_byte(iy & 0xffu) << 1) | (spread3_byte(iz & 0xffu) << 2);
  const uint32_t hi = spread3_byte(ix >> 8) | (spread3_byte(iy >> 8) << 1) | (spread3_byte(iz >> 8) << 2);
  return (uint64_t)lo | ((uint64_t)hi << 24);
}

// 24 bits, every third one -> 8 bits (inverse of spread3_byte)
__device__ __forceinline__ uint32_t compact3_byte(uint32_t x) {
  x &= 0x00249249u;
  x = (x | (x >> 2)) & 0x000c30c3u;
  x = (x | (x >> 4)) & 0x0000f00fu;
  x = (x | (x >> 8)) & 0x000000ffu;
  return x;
}

// packed keys (x | y << 16 | z << 32) <-> 48-bit Morton code
__device__ __forceinline__ uint64_t morton_of_key(uint64_t key) {
  return morton48((uint32_t)key & 0xffffu, ((uint32_t)key >> 16) & 0xffffu, (uint32_t)(key >> 32) & 0xffffu);
}
__device__ __forceinline__ uint64_t key_of_morton(uint64_t code) {
  const uint32_t lo = (uint32_t)code & 0xffffffu, hi = (uint32_t)(code >> 24) & 0xffffffu;
  const uint32_t ix = compact3_byte(lo) | (compact3_byte(hi) << 8);
  const uint32_t iy = compact3_byte(lo >> 1) | (compact3_byte(hi >> 1) << 8);
  const uint32_t iz = compact3_byte(lo >> 2) | (compact3_byte(hi >> 2) << 8);
  return (uint64_t)(ix | (iy << 16)) | ((uint64_t)iz << 32);
}

// Keys in fp64 exactly as OcTreeBaseImpl::coordToKey computes them from FLOAT coordinates.  (An exact fp32 formulation -- t =
// round(x f), r = fma(x, f, -t), floor(t) corrected when t is an integer and r < 0; valid when 1/res is exactly a float --
// passed every face / ulp / denormal test in round 3 and was 4 % SLOWER on the scan benchmark: the kernel does not wait for
// its ALUs.  Not shipped.)
__device__ __forceinline__ bool voxel_key(float x, float y, float z, double factor, uint64_t* key) {
  const double dx = floor(factor * (double)x), dy = floor(factor * (double)y), dz = floor(factor * (double)z);
  // rejects NaN/inf and anything outside the 2^16 key range
  const bool ok = dx >= -(double)kTreeMaxVal && dx < (double)kTreeMaxVal && dy >= -(double)kTreeMaxVal &&
                  dy < (double)kTreeMaxVal && dz >= -(double)kTreeMaxVal && dz < (double)kTreeMaxVal;
  if (!ok) return false;
  const uint32_t ix = (uint32_t)((int)dx + kTreeMaxVal), iy = (uint32_t)((int)dy + kTreeMaxVal), iz = (uint32_t)((int)dz + kTreeMaxVal);
  *key = (uint64_t)(ix | (iy << 16)) | ((uint64_t)iz << 32);
  return true;
}

// the value the previous lane of the wave holds (lane 0: its own), by DPP wave_shr:1 -- one VALU move per 32 bits instead of a
// ds_bpermute round trip through the LDS crossbar (__shfl_up compiles to two of those for a 64-bit value)
__device__ __forceinline__ uint64_t prev_lane_u64(uint64_t v) {
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
  return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32);
}

// Every LDS operation of this wave has been performed.  To be called in front of a __syncthreads() that a result-less LDS
// atomic (ds_add_u32 ...) can reach along a loop's back edge: hipcc (ROCm 7.2, gfx950) leaves the s_waitcnt of the barrier's
// release fence out on that path, the add may still sit in its SIMD's LDS queue when another wave's read behind the barrier
// is served, the waves then disagree about a count that has to be workgroup-uniform and take different numbers of barriers.
// Seen in round 3 (fuse_voxel_kernel: wrong colour words and lost codes at 20 M points; voxel_insert_kernel had the same
// hole in its ISA without ever failing a test).  tools/isa_barrier_check.py scans the compiled kernels for the pattern.
__device__ __forceinline__ void lds_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// A workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is also a fence for global memory: hipcc puts
// s_waitcnt vmcnt(0) in front of its s_barrier, i.e. every global load and store of the wave has to have completed -- which
// ends any overlap between a loop iteration's write-back / the next iteration's prefetch and the LDS work in between.  Where a
// barrier protects nothing but an LDS array this one leaves the wave's global operations in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Three per-thread counts reach three global counters as ONE atomicAdd per WORKGROUP and non-zero counter.  Adds to one
// address complete at ~0.09 G/s on this chip (11 ns each, whoever issues them): a SHORT kernel with many workgroups that each
// end with per-wave adds waits for them (the sort-merge insert's merge: 32768 adds = 360 of its 455 us; it now leaves per-
// workgroup partials instead).  Long kernels whose waves end spread over milliseconds do not -- see voxel_insert_kernel.
// `wg`: three words of LDS; every thread of the workgroup calls this once, at the end.
__device__ __forceinline__ void flush_counts(unsigned n_new, unsigned n_ignored, unsigned n_over, unsigned* wg,
                                             unsigned long long* __restrict__ counters) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_new += __shfl_down(n_new, off, 64);
    n_ignored += __shfl_down(n_ignored, off, 64);
    n_over += __shfl_down(n_over, off, 64);
  }
  lds_barrier();
  if (threadIdx.x < 3) wg[threadIdx.x] = 0;
  lds_barrier();
  if ((threadIdx.x & 63) == 0) {
    if (n_new) atomicAdd(&wg[0], n_new);
    if (n_ignored) atomicAdd(&wg[1], n_ignored);
    if (n_over) atomicAdd(&wg[2], n_over);
  }
  lds_barrier();
  if (threadIdx.x < 3 && wg[threadIdx.x]) atomicAdd(&counters[threadIdx.x], (unsigned long long)wg[threadIdx.x]);
}

// Where a packed key lives in the table: the top log2cap bits of h48 = key * G mod 2^48.  G is odd, so key -> h48 is a
// BIJECTION of the 48-bit keys (key = h48 * G^-1 mod 2^48): the top 16 bits of h48 name one of 65536 consecutive pieces of the
// table and the low 32 bits say which of the 2^32 keys of that piece it is.  The sort-merge insert (r3d_voxel.hip) lives on
// that: once the keys are in piece order a key is its 32-bit remainder -- the sort moves 4-byte words, not 8-byte ones.
// (Rounds 2-4 hashed with the 64-bit golden-ratio multiplier; the table is internal, its layout was free to change.)
constexpr uint64_t kMask48 = ((uint64_t)1 << 48) - 1;
constexpr uint64_t kHashMul48 = 0x9E3779B97F4Bull;   // ~ 2^48 / golden ratio, odd
constexpr uint64_t inverse_mod_2_64(uint64_t a) {      // Newton: x <- x (2 - a x) doubles the correct low bits (a odd)
  uint64_t x = a;
  for (int k = 0; k < 6; ++k) x *= 2 - a * x;
  return x;
}
constexpr uint64_t kHashInv48 = inverse_mod_2_64(kHashMul48) & kMask48;
static_assert(((kHashMul48 * kHashInv48) & kMask48) == 1, "48-bit multiplicative inverse");
__host__ __device__ __forceinline__ uint64_t hash48(uint64_t key) { return (key * kHashMul48) & kMask48; }
__host__ __device__ __forceinline__ uint64_t unhash48(uint64_t h48) { return (h48 * kHashInv48) & kMask48; }
__host__ __device__ __forceinline__ uint64_t home_slot(uint64_t key, int log2cap) { return hash48(key) >> (48 - log2cap); }

// One code into the global open-addressing table (64-bit CAS, linear probing).  Returns 1 new, 0 already there, -1 no slot.
__device__ __forceinline__ int table_insert(uint64_t* __restrict__ table, uint64_t mask, int log2cap, uint64_t code) {
  uint64_t slot = home_slot(code, log2cap);
  for (uint64_t probe = 0; probe <= mask; ++probe) {
    const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[slot]), (unsigned long long)kEmpty,
                                   (unsigned long long)code);
    if (old == kEmpty) return 1;
    if (old == code) return 0;
    slot = (slot + 1) & mask;
  }
  return -1;
}

// One code into a workgroup's LDS set.  *mine: this lane put it there; returns false only when the set is full (the caller
// then sends the code to the global table directly).  The shape of this code matters more than its instruction count:
//  * one exit and a `done` flag -- the same logic with early returns compiled to a 13 % slower insert kernel (214 -> 186
//    Gpoints/s on the scan benchmark);
//  * one round at a time -- reading the four rounds' first slots back to back (one LDS round trip per tile instead of four)
//    and resolving them afterwards LOST 14 % on scans (222 -> 188) and 11 % on the worst case (15.1 -> 13.4 G inserts/s).
__device__ __forceinline__ bool lds_set_claim(unsigned long long* local_set, uint64_t code, bool* mine_out) {
  // 11 bits from two 32-bit multiplicative hashes of the packed key's halves (a 64-bit multiply is five instructions)
  uint32_t slot = (((uint32_t)code * 0x9E3779B1u) ^ ((uint32_t)(code >> 32) * 0x85EBCA77u)) >> 21;
  bool mine = false, done = false;
  // most codes of a scan are already in the set (the previous rows put them there): a plain LDS read settles those
  // without a compare-and-swap
  if (local_set[slot] == code) done = true;
  for (int probe = 0; probe < kLdsSlots && !done; ++probe) {
    const unsigned long long old = atomicCAS(&local_set[slot], (unsigned long long)kEmpty, (unsigned long long)code);
    if (old == kEmpty) {
      mine = true;
      done = true;
    } else if (old == code) {
      done = true;  // already in the set: it reaches the global table with the next flush
    } else {
      slot = (slot + 1) & (kLdsSlots - 1);
    }
  }
  *mine_out = mine;
  return done;
}

}  // namespace r3d_vox
