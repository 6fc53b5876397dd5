// Scattered 8-byte memory-op rates on gfx950 over a 1 GiB table: what a hash-set insert can hope for.
// hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o tools/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

template <int OP>
__global__ __launch_bounds__(256) void k(u64* table, u64 mask, long n, u64* sink) {
  u64 acc = 0;
  for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const u64 code = mix(i) & 0xffffffffffffull;
    const u64 slot = (code * 0x9E3779B97F4A7C15ull) >> 37 & mask;
    if (OP == 0) acc += atomicCAS(&table[slot], ~0ull, code);                                  // returning CAS
    if (OP == 1) __hip_atomic_fetch_min(&table[slot], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // no-return umin x2
    if (OP == 2) __hip_atomic_fetch_or(reinterpret_cast<unsigned*>(table) + (slot * 2 & (mask * 2 + 1)), 1u << (code & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // no-return or b32
    if (OP == 3) table[slot] = code;                                                            // plain store
    if (OP == 4) acc += table[slot];                                                            // plain load
    if (OP == 5) __builtin_nontemporal_store(code, &table[slot]);                               // nt store
  }
  if (acc == 0x1234567) *sink = acc;
}
int main() {
  const u64 slots = 1ull << 27;  // 128 Mi x 8 B = 1 GiB
  const long n = 48000000;
  u64 *table, *sink;
  CK(hipMalloc(&table, slots * 8)); CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"atomicCAS x2 (returning)", "atomic umin x2 (no return)", "atomic or b32 (no return)", "plain 8-B store", "plain 8-B load", "nt 8-B store"};
  for (int op = 0; op < 6; ++op) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(table, 0xff, slots * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (op == 0) k<0><<<2048, 256>>>(table, slots - 1, n, sink);
      if (op == 1) k<1><<<2048, 256>>>(table, slots - 1, n, sink);
      if (op == 2) k<2><<<2048, 256>>>(table, slots - 1, n, sink);
      if (op == 3) k<3><<<2048, 256>>>(table, slots - 1, n, sink);
      if (op == 4) k<4><<<2048, 256>>>(table, slots - 1, n, sink);
      if (op == 5) k<5><<<2048, 256>>>(table, slots - 1, n, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("%-28s %8.3f ms for %ld scattered ops -> %.2f Gops/s\n", names[op], best, n, n / best / 1e6);
  }
  return 0;
}
