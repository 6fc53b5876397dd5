// Micro-benchmark: a 12 B/point nontemporal x3 store stream (the fused kernels' output) fed by a contiguous input stream
// of B bytes/point read B bytes per lane -- how much does an HBM-resident (larger than the 256 MiB Infinity Cache) input
// cost next to the store stream, by load width, cache policy and occupancy?   make -C tools rw_mix && tools/rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int B> struct In;
template <> struct In<1> { using T = uint8_t; static __device__ float f(T v) { return (float)v; } };
template <> struct In<2> { using T = uint16_t; static __device__ float f(T v) { return (float)v; } };
template <> struct In<4> { using T = float; static __device__ float f(T v) { return v; } };
template <> struct In<8> { using T = f32x2; static __device__ float f(T v) { return v.x + v.y; } };
template <> struct In<16> { using T = f32x4; static __device__ float f(T v) { return v.x + v.y + v.z + v.w; } };
template <int B, bool NT, int PX>
__global__ __launch_bounds__(256) void k(const typename In<B>::T* __restrict__ in, float* __restrict__ out, uint64_t n) {
  const uint64_t n_tiles = (n + 256 * PX - 1) / (256 * PX);
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    typename In<B>::T raw[PX];
#pragma unroll
    for (int r = 0; r < PX; ++r) {
      const uint64_t p = tile * 256 * PX + r * 256 + threadIdx.x;
      if (p < n) raw[r] = NT ? __builtin_nontemporal_load(in + p) : in[p];
    }
#pragma unroll
    for (int r = 0; r < PX; ++r) {
      const uint64_t p = tile * 256 * PX + r * 256 + threadIdx.x;
      if (p < n) {
        const float z = In<B>::f(raw[r]);
        asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(out + p * 3), "v"(f32x3{z, z * 2.f, z + 1.f}) : "memory");
      }
    }
  }
}
template <typename F> float time_ms(F&& f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 300; ++i) f();
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(a)); for (int i = 0; i < 50; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms / 50 < best) best = ms / 50; }
  return best;
}
template <int B, bool NT, int PX> void run(const void* in, float* out, uint64_t n, int blocks_per_cu) {
  const uint64_t n_tiles = (n + 256 * PX - 1) / (256 * PX);
  unsigned grid = blocks_per_cu > 0 ? 256u * blocks_per_cu : (unsigned)n_tiles;
  float ms = time_ms([&] { hipLaunchKernelGGL((k<B, NT, PX>), dim3(grid), dim3(256), 0, 0, (const typename In<B>::T*)in, out, n); });
  printf("  in %2d B/pt (%6.1f MB) %-3s px/lane %d grid %-9s : %.4f ms  %.2f TB/s total, store stream alone would be %.4f ms at 6.1 TB/s\n", B, n * (double)B / 1e6,
         NT ? "nt" : "", PX, blocks_per_cu > 0 ? (blocks_per_cu == 8 ? "8/CU" : "16/CU") : "1 tile/WG", ms, n * (12.0 + B) / ms / 1e9, n * 12.0 / 6.1e9);
}
int main() {
  for (uint64_t n : {(uint64_t)49152000, (uint64_t)103680000}) {
    void* in; float* out;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 12)); CK(hipMemset(in, 1, n * 16));
    printf("n = %.1f M points\n", n / 1e6);
    run<1, false, 4>(in, out, n, 8); run<2, false, 4>(in, out, n, 8); run<4, false, 4>(in, out, n, 8); run<4, true, 4>(in, out, n, 8);
    run<4, false, 4>(in, out, n, 0); run<4, true, 4>(in, out, n, 0); run<4, false, 2>(in, out, n, 0); run<4, false, 8>(in, out, n, 8);
    run<8, false, 4>(in, out, n, 8); run<16, false, 4>(in, out, n, 8); run<16, true, 4>(in, out, n, 8); run<16, true, 4>(in, out, n, 0);
    CK(hipFree(in)); CK(hipFree(out));
  }
  return 0;
}
