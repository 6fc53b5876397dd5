// Experiment (round 3): is the DRAM's read/write-mix penalty on config 5's fused launch (7 B/point read,
// 16 B/point written: 30 % reads) avoidable by separating reads and writes IN TIME inside the kernel instead of sprinkling
// them?  Compute-free models of the launch (8 B/point in as one 8-byte load, 12 + 4 B/point out with the fused kernel's
// store shapes), inputs far larger than the 256 MiB Infinity Cache:
//   mixed      one 1024-pixel tile per workgroup, load -> store, like fuse_rgb_kernel today
//   front T    2048 resident workgroups per launch; every lane first issues ALL loads of its T tiles (registers), then
//              stores them: the whole chip reads for the first microseconds of a launch and then only writes; launches of
//              2048 x T tiles follow one another (launch boundaries = the chip-wide phase barrier)
//   reads/writes alone  the two streams by themselves (what perfect separation could reach: their sum)
//   make -C tools rw_phase && tools/rw_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_point(float* xyz, uint32_t* rgba, uint64_t p, f32x2 v) {
  asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(xyz + p * 3), "v"(f32x3{v.x, v.x * 2.f, v.y + 1.f}) : "memory");
  asm volatile("global_store_dword %0, %1, off nt" ::"v"(rgba + p), "v"(__float_as_uint(v.y)) : "memory");
}

__global__ __launch_bounds__(256) void mixed(const f32x2* __restrict__ in, float* __restrict__ xyz, uint32_t* __restrict__ rgba, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  f32x2 raw[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    raw[r] = in[p < n ? p : n - 1];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    if (p < n) store_point(xyz, rgba, p, raw[r]);
  }
}

template <int T>
__global__ __launch_bounds__(256) void front(const f32x2* __restrict__ in, float* __restrict__ xyz, uint32_t* __restrict__ rgba, uint64_t n,
                                             uint64_t tile0) {
  const uint64_t base = (tile0 + (uint64_t)blockIdx.x * T) * 1024 + threadIdx.x;
  f32x2 raw[T * 4];
#pragma unroll
  for (int k = 0; k < T * 4; ++k) {
    const uint64_t p = base + (uint64_t)k * 256;
    raw[k] = in[p < n ? p : n - 1];
  }
#pragma unroll
  for (int k = 0; k < T * 4; ++k) {
    const uint64_t p = base + (uint64_t)k * 256;
    if (p < n) store_point(xyz, rgba, p, raw[k]);
  }
}

__global__ __launch_bounds__(256) void reads_only(const f32x2* __restrict__ in, float* __restrict__ sink, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    const f32x2 v = in[p < n ? p : n - 1];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e30f) *sink = acc;
}

__global__ __launch_bounds__(256) void writes_only(float* __restrict__ xyz, uint32_t* __restrict__ rgba, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    if (p < n) store_point(xyz, rgba, p, f32x2{(float)p, 1.f});
  }
}

// the same two streams, but the colour words of FOUR consecutive pixels leave as one 16-byte store per lane (lane t of a
// tile owns pixels 4t .. 4t+3 of it for the colour stream; the xyz stream keeps its 12 B-per-lane shape)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_xyz(float* xyz, uint64_t p, f32x2 v) {
  asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(xyz + p * 3), "v"(f32x3{v.x, v.x * 2.f, v.y + 1.f}) : "memory");
}
__device__ __forceinline__ void store_rgba4(uint32_t* rgba, uint64_t p4, uint64_t n, f32x2 a, f32x2 b) {
  if (p4 + 3 < n) {
    const u32x4 w = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.y)};
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(rgba + p4), "v"(w) : "memory");
  }
}
__global__ __launch_bounds__(256) void writes_only_v4(float* __restrict__ xyz, uint32_t* __restrict__ rgba, uint64_t n) {
  const uint64_t t0 = (uint64_t)blockIdx.x * 1024, base = t0 + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    if (p < n) store_xyz(xyz, p, f32x2{(float)p, 1.f});
  }
  store_rgba4(rgba, t0 + 4 * (uint64_t)threadIdx.x, n, f32x2{1.f, 2.f}, f32x2{3.f, 4.f});
}
__global__ __launch_bounds__(256) void mixed_v4(const f32x2* __restrict__ in, float* __restrict__ xyz, uint32_t* __restrict__ rgba, uint64_t n) {
  const uint64_t t0 = (uint64_t)blockIdx.x * 1024, base = t0 + threadIdx.x;
  f32x2 raw[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    raw[r] = in[p < n ? p : n - 1];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    if (p < n) store_xyz(xyz, p, raw[r]);
  }
  store_rgba4(rgba, t0 + 4 * (uint64_t)threadIdx.x, n, raw[0], raw[1]);
}
__global__ __launch_bounds__(256) void writes_only_xyz(float* __restrict__ xyz, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    if (p < n) store_xyz(xyz, p, f32x2{(float)p, 1.f});
  }
}

// median of 9 groups of 10 launches after >= 150 ms of warm-up on the same kernel (an idle GPU boosts, dips, then settles;
// best-of-five after 20 launches gave run-to-run swings of 15 %)
template <typename F> float time_ms(F&& f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  float warm = 0.f;
  while (warm < 150.f) { for (int i = 0; i < 20; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&warm, a, b)); }
  float t[9];
  for (int rep = 0; rep < 9; ++rep) { CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&t[rep], a, b)); t[rep] /= 10; }
  for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) if (t[j] < t[i]) { float x = t[i]; t[i] = t[j]; t[j] = x; }
  return t[4];
}

template <int T> void run_front(const f32x2* in, float* xyz, uint32_t* rgba, uint64_t n, unsigned wgs) {
  const uint64_t n_tiles = (n + 1023) / 1024;
  const uint64_t per_launch = (uint64_t)wgs * T;
  float ms = time_ms([&] {
    for (uint64_t t0 = 0; t0 < n_tiles; t0 += per_launch) {
      const uint64_t left = n_tiles - t0;
      const unsigned grid = (unsigned)((left < per_launch ? left : per_launch + T - 1) / T);
      hipLaunchKernelGGL((front<T>), dim3(grid < 1 ? 1 : grid), dim3(256), 0, 0, in, xyz, rgba, n, t0);
    }
  });
  printf("  front T=%d, %4u workgroups per launch (%3llu launches of %5.1f MB in / %5.1f MB out): %.4f ms  %.2f TB/s\n", T, wgs,
         (unsigned long long)((n_tiles + per_launch - 1) / per_launch), per_launch * 1024 * 8.0 / 1e6, per_launch * 1024 * 16.0 / 1e6, ms,
         n * 24.0 / ms / 1e9);
}

int main() {
 for (int round = 0; round < 2; ++round) {   // everything twice: the two rounds must agree
  const uint64_t n = 103680000;   // 50 frames of 1920 x 1080
  f32x2* in; float* xyz; uint32_t* rgba; float* sink;
  CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&xyz, n * 12)); CK(hipMalloc(&rgba, n * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(in, 1, n * 8));
  const unsigned tiles = (unsigned)((n + 1023) / 1024);
  printf("n = %.1f M points, 8 B/point in (%.0f MB), 16 B/point out (%.0f MB)\n", n / 1e6, n * 8.0 / 1e6, n * 16.0 / 1e6);
  float r = time_ms([&] { hipLaunchKernelGGL(reads_only, dim3(tiles), dim3(256), 0, 0, in, sink, n); });
  float w = time_ms([&] { hipLaunchKernelGGL(writes_only, dim3(tiles), dim3(256), 0, 0, xyz, rgba, n); });
  float m = time_ms([&] { hipLaunchKernelGGL(mixed, dim3(tiles), dim3(256), 0, 0, in, xyz, rgba, n); });
  printf("  reads alone  %.4f ms  %.2f TB/s\n  writes alone %.4f ms  %.2f TB/s\n  sum (perfect separation) %.4f ms  %.2f TB/s\n", r, n * 8.0 / r / 1e9, w,
         n * 16.0 / w / 1e9, r + w, n * 24.0 / (r + w) / 1e9);
  printf("  mixed (one tile per workgroup): %.4f ms  %.2f TB/s\n", m, n * 24.0 / m / 1e9);
  float w3 = time_ms([&] { hipLaunchKernelGGL(writes_only_xyz, dim3(tiles), dim3(256), 0, 0, xyz, n); });
  float w4 = time_ms([&] { hipLaunchKernelGGL(writes_only_v4, dim3(tiles), dim3(256), 0, 0, xyz, rgba, n); });
  float m4 = time_ms([&] { hipLaunchKernelGGL(mixed_v4, dim3(tiles), dim3(256), 0, 0, in, xyz, rgba, n); });
  printf("  writes alone, xyz stream only (12 B/point): %.4f ms  %.2f TB/s\n", w3, n * 12.0 / w3 / 1e9);
  printf("  writes alone, colour as one 16-byte store per 4 pixels: %.4f ms  %.2f TB/s\n", w4, n * 16.0 / w4 / 1e9);
  printf("  mixed, colour as one 16-byte store per 4 pixels: %.4f ms  %.2f TB/s\n", m4, n * 24.0 / m4 / 1e9);
  run_front<2>(in, xyz, rgba, n, 2048); run_front<4>(in, xyz, rgba, n, 2048); run_front<6>(in, xyz, rgba, n, 2048);
  run_front<4>(in, xyz, rgba, n, 1024); run_front<8>(in, xyz, rgba, n, 1024); run_front<12>(in, xyz, rgba, n, 1024);
  run_front<8>(in, xyz, rgba, n, 512);
  CK(hipFree(in)); CK(hipFree(xyz)); CK(hipFree(rgba)); CK(hipFree(sink));
 }
  return 0;
}
