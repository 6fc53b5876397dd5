// Experiment (round 4, VERDICT weak point 8): the write-only leg of config 5's launch (12 B/point xyz + 4 B/point rgba, 103.7 M
// points) was bimodal between allocations (5.9 vs 6.9 TB/s, profiles/r03_rw_phase.log).  Is it the RELATIVE placement of the two
// output arrays (something a caller or the library could choose), the alignment of each, or the physical backing of an
// allocation (something nobody chooses)?  One big allocation, the two arrays placed inside it at swept offsets; then the same
// placement in a fresh allocation, several times.
//   make -C tools placement && tools/placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x3 __attribute__((ext_vector_type(3)));

__global__ __launch_bounds__(256) void writes_two(float* __restrict__ xyz, uint32_t* __restrict__ rgba, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    if (p < n) {
      asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(xyz + p * 3), "v"(f32x3{(float)p, 2.f, 3.f}) : "memory");
      asm volatile("global_store_dword %0, %1, off nt" ::"v"(rgba + p), "v"((uint32_t)p) : "memory");
    }
  }
}
__global__ __launch_bounds__(256) void writes_xyz(float* __restrict__ xyz, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t p = base + r * 256;
    if (p < n) asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(xyz + p * 3), "v"(f32x3{(float)p, 2.f, 3.f}) : "memory");
  }
}

template <typename F> float time_ms(F&& f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  float warm = 0.f;
  while (warm < 60.f) { for (int i = 0; i < 20; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&warm, a, b)); }
  float t[7];
  for (int rep = 0; rep < 7; ++rep) { CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&t[rep], a, b)); t[rep] /= 10; }
  for (int i = 0; i < 7; ++i) for (int j = i + 1; j < 7; ++j) if (t[j] < t[i]) { float x = t[i]; t[i] = t[j]; t[j] = x; }
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return t[3];
}

int main() {
  const uint64_t n = 103680000;   // 50 frames of 1920 x 1080
  const unsigned tiles = (unsigned)((n + 1023) / 1024);
  const size_t slack = (size_t)256 << 20;
  char* big;
  CK(hipMalloc(&big, n * 16 + 2 * slack));
  printf("one allocation at %p; xyz at +0, rgba at +xyz bytes + delta\n", (void*)big);
  const size_t deltas[] = {0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 8 << 20, 16 << 20,
                           (size_t)32 << 20, (size_t)33 << 20, (size_t)64 << 20, (size_t)100 << 20, (size_t)128 << 20, (size_t)200 << 20};
  for (size_t d : deltas) {
    float* xyz = reinterpret_cast<float*>(big);
    uint32_t* rgba = reinterpret_cast<uint32_t*>(big + n * 12 + d);
    float w = time_ms([&] { hipLaunchKernelGGL(writes_two, dim3(tiles), dim3(256), 0, 0, xyz, rgba, n); });
    printf("  delta %10zu B : %.4f ms  %.2f TB/s\n", d, w, n * 16.0 / w / 1e9);
  }
  printf("xyz base shifted inside the allocation (rgba right behind it)\n");
  const size_t shifts[] = {0, 256, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, (size_t)100 << 20};
  for (size_t sft : shifts) {
    float* xyz = reinterpret_cast<float*>(big + sft);
    uint32_t* rgba = reinterpret_cast<uint32_t*>(big + sft + n * 12);
    float w = time_ms([&] { hipLaunchKernelGGL(writes_two, dim3(tiles), dim3(256), 0, 0, xyz, rgba, n); });
    float w1 = time_ms([&] { hipLaunchKernelGGL(writes_xyz, dim3(tiles), dim3(256), 0, 0, xyz, n); });
    printf("  shift %10zu B : two streams %.4f ms  %.2f TB/s ; xyz alone %.4f ms  %.2f TB/s\n", sft, w, n * 16.0 / w / 1e9, w1, n * 12.0 / w1 / 1e9);
  }
  printf("layouts inside the one allocation, offsets rounded to 2 MiB\n");
  {
    const size_t M2 = (size_t)2 << 20;
    const size_t xyz_b = (n * 12 + M2 - 1) / M2 * M2, rgba_b = (n * 4 + M2 - 1) / M2 * M2;
    struct { const char* what; size_t xyz_off, rgba_off; } lay[] = {
        {"xyz at +0, rgba at the next 2 MiB boundary behind it", 0, xyz_b},
        {"rgba at +0, xyz at the next 2 MiB boundary behind it", rgba_b, 0},
        {"rgba at +0, xyz right behind it (4 KiB aligned only)", n * 4, 0},
        {"xyz at +0, rgba 2 MiB-aligned + 1 MiB", 0, xyz_b + (M2 >> 1)},
        {"xyz at +1 MiB, rgba at a 2 MiB boundary", M2 >> 1, xyz_b + M2},
    };
    for (auto& L : lay) {
      float* xyz = reinterpret_cast<float*>(big + L.xyz_off);
      uint32_t* rgba = reinterpret_cast<uint32_t*>(big + L.rgba_off);
      float w = time_ms([&] { hipLaunchKernelGGL(writes_two, dim3(tiles), dim3(256), 0, 0, xyz, rgba, n); });
      printf("  %-58s: %.4f ms  %.2f TB/s\n", L.what, w, n * 16.0 / w / 1e9);
    }
  }
  CK(hipFree(big));
  printf("two allocations, rgba FIRST then xyz / xyz first then rgba (allocation order decides which lies lower)\n");
  for (int order = 0; order < 2; ++order) {
    float* xyz; uint32_t* rgba;
    if (order == 0) { CK(hipMalloc(&rgba, n * 4)); CK(hipMalloc(&xyz, n * 12)); } else { CK(hipMalloc(&xyz, n * 12)); CK(hipMalloc(&rgba, n * 4)); }
    float w = time_ms([&] { hipLaunchKernelGGL(writes_two, dim3(tiles), dim3(256), 0, 0, xyz, rgba, n); });
    printf("  xyz %p rgba %p : %.2f TB/s\n", (void*)xyz, (void*)rgba, n * 16.0 / w / 1e9);
    CK(hipFree(xyz)); CK(hipFree(rgba));
  }
  printf("fresh pairs of allocations (what a caller gets)\n");
  for (int k = 0; k < 6; ++k) {
    float* xyz; uint32_t* rgba; void* pad = nullptr;
    if (k & 1) CK(hipMalloc(&pad, ((size_t)37 << 20) * (k + 1)));
    CK(hipMalloc(&xyz, n * 12)); CK(hipMalloc(&rgba, n * 4));
    float w = time_ms([&] { hipLaunchKernelGGL(writes_two, dim3(tiles), dim3(256), 0, 0, xyz, rgba, n); });
    float w1 = time_ms([&] { hipLaunchKernelGGL(writes_xyz, dim3(tiles), dim3(256), 0, 0, xyz, n); });
    printf("  xyz %p rgba %p (rgba - xyz - bytes = %lld): two streams %.2f TB/s ; xyz alone %.2f TB/s\n", (void*)xyz, (void*)rgba,
           (long long)((char*)rgba - (char*)xyz) - (long long)(n * 12), n * 16.0 / w / 1e9, n * 12.0 / w1 / 1e9);
    CK(hipFree(xyz)); CK(hipFree(rgba)); if (pad) CK(hipFree(pad));
  }
  return 0;
}
