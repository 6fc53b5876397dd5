// Calibration: what does this MI355X sustain for a pure 16-B/lane store stream, a copy and a
// 1:12 read:write mix (the fused kernel's shape)?  hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill_k(f32x4* __restrict__ out, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f32x4{v, v, v, v};
}
__global__ void fill_nt_k(f32x4* __restrict__ out, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(f32x4{v, v, v, v}, out + i);
}
__global__ void copy_k(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// one dword read per three 16-B stores, block-contiguous like the fused kernel's tiles
__global__ void mix_k(const unsigned* __restrict__ in, f32x4* __restrict__ out, size_t n_in) {
  for (size_t t = blockIdx.x; t * 256 < n_in; t += gridDim.x) {
    const unsigned w = in[t * 256 + threadIdx.x];
    const float f = (float)w;
    f32x4* o = out + t * 768;
    o[threadIdx.x] = f32x4{f, f, f, f};
    o[256 + threadIdx.x] = f32x4{f, f, f, f};
    o[512 + threadIdx.x] = f32x4{f, f, f, f};
  }
}
int main() {
  const size_t out_bytes = 589824000, n16 = out_bytes / 16, n_in = out_bytes / 12 / 4;  // C2: 49,152,000 points
  void *a, *b;
  CK(hipMalloc(&a, out_bytes)); CK(hipMalloc(&b, out_bytes));
  CK(hipMemset(a, 1, out_bytes)); CK(hipMemset(b, 1, out_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {1024, 2048, 4096, 16384, 65536};
  for (int g : grids) {
    float best[5] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 5; ++rep) {
      for (int k = 0; k < 5; ++k) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 5; ++it) {
          if (k == 0) fill_k<<<g, 256>>>((f32x4*)a, n16, 1.f);
          if (k == 1) fill_nt_k<<<g, 256>>>((f32x4*)a, n16, 1.f);
          if (k == 2) copy_k<<<g, 256>>>((const f32x4*)a, (f32x4*)b, n16);
          if (k == 3) mix_k<<<g, 256>>>((const unsigned*)b, (f32x4*)a, n_in);
          if (k == 4) CK(hipMemsetAsync(a, 0, out_bytes));
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best[k] = std::min(best[k], ms / 5);
      }
    }
    printf("grid %6d: fill %.4f ms %.0f GB/s | fill_nt %.4f ms %.0f GB/s | copy %.4f ms %.0f GB/s (r+w) | mix1:12 %.4f ms %.0f GB/s | hipMemset %.4f ms %.0f GB/s\n", g,
           best[0], out_bytes / best[0] / 1e6, best[1], out_bytes / best[1] / 1e6, best[2], 2.0 * out_bytes / best[2] / 1e6,
           best[3], (out_bytes + n_in * 4.0) / best[3] / 1e6, best[4], out_bytes / best[4] / 1e6);
  }
  return 0;
}
