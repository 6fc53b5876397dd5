// What is the store ceiling of this MI355X for the fused kernel's output pattern?  Pure-store kernels over 589.8 MB.
// hipcc --offload-arch=gfx950 -O3 tools/storebw.hip -o tools/storebw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: 12-byte nt store per lane, 4 rounds of 256 lanes per 1024-point tile (the fused kernel's shape, no math)
__global__ __launch_bounds__(256) void k_x3nt(float* __restrict__ out, long n_pts) {
  const long n_tiles = n_pts / 1024;
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* dst = out + (t * 1024 + r * 256 + threadIdx.x) * 3;
      asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(f32x3{1.f, 2.f, 3.f}) : "memory");
    }
  }
}
// B: same bytes as 16-byte nt stores, tile-linear (768 pieces per tile)
__global__ __launch_bounds__(256) void k_x4nt(float* __restrict__ out, long n_pts) {
  const long n_tiles = n_pts / 1024;
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
#pragma unroll
    for (int r = 0; r < 3; ++r) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, reinterpret_cast<f32x4*>(out + t * 3072) + r * 256 + threadIdx.x);
  }
}
// C: 16-byte plain stores, tile-linear
__global__ __launch_bounds__(256) void k_x4(float* __restrict__ out, long n_pts) {
  const long n_tiles = n_pts / 1024;
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
#pragma unroll
    for (int r = 0; r < 3; ++r) reinterpret_cast<f32x4*>(out + t * 3072)[r * 256 + threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
  }
}
// D: every wave owns 4 KiB contiguous per iteration (4 back-to-back 1-KiB nt stores)
__global__ __launch_bounds__(256) void k_wave4k(float* __restrict__ out, long n_pts) {
  const long n_chunks = n_pts * 12 / 4096;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  for (long c = wave; c < n_chunks; c += n_waves) {
    f32x4* base = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + c * 4096);
#pragma unroll
    for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, base + r * 64 + lane);
  }
}
// E: x3 nt with 8 rounds per tile (2048-point tiles)
__global__ __launch_bounds__(256) void k_x3nt8(float* __restrict__ out, long n_pts) {
  const long n_tiles = n_pts / 2048;
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float* dst = out + (t * 2048 + r * 256 + threadIdx.x) * 3;
      asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(f32x3{1.f, 2.f, 3.f}) : "memory");
    }
  }
}
// F: x3 with sc0 sc1 nt bits (write-through system scope)
__global__ __launch_bounds__(256) void k_x3sc(float* __restrict__ out, long n_pts) {
  const long n_tiles = n_pts / 1024;
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* dst = out + (t * 1024 + r * 256 + threadIdx.x) * 3;
      asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1" ::"v"(dst), "v"(f32x3{1.f, 2.f, 3.f}) : "memory");
    }
  }
}
// G: rocclr-fill shape: flat grid-stride over 16-byte pieces, few workgroups, unrolled
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_flat(f32x4* __restrict__ out, long n16) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, out + i + u * stride);
      else out[i + u * stride] = f32x4{1.f, 2.f, 3.f, 4.f};
    }
  }
  for (; i < n16; i += stride) out[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
// H: flat grid-stride over 12-byte points with x3 nt stores, unrolled
template <int UNROLL>
__global__ __launch_bounds__(256) void k_flat3(float* __restrict__ out, long n_pts) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n_pts; i += UNROLL * stride) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(out + (i + u * stride) * 3), "v"(f32x3{1.f, 2.f, 3.f}) : "memory");
  }
  for (; i < n_pts; i += stride) asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(out + i * 3), "v"(f32x3{1.f, 2.f, 3.f}) : "memory");
}
int main() {
  const long n_pts = 49152000;
  const size_t bytes = (size_t)n_pts * 12;
  float* a;
  CK(hipMalloc(&a, bytes));
  CK(hipMemset(a, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"x3 nt (kernel shape)", "x4 nt tile-linear", "x4 plain tile-linear", "wave-owned 4 KiB nt", "x3 nt, 2048-pt tiles", "x3 sc0 sc1", "hipMemsetAsync"};
  const int grids[] = {1024, 2048, 4096, 8192, 48000};
  for (int g : grids) {
    printf("grid %5d:", g);
    for (int k = 0; k < 7; ++k) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) {
          if (k == 0) k_x3nt<<<g, 256>>>(a, n_pts);
          if (k == 1) k_x4nt<<<g, 256>>>(a, n_pts);
          if (k == 2) k_x4<<<g, 256>>>(a, n_pts);
          if (k == 3) k_wave4k<<<g, 256>>>(a, n_pts);
          if (k == 4) k_x3nt8<<<g, 256>>>(a, n_pts);
          if (k == 5) k_x3sc<<<g, 256>>>(a, n_pts);
          if (k == 6) CK(hipMemsetAsync(a, 0, bytes));
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 10);
      }
      printf("  %s %.4f ms %.0f GB/s |", names[k], best, bytes / best / 1e6);
    }
    printf("\n");
  }
  printf("flat grid-stride fills (rocclr shape):\n");
  for (int g : {256, 512, 1024, 2048}) {
    printf("grid %5d:", g);
    for (int k = 0; k < 6; ++k) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) {
          if (k == 0) k_flat<1, false><<<g, 256>>>((f32x4*)a, bytes / 16);
          if (k == 1) k_flat<4, false><<<g, 256>>>((f32x4*)a, bytes / 16);
          if (k == 2) k_flat<8, false><<<g, 256>>>((f32x4*)a, bytes / 16);
          if (k == 3) k_flat<8, true><<<g, 256>>>((f32x4*)a, bytes / 16);
          if (k == 4) k_flat3<4><<<g, 256>>>(a, n_pts);
          if (k == 5) k_flat3<8><<<g, 256>>>(a, n_pts);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 10);
      }
      const char* nm[] = {"x4 u1", "x4 u4", "x4 u8", "x4 nt u8", "x3 nt u4", "x3 nt u8"};
      printf("  %s %.4f ms %.0f GB/s |", nm[k], best, bytes / best / 1e6);
    }
    printf("\n");
  }
  return 0;
}
