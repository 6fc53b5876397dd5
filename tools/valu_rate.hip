// How many cycles does one wave64 v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 / v_min3_f32 cost per SIMD on gfx950?
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kIters = 4096, kUnroll = 16;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[kUnroll];
  f2 p[kUnroll];
  for (int i = 0; i < kUnroll; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.f}; }
  const float b = seed * 0.999f, c = seed * 0.001f;
  const f2 pb = f2{b, b}, pc = f2{c, c};
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if (OP == 3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
    }
  }
  float s = 0;
  for (int i = 0; i < kUnroll; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_min3_f32", "v_sub_f32", "v_pk_mul_f32"};
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("clockRate %d kHz, CUs %d\n", prop.clockRate, prop.multiProcessorCount);
  for (int wpb = 1; wpb <= 8; wpb *= 2) {           // workgroups per CU (4 waves each => wpb waves per SIMD)
    for (int op = 0; op < 6; ++op) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        const int grid = 256 * wpb;
        if (op == 0) k<0><<<grid, 256>>>(out, 1.f);
        if (op == 1) k<1><<<grid, 256>>>(out, 1.f);
        if (op == 2) k<2><<<grid, 256>>>(out, 1.f);
        if (op == 3) k<3><<<grid, 256>>>(out, 1.f);
        if (op == 4) k<4><<<grid, 256>>>(out, 1.f);
        if (op == 5) k<5><<<grid, 256>>>(out, 1.f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      // instructions per SIMD = wpb waves * kIters * kUnroll
      const double instr = (double)wpb * kIters * kUnroll;
      printf("waves/SIMD %d  %-14s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", wpb,
             names[op], best, best * 1e6 / instr, best * 1e6 / instr * 2.4);
    }
  }
  return 0;
}
