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
  double d[kUnroll];
  unsigned u[kUnroll];
  for (int i = 0; i < kUnroll; ++i) { d[i] = a[i]; u[i] = (unsigned)a[i]; }
  const double db = b, dc = c;
  const unsigned ub = 0x9E3779B9u;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if (OP == 3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
      if (OP == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      if (OP == 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      if (OP == 9) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
      if (OP == 10) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
      if (OP == 11) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(u[i]));
      if (OP == 12) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
      if (OP == 13) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(ub));
      if (OP == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 15) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  float s = 0;
  for (int i = 0; i < kUnroll; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i] + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_min3_f32", "v_sub_f32", "v_pk_mul_f32", "v_fma_f64", "v_mul_f64",
                         "v_add_f64", "v_cvt_f64_u32", "v_cvt_f32_f64", "v_cvt_f32_ubyte0", "v_cvt_f64_f32", "v_mul_hi_u32", "v_fmac_f32", "v_mul_f32"};
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("clockRate %d kHz, CUs %d\n", prop.clockRate, prop.multiProcessorCount);
  for (int wpb = 8; wpb <= 8; wpb *= 2) {           // workgroups per CU (4 waves each => wpb waves per SIMD)
    for (int op = 0; op < 16; ++op) {
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
        if (op == 6) k<6><<<grid, 256>>>(out, 1.f);
        if (op == 7) k<7><<<grid, 256>>>(out, 1.f);
        if (op == 8) k<8><<<grid, 256>>>(out, 1.f);
        if (op == 9) k<9><<<grid, 256>>>(out, 1.f);
        if (op == 10) k<10><<<grid, 256>>>(out, 1.f);
        if (op == 11) k<11><<<grid, 256>>>(out, 1.f);
        if (op == 12) k<12><<<grid, 256>>>(out, 1.f);
        if (op == 13) k<13><<<grid, 256>>>(out, 1.f);
        if (op == 14) k<14><<<grid, 256>>>(out, 1.f);
        if (op == 15) k<15><<<grid, 256>>>(out, 1.f);
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
