// What does a scattered run of R 4-byte words cost by its alignment?  The write pattern of the voxel sort's first pass
// (r3d_voxel.hip, voxel_bin_kernel): G workgroups, each owning a segment per bin (256 bins), walk T tiles of 4096 words and
// append, per tile and bin, a run of words at the bin's cursor.  Variants: runs of 16 words starting anywhere (what ranking a
// tile gives), runs of 16 on 64-byte boundaries, whole 128-byte lines (32 words to half the bins per tile), 32-byte sectors.
// hipcc --offload-arch=gfx950 -O3 tools/scatter_runs.hip -o tools/scatter_runs
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// run: words per (tile, bin) run; skew: first cursor of a segment (words); half: only bins with (bin + tile) even get a run (of 2 x run)
template <int RUN, bool HALF>
__global__ __launch_bounds__(256) void k(uint32_t* __restrict__ out, int tiles, int n_groups, int cap, int skew_mask) {
  const int g = blockIdx.x;
  for (int t = 0; t < tiles; ++t) {
#pragma unroll 4
    for (int j = threadIdx.x; j < 4096; j += 256) {
      int d, at;
      if (HALF) {
        const int k2 = j / (2 * RUN);            // 0..127: which of the tile's active bins
        d = 2 * k2 + (t & 1);
        at = (t >> 1) * 2 * RUN + j % (2 * RUN);
      } else {
        d = j / RUN;
        at = t * RUN + j % RUN;
      }
      const uint32_t skew = ((uint32_t)(d * 2654435761u + g * 40503u) >> 7) & skew_mask;
      out[((size_t)d * n_groups + g) * cap + skew + at] = j;
    }
  }
}

int main() {
  const int n_groups = 1200, tiles = 10, cap = 640;
  const size_t words = (size_t)256 * n_groups * cap;
  uint32_t* out;
  CK(hipMalloc(&out, words * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"runs of 16 words, any start", "runs of 16 words on 64 B", "whole 128 B lines (32 words, half the bins)", "runs of 16, start on 32 B", "runs of 8 words on 32 B... (16 per bin as 2 tiles)"};
  for (int v = 0; v < 4; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(out, 0, words * 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (v == 0) k<16, false><<<n_groups, 256>>>(out, tiles, n_groups, cap, 31);
      if (v == 1) k<16, false><<<n_groups, 256>>>(out, tiles, n_groups, cap, 0);
      if (v == 2) k<16, true><<<n_groups, 256>>>(out, tiles, n_groups, cap, 0);
      if (v == 3) k<16, false><<<n_groups, 256>>>(out, tiles, n_groups, cap, 8);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    const double bytes = (double)n_groups * tiles * 4096 * 4;
    printf("%-48s %7.1f us  %6.0f GB/s of payload\n", names[v], best * 1e3, bytes / best / 1e6);
  }
  return 0;
}
