// Rigid point-to-plane ICP for gfx950 (MI355X): the relative pose of two PARTIALLY OVERLAPPING single-view depth
// clouds -- the reference's own use of ICP ("match the point clouds corresponding to two images", readme.md:25; the
// consumer merges ./point/0.txt with ./point/24.txt, other_tools/transfer_T_icp.py:107-108).  NOT IN THE REFERENCE
// (it did this step by hand in CloudCompare, readme.md:54): build-defined, specified in include/r3d.h.
//
//   normals_kernel         normals of an ORGANISED cloud (the H x W raster order gentxtcord emits, p2c:34-44): central
//                          differences of the four raster neighbours, fp64 cross product; raster borders, missing depth
//                          and depth jumps give the zero vector = "no plane here" (HBM-bound: 12 B read + 12 B written)
//   plane_residual_kernel  r^2 = (n . (p - q))^2 per matched pair, +inf for pairs that cannot take part
//   select_*               exact order statistic of an fp32 array per class: four passes of 8-bit digits (LDS counts + a pick launch), all on
//                          the GPU (rank trimming needs the q-quantile of r^2 every iteration; no D2H, no host sort)
//   plane_accumulate_kernel the 29 fp64 sums of the linearised normal equations over the kept pairs: per-lane
//                          accumulators -> wave shuffle tree -> LDS across waves -> one row per workgroup
//   plane_finish_kernel    ONE workgroup: rows in fixed order -> 29 sums -> Cholesky solve + exponential map -> the ICP
//                          state (T_total, T_step, history) in HBM, so that whole iterations run without the host
// No float atomics anywhere: bitwise repeatable run to run.
#include <cmath>

#include "r3d_icp_sums.h"
#include "r3d_internal.h"
#include "r3d_plane_sums.h"

namespace {

constexpr int kThreads = 256;
using r3d_plane::kSums;

struct __attribute__((packed, aligned(4))) P3 {
  float x, y, z;
};

// ---- normals of an organised cloud -------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void normals_kernel(const float* __restrict__ xyz, int64_t n_frames, int h, int w,
                                                           float max_jump, double vx, double vy, double vz,
                                                           float* __restrict__ out) {
  const int64_t per = (int64_t)h * w;
  const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (k >= n_frames * per) return;
  const int64_t in_frame = k % per;
  const int j = (int)(in_frame / w), i = (int)(in_frame % w);
  P3 o = {0.f, 0.f, 0.f};
  if (j > 0 && j < h - 1 && i > 0 && i < w - 1) {
    const P3* P = reinterpret_cast<const P3*>(xyz);
    const P3 c = P[k], l = P[k - 1], r = P[k + 1], u = P[k - w], d = P[k + w];
    // the raster neighbours must be the same surface: every one present, finite and no farther from the centre's range
    // than max_jump x that range (range = distance from the viewpoint along the ray, in fp64)
    const double cx = (double)c.x - vx, cy = (double)c.y - vy, cz = (double)c.z - vz;
    const double rc = sqrt(cx * cx + cy * cy + cz * cz);
    bool ok = isfinite(rc) && rc > 0.0;
    const P3 nb[4] = {l, r, u, d};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double ex = (double)nb[m].x - vx, ey = (double)nb[m].y - vy, ez = (double)nb[m].z - vz;
      const double rn = sqrt(ex * ex + ey * ey + ez * ez);
      ok = ok && isfinite(rn) && rn > 0.0 && fabs(rn - rc) <= (double)max_jump * rc;
    }
    if (ok) {
      const double ax = (double)r.x - (double)l.x, ay = (double)r.y - (double)l.y, az = (double)r.z - (double)l.z;
      const double bx = (double)d.x - (double)u.x, by = (double)d.y - (double)u.y, bz = (double)d.z - (double)u.z;
      double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
      const double len = sqrt(nx * nx + ny * ny + nz * nz);
      if (len > 0.0 && isfinite(len)) {
        nx /= len;
        ny /= len;
        nz /= len;
        if (nx * cx + ny * cy + nz * cz > 0.0) {   // towards the viewpoint
          nx = -nx;
          ny = -ny;
          nz = -nz;
        }
        o.x = (float)nx;
        o.y = (float)ny;
        o.z = (float)nz;
      }
    }
  }
  reinterpret_cast<P3*>(out)[k] = o;
}

// ---- exact order statistics of an fp32 array, per bucket ---------------------------------------------------------
// Every element belongs to one of n_buckets classes (bucket[i]; NULL = contiguous blocks of per_class elements, or one
// class).  Per class: the element of rank floor(q (m - 1)) among its m finite values.  Four passes over the keys' 8-bit
// digits, most significant first; per pass a COUNT launch -- every workgroup counts its elements into an LDS histogram of all
// classes (32 x 256 counters) and adds its non-zero counters to the global one -- and a PICK launch: one workgroup, a wave
// per class, four bins per lane, one shuffle scan; it narrows the prefixes and clears the histogram for the next pass.
// Per pass on a 300k-value array of 24 classes (rocprofv3, tools/plane_once.py), in the order they were tried in round 3:
//   11 / 11 / 10-bit digits, counters straight in HBM (every wave of the grid wants the same few words) + pick   37.8 + 5.2 us  (x 3)
//   8-bit digits in LDS, the workgroup that finishes LAST (fence + ticket) picks: one launch per pass              21.2 us       (x 4)
//   all four passes in one launch, the last arriver picks and publishes, the others poll (bounded)                 104 - 127 us in all
//   8-bit digits in LDS + a pick launch: no fence anywhere                                                         7.4 + 6.4 us  (x 4)  <- this
//   the same, the pick of pass p done by EVERY workgroup at the head of pass p + 1 (a histogram and state per pass)  11.2 us (x 4) + 7.0
//     (51.8 against 55.2 us per selection: the redundant pick sits on every workgroup's critical path for nearly as long as the
//      launch it replaces; not worth a histogram per pass -- dropped)
// An agent-scope release / acquire pair inside a kernel has to make the eight XCDs' L2 caches agree; a kernel boundary does
// that anyway.  The "last workgroup done" idiom costs more than the launch it saves on this chip.
constexpr int kBins = 256;
constexpr int kSelectPasses = 4;
constexpr int kMaxBuckets = 32;
// per class: state[4] = {rank still to go, key prefix, prefix mask, finite count}
struct SelectOut {
  float value;
  unsigned count;
};

__device__ __forceinline__ unsigned order_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u >> 31) ? ~u : (u | 0x80000000u);
}

// counts this workgroup's elements of one pass into the LDS histogram `local` (zeroed, n_buckets x 256)
__device__ __forceinline__ void select_count(const float* __restrict__ v, const unsigned char* __restrict__ bucket, int64_t per_class,
                                             int64_t n, int n_buckets, int shift, const unsigned* s_prefix, const unsigned* s_mask,
                                             unsigned* local) {
  const int lane = threadIdx.x & 63;
  // whole waves stay in the loop together (the aggregation below votes across the wave); four elements per lane are
  // requested before the first is counted
  constexpr int kAhead = 4;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i0 = (int64_t)blockIdx.x * kThreads + (threadIdx.x & ~63); i0 < n; i0 += stride * kAhead) {
    float f[kAhead];
    int cls[kAhead];
#pragma unroll
    for (int a = 0; a < kAhead; ++a) {
      const int64_t i = i0 + a * stride + lane;
      const int64_t ic = i < n ? i : n - 1;   // clamped: unconditional loads
      f[a] = v[ic];
      // class of element i: its byte in `bucket`, or the contiguous block of per_class elements it lies in, or the only one
      cls[a] = bucket ? (int)bucket[ic] : per_class > 0 ? (int)(ic / per_class) : 0;
    }
#pragma unroll
    for (int a = 0; a < kAhead; ++a) {
      const int64_t i = i0 + a * stride + lane;
      bool live = i < n && (__float_as_uint(f[a]) & 0x7f800000u) != 0x7f800000u && cls[a] < n_buckets;   // inf / NaN never count
      unsigned slot = 0;
      if (live) {
        const unsigned key = order_key(f[a]);
        live = (key & s_mask[cls[a]]) == s_prefix[cls[a]];
        slot = (unsigned)cls[a] * kBins + ((key >> shift) & (kBins - 1u));
      }
      // Wave-aggregated: the residuals of one wall share their leading digits, so a wave's 64 lanes mostly want the same few
      // counters -- one LDS atomic per DISTINCT counter (up to four rounds), plain atomics for what is left
      unsigned long long todo = __ballot(live);
      for (int round = 0; round < 4 && todo; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned want = __shfl(slot, leader, 64);
        const unsigned long long same = __ballot(live && slot == want);
        if (lane == leader) atomicAdd(&local[want], (unsigned)__popcll(same));
        if (live && slot == want) live = false;
        todo &= ~same;
      }
      if (live) atomicAdd(&local[slot], 1u);
    }
  }
}

// One workgroup, every class (a wave per class, four bins per lane, one shuffle scan): the digit that holds the wanted rank,
// the narrowed prefix, after the last pass the value.  Everything it reads was written by OTHER workgroups (this launch or the
// previous one): agent-scope loads, past this CU's L1.  clear: zero the histogram afterwards (it is reused by the next pass).
__device__ __forceinline__ void select_pick_all(unsigned* hist, unsigned long long* state, SelectOut* __restrict__ out_all,
                                                int n_buckets, int pass, double q, bool clear) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int shift = 8 * (kSelectPasses - 1 - pass);
  const int last = pass == kSelectPasses - 1;
  // every class of this wave is requested before the first is scanned (a wave has up to eight classes: eight dependent L2
  // round trips otherwise)
  constexpr int kWaves = kThreads / 64, kPerWave = kMaxBuckets / kWaves;
  unsigned cnt_all[kPerWave][4];
  unsigned long long st_all[kPerWave][4];
#pragma unroll
  for (int j = 0; j < kPerWave; ++j) {
    const int c = wave + j * kWaves;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      cnt_all[j][m] = c < n_buckets ? __hip_atomic_load(&hist[(size_t)c * kBins + lane * 4 + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      st_all[j][m] = c < n_buckets ? __hip_atomic_load(&state[(size_t)c * 4 + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
  }
#pragma unroll
  for (int j = 0; j < kPerWave; ++j) {
    const int c = wave + j * kWaves;
    if (c >= n_buckets) break;   // wave-uniform
    unsigned* h = hist + (size_t)c * kBins;
    unsigned long long* st = state + (size_t)c * 4;
    unsigned cnt[4];
    unsigned mine = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      cnt[m] = cnt_all[j][m];
      mine += cnt[m];
    }
    unsigned inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    const unsigned long long total = __shfl(inc, 63, 64);
    unsigned long long k;
    if (pass == 0) {
      // rank of the q-quantile among the finite values, "lower" rule: floor(q (m - 1))
      double r = floor(q * (double)(total > 0 ? total - 1 : 0));
      if (!(r >= 0.0)) r = 0.0;
      k = (unsigned long long)r;
      if (total > 0 && k > total - 1) k = total - 1;
    } else {
      k = st_all[j][0];
    }
    const unsigned long long old_prefix = pass ? st_all[j][1] : 0ull, old_mask = pass ? st_all[j][2] : 0ull, old_count = st_all[j][3];
    if (total > 0) {
      unsigned long long before = inc - mine;
      if (k >= before && k < before + mine) {   // exactly one lane
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (k < before + cnt[m]) {
            const unsigned bin = (unsigned)(lane * 4 + m);
            const unsigned prefix = (unsigned)old_prefix | (bin << shift);
            st[0] = k - before;
            st[1] = prefix;
            st[2] = (unsigned)old_mask | ((kBins - 1u) << shift);
            if (pass == 0) st[3] = total;
            if (last) {
              const unsigned u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
              out_all[c].value = __uint_as_float(u);
              out_all[c].count = (unsigned)(pass == 0 ? total : old_count);
            }
            break;
          }
          before += cnt[m];
        }
      }
    } else if (lane == 0) {
      if (pass == 0) {
        st[0] = 0;
        st[3] = 0;
      }
      if (last) {
        out_all[c].value = INFINITY;   // no finite value in this class: nothing passes a "<= gate" test anyway
        out_all[c].count = 0;
      }
    }
    if (clear) {
#pragma unroll
      for (int m = 0; m < 4; ++m) h[lane * 4 + m] = 0;
    }
  }
}

__global__ __launch_bounds__(kThreads) void select_pass_kernel(const float* __restrict__ v, const unsigned char* __restrict__ bucket,
                                                               int64_t per_class, int64_t n, int n_buckets, int pass,
                                                               unsigned* hist, const unsigned long long* __restrict__ state) {
  __shared__ unsigned local[kMaxBuckets * kBins];
  __shared__ unsigned s_prefix[kMaxBuckets], s_mask[kMaxBuckets];
  const int n_slots = n_buckets * kBins;
  for (int b = threadIdx.x; b < n_slots; b += kThreads) local[b] = 0;
  if (threadIdx.x < n_buckets) {   // (pass 0 starts from nothing: the state words still hold the previous selection's)
    s_prefix[threadIdx.x] = pass ? (unsigned)state[threadIdx.x * 4 + 1] : 0u;
    s_mask[threadIdx.x] = pass ? (unsigned)state[threadIdx.x * 4 + 2] : 0u;
  }
  __syncthreads();
  select_count(v, bucket, per_class, n, n_buckets, 8 * (kSelectPasses - 1 - pass), s_prefix, s_mask, local);
  __syncthreads();
  for (int b = threadIdx.x; b < n_slots; b += kThreads)
    if (local[b]) atomicAdd(&hist[b], local[b]);
}

// one workgroup: every class's digit from the counters the pass kernel left (the kernel boundary publishes them)
__global__ __launch_bounds__(kThreads) void select_pick_kernel(unsigned* hist, unsigned long long* state, SelectOut* __restrict__ out_all,
                                                               int n_buckets, int pass, double q) {
  select_pick_all(hist, state, out_all, n_buckets, pass, q, true);
}

// ---- point-to-plane pairs ----------------------------------------------------------------------------------
// Direction class of a target normal (f32 components as stored): which axis is largest in magnitude (the lowest on ties),
// its sign, and the signs of the other two in cyclic order: 24 classes.  Rank statistics of the residual are taken PER
// CLASS, so that a wall whose pairs all disagree with the current pose keeps its say against the walls that already fit --
// trimming over all pairs together drops exactly the family that carries the missing constraint and the pose slides.
constexpr int kNormalClasses = 24;
__device__ __forceinline__ int normal_class(float nx, float ny, float nz) {
  const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
  int major = 0;
  float n0 = nx, n1 = ny, n2 = nz;
  if (ay > ax && ay >= az) {
    major = 1; n0 = ny; n1 = nz; n2 = nx;
  } else if (az > ax && az > ay) {
    major = 2; n0 = nz; n1 = nx; n2 = ny;
  }
  return major * 8 + (n0 < 0.f ? 4 : 0) + (n1 < 0.f ? 2 : 0) + (n2 < 0.f ? 1 : 0);
}

// Can pair (source i, target j) take part at all?  Finite source point, a target with a plane (non-zero normal), inside
// the point-to-point gate.  Returns r = n . (p - q) through *r_out and the normal's direction class through *cls.
__device__ __forceinline__ bool plane_pair(const float* __restrict__ src, const float* __restrict__ tgt,
                                           const float* __restrict__ nrm, int64_t n_tgt, const uint32_t* __restrict__ idx,
                                           const float* __restrict__ d2, float max_d2, int64_t i, double p[3], double n[3],
                                           double* r_out, int* cls) {
  const int64_t j = (int64_t)idx[i];
  if (j >= n_tgt) return false;                                  // a caller's index array is data
  if (max_d2 >= 0.f && !(d2[i] <= max_d2)) return false;
  const P3 ps = reinterpret_cast<const P3*>(src)[i], qs = reinterpret_cast<const P3*>(tgt)[j],
           ns = reinterpret_cast<const P3*>(nrm)[j];
  p[0] = (double)ps.x; p[1] = (double)ps.y; p[2] = (double)ps.z;
  n[0] = (double)ns.x; n[1] = (double)ns.y; n[2] = (double)ns.z;
  const double q[3] = {(double)qs.x, (double)qs.y, (double)qs.z};
  if (ns.x == 0.f && ns.y == 0.f && ns.z == 0.f) return false;   // no plane at this target point
  const double r = r3d_plane::plane_residual(p, q, n);
  if (!isfinite(r) || !isfinite((p[0] + p[1]) + p[2])) return false;
  *r_out = r;
  *cls = normal_class(ns.x, ns.y, ns.z);
  return true;
}

__global__ __launch_bounds__(kThreads) void plane_residual_kernel(const float* __restrict__ src, int64_t n_src,
                                                                  const float* __restrict__ tgt, const float* __restrict__ nrm,
                                                                  int64_t n_tgt, const uint32_t* __restrict__ idx,
                                                                  const float* __restrict__ d2, float max_d2,
                                                                  float* __restrict__ r2_out, unsigned char* __restrict__ cls_out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_src) return;
  double p[3], n[3], r = 0.0;
  int cls = 0;
  const bool ok = plane_pair(src, tgt, nrm, n_tgt, idx, d2, max_d2, i, p, n, &r, &cls);
  r2_out[i] = ok ? (float)(r * r) : INFINITY;
  if (cls_out) cls_out[i] = ok ? (unsigned char)cls : (unsigned char)255;
}

// gates: per direction class the order statistic of (float)(r r); a pair takes part when its value is <= gate x gate_scale
// (one fp32 multiply).  gates == NULL: every admissible pair takes part.
__global__ __launch_bounds__(kThreads) void plane_accumulate_kernel(const float* __restrict__ src, int64_t n_src,
                                                                    const float* __restrict__ tgt,
                                                                    const float* __restrict__ nrm, int64_t n_tgt,
                                                                    const uint32_t* __restrict__ idx,
                                                                    const float* __restrict__ d2, float max_d2,
                                                                    const SelectOut* __restrict__ gates, float gate_scale,
                                                                    double* __restrict__ partials) {
  __shared__ double red[kThreads / 64][kSums];
  __shared__ float s_gate[kNormalClasses];
  if (threadIdx.x < kNormalClasses) s_gate[threadIdx.x] = gates ? gates[threadIdx.x].value * gate_scale : INFINITY;
  __syncthreads();
  double acc[kSums];
#pragma unroll
  for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_src; i += (int64_t)gridDim.x * kThreads) {
    double p[3], n[3], r = 0.0;
    int cls = 0;
    if (!plane_pair(src, tgt, nrm, n_tgt, idx, d2, max_d2, i, p, n, &r, &cls)) continue;
    if (!((float)(r * r) <= s_gate[cls])) continue;   // the same fp32 value the selection ranked
    r3d_plane::pair_accumulate(acc, 1.0, p, n, r);
  }
  r3d_plane::block_reduce_store(acc, red, partials + (int64_t)blockIdx.x * kSums);
}

// One thread: step from the sums; T_total <- step . T_total; history (same state layout as the similarity loop).
__device__ void plane_solve_step(const double* s, double* __restrict__ st) {
  double T[16], rms = 0.0;
  const int bad = r3d_plane::step_from_sums(s, T, &rms);
  double tot[16], nt[16];
  for (int k = 0; k < 16; ++k) tot[k] = st[r3d_icp::kStateTTotal + k];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double v = 0.0;
      for (int m = 0; m < 4; ++m) v += T[4 * r + m] * tot[4 * m + c];
      nt[4 * r + c] = v;
    }
  for (int k = 0; k < 16; ++k) {
    st[r3d_icp::kStateTStep + k] = T[k];
    st[r3d_icp::kStateTTotal + k] = nt[k];
  }
  const int it = (int)st[r3d_icp::kStateIters];
  if (r3d_icp::kStateHistory + it < r3d_icp::kStateDoubles) st[r3d_icp::kStateHistory + it] = rms;
  st[r3d_icp::kStateIters] = (double)(it + 1);
  if (bad) st[r3d_icp::kStateStatus] = 1.0;
  st[r3d_icp::kStateRms] = rms;
  st[r3d_icp::kStatePairs] = s[0];
}

__global__ __launch_bounds__(kThreads) void plane_finish_kernel(const double* __restrict__ partials, int n_rows,
                                                                double* __restrict__ sums_out, double* __restrict__ state) {
  __shared__ double red[kThreads / 64][kSums];
  __shared__ double total[kSums];
  double acc[kSums];
#pragma unroll
  for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < n_rows; b += kThreads) {
#pragma unroll
    for (int k = 0; k < kSums; ++k) acc[k] += partials[(int64_t)b * kSums + k];
  }
  r3d_plane::block_reduce_store(acc, red, total);
  __syncthreads();
  if (threadIdx.x < kSums && sums_out) sums_out[threadIdx.x] = total[threadIdx.x];
  if (threadIdx.x == 0 && state != nullptr) {
    double s[kSums];
    for (int k = 0; k < kSums; ++k) s[k] = total[k];
    plane_solve_step(s, state);
  }
}

// workspace in scratch slot 6: [hist n_buckets x 256 u32][state n_buckets x 4 u64][SelectOut x n_buckets][pad], then the
// partial rows
struct Workspace {
  unsigned* hist;
  unsigned long long* state;
  SelectOut* out;
  double* rows;
  size_t clear_bytes;   // hist + state
};

int workspace(r3d_ctx* ctx, int n_buckets, int n_rows, Workspace* ws) {
  void* p = nullptr;
  const size_t hist_b = (size_t)n_buckets * kBins * sizeof(unsigned), state_b = (size_t)n_buckets * 4 * sizeof(unsigned long long);
  const size_t head = hist_b + state_b + (size_t)n_buckets * sizeof(SelectOut) + 64;
  const size_t head_al = (head + 255) & ~(size_t)255;
  int rc = r3d_scratch(ctx, 6, head_al + (size_t)(n_rows + 1) * kSums * sizeof(double), &p);
  if (rc) return rc;
  // The "histogram is known to be zero" record (select_enqueue) describes ONE layout.  A workspace laid out for another class
  // count puts its state words, picks or partial rows where that layout has its histogram -- e.g. the untrimmed sums pass
  // (1 class) writes fp64 rows at byte 1280, inside the 24-class histogram -- so the record dies here, whether or not a
  // selection follows (round-3 advisor finding: trim -> no trim -> trim on one ctx ranked against leftover rows).
  if (ctx->select_ws_buckets != n_buckets) ctx->select_ws = nullptr;
  char* c = static_cast<char*>(p);
  ws->hist = reinterpret_cast<unsigned*>(c);
  ws->state = reinterpret_cast<unsigned long long*>(c + hist_b);
  ws->out = reinterpret_cast<SelectOut*>(c + hist_b + state_b);
  ws->rows = reinterpret_cast<double*>(c + head_al);
  ws->clear_bytes = hist_b + state_b;
  return R3D_OK;
}

// d_bucket != NULL: class per element; else per_class > 0: classes are contiguous blocks of that many elements; else one class
int select_enqueue(r3d_ctx* ctx, const float* d_values, const unsigned char* d_bucket, int n_buckets, int64_t n, double q,
                   const Workspace& ws, int64_t per_class = 0) {
  hipStream_t st = ctx->stream;
  // The histogram has to be all zero in front of pass 0.  Every pick leaves it so, and pass 0 ignores the state words, so only a
  // workspace that has not been through a selection of this shape yet is cleared (a 5 us launch per ICP iteration otherwise).
  if (ctx->select_ws != ws.hist || ctx->select_ws_buckets != n_buckets) {
    R3D_HIP(hipMemsetAsync(ws.hist, 0, ws.clear_bytes, st));
    ctx->select_ws = ws.hist;
    ctx->select_ws_buckets = n_buckets;
  }
  int blocks = (int)std::min<int64_t>((n + kThreads * 8 - 1) / (kThreads * 8), (int64_t)ctx->num_cus);
  if (blocks < 1) blocks = 1;
  for (int pass = 0; pass < kSelectPasses; ++pass) {
    hipLaunchKernelGGL(select_pass_kernel, dim3(blocks), dim3(kThreads), 0, st, d_values, d_bucket, per_class, n, n_buckets, pass,
                       ws.hist, (const unsigned long long*)ws.state);
    hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(kThreads), 0, st, ws.hist, ws.state, ws.out, n_buckets, pass, q);
  }
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

// one workgroup per class (a contiguous block of per_class values): sum and count of the finite values <= the class's
// selected order statistic, fp64, fixed order
__global__ __launch_bounds__(kThreads) void class_sum_below_kernel(const float* __restrict__ v, int64_t per_class, int64_t n,
                                                                   const SelectOut* __restrict__ gates, double* __restrict__ out2) {
  __shared__ double red[2][kThreads / 64];
  const int c = blockIdx.x;
  const float g = gates[c].value;
  const int64_t lo = (int64_t)c * per_class, hi = lo + per_class < n ? lo + per_class : n;
  double sum = 0.0, cnt = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
    const float f = v[i];
    if ((__float_as_uint(f) & 0x7f800000u) == 0x7f800000u) continue;
    if (f <= g) {
      sum += (double)f;
      cnt += 1.0;
    }
  }
  sum = r3d_plane::wave_sum(sum);
  cnt = r3d_plane::wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sum;
    red[1][threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0, k = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) {
      s += red[0][w];
      k += red[1][w];
    }
    out2[2 * c] = s;
    out2[2 * c + 1] = k;
  }
}

// one workgroup per CU: every workgroup pays a 29-value fp64 tree reduction and leaves a row for plane_finish_kernel to read
// (four per CU: accumulate 20 us + finish 19 us per iteration on a 307k-pair cloud, mostly those reductions)
int accumulate_blocks(r3d_ctx* ctx, int64_t n_src) {
  int blocks = (int)std::min<int64_t>((n_src + kThreads - 1) / kThreads, (int64_t)ctx->num_cus);
  return blocks < 1 ? 1 : blocks;
}

}  // namespace

extern "C" {

int r3d_normals_organized(r3d_ctx* ctx, const float* d_xyz, int64_t n_frames, int height, int width, float max_jump,
                          const double* h_viewpoint, float* d_normals_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_frames >= 0 && height >= 0 && width >= 0, "negative raster size");
  R3D_REQUIRE(max_jump >= 0.f, "max_jump must be >= 0");
  const int64_t n = n_frames * (int64_t)height * width;
  if (n == 0) return R3D_OK;
  R3D_REQUIRE(d_xyz && d_normals_out, "NULL device pointer");
  R3D_REQUIRE(d_xyz != d_normals_out, "normals cannot be written over the cloud they are taken from");
  const double v[3] = {h_viewpoint ? h_viewpoint[0] : 0.0, h_viewpoint ? h_viewpoint[1] : 0.0, h_viewpoint ? h_viewpoint[2] : 0.0};
  hipLaunchKernelGGL(normals_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d_xyz,
                     n_frames, height, width, max_jump, v[0], v[1], v[2], d_normals_out);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_select_quantile_f32(r3d_ctx* ctx, const float* d_values, int64_t n, double q, float* h_value_out, int64_t* h_count_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n >= 0, "negative count");
  R3D_REQUIRE(q >= 0.0 && q <= 1.0, "q must be in [0, 1]");
  R3D_REQUIRE(n == 0 || d_values != nullptr, "NULL device pointer");
  Workspace ws;
  if ((rc = workspace(ctx, 1, 0, &ws))) return rc;
  if ((rc = select_enqueue(ctx, d_values, nullptr, 1, n, q, ws))) return rc;
  SelectOut o;
  R3D_HIP(hipMemcpyAsync(&o, ws.out, sizeof(o), hipMemcpyDeviceToHost, ctx->stream));
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  if (h_value_out) *h_value_out = o.value;
  if (h_count_out) *h_count_out = (int64_t)o.count;
  return R3D_OK;
}

int r3d_select_quantile_f32_dev(r3d_ctx* ctx, const float* d_values, int64_t n, double q, void* d_out8) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n >= 0, "negative count");
  R3D_REQUIRE(q >= 0.0 && q <= 1.0, "q must be in [0, 1]");
  R3D_REQUIRE((n == 0 || d_values != nullptr) && d_out8 != nullptr, "NULL device pointer");
  Workspace ws;
  if ((rc = workspace(ctx, 1, 0, &ws))) return rc;
  if ((rc = select_enqueue(ctx, d_values, nullptr, 1, n, q, ws))) return rc;
  R3D_HIP(hipMemcpyAsync(d_out8, ws.out, sizeof(SelectOut), hipMemcpyDeviceToDevice, ctx->stream));
  return R3D_OK;
}

int r3d_trimmed_means_f32(r3d_ctx* ctx, const float* d_values, int n_classes, int64_t per_class, double keep, double* h_means_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_classes >= 1 && n_classes <= kMaxBuckets, "1..%d classes", kMaxBuckets);
  R3D_REQUIRE(per_class >= 1, "per_class must be >= 1");
  R3D_REQUIRE(keep >= 0.0 && keep <= 1.0, "keep must be in [0, 1]");
  R3D_REQUIRE(d_values && h_means_out, "NULL argument");
  const int64_t n = (int64_t)n_classes * per_class;
  Workspace ws;
  if ((rc = workspace(ctx, n_classes, 2 * n_classes / kSums + 2, &ws))) return rc;
  if ((rc = select_enqueue(ctx, d_values, nullptr, n_classes, n, keep, ws, per_class))) return rc;
  hipLaunchKernelGGL(class_sum_below_kernel, dim3(n_classes), dim3(kThreads), 0, ctx->stream, d_values, per_class, n,
                     (const SelectOut*)ws.out, ws.rows);
  R3D_HIP(hipGetLastError());
  double pairs[2 * kMaxBuckets];
  R3D_HIP(hipMemcpyAsync(pairs, ws.rows, (size_t)n_classes * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  for (int c = 0; c < n_classes; ++c) h_means_out[c] = pairs[2 * c + 1] > 0.0 ? pairs[2 * c] / pairs[2 * c + 1] : INFINITY;
  return R3D_OK;
}

int r3d_icp_plane_residuals(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, const float* d_tgt_normals,
                            int64_t n_tgt, const uint32_t* d_idx, const float* d_d2, float max_d2, float* d_r2_out,
                            unsigned char* d_class_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_src >= 0 && n_tgt >= 0, "negative cloud size");
  if (n_src == 0) return R3D_OK;
  R3D_REQUIRE(d_src && d_tgt && d_tgt_normals && d_idx && d_r2_out, "NULL device pointer");
  R3D_REQUIRE(!(max_d2 >= 0.f) || d_d2 != nullptr, "max_d2 >= 0 needs the d2 array");
  hipLaunchKernelGGL(plane_residual_kernel, dim3((unsigned)((n_src + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                     d_src, n_src, d_tgt, d_tgt_normals, n_tgt, d_idx, d_d2, max_d2 >= 0.f ? max_d2 : -1.f, d_r2_out, d_class_out);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

// 0 < trim_q < 1: per direction class only the pairs up to gate_scale x that class's trim_q order statistic of r^2
static int plane_sums_impl(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, const float* d_nrm, int64_t n_tgt,
                           const uint32_t* d_idx, const float* d_d2, float max_d2, float trim_q, float gate_scale,
                           double* d_sums_out, double* d_state) {
  const int blocks = accumulate_blocks(ctx, n_src);
  const bool trim = trim_q > 0.f && trim_q < 1.f;
  Workspace ws;
  int rc = workspace(ctx, trim ? kNormalClasses : 1, blocks, &ws);
  if (rc) return rc;
  const float gate_d2 = max_d2 >= 0.f ? max_d2 : -1.f;
  const SelectOut* gates = nullptr;
  if (trim) {
    void* r2 = nullptr;   // [n_src] f32 followed by [n_src] u8
    if ((rc = r3d_scratch(ctx, 7, (size_t)n_src * 5 + 16, &r2))) return rc;
    unsigned char* cls = reinterpret_cast<unsigned char*>(static_cast<float*>(r2) + n_src);
    if ((rc = r3d_icp_plane_residuals(ctx, d_src, n_src, d_tgt, d_nrm, n_tgt, d_idx, d_d2, gate_d2, (float*)r2, cls))) return rc;
    if ((rc = select_enqueue(ctx, (const float*)r2, cls, kNormalClasses, n_src, (double)trim_q, ws))) return rc;
    gates = ws.out;
  }
  hipLaunchKernelGGL(plane_accumulate_kernel, dim3(blocks), dim3(kThreads), 0, ctx->stream, d_src, n_src, d_tgt, d_nrm, n_tgt,
                     d_idx, d_d2, gate_d2, gates, gate_scale, ws.rows);
  hipLaunchKernelGGL(plane_finish_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double*)ws.rows, blocks, d_sums_out,
                     d_state);
  R3D_HIP(hipGetLastError());
  return R3D_OK;
}

int r3d_icp_plane_accumulate(r3d_ctx* ctx, const float* d_src, int64_t n_src, const float* d_tgt, const float* d_tgt_normals,
                             int64_t n_tgt, const uint32_t* d_idx, const float* d_d2, float max_d2, float trim_q,
                             float gate_scale, double* d_sums_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(n_src >= 0 && n_tgt >= 0, "negative cloud size");
  R3D_REQUIRE(d_sums_out != nullptr, "d_sums_out is NULL");
  R3D_REQUIRE(gate_scale > 0.f, "gate_scale must be positive");
  if (n_src == 0) {
    R3D_HIP(hipMemsetAsync(d_sums_out, 0, kSums * sizeof(double), ctx->stream));
    return R3D_OK;
  }
  R3D_REQUIRE(d_src && d_tgt && d_tgt_normals && d_idx, "NULL device pointer");
  R3D_REQUIRE(!(max_d2 >= 0.f) || d_d2 != nullptr, "max_d2 >= 0 needs the d2 array");
  return plane_sums_impl(ctx, d_src, n_src, d_tgt, d_tgt_normals, n_tgt, d_idx, d_d2, max_d2, trim_q, gate_scale, d_sums_out,
                         nullptr);
}

int r3d_plane_step_from_sums(const double* h_sums, double* h_T, double* h_rms_out) {
  R3D_REQUIRE(h_sums && h_T, "NULL argument");
  const int bad = r3d_plane::step_from_sums(h_sums, h_T, h_rms_out);
  if (bad) {
    r3d_set_error("point-to-plane step undefined: %g pairs (need >= 6) or the matched normals leave a freedom unconstrained",
                  h_sums[0]);
    return R3D_ERR_INVALID;
  }
  return R3D_OK;
}

int r3d_icp_iterate_plane(r3d_ctx* ctx, r3d_nn_index* index, const float* d_src_orig, float* d_src, int64_t n_src,
                          const float* d_tgt_normals, uint32_t* d_idx, float* d_d2, int n_iters, float trim_q, float gate_scale,
                          float max_d2, double* d_state) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(index != nullptr, "nn index is NULL");
  R3D_REQUIRE(n_iters >= 0, "n_iters must be >= 0");
  R3D_REQUIRE(n_src >= 6, "need at least 6 source points");
  R3D_REQUIRE(gate_scale > 0.f, "gate_scale must be positive");
  R3D_REQUIRE(d_src && d_tgt_normals && d_idx && d_d2 && d_state, "NULL device pointer");
  R3D_REQUIRE(d_src_orig != d_src, "d_src_orig must be a separate buffer (or NULL)");
  const float* d_tgt = nullptr;
  int64_t n_tgt = 0;
  r3d_ctx* ictx = nullptr;
  if ((rc = r3d_nn_index_target(index, &d_tgt, &n_tgt, &ictx))) return rc;
  R3D_REQUIRE(ictx == ctx, "the index belongs to another context");
  // the same loop going on (r3d_internal.h, r3d_ctx::loop_*): its first iteration starts from the previous matches too
  const bool going_on = ctx->loop_state == d_state && ctx->loop_src == d_src && ctx->loop_idx == d_idx && ctx->loop_index == index;
  // With the original cloud at hand every iteration moves IT by the accumulated pose: one rounding per point however many
  // steps were taken (moving the moved cloud again and again lets fp32 rounding drift by ~1e-7 per step).  That move is
  // idempotent -- d_src = T_total . d_src_orig -- so an iteration that runs the wave-local search lets the search kernel do it
  // (r3d_nn_index_query_step) and the move launch of the iteration before is left out; only the LAST iteration of a call
  // moves explicitly, so that the caller finds d_src where the state says it is.
  bool pending_move = false;   // the state has a pose that d_src does not show yet
  auto one_iteration = [&](bool warm, bool last) -> int {
    int rc2, moved = 0;
    // sources are kept in the index's Morton order by the caller (r3d_nn_index_sort_cloud; rigid moves preserve it)
    if (pending_move && !(warm && d_src_orig)) {   // (cannot happen: a pending move implies both; kept for safety)
      if ((rc2 = r3d_apply_T_dev(ctx, d_src_orig, R3D_F32, n_src, d_state + r3d_icp::kStateTTotal, d_src, R3D_F32))) return rc2;
      pending_move = false;
    }
    if ((rc2 = r3d_nn_index_query_step(index, d_src, n_src, d_idx, d_d2, warm, pending_move ? d_src_orig : nullptr,
                                       pending_move ? d_state + r3d_icp::kStateTTotal : nullptr, &moved)))
      return rc2;
    if (pending_move && !moved) {   // the query took the other kernel after all: it searched the unmoved cloud -- redo properly
      if ((rc2 = r3d_apply_T_dev(ctx, d_src_orig, R3D_F32, n_src, d_state + r3d_icp::kStateTTotal, d_src, R3D_F32))) return rc2;
      if ((rc2 = r3d_nn_index_query_step(index, d_src, n_src, d_idx, d_d2, warm, nullptr, nullptr, nullptr))) return rc2;
    }
    pending_move = false;
    if ((rc2 = plane_sums_impl(ctx, d_src, n_src, d_tgt, d_tgt_normals, n_tgt, d_idx, d_d2, max_d2, trim_q, gate_scale, nullptr,
                               d_state)))
      return rc2;
    if (d_src_orig && !last) {
      pending_move = true;   // the next iteration's search moves the cloud
      return R3D_OK;
    }
    if (d_src_orig) return r3d_apply_T_dev(ctx, d_src_orig, R3D_F32, n_src, d_state + r3d_icp::kStateTTotal, d_src, R3D_F32);
    return r3d_apply_T_dev(ctx, d_src, R3D_F32, n_src, d_state + r3d_icp::kStateTStep, d_src, R3D_F32);
  };
  // (Replaying one captured iteration as a hipGraph -- 14 launches per iteration, each dependent on the one before -- was
  // measured in round 3: 190-198 us per iteration against 184-186 us launch by launch, capture and instantiation included.  The
  // host is ahead of the GPU either way; what separates two dependent kernels is the GPU's own dispatch latency.)
  for (int it = 0; it < n_iters; ++it)
    if ((rc = one_iteration(it > 0 || going_on, it == n_iters - 1))) return rc;
  if (n_iters > 0) {
    ctx->loop_state = d_state;
    ctx->loop_src = d_src;
    ctx->loop_src_bytes = (size_t)n_src * 12;
    ctx->loop_idx = d_idx;
    ctx->loop_index = index;
  }
  return R3D_OK;
}

}  // extern "C"
