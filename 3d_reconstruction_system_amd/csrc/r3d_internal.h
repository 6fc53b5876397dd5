// Internal declarations shared by the translation units of libr3d_hip.so.
// Public surface: include/r3d.h.  gfx950 (MI355X) only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>

#include "r3d.h"
#include "r3d_internal_api.h"

struct r3d_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int num_cus = 256;
  // tuning knobs (r3d_ctx_set_tuning)
  int fuse_blocks = 0;   // 0 auto
  int fuse_prefetch = 0; // 0 auto, 1 off, 2 on: stage the inputs of a launch in the Infinity Cache with a read-only sweep first
  int fuse_chunk_mb = 0; // 0 auto: input bytes staged (and fused) per step when the prefetch is on
  // fuse_prefetch auto: stage a launch's small-share inputs when they exceed this many MB.  Round 2 had 64 ("less than that
  // is usually still in the cache from its producer"); measured in round 3 (bench.py --workload regimes): a 49 MB raster that
  // an H2D copy has just written is NOT in the Infinity Cache -- the launch runs at 0.49 of peak plain, 0.82 staged -- while
  // staging a raster that IS cached costs 4 %.  Below ~8 MB the extra launch eats the gain.
  int fuse_stage_auto_mb = 8;
  // ... and, since round 4, only when those inputs are NOT presumed to sit in the Infinity Cache already: the library keeps a
  // per-device record of the input ranges its launches have read lately (r3d_inputs_* below) and forgets a range the moment
  // it writes it from outside the cache (H2D copies, the host pipeline's uploads, collective receives, frees).  This many MB
  // of inputs are presumed to survive in the 256 MiB cache while a launch's write stream passes through it.
  int fuse_resident_mb = 128;
  int fuse_sweeps = 0;        // read-only statistic: staging sweeps the fused launches of this ctx have enqueued so far
  int fuse_inputs_fresh = 0;  // write-only knob: a foreign producer (torch, another library) has rewritten input buffers
  int nn_variant = 0;     // sources per lane (1, 2, 4; 0 = auto)
  // the ICP loop that ran last on this ctx (r3d_icp_iterate / _plane): a call with the same state, source, match buffer and
  // index and no state reset / write to that source through the library in between is the SAME loop going on -- its first
  // iteration may start from the previous matches like every other one (nn_warm_kernel)
  const void *loop_state = nullptr, *loop_src = nullptr, *loop_idx = nullptr, *loop_index = nullptr;
  size_t loop_src_bytes = 0;          // ... and the extent of that source cloud: r3d_wrote() ends the loop on ANY overlapping write
  const void* select_ws = nullptr;   // the selection workspace (r3d_plane.hip) whose histogram is known to be all zero ...
  int select_ws_buckets = 0;         // ... for this many classes
  int nn_warm = 0;        // 0 auto: repeated presorted queries start from the previous matches' distances, ICP loops use
                          // nn_warm_kernel from their second iteration on; 1 off; 2 / 3: never / always nn_warm_kernel (A/B)
  int apply_blocks = 0;
  int voxel_path = 0;     // big inserts: 0 auto (a sample of the cloud decides), 1 LDS-set + CAS kernel, 2 sort-merge (r3d_voxel.hip)
  int voxel_last_path = 0; // read-only: the path the last r3d_voxelset_insert took (1 / 2)
  int voxel_dedupe = 0;   // 0 auto (on), 1 off, 2 on: per-workgroup LDS dedupe in front of the global hash set; 3: on, with the
                          // flush barrier inside its `if` (A/B against DESIGN 4.5b's finding only)
  // HIP-event stopwatch
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  // grow-only scratch buffers for the *_host entry points and reductions
  static constexpr int kScratchSlots = 8;   // 6, 7: point-to-plane / selection workspace (r3d_plane.hip)
  void* scratch[kScratchSlots] = {};
  size_t scratch_bytes[kScratchSlots] = {};
  // pinned staging buffers of the host pipeline: slot = ((direction * kPipeBufs + array) * 2 + parity)
  static constexpr int kPipeBufs = 2;   // arrays per direction (depth + colour in, xyz + rgba out)
  static constexpr int kPinnedSlots = 2 * kPipeBufs * 2;
  void* pinned[kPinnedSlots] = {};
  size_t pinned_bytes[kPinnedSlots] = {};
  hipEvent_t ev_pipe[6] = {};           // host pipeline: done[2], uploaded[2], entry, spare
  hipStream_t upload_stream = nullptr;  // H2D side of the host pipeline (full-duplex PCIe)
};

struct r3d_camera {
  r3d_ctx* ctx = nullptr;
  int device = 0;  // for destroy, which must not dereference ctx
  int height = 0, width = 0;
  double fx = 0, fy = 0, cx = 0, cy = 0;
  double* d_u = nullptr;  // [width]  (i-cx)/fx
  double* d_v = nullptr;  // [height] (j-cy)/fy
};

// thread-local error text
void r3d_set_error(const char* fmt, ...);
int r3d_fail_hip(hipError_t e, const char* what, const char* file, int line);

#define R3D_HIP(call)                                                       \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) return r3d_fail_hip(e_, #call, __FILE__, __LINE__); \
  } while (0)

#define R3D_REQUIRE(cond, ...)  \
  do {                          \
    if (!(cond)) {              \
      r3d_set_error(__VA_ARGS__); \
      return R3D_ERR_INVALID;   \
    }                           \
  } while (0)

// make the ctx's device current for the calling thread
int r3d_ctx_enter(r3d_ctx* ctx);
// the context a communicator was created on (its collectives run on that context's stream); r3d_comm.hip
struct r3d_comm;
r3d_ctx* r3d_comm_context(const r3d_comm* comm);

// Which input ranges are presumed to sit in the device's Infinity Cache (r3d_ctx.hip; one record per device, shared by its
// contexts).  _resident: [p, p+bytes) was read by a launch of the library and fewer than `budget_bytes` of other inputs have
// gone through since.  _read: a launch has just been enqueued that reads the range.  _written: the range has been (or is
// about to be) written from outside the cache, or freed -- forget it.  _forget_all: the hint for foreign producers.
bool r3d_inputs_resident(r3d_ctx* ctx, const void* p, size_t bytes, size_t budget_bytes);
void r3d_inputs_read(r3d_ctx* ctx, const void* p, size_t bytes, size_t budget_bytes);
void r3d_inputs_written(r3d_ctx* ctx, const void* p, size_t bytes);
void r3d_inputs_forget_all(r3d_ctx* ctx);
int r3d_inputs_tracked(r3d_ctx* ctx);

// Every entry point that writes a CALLER's device range (or frees one) says so here: the range is no longer presumed cached
// (r3d_inputs_written) and an ICP loop whose source cloud it overlaps is over (its next call starts cold, not "warm" from
// matches that describe other points).  Results never depend on either record; speed does.
void r3d_wrote(r3d_ctx* ctx, const void* p, size_t bytes);

// grow-only scratch slot (device memory); returns device pointer in *p
int r3d_scratch(r3d_ctx* ctx, int slot, size_t bytes, void** p);

// Chunked, double-buffered host<->device pipeline (r3d_hostpipe.hip).  Items [lo, lo+n) of the batch are uploaded
// to d_in + lo*in_item_bytes, `launch(lo, n)` enqueues the kernel for them, and d_out + lo*out_item_bytes comes back.
int r3d_host_pipeline(r3d_ctx* ctx, int64_t n_items, size_t in_item_bytes, size_t out_item_bytes, const void* h_in,
                      void* h_out, void* d_in, void* d_out, const std::function<int(int64_t, int64_t)>& launch);
// The same with up to r3d_ctx::kPipeBufs arrays per direction travelling together (e.g. depth + colour in, xyz + rgba out).
struct r3d_pipe_buf {
  void* h;            // host array (pageable or pinned); inputs are only read
  void* d;            // whole-batch device array
  size_t item_bytes;  // bytes per item (frame)
};
int r3d_host_pipeline_multi(r3d_ctx* ctx, int64_t n_items, const r3d_pipe_buf* ins, int n_in, const r3d_pipe_buf* outs,
                            int n_out, const std::function<int(int64_t, int64_t)>& launch);

// r3d_voxel.hip: what a kernel outside that file needs to insert into a voxel set (r3d_fuse.hip's fused RGBD + voxel launch)
int r3d_voxelset_device_view(r3d_voxelset* vs, r3d_ctx** ctx, double* factor, uint64_t** d_table, int* log2cap,
                             unsigned long long** d_counters);

// the two insert paths of a voxel set, and how a big insert chooses between them (r3d_voxel.hip)
bool r3d_voxelset_sort_feasible(const r3d_voxelset* vs, int64_t n_points, bool forced);
int r3d_voxelset_sample(r3d_voxelset* vs, const float* d_xyz, int64_t n_points, int64_t n_insert, double cas_base_ps, bool* sort_out);
int r3d_voxelset_insert_path(r3d_voxelset* vs, const float* d_xyz, int64_t n_points, int path);

// Device -> pageable host memory through pinned staging chunks (r3d_hostpipe.hip); synchronous.
int r3d_download_pageable(r3d_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

// Stable LSD radix sort of 64-bit keys by their bits [first_bit, bits) (r3d_sort.hip); d_tmp holds n keys.
int r3d_radix_sort_u64(r3d_ctx* ctx, uint64_t* d_keys, uint64_t* d_tmp, int64_t n, int bits, int first_bit = 0,
                       uint64_t** d_result = nullptr, bool first_hist_done = false);
constexpr int kSortTile = 4096;   // keys per workgroup of every sort kernel
int r3d_radix_sort_workspace(r3d_ctx* ctx, int64_t n, uint32_t** hist_out, int* n_blocks_out);
static inline int r3d_sort_stride(int n_blocks) { return (n_blocks + 7) & ~7; }   // counters per histogram row (32-byte aligned rows)
// hist[bin][tile] (256 rows of `stride` counters) -> exclusive prefixes over the tiles, in place; totals[bin] = the bin's count
void r3d_sort_launch_scan(r3d_ctx* ctx, uint32_t* hist, int n_blocks, int stride, uint32_t* totals);

// r3d_nnindex.hip: the index's own copy of the target cloud (original order), its size and its context
int r3d_nn_index_target(r3d_nn_index* index, const float** d_tgt, int64_t* n_tgt, r3d_ctx** ctx);
// r3d_icp.hip: last kernel of a sums pass (partial rows -> 18 sums [-> solve + ICP state])
int r3d_icp_sums_finish(r3d_ctx* ctx, const double* d_partials, int n_rows, double* d_sums_out, int with_scale, double* d_state);
// r3d_nnindex.hip: presorted culled query + fused sums + device solve (one iteration's worth, used by r3d_icp_iterate)
int r3d_nn_index_query_solve(r3d_nn_index* index, const float* d_src, int64_t n_src, uint32_t* d_idx_out, float* d_d2_out,
                             float max_d2, double* d_sums_out, int with_scale, double* d_state, int small_motion);
// presorted query of an ICP loop; small_motion: the sources moved by one ICP step since this same query ran last (its matches,
// still in d_idx_out, bound the new search tightly: wave-local kernel).  d_src_orig + d_T_move (device, 12 doubles of a 4x4):
// when the wave-local kernel runs, it first writes d_src = T . d_src_orig itself (*moved_out = 1) -- the caller then skips
// its move launch; otherwise d_src is searched as it is (*moved_out = 0).
int r3d_nn_index_query_step(r3d_nn_index* index, const float* d_src, int64_t n_src, uint32_t* d_idx_out, float* d_d2_out,
                            int small_motion, const float* d_src_orig, const double* d_T_move, int* moved_out);

static inline size_t r3d_depth_size(int dt) { return dt == R3D_DEPTH_U8 ? 1 : dt == R3D_DEPTH_U16 ? 2 : 4; }
static inline size_t r3d_xyz_size(int dt) { return dt == R3D_F32 ? 4 : 8; }
