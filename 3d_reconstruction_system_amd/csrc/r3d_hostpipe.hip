// Host-buffer pipeline shared by the *_host entry points.
//
// The boundary hands over pageable host buffers (NumPy arrays).  A plain hipMemcpy from pageable
// memory runs at ~11 GB/s on the MI355X host, which would cap the fused path at ~1 Gpoint/s.  Here the
// batch is cut into chunks that flow through two pinned staging buffers per direction:
//
//   CPU: user -> pinned_in[b]   |  GPU stream: H2D, kernel, D2H -> pinned_out[b], event[b]
//   CPU: wait event[b]; threads copy pinned_out[b] -> user      (while the GPU works on chunk c+1)
//
// so PCIe runs at its pinned rate and the pageable copies are spread over host threads.  Buffers that are
// already pinned (r3d_host_alloc, or hipHostRegister'ed by the caller) skip the staging copies.
#include <algorithm>
#include <memory>
#include <thread>
#include <vector>

#include "r3d_hostpool.h"
#include "r3d_internal.h"

namespace {

// How many threads share a chunk's pageable <-> pinned copy.  A 32 MiB chunk is ~1 ms of copying for ONE thread pair of
// streams; what the copy competes with is not bandwidth but the hand-over (wake the crew, wait for the last member): on the
// box C2's pageable round trip took 11.8-13.7 ms with 6-8 members, 13.7-14.5 with 16, 13.9 with 32, 16.9 with 2 (pinned
// buffers, no copies: 11.3).
constexpr unsigned kCopyCrew = 8;

// The call's crew, started when the first pageable copy needs it (pinned buffers never do).
struct LazyCrew {
  unsigned n_threads;
  std::unique_ptr<r3d_host::Crew> crew;
  // false: no thread to be had
  bool copy(void* dst, const void* src, size_t bytes) {
    if (bytes < ((size_t)4 << 20) || n_threads <= 1) {
      memcpy(dst, src, bytes);
      return true;
    }
    if (!crew) {
      try {
        crew.reset(new r3d_host::Crew(n_threads));
      } catch (const std::exception&) {
        return false;
      }
    }
    const unsigned n = crew->size();
    // a member's share: ceil(bytes / n) rounded up to a page.  (Until round 4 this was floor(bytes / n) rounded up, which
    // leaves up to n - 1 bytes at the end uncopied whenever bytes / n is itself a whole number of pages and bytes is not
    // a multiple of n: e.g. 65536 k + 12 bytes over 16 threads.  tests/test_gpu_fusion.py pins such sizes now.)
    const size_t per = (((bytes + n - 1) / n) + 4095) & ~(size_t)4095;
    crew->run([=](unsigned t) {
      const size_t lo = (size_t)t * per;
      if (lo < bytes) memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, std::min(per, bytes - lo));
    });
    return true;
  }
};

bool is_pinned(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable pointers report an error: clear it
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int pinned_slot(r3d_ctx* ctx, int slot, size_t bytes, void** p) {
  if (ctx->pinned_bytes[slot] < bytes) {
    if (ctx->pinned[slot]) {
      R3D_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->upload_stream) R3D_HIP(hipStreamSynchronize(ctx->upload_stream));
      R3D_HIP(hipHostFree(ctx->pinned[slot]));
      ctx->pinned[slot] = nullptr;
      ctx->pinned_bytes[slot] = 0;
    }
    R3D_HIP(hipHostMalloc(&ctx->pinned[slot], bytes, hipHostMallocDefault));
    ctx->pinned_bytes[slot] = bytes;
  }
  *p = ctx->pinned[slot];
  return R3D_OK;
}

}  // namespace

int r3d_host_pipeline_multi(r3d_ctx* ctx, int64_t n_items, const r3d_pipe_buf* ins, int n_in, const r3d_pipe_buf* outs,
                            int n_out, const std::function<int(int64_t, int64_t)>& launch) {
  if (n_items <= 0) return R3D_OK;
  R3D_REQUIRE(n_in >= 1 && n_in <= r3d_ctx::kPipeBufs && n_out >= 1 && n_out <= r3d_ctx::kPipeBufs,
              "host pipeline takes 1..%d arrays per direction", r3d_ctx::kPipeBufs);
  int rc;
  LazyCrew crew{std::min(std::max(1u, r3d_host::cpu_budget()), kCopyCrew), nullptr};
  bool in_pinned[r3d_ctx::kPipeBufs], out_pinned[r3d_ctx::kPipeBufs];
  size_t big = 0;
  for (int k = 0; k < n_in; ++k) {
    in_pinned[k] = is_pinned(ins[k].h);
    big = std::max(big, ins[k].item_bytes);
  }
  for (int k = 0; k < n_out; ++k) {
    out_pinned[k] = is_pinned(outs[k].h);
    big = std::max(big, outs[k].item_bytes);
  }
  // ~32 MiB of the largest array per chunk, at least 4 chunks when the batch allows it
  int64_t chunk = std::max<int64_t>(1, (int64_t)(((size_t)32 << 20) / std::max<size_t>(big, 1)));
  chunk = std::min(chunk, std::max<int64_t>(1, (n_items + 3) / 4));
  const int64_t n_chunks = (n_items + chunk - 1) / chunk;
  // staging: slot (direction, array, parity)
  void* pin_in[r3d_ctx::kPipeBufs][2] = {};
  void* pin_out[r3d_ctx::kPipeBufs][2] = {};
  for (int b = 0; b < 2; ++b) {
    for (int k = 0; k < n_in; ++k)
      if (!in_pinned[k] && (rc = pinned_slot(ctx, (0 * r3d_ctx::kPipeBufs + k) * 2 + b, (size_t)chunk * ins[k].item_bytes, &pin_in[k][b])))
        return rc;
    for (int k = 0; k < n_out; ++k)
      if (!out_pinned[k] &&
          (rc = pinned_slot(ctx, (1 * r3d_ctx::kPipeBufs + k) * 2 + b, (size_t)chunk * outs[k].item_bytes, &pin_out[k][b])))
        return rc;
  }
  if (!ctx->ev_pipe[0]) {
    for (int k = 0; k < 6; ++k) R3D_HIP(hipEventCreateWithFlags(&ctx->ev_pipe[k], hipEventDisableTiming));
    R3D_HIP(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
  }
  // uploads ride their own stream so H2D of chunk c+1 overlaps the kernel + D2H of chunk c (PCIe is full duplex):
  //   upload_stream: H2D -> up(b)        main stream: [wait up(b)] kernel, D2H -> done(b)
  // the device arrays hold the whole batch, so uploads never overwrite data a kernel still reads; the first upload waits
  // for whatever the main stream did before this call.
  hipEvent_t* ev_done = ctx->ev_pipe;      // [0,1] results of chunk b landed in pinned_out[b] / user memory
  hipEvent_t* ev_up = ctx->ev_pipe + 2;    // [2,3] chunk uploaded
  hipEvent_t ev_entry = ctx->ev_pipe[4];
  R3D_HIP(hipEventRecord(ev_entry, ctx->stream));
  R3D_HIP(hipStreamWaitEvent(ctx->upload_stream, ev_entry, 0));
  auto issue = [&](int64_t c) -> int {
    const int b = (int)(c & 1);
    const int64_t lo = c * chunk, n = std::min(chunk, n_items - lo);
    for (int k = 0; k < n_in; ++k) {
      const size_t ib = ins[k].item_bytes;
      const char* src = static_cast<const char*>(ins[k].h) + (size_t)lo * ib;
      if (!in_pinned[k]) {
        R3D_REQUIRE(crew.copy(pin_in[k][b], src, (size_t)n * ib), "no host thread for the staging copies");
        src = static_cast<const char*>(pin_in[k][b]);
      }
      r3d_wrote(ctx, static_cast<char*>(ins[k].d) + (size_t)lo * ib, (size_t)n * ib);   // fresh from the host
      R3D_HIP(hipMemcpyAsync(static_cast<char*>(ins[k].d) + (size_t)lo * ib, src, (size_t)n * ib, hipMemcpyHostToDevice,
                             ctx->upload_stream));
    }
    R3D_HIP(hipEventRecord(ev_up[b], ctx->upload_stream));
    R3D_HIP(hipStreamWaitEvent(ctx->stream, ev_up[b], 0));
    int r = launch(lo, n);
    if (r) return r;
    for (int k = 0; k < n_out; ++k) {
      const size_t ob = outs[k].item_bytes;
      void* dst = out_pinned[k] ? static_cast<void*>(static_cast<char*>(outs[k].h) + (size_t)lo * ob) : pin_out[k][b];
      R3D_HIP(hipMemcpyAsync(dst, static_cast<char*>(outs[k].d) + (size_t)lo * ob, (size_t)n * ob, hipMemcpyDeviceToHost,
                             ctx->stream));
    }
    R3D_HIP(hipEventRecord(ev_done[b], ctx->stream));
    return R3D_OK;
  };
  auto drain = [&](int64_t c) -> int {
    const int b = (int)(c & 1);
    const int64_t lo = c * chunk, n = std::min(chunk, n_items - lo);
    R3D_HIP(hipEventSynchronize(ev_done[b]));
    for (int k = 0; k < n_out; ++k)
      if (!out_pinned[k])
        R3D_REQUIRE(crew.copy(static_cast<char*>(outs[k].h) + (size_t)lo * outs[k].item_bytes, pin_out[k][b], (size_t)n * outs[k].item_bytes),
                    "no host thread for the staging copies");
    return R3D_OK;
  };
  // On ANY failure copies may still be in flight into the caller's buffers or the staging ring: quiesce both streams
  // before handing control (and ownership of those buffers) back.
  auto bail = [&](int code) -> int {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->upload_stream);
    return code;
  };
  for (int64_t c = 0; c < n_chunks; ++c) {
    if (c >= 2 && (rc = drain(c - 2))) return bail(rc);  // frees staging pair b before it is reused
    if ((rc = issue(c))) return bail(rc);
  }
  for (int64_t c = std::max<int64_t>(0, n_chunks - 2); c < n_chunks; ++c)
    if ((rc = drain(c))) return bail(rc);
  return R3D_OK;
}

// Device -> pageable host memory at the pinned PCIe rate: 32 MiB chunks through the two pinned staging buffers of the
// pipeline's first output array, the pageable copy of chunk c spread over host threads while chunk c+1 crosses the bus.
// Synchronous.  (A plain hipMemcpy into a fresh NumPy array runs at ~12 GB/s: 31 of the 36 ms of a 48 M-code voxel list.)
int r3d_download_pageable(r3d_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (bytes == 0) return R3D_OK;
  if (bytes < ((size_t)8 << 20) || is_pinned(h_dst)) {
    R3D_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(hipStreamSynchronize(ctx->stream));
    return R3D_OK;
  }
  int rc;
  LazyCrew crew{std::min(std::max(1u, r3d_host::cpu_budget()), kCopyCrew), nullptr};
  const size_t chunk = (size_t)32 << 20;
  void* pin[2] = {};
  for (int b = 0; b < 2; ++b)
    if ((rc = pinned_slot(ctx, (1 * r3d_ctx::kPipeBufs + 0) * 2 + b, chunk, &pin[b]))) return rc;
  if (!ctx->ev_pipe[0]) {
    for (int k = 0; k < 6; ++k) R3D_HIP(hipEventCreateWithFlags(&ctx->ev_pipe[k], hipEventDisableTiming));
    R3D_HIP(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
  }
  const size_t n_chunks = (bytes + chunk - 1) / chunk;
  auto drain = [&](size_t c) -> int {
    const size_t lo = c * chunk, n = std::min(chunk, bytes - lo);
    const hipError_t e = hipEventSynchronize(ctx->ev_pipe[c & 1]);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(ctx->stream);
      return r3d_fail_hip(e, "download", __FILE__, __LINE__);
    }
    if (!crew.copy(static_cast<char*>(h_dst) + lo, pin[c & 1], n)) {
      (void)hipStreamSynchronize(ctx->stream);
      r3d_set_error("no host thread for the staging copies");
      return R3D_ERR_NOMEM;
    }
    return R3D_OK;
  };
  for (size_t c = 0; c < n_chunks; ++c) {
    if (c >= 2 && (rc = drain(c - 2))) return rc;
    const size_t lo = c * chunk, n = std::min(chunk, bytes - lo);
    hipError_t e = hipMemcpyAsync(pin[c & 1], static_cast<const char*>(d_src) + lo, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_pipe[c & 1], ctx->stream);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(ctx->stream);
      return r3d_fail_hip(e, "download", __FILE__, __LINE__);
    }
  }
  for (size_t c = n_chunks >= 2 ? n_chunks - 2 : 0; c < n_chunks; ++c)
    if ((rc = drain(c))) return rc;
  return R3D_OK;
}

int r3d_host_pipeline(r3d_ctx* ctx, int64_t n_items, size_t in_item_bytes, size_t out_item_bytes, const void* h_in,
                      void* h_out, void* d_in, void* d_out, const std::function<int(int64_t, int64_t)>& launch) {
  const r3d_pipe_buf in{const_cast<void*>(h_in), d_in, in_item_bytes}, out{h_out, d_out, out_item_bytes};
  return r3d_host_pipeline_multi(ctx, n_items, &in, 1, &out, 1, launch);
}

extern "C" {

int r3d_download(r3d_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  if (bytes == 0) return R3D_OK;
  R3D_REQUIRE(h_dst && d_src, "NULL pointer with bytes > 0");
  return r3d_download_pageable(ctx, h_dst, d_src, bytes);
}

int r3d_host_alloc(r3d_ctx* ctx, size_t bytes, void** h_ptr_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(h_ptr_out != nullptr, "h_ptr_out is NULL");
  *h_ptr_out = nullptr;
  R3D_HIP(hipHostMalloc(h_ptr_out, bytes ? bytes : 16, hipHostMallocDefault));
  return R3D_OK;
}

int r3d_host_free(r3d_ctx* ctx, void* h_ptr) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  if (!h_ptr) return R3D_OK;
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  R3D_HIP(hipHostFree(h_ptr));
  return R3D_OK;
}

}  // extern "C"
