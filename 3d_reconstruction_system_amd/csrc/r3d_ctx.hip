// Context, device memory, HIP-event stopwatch and camera tables of libr3d_hip.so.
#include <cstdlib>
#include <mutex>
#include <vector>

#include "r3d_internal.h"

static thread_local char g_err[512] = "";

void r3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int r3d_fail_hip(hipError_t e, const char* what, const char* file, int line) {
  r3d_set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  if (e == hipErrorOutOfMemory) return R3D_ERR_NOMEM;
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return R3D_ERR_NODEVICE;
  return R3D_ERR_HIP;
}

int r3d_ctx_enter(r3d_ctx* ctx) {
  R3D_REQUIRE(ctx != nullptr, "ctx is NULL");
  R3D_HIP(hipSetDevice(ctx->device));
  // HIP's "last error" is per thread and sticky: another component of the process (torch probing a pointer, a
  // failed query) may have left one behind.  Every entry point starts here, so the hipGetLastError() after our own
  // launches reports our launches only.
  (void)hipGetLastError();
  return R3D_OK;
}

// ---- which inputs are presumed cached ---------------------------------------------------------------------------
// The Infinity Cache belongs to the device, so the record does too (two contexts of one device -- e.g. a compute stream and
// an exchange stream -- see each other's reads and writes).  A small LRU over byte ranges with a byte clock: a MISS advances
// the clock by the range's size and stamps the range with it, a HIT only refreshes the stamp; a range is presumed cached
// while clock - stamp + size <= budget.  Ranges larger than the budget are never presumed cached.
namespace {
struct Residency {
  struct Entry {
    uintptr_t lo, hi;
    uint64_t stamp;
  };
  static constexpr int kEntries = 32;
  std::mutex mu;
  Entry e[kEntries];
  int n = 0;
  uint64_t clock = 0;
  void drop(int k) { e[k] = e[--n]; }
};
constexpr int kMaxDevices = 64;
Residency g_residency[kMaxDevices];
Residency* residency(r3d_ctx* ctx) { return ctx && ctx->device >= 0 && ctx->device < kMaxDevices ? &g_residency[ctx->device] : nullptr; }
}  // namespace

bool r3d_inputs_resident(r3d_ctx* ctx, const void* p, size_t bytes, size_t budget) {
  Residency* r = residency(ctx);
  if (!r || !bytes) return false;
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  std::lock_guard<std::mutex> g(r->mu);
  for (int k = 0; k < r->n; ++k)
    if (r->e[k].lo <= lo && hi <= r->e[k].hi) return r->clock - r->e[k].stamp + (r->e[k].hi - r->e[k].lo) <= budget;
  return false;
}

void r3d_inputs_read(r3d_ctx* ctx, const void* p, size_t bytes, size_t budget) {
  Residency* r = residency(ctx);
  if (!r || !bytes) return;
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  std::lock_guard<std::mutex> g(r->mu);
  for (int k = 0; k < r->n; ++k)
    if (r->e[k].lo <= lo && hi <= r->e[k].hi && r->clock - r->e[k].stamp + (r->e[k].hi - r->e[k].lo) <= budget) {
      if (r->e[k].lo == lo && r->e[k].hi == hi) r->e[k].stamp = r->clock;   // hit on the whole range: refreshed
      return;                                                              // hit on a part: the rest ages as before
    }
  r->clock += bytes;
  for (int k = r->n - 1; k >= 0; --k)
    if (r->e[k].lo < hi && lo < r->e[k].hi) r->drop(k);   // superseded (a stale or partial record of the same bytes)
  if (bytes > budget) return;
  if (r->n == Residency::kEntries) {
    int oldest = 0;
    for (int k = 1; k < r->n; ++k)
      if (r->e[k].stamp < r->e[oldest].stamp) oldest = k;
    r->drop(oldest);
  }
  r->e[r->n++] = {lo, hi, r->clock};
}

void r3d_inputs_written(r3d_ctx* ctx, const void* p, size_t bytes) {
  Residency* r = residency(ctx);
  if (!r || !bytes) return;
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  std::lock_guard<std::mutex> g(r->mu);
  for (int k = r->n - 1; k >= 0; --k)
    if (r->e[k].lo < hi && lo < r->e[k].hi) r->drop(k);
}

void r3d_inputs_forget_all(r3d_ctx* ctx) {
  Residency* r = residency(ctx);
  if (!r) return;
  std::lock_guard<std::mutex> g(r->mu);
  r->n = 0;
}

void r3d_wrote(r3d_ctx* ctx, const void* p, size_t bytes) {
  if (!ctx || !bytes) return;
  r3d_inputs_written(ctx, p, bytes);
  if (ctx->loop_src) {
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes, slo = (uintptr_t)ctx->loop_src;
    const uintptr_t shi = slo + (ctx->loop_src_bytes ? ctx->loop_src_bytes : 1);
    if (lo < shi && slo < hi) ctx->loop_src = nullptr;
  }
}

int r3d_inputs_tracked(r3d_ctx* ctx) {
  Residency* r = residency(ctx);
  if (!r) return 0;
  std::lock_guard<std::mutex> g(r->mu);
  return r->n;
}

int r3d_scratch(r3d_ctx* ctx, int slot, size_t bytes, void** p) {
  R3D_REQUIRE(slot >= 0 && slot < r3d_ctx::kScratchSlots, "bad scratch slot");
  if (bytes == 0) bytes = 16;
  if (ctx->scratch_bytes[slot] < bytes) {
    if (ctx->scratch[slot]) {
      R3D_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->upload_stream) R3D_HIP(hipStreamSynchronize(ctx->upload_stream));
      r3d_inputs_written(ctx, ctx->scratch[slot], ctx->scratch_bytes[slot]);   // the addresses may come back as anything
      R3D_HIP(hipFree(ctx->scratch[slot]));
      ctx->scratch[slot] = nullptr;
      ctx->scratch_bytes[slot] = 0;
    }
    R3D_HIP(hipMalloc(&ctx->scratch[slot], bytes));
    ctx->scratch_bytes[slot] = bytes;
    if (slot == 6) ctx->select_ws = nullptr;   // fresh memory: the selection's histogram is not known to be zero (r3d_plane.hip)
  }
  *p = ctx->scratch[slot];
  return R3D_OK;
}

extern "C" {

int r3d_version(void) { return R3D_VERSION; }

const char* r3d_last_error(void) { return g_err; }

int r3d_device_count(int* n_out) {
  R3D_REQUIRE(n_out != nullptr, "n_out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *n_out = 0;
    return r3d_fail_hip(e, "hipGetDeviceCount", __FILE__, __LINE__);
  }
  *n_out = n;
  return R3D_OK;
}

int r3d_ctx_create(int device, void* stream, int flags, r3d_ctx** ctx_out) {
  R3D_REQUIRE(ctx_out != nullptr, "ctx_out is NULL");
  *ctx_out = nullptr;
  R3D_REQUIRE((flags & ~R3D_CTX_EXTERNAL_STREAM) == 0, "unknown ctx flags 0x%x", flags);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    r3d_set_error("no HIP device visible (hipGetDeviceCount -> %d, n=%d)", (int)e, n);
    return R3D_ERR_NODEVICE;
  }
  if (device < 0 || device >= n) {
    r3d_set_error("device %d out of range [0,%d)", device, n);
    return R3D_ERR_NODEVICE;
  }
  R3D_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  R3D_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    r3d_set_error("device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
    return R3D_ERR_NODEVICE;
  }
  r3d_ctx* c = new (std::nothrow) r3d_ctx();
  if (!c) {
    r3d_set_error("host allocation failed");
    return R3D_ERR_NOMEM;
  }
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  if (flags & R3D_CTX_EXTERNAL_STREAM) {
    c->stream = (hipStream_t)stream;  // NULL = the device's default stream
    c->owns_stream = false;
  } else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return r3d_fail_hip(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
    }
    c->owns_stream = true;
  }
  e = hipEventCreate(&c->ev_start);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_stop);
  if (e != hipSuccess) {
    r3d_ctx_destroy(c);
    return r3d_fail_hip(e, "hipEventCreate", __FILE__, __LINE__);
  }
  *ctx_out = c;
  return R3D_OK;
}

int r3d_ctx_destroy(r3d_ctx* ctx) {
  if (!ctx) return R3D_OK;
  // teardown is best effort: every call below may legitimately fail once the device is gone
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->upload_stream) (void)hipStreamSynchronize(ctx->upload_stream);
  for (int i = 0; i < r3d_ctx::kScratchSlots; ++i)
    if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
  for (int i = 0; i < r3d_ctx::kPinnedSlots; ++i)
    if (ctx->pinned[i]) (void)hipHostFree(ctx->pinned[i]);
  for (int i = 0; i < 6; ++i)
    if (ctx->ev_pipe[i]) (void)hipEventDestroy(ctx->ev_pipe[i]);
  if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
  if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
  if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return R3D_OK;
}

int r3d_ctx_sync(r3d_ctx* ctx) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  return R3D_OK;
}

int r3d_ctx_stream(r3d_ctx* ctx, void** stream_out) {
  R3D_REQUIRE(ctx && stream_out, "NULL argument");
  *stream_out = (void*)ctx->stream;
  return R3D_OK;
}

static int* tuning_slot(r3d_ctx* ctx, const char* key) {
  if (!strcmp(key, "fuse_blocks")) return &ctx->fuse_blocks;
  if (!strcmp(key, "fuse_prefetch")) return &ctx->fuse_prefetch;
  if (!strcmp(key, "fuse_chunk_mb")) return &ctx->fuse_chunk_mb;
  if (!strcmp(key, "fuse_stage_auto_mb")) return &ctx->fuse_stage_auto_mb;
  if (!strcmp(key, "fuse_resident_mb")) return &ctx->fuse_resident_mb;
  if (!strcmp(key, "fuse_inputs_fresh")) return &ctx->fuse_inputs_fresh;
  if (!strcmp(key, "fuse_sweeps")) return &ctx->fuse_sweeps;
  if (!strcmp(key, "nn_variant")) return &ctx->nn_variant;
  if (!strcmp(key, "nn_warm")) return &ctx->nn_warm;
  if (!strcmp(key, "apply_blocks")) return &ctx->apply_blocks;
  if (!strcmp(key, "voxel_dedupe")) return &ctx->voxel_dedupe;
  if (!strcmp(key, "voxel_path")) return &ctx->voxel_path;
  if (!strcmp(key, "voxel_last_path")) return &ctx->voxel_last_path;
  return nullptr;
}

int r3d_ctx_set_tuning(r3d_ctx* ctx, const char* key, int value) {
  R3D_REQUIRE(ctx && key, "NULL argument");
  int* s = tuning_slot(ctx, key);
  R3D_REQUIRE(s != nullptr, "unknown tuning key '%s'", key);
  R3D_REQUIRE(value >= 0, "tuning value must be >= 0");
  if (s == &ctx->fuse_inputs_fresh) {   // an event, not a state: nothing on this device is presumed cached any more
    if (value) r3d_inputs_forget_all(ctx);
    return R3D_OK;
  }
  *s = value;
  return R3D_OK;
}

int r3d_ctx_get_tuning(r3d_ctx* ctx, const char* key, int* value_out) {
  R3D_REQUIRE(ctx && key && value_out, "NULL argument");
  int* s = tuning_slot(ctx, key);
  R3D_REQUIRE(s != nullptr, "unknown tuning key '%s'", key);
  if (s == &ctx->fuse_inputs_fresh) {   // reads back how many input ranges are on record for this device
    *value_out = r3d_inputs_tracked(ctx);
    return R3D_OK;
  }
  *value_out = *s;
  return R3D_OK;
}

int r3d_dev_alloc(r3d_ctx* ctx, size_t bytes, void** d_ptr_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(d_ptr_out != nullptr, "d_ptr_out is NULL");
  *d_ptr_out = nullptr;
  R3D_HIP(hipMalloc(d_ptr_out, bytes ? bytes : 16));
  return R3D_OK;
}

int r3d_dev_free(r3d_ctx* ctx, void* d_ptr) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  if (!d_ptr) return R3D_OK;
  R3D_HIP(hipStreamSynchronize(ctx->stream));
  {
    void* base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, d_ptr) == hipSuccess) r3d_wrote(ctx, base, size);
    else {
      (void)hipGetLastError();
      r3d_inputs_forget_all(ctx);
      ctx->loop_src = nullptr;
    }
  }
  R3D_HIP(hipFree(d_ptr));
  return R3D_OK;
}

int r3d_memcpy_h2d(r3d_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  if (bytes == 0) return R3D_OK;
  R3D_REQUIRE(d_dst && h_src, "NULL pointer with bytes > 0");
  r3d_wrote(ctx, d_dst, bytes);   // DMA lands in HBM, not in the Infinity Cache (measured: bench.py regimes)
  R3D_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return R3D_OK;
}

int r3d_memcpy_d2h(r3d_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  if (bytes == 0) return R3D_OK;
  R3D_REQUIRE(h_dst && d_src, "NULL pointer with bytes > 0");
  R3D_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return R3D_OK;
}

int r3d_memcpy_d2d(r3d_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  if (bytes == 0) return R3D_OK;
  R3D_REQUIRE(d_dst && d_src, "NULL pointer with bytes > 0");
  r3d_wrote(ctx, d_dst, bytes);
  R3D_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return R3D_OK;
}

int r3d_memset(r3d_ctx* ctx, void* d_dst, int byte_value, size_t bytes) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  if (bytes == 0) return R3D_OK;
  R3D_REQUIRE(d_dst != nullptr, "NULL pointer with bytes > 0");
  r3d_wrote(ctx, d_dst, bytes);
  R3D_HIP(hipMemsetAsync(d_dst, byte_value, bytes, ctx->stream));
  return R3D_OK;
}

int r3d_timer_start(r3d_ctx* ctx) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
  return R3D_OK;
}

int r3d_timer_stop(r3d_ctx* ctx, float* ms_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(ms_out != nullptr, "ms_out is NULL");
  R3D_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
  R3D_HIP(hipEventSynchronize(ctx->ev_stop));
  R3D_HIP(hipEventElapsedTime(ms_out, ctx->ev_start, ctx->ev_stop));
  return R3D_OK;
}

int r3d_camera_create(r3d_ctx* ctx, int height, int width, double fx, double fy, double cx, double cy,
                      r3d_camera** cam_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(cam_out != nullptr, "cam_out is NULL");
  *cam_out = nullptr;
  R3D_REQUIRE(height > 0 && width > 0, "raster must be at least 1x1 (got %dx%d)", height, width);
  R3D_REQUIRE((int64_t)height * width < ((int64_t)1 << 31), "raster too large (%dx%d)", height, width);
  R3D_REQUIRE(fx != 0.0 && fy != 0.0, "fx and fy must be non-zero");
  r3d_camera* cam = new (std::nothrow) r3d_camera();
  if (!cam) {
    r3d_set_error("host allocation failed");
    return R3D_ERR_NOMEM;
  }
  cam->ctx = ctx;
  cam->device = ctx->device;
  cam->height = height;
  cam->width = width;
  cam->fx = fx;
  cam->fy = fy;
  cam->cx = cx;
  cam->cy = cy;
  // Reference evaluation order (camera_to_world.py:78-79): (i - cx)/fx first, the product
  // with Z later.  The quotient is Z-independent, so it is tabulated once, in IEEE fp64.
  // Tables are padded to a multiple of 4 entries so 4-wide loads never run off the end.
  const int wp = (width + 3) & ~3, hp = (height + 3) & ~3;
  std::vector<double> u(wp, 0.0), v(hp, 0.0);
  for (int i = 0; i < width; ++i) u[i] = ((double)i - cx) / fx;
  for (int j = 0; j < height; ++j) v[j] = ((double)j - cy) / fy;
  hipError_t e = hipMalloc((void**)&cam->d_u, wp * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&cam->d_v, hp * sizeof(double));
  // synchronous copies: the host vectors die at return
  if (e == hipSuccess) e = hipMemcpy(cam->d_u, u.data(), wp * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(cam->d_v, v.data(), hp * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    r3d_camera_destroy(cam);
    return r3d_fail_hip(e, "camera table upload", __FILE__, __LINE__);
  }
  *cam_out = cam;
  return R3D_OK;
}

int r3d_camera_destroy(r3d_camera* cam) {
  if (!cam) return R3D_OK;
  (void)hipSetDevice(cam->device);
  if (cam->d_u) (void)hipFree(cam->d_u);
  if (cam->d_v) (void)hipFree(cam->d_v);
  delete cam;
  return R3D_OK;
}

}  // extern "C"
