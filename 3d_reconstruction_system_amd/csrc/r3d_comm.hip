// Multi-GPU exchange step of the fusion path behind the C ABI: one process per GPU, RCCL over xGMI.
//
// The reference has no distributed code at all; north_star shards the frame loop
// (camera_to_world.py:149-172 carries no state between frames) one block of frames per GPU and assembles the fused
// world cloud with an all-gather.  This file gives a ctypes / plain-C host that exchange without torch:
//   r3d_comm_unique_id / r3d_comm_create   ncclGetUniqueId / ncclCommInitRank on the ctx's GPU
//   r3d_comm_allgather                     byte shards of UNEQUAL length, rank order, on the ctx's stream:
//                                          algo 1 = ncclAllGather (equal shards; RCCL picks ring/tree),
//                                          algo 2 = "direct": one grouped ncclSend/ncclRecv pair per peer, so each peer's
//                                          shard crosses its own xGMI link once (the MI355X node is a full mesh of
//                                          7 links x ~153 GB/s per GPU; a ring is bound by ONE link over world-1 steps),
//                                          algo 0 = 1 when the shards are equal, else 2.
//   r3d_allgather_xyz / r3d_allgather_inputs   the two assemblies of dist.py expressed in points / frames.
//   r3d_comm_allreduce_sum_f64             the 18 ICP sums if a source cloud is ever sharded (144 bytes).
//
// librccl is loaded lazily with dlopen, preferring an RCCL that is ALREADY in the process (torch's bundled copy when a
// Python host imported torch), so that a process never runs two RCCL instances; a host that never calls r3d_comm_*
// never loads it, and the library keeps no link-time dependency on RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
#include <vector>

#include "r3d_internal.h"

struct r3d_comm {
  r3d_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

namespace {

struct RcclApi {
  void* lib = nullptr;
  const char* origin = "";
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  // optional (r3d_comm_rccl_report): what the communicator says about itself
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};

RcclApi g_api;
std::once_flag g_once;
char g_load_error[256] = "";

void load_rccl() {
  struct Try {
    const char* name;
    int flags;
    const char* origin;
  };
  const char* env = getenv("R3D_RCCL_PATH");
  std::vector<Try> tries;
  if (env && *env) tries.push_back({env, RTLD_NOW | RTLD_LOCAL, "R3D_RCCL_PATH"});
  // an RCCL some other component of this process already loaded (torch links "librccl.so")
  tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD, "already loaded (librccl.so)"});
  tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD, "already loaded (librccl.so.1)"});
  tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL, "librccl.so.1"});
  tries.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL, "/opt/rocm/lib/librccl.so.1"});
  void* h = nullptr;
  const char* origin = "";
  for (const Try& t : tries) {
    h = dlopen(t.name, t.flags);
    if (h) {
      origin = t.origin;
      break;
    }
  }
  if (!h) {
    const char* why = dlerror();
    snprintf(g_load_error, sizeof(g_load_error), "cannot load librccl (%s); set R3D_RCCL_PATH", why ? why : "?");
    return;
  }
  RcclApi a;
  a.lib = h;
  a.origin = origin;
#define R3D_SYM(field, sym)                                                        \
  *(void**)(&a.field) = dlsym(h, sym);                                              \
  if (!a.field) {                                                                  \
    snprintf(g_load_error, sizeof(g_load_error), "librccl lacks symbol %s", sym);  \
    return;                                                                        \
  }
  R3D_SYM(GetUniqueId, "ncclGetUniqueId")
  R3D_SYM(CommInitRank, "ncclCommInitRank")
  R3D_SYM(CommDestroy, "ncclCommDestroy")
  R3D_SYM(AllGather, "ncclAllGather")
  R3D_SYM(AllReduce, "ncclAllReduce")
  R3D_SYM(Send, "ncclSend")
  R3D_SYM(Recv, "ncclRecv")
  R3D_SYM(GroupStart, "ncclGroupStart")
  R3D_SYM(GroupEnd, "ncclGroupEnd")
  R3D_SYM(GetErrorString, "ncclGetErrorString")
#undef R3D_SYM
  *(void**)(&a.CommCount) = dlsym(h, "ncclCommCount");
  *(void**)(&a.CommUserRank) = dlsym(h, "ncclCommUserRank");
  *(void**)(&a.CommCuDevice) = dlsym(h, "ncclCommCuDevice");
  *(void**)(&a.GetVersion) = dlsym(h, "ncclGetVersion");
  g_api = a;
}

const RcclApi* rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_api.lib) {
    r3d_set_error("%s", g_load_error[0] ? g_load_error : "librccl unavailable");
    return nullptr;
  }
  return &g_api;
}

#define R3D_NCCL(api, call)                                                                              \
  do {                                                                                                   \
    ncclResult_t r_ = (call);                                                                            \
    if (r_ != ncclSuccess) {                                                                             \
      r3d_set_error("RCCL error %d (%s) in %s at %s:%d", (int)r_, (api)->GetErrorString(r_), #call, __FILE__, __LINE__); \
      return R3D_ERR_HIP;                                                                                \
    }                                                                                                    \
  } while (0)

int check_comm(const r3d_comm* c) {
  R3D_REQUIRE(c != nullptr && c->ctx != nullptr, "comm is NULL");
  return r3d_ctx_enter(c->ctx);
}

// Shard layout shared by every assembly: rank r's block starts at the sum of the counts before it.
void r3d_shard_offsets(const int64_t* counts, int world, int64_t* offsets_out /* world + 1 */) {
  offsets_out[0] = 0;
  for (int r = 0; r < world; ++r) offsets_out[r + 1] = offsets_out[r] + counts[r];
}

}  // namespace

r3d_ctx* r3d_comm_context(const r3d_comm* comm) { return comm ? comm->ctx : nullptr; }

extern "C" {

int r3d_comm_unique_id(void* id_out) {
  R3D_REQUIRE(id_out != nullptr, "id_out is NULL");
  const RcclApi* api = rccl();
  if (!api) return R3D_ERR_UNSUPPORTED;
  ncclUniqueId id;
  R3D_NCCL(api, api->GetUniqueId(&id));
  memcpy(id_out, id.internal, R3D_COMM_ID_BYTES);
  return R3D_OK;
}

int r3d_comm_create(r3d_ctx* ctx, const void* id, int rank, int world, r3d_comm** comm_out) {
  int rc = r3d_ctx_enter(ctx);
  if (rc) return rc;
  R3D_REQUIRE(comm_out != nullptr, "comm_out is NULL");
  *comm_out = nullptr;
  R3D_REQUIRE(id != nullptr, "unique id is NULL");
  R3D_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
  static_assert(R3D_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  const RcclApi* api = rccl();
  if (!api) return R3D_ERR_UNSUPPORTED;
  r3d_comm* c = new (std::nothrow) r3d_comm();
  if (!c) {
    r3d_set_error("host allocation failed");
    return R3D_ERR_NOMEM;
  }
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  ncclUniqueId uid;
  memcpy(uid.internal, id, R3D_COMM_ID_BYTES);
  ncclResult_t r = api->CommInitRank(&c->comm, world, uid, rank);  // collective: every rank calls it with the same id
  if (r != ncclSuccess) {
    r3d_set_error("ncclCommInitRank(rank %d of %d) failed: %d (%s)", rank, world, (int)r, api->GetErrorString(r));
    delete c;
    return R3D_ERR_HIP;
  }
  *comm_out = c;
  return R3D_OK;
}

int r3d_comm_destroy(r3d_comm* comm) {
  if (!comm) return R3D_OK;
  if (comm->comm && g_api.lib) {
    (void)hipSetDevice(comm->ctx ? comm->ctx->device : 0);
    (void)g_api.CommDestroy(comm->comm);
  }
  delete comm;
  return R3D_OK;
}

int r3d_comm_info(const r3d_comm* comm, int* rank_out, int* world_out, const char** rccl_origin_out) {
  R3D_REQUIRE(comm != nullptr, "comm is NULL");
  if (rank_out) *rank_out = comm->rank;
  if (world_out) *world_out = comm->world;
  if (rccl_origin_out) *rccl_origin_out = g_api.origin;
  return R3D_OK;
}

int r3d_comm_rccl_report(const r3d_comm* comm, int* count_out, int* user_rank_out, int* device_out, int* version_out) {
  R3D_REQUIRE(comm != nullptr && comm->comm != nullptr, "comm is NULL");
  const RcclApi* api = rccl();
  if (!api) return R3D_ERR_UNSUPPORTED;
  int v = -1;
  if (count_out) *count_out = (api->CommCount && api->CommCount(comm->comm, &v) == ncclSuccess) ? v : -1;
  if (user_rank_out) *user_rank_out = (api->CommUserRank && api->CommUserRank(comm->comm, &v) == ncclSuccess) ? v : -1;
  if (device_out) *device_out = (api->CommCuDevice && api->CommCuDevice(comm->comm, &v) == ncclSuccess) ? v : -1;
  if (version_out) *version_out = (api->GetVersion && api->GetVersion(&v) == ncclSuccess) ? v : -1;
  return R3D_OK;
}

int r3d_comm_allgather(r3d_comm* comm, const void* d_send, const int64_t* h_counts, void* d_recv, int algo) {
  int rc = check_comm(comm);
  if (rc) return rc;
  R3D_REQUIRE(h_counts != nullptr, "counts is NULL");
  R3D_REQUIRE(algo >= 0 && algo <= 2, "unknown all-gather algorithm %d", algo);
  const RcclApi* api = rccl();
  if (!api) return R3D_ERR_UNSUPPORTED;
  const int W = comm->world, me = comm->rank;
  std::vector<int64_t> off((size_t)W + 1);
  bool equal = true;
  for (int r = 0; r < W; ++r) {
    R3D_REQUIRE(h_counts[r] >= 0, "negative shard size for rank %d", r);
    if (h_counts[r] != h_counts[0]) equal = false;
  }
  r3d_shard_offsets(h_counts, W, off.data());
  if (off[W] == 0) return R3D_OK;
  R3D_REQUIRE(d_recv != nullptr && (h_counts[me] == 0 || d_send != nullptr), "NULL device pointer");
  R3D_REQUIRE(algo != 1 || equal, "ncclAllGather needs equal shards; use algo 0 or 2");
  char* recv = static_cast<char*>(d_recv);
  hipStream_t st = comm->ctx->stream;
  // what arrives over the fabric is not in this device's Infinity Cache: a fused launch that reads it stages it first
  r3d_wrote(comm->ctx, recv, (size_t)off[W]);
  if (algo == 1 || (algo == 0 && equal)) {
    // in place when d_send is already this rank's slot
    R3D_NCCL(api, api->AllGather(d_send, d_recv, (size_t)h_counts[0], ncclUint8, comm->comm, st));
    return R3D_OK;
  }
  // direct: every pair of ranks exchanges its shards point to point, all inside one group (one fused launch)
  if (h_counts[me] > 0 && d_send != recv + off[me])
    R3D_HIP(hipMemcpyAsync(recv + off[me], d_send, (size_t)h_counts[me], hipMemcpyDeviceToDevice, st));
  if (W == 1) return R3D_OK;
  R3D_NCCL(api, api->GroupStart());
  ncclResult_t bad = ncclSuccess;  // a failed enqueue must still close the group, or every later RCCL call of this thread queues into it
  for (int step = 1; step < W && bad == ncclSuccess; ++step) {
    const int to = (me + step) % W, from = (me - step + W) % W;  // staggered peers: no two ranks start on the same target
    if (h_counts[me] > 0) bad = api->Send(d_send, (size_t)h_counts[me], ncclUint8, to, comm->comm, st);
    if (bad == ncclSuccess && h_counts[from] > 0)
      bad = api->Recv(recv + off[from], (size_t)h_counts[from], ncclUint8, from, comm->comm, st);
  }
  if (bad != ncclSuccess) {
    (void)api->GroupEnd();
    r3d_set_error("RCCL error %d (%s) while queueing the direct exchange", (int)bad, api->GetErrorString(bad));
    return R3D_ERR_HIP;
  }
  R3D_NCCL(api, api->GroupEnd());
  return R3D_OK;
}

int r3d_allgather_xyz(r3d_comm* comm, const void* d_shard, const int64_t* h_points_per_rank, int dtype, void* d_full,
                      int algo) {
  R3D_REQUIRE(comm != nullptr && h_points_per_rank != nullptr, "NULL argument");
  R3D_REQUIRE(dtype == R3D_F32 || dtype == R3D_F64, "unknown point dtype %d", dtype);
  std::vector<int64_t> bytes((size_t)comm->world);
  for (int r = 0; r < comm->world; ++r) bytes[r] = h_points_per_rank[r] * 3 * (int64_t)r3d_xyz_size(dtype);
  return r3d_comm_allgather(comm, d_shard, bytes.data(), d_full, algo);
}

int r3d_allgather_inputs(r3d_comm* comm, const void* d_depth, int depth_dtype, const int64_t* h_frames_per_rank, int height,
                         int width, const double* d_pose, void* d_depth_all, double* d_pose_all, int algo) {
  R3D_REQUIRE(comm != nullptr && h_frames_per_rank != nullptr, "NULL argument");
  R3D_REQUIRE(depth_dtype >= R3D_DEPTH_U8 && depth_dtype <= R3D_DEPTH_F32, "unknown depth dtype %d", depth_dtype);
  R3D_REQUIRE(height > 0 && width > 0, "bad raster size");
  std::vector<int64_t> db((size_t)comm->world), pb((size_t)comm->world);
  for (int r = 0; r < comm->world; ++r) {
    db[r] = h_frames_per_rank[r] * (int64_t)height * width * (int64_t)r3d_depth_size(depth_dtype);
    pb[r] = h_frames_per_rank[r] * 12 * (int64_t)sizeof(double);
  }
  int rc = r3d_comm_allgather(comm, d_depth, db.data(), d_depth_all, algo);
  if (rc) return rc;
  if (d_pose_all) rc = r3d_comm_allgather(comm, d_pose, pb.data(), d_pose_all, algo);
  return rc;
}

int r3d_comm_allreduce_sum_f64(r3d_comm* comm, double* d_buf, int64_t n) {
  int rc = check_comm(comm);
  if (rc) return rc;
  R3D_REQUIRE(n >= 0, "negative count");
  if (n == 0) return R3D_OK;
  R3D_REQUIRE(d_buf != nullptr, "NULL device pointer");
  const RcclApi* api = rccl();
  if (!api) return R3D_ERR_UNSUPPORTED;
  R3D_NCCL(api, api->AllReduce(d_buf, d_buf, (size_t)n, ncclFloat64, ncclSum, comm->comm, comm->ctx->stream));
  return R3D_OK;
}

}  // extern "C"
