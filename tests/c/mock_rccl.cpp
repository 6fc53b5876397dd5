// TEST INFRASTRUCTURE, not product code: a stand-in for librccl that moves bytes between PROCESSES THAT SHARE ONE GPU
// through files in /dev/shm, so that the N > 1 logic of csrc/r3d_comm.hip (shard offsets, ragged shards, in-place slots,
// the direct send/recv schedule, the all-reduce) runs for real on the one-GPU test box, where RCCL itself refuses two
// ranks on one device.  Loaded through R3D_RCCL_PATH; implements the ten symbols r3d_comm.hip needs and two of
// the four optional ones (ncclCommCount, ncclCommUserRank; ncclCommCuDevice / ncclGetVersion are left out on purpose: -1).
// Semantics kept from NCCL: rank order, in-place all-gather when sendbuff == recvbuff + rank*count, grouped p2p without
// deadlock.  Not kept: asynchrony (every call synchronises the stream), speed.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <thread>
#include <vector>

struct ncclComm {
  int rank, world;
  std::string id;
  std::map<std::pair<int, int>, long> seq;  // per (src, dst) message counter
  int group = 0;
  struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    hipStream_t st;
  };
  std::vector<Op> pending;
};

namespace {

size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

std::string path_for(ncclComm* c, int src, int dst, long k) {
  char b[256];
  snprintf(b, sizeof(b), "/dev/shm/r3dmock_%s_%d_%d_%ld", c->id.c_str(), src, dst, k);
  return b;
}

// Messages of config-4 size (gigabytes: the byte offsets beyond 2^32 are what the rehearsal is after) pass through a
// 64 MB host buffer piece by piece -- the only thing of message size is the /dev/shm file itself.
constexpr size_t kPiece = (size_t)64 << 20;

bool do_send(ncclComm* c, const void* d_buf, size_t bytes, int peer, hipStream_t st) {
  std::vector<char> h(bytes < kPiece ? bytes : kPiece);
  if (hipStreamSynchronize(st) != hipSuccess) return false;
  const long k = c->seq[{c->rank, peer}]++;
  const std::string p = path_for(c, c->rank, peer, k), tmp = p + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  bool ok = true;
  for (size_t at = 0; at < bytes && ok; at += kPiece) {
    const size_t n = bytes - at < kPiece ? bytes - at : kPiece;
    ok = hipMemcpy(h.data(), static_cast<const char*>(d_buf) + at, n, hipMemcpyDeviceToHost) == hipSuccess &&
         fwrite(h.data(), 1, n, f) == n;
  }
  fclose(f);
  return ok && rename(tmp.c_str(), p.c_str()) == 0;
}

bool do_recv(ncclComm* c, void* d_buf, size_t bytes, int peer, hipStream_t st) {
  const long k = c->seq[{peer, c->rank}]++;
  const std::string p = path_for(c, peer, c->rank, k);
  const auto t0 = std::chrono::steady_clock::now();
  struct stat sb;
  while (stat(p.c_str(), &sb) != 0) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;  // a lost peer must not hang a test
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  if ((size_t)sb.st_size != bytes) return false;   // (the sender renames the file into place when it is complete)
  std::vector<char> h(bytes < kPiece ? bytes : kPiece);
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return false;
  bool ok = hipStreamSynchronize(st) == hipSuccess;
  for (size_t at = 0; at < bytes && ok; at += kPiece) {
    const size_t n = bytes - at < kPiece ? bytes - at : kPiece;
    ok = fread(h.data(), 1, n, f) == n &&
         hipMemcpy(static_cast<char*>(d_buf) + at, h.data(), n, hipMemcpyHostToDevice) == hipSuccess;
  }
  fclose(f);
  unlink(p.c_str());
  return ok;
}

ncclResult_t flush(ncclComm* c) {
  for (auto& o : c->pending)
    if (o.send && !do_send(c, o.buf, o.bytes, o.peer, o.st)) return ncclSystemError;
  for (auto& o : c->pending)
    if (!o.send && !do_recv(c, o.buf, o.bytes, o.peer, o.st)) return ncclSystemError;
  c->pending.clear();
  return ncclSuccess;
}

ncclComm* g_group_comm = nullptr;  // ncclGroupStart/End carry no communicator: remember the one used inside the group
int g_group_depth = 0;

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id->internal, 0, sizeof(id->internal));
  std::random_device rd;
  snprintf(id->internal, sizeof(id->internal), "%08x%08x", rd(), rd());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->world = nranks;
  c->id = std::string(id.internal, strnlen(id.internal, 32));
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return ncclSuccess;
}

// the communicator's own account of itself (r3d_comm_rccl_report binds these when the library has them)
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  *count = comm->world;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  *rank = comm->rank;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++g_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (--g_group_depth > 0) return ncclSuccess;
  ncclComm* c = g_group_comm;
  g_group_comm = nullptr;
  return c ? flush(c) : ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  c->pending.push_back({true, const_cast<void*>(buf), count * dtype_size(t), peer, st});
  if (g_group_depth > 0) {
    g_group_comm = c;
    return ncclSuccess;
  }
  return flush(c);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  c->pending.push_back({false, buf, count * dtype_size(t), peer, st});
  if (g_group_depth > 0) {
    g_group_comm = c;
    return ncclSuccess;
  }
  return flush(c);
}

// MOCK_RCCL_STALL_RANK=r MOCK_RCCL_STALL_AFTER=k: rank r never returns from its k-th all-gather -- how a wedged fabric
// looks to the host (bench.py's watchdog test)
static void maybe_stall(ncclComm_t c) {
  static int calls = 0;
  const char* r = getenv("MOCK_RCCL_STALL_RANK");
  const char* k = getenv("MOCK_RCCL_STALL_AFTER");
  if (r && k && atoi(r) == c->rank && ++calls > atoi(k))
    for (;;) sleep(1);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
  maybe_stall(c);
  const size_t bytes = count * dtype_size(t);
  char* r = static_cast<char*>(recv);
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (send != r + bytes * c->rank && bytes && hipMemcpy(r + bytes * c->rank, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess)
    return ncclUnhandledCudaError;
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && !do_send(c, send, bytes, p, st)) return ncclSystemError;
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && !do_recv(c, r + bytes * p, bytes, p, st)) return ncclSystemError;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t st) {
  if (t != ncclFloat64 || op != ncclSum) return ncclInvalidArgument;
  std::vector<double> mine(count), acc(count), other(count);
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(mine.data(), send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  void* d_tmp = nullptr;
  if (hipMalloc(&d_tmp, count * 8) != hipSuccess) return ncclUnhandledCudaError;
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && !do_send(c, send, count * 8, p, st)) return ncclSystemError;
  // rank order summation: every rank computes the same bits
  for (size_t k = 0; k < count; ++k) acc[k] = 0.0;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) {
      other = mine;
    } else {
      if (!do_recv(c, d_tmp, count * 8, p, st)) return ncclSystemError;
      if (hipMemcpy(other.data(), d_tmp, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    }
    for (size_t k = 0; k < count; ++k) acc[k] += other[k];
  }
  (void)hipFree(d_tmp);
  return hipMemcpy(recv, acc.data(), count * 8, hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

}  // extern "C"
