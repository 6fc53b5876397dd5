// What the library's host-side pools share (file decoders r3d_png.cpp / r3d_jpeg.cpp, text formatters r3d_format.cpp, staging
// copies r3d_hostpipe.hip, the .bt writer r3d_voxel.hip): how many threads to start (cpu_budget), where they start (Spread, opt-in),
// recycled scratch memory (Scratch), and run_batch -- n files, one task each, first failure wins.
#pragma once

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "r3d.h"

void r3d_set_error(const char* fmt, ...);

namespace r3d_host {

// How many threads a pool may usefully run: the CPUs this process is ALLOWED, not the ones the machine has --
// hardware_concurrency() capped by the affinity mask and by TWICE the cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us).
// The MI355X box shows 256 CPUs and grants a quota of 16.  Pools sized by the first number started 128 formatter threads per
// slab (and 15 per txt file, four files at a time) for no gain; pools of exactly 16 lose to pools of 32 -- the quota is CPU
// time per 100 ms period, and a pool that lives for 10-200 ms may burst above its average: 100 PNGs decode in 14.5 ms with 16
// threads, 9.0 with 32 (9.2 with 64); 100 camera txt files take 162-173 ms with 16, 92-142 with 32, no better beyond
// (tools/bench_host_io.py under R3D_HOST_THREADS).  Hence twice the quota.  R3D_HOST_THREADS overrides.  Read once.
inline unsigned cpu_budget() {
  static const unsigned budget = [] {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (pthread_getaffinity_np(pthread_self(), sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
      fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (fscanf(g, "%lld", &quota) != 1) quota = -1;
      fclose(g);
      if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(h, "%lld", &period) != 1) period = 0;
        fclose(h);
      }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, 2 * ((quota + period - 1) / period)));
    if (const char* e = getenv("R3D_HOST_THREADS")) {
      const int v = atoi(e);
      if (v > 0) n = (unsigned)std::min(v, 1024);
    }
    return n;
  }();
  return budget;
}

// Where a pool's workers START -- an OPT-IN hint (R3D_HOST_SPREAD=1).  A thread is born on its creator's CPU and the kernel's
// load balancer moves it later -- on some guests much later: on the build container of this repo (8 vCPUs, Linux 6.18) eight
// fresh threads of 15 ms of pure arithmetic each ran one after the other on ONE CPU (117 ms; 16 ms once placed), so every
// short pool of this library -- file decoders, staging copies, text formatters -- ran serially (8 x 1080p JPEG: 120 ms, 21 ms
// with the hint).  place(t), called by worker t first thing, moves it to the t-th CPU of the creator's affinity mask, counted
// from the creator's own CPU, by narrowing the thread's mask to that CPU and restoring the full mask at once: a placement,
// not a pin.  Off by default because a host whose balancer works loses by it: on the MI355X box (256 CPUs allowed, a quota of
// 16) concurrent pools land on the same CPUs -- the drop-in's 100-frame run took 1.40-1.67 s with the hint, 1.25-1.27 s
// without, the PLY formatter 73-137 ms vs 61-65 ms, the decoders the same either way (tools/host_spread_ab.py).
struct Spread {
  cpu_set_t allowed;
  int cpus[CPU_SETSIZE];
  int n = 0, home = 0;
  Spread() {
    const char* e = getenv("R3D_HOST_SPREAD");
    if (!e || e[0] != '1') return;
    CPU_ZERO(&allowed);
    if (pthread_getaffinity_np(pthread_self(), sizeof(allowed), &allowed) != 0) return;
    const int here = sched_getcpu();
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &allowed)) {
        if (c == here) home = n;
        cpus[n++] = c;
      }
  }
  void place(unsigned t) const {
    if (n < 2) return;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpus[(home + t) % (unsigned)n], &one);
    if (pthread_setaffinity_np(pthread_self(), sizeof(one), &one) == 0) pthread_setaffinity_np(pthread_self(), sizeof(allowed), &allowed);
  }
};

// Threads that live for ONE library call and take its many small jobs -- the 32 MiB chunks of a host pipeline, each copied
// between pageable and pinned memory by all of them: starting sixteen threads per chunk cost more than the chunk's copy
// (19 chunks of C2: 15.0 -> see DESIGN 5 for the figure).  run(work) calls work(t) for t in [0, size()) -- t = 0 on the
// calling thread -- and returns when all are done.  Jobs must not throw.
class Crew {
 public:
  explicit Crew(unsigned n) : n_(std::max(1u, n)) {
    const Spread spread;
    try {
      for (unsigned t = 1; t < n_; ++t)
        helpers_.emplace_back([this, t, spread]() {
          spread.place(t);
          loop(t);
        });
    } catch (...) {
      stop();
      throw;
    }
  }
  ~Crew() { stop(); }
  Crew(const Crew&) = delete;
  Crew& operator=(const Crew&) = delete;
  unsigned size() const { return n_; }
  void run(const std::function<void(unsigned)>& work) {
    if (helpers_.empty()) {
      work(0);
      return;
    }
    {
      std::lock_guard<std::mutex> lock(m_);
      job_ = &work;
      pending_ = (unsigned)helpers_.size();
      ++generation_;
    }
    wake_.notify_all();
    work(0);
    std::unique_lock<std::mutex> lock(m_);
    done_.wait(lock, [this] { return pending_ == 0; });
  }

 private:
  void loop(unsigned t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned)>* job;
      {
        std::unique_lock<std::mutex> lock(m_);
        wake_.wait(lock, [&] { return generation_ != seen; });
        seen = generation_;
        if (stopping_) return;
        job = job_;
      }
      (*job)(t);
      std::lock_guard<std::mutex> lock(m_);
      if (--pending_ == 0) done_.notify_one();
    }
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lock(m_);
      stopping_ = true;
      ++generation_;
    }
    wake_.notify_all();
    for (auto& t : helpers_) t.join();
    helpers_.clear();
  }
  const unsigned n_;
  std::vector<std::thread> helpers_;
  std::mutex m_;
  std::condition_variable wake_, done_;
  const std::function<void(unsigned)>* job_ = nullptr;
  uint64_t generation_ = 0;
  unsigned pending_ = 0;
  bool stopping_ = false;
};

// Scratch memory of the decoders (file bytes, inflated rows, component planes), recycled across frames AND batches.  A frame
// needs a few buffers of 0.5 ... 6 MB; as fresh std::vectors each is an mmap, a run of first-touch page faults and a munmap
// -- per frame and thread, all of them through the process's one address-space lock.  With recycled buffers a worker touches
// fresh pages once per process, not once per frame.  Vectors keep their capacity; users clear() / resize() them as if they
// were new.  At most kKeep scratches of at most kKeepBytes each, and at most kKeepTotal bytes over all of them, are held.
struct Scratch {
  std::vector<unsigned char> buf[6];
  size_t capacity() const {
    size_t c = 0;
    for (const auto& b : buf) c += b.capacity();
    return c;
  }
};
constexpr size_t kKeep = 64, kKeepBytes = (size_t)256 << 20, kKeepTotal = (size_t)1 << 30;
inline std::mutex g_scratch_mutex;
inline std::vector<std::unique_ptr<Scratch>> g_scratch_free;
inline size_t g_scratch_held = 0;   // capacity of everything on the free list (under g_scratch_mutex)
inline thread_local Scratch* t_scratch = nullptr;

// The calling thread holds a scratch for the lifetime of the outermost ScratchScope on its stack.
struct ScratchScope {
  bool own = false;
  ScratchScope() {
    if (t_scratch) return;
    {
      std::lock_guard<std::mutex> lock(g_scratch_mutex);
      if (!g_scratch_free.empty()) {
        t_scratch = g_scratch_free.back().release();
        g_scratch_free.pop_back();
        g_scratch_held -= std::min(g_scratch_held, t_scratch->capacity());
      }
    }
    if (!t_scratch) t_scratch = new Scratch;
    own = true;
  }
  ~ScratchScope() {
    if (!own) return;
    std::unique_ptr<Scratch> s(t_scratch);
    t_scratch = nullptr;
    const size_t cap = s->capacity();
    if (cap > kKeepBytes) return;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_free.size() < kKeep && g_scratch_held + cap <= kKeepTotal) {
      g_scratch_held += cap;
      g_scratch_free.push_back(std::move(s));
    }
  }
  ScratchScope(const ScratchScope&) = delete;
  ScratchScope& operator=(const ScratchScope&) = delete;
};
// buffer i of the calling thread's scratch, emptied (capacity kept); a ScratchScope must be alive
inline std::vector<unsigned char>& scratch(int i) {
  std::vector<unsigned char>& b = t_scratch->buf[i];
  b.clear();
  return b;
}

// runs decode_one(k) for k in [0, n) on a thread pool; first failure wins
template <typename F>
inline int run_batch(int n_files, const char* what, F&& decode_one) {
  const unsigned n_threads = std::max(1u, std::min<unsigned>(cpu_budget(), std::min(32, n_files)));
  std::atomic<int> next{0}, first_rc{R3D_OK};
  std::string first_msg;
  std::atomic<bool> have_msg{false};
  const Spread spread;
  auto worker = [&](unsigned t) {
    if (t) spread.place(t);
    ScratchScope scope;   // one scratch per worker for the whole batch
    for (;;) {
      const int k = next.fetch_add(1);
      if (k >= n_files || first_rc.load() != R3D_OK) return;
      std::string msg;
      const int rc = decode_one(k, &msg);
      if (rc != R3D_OK) {
        int expected = R3D_OK;
        if (first_rc.compare_exchange_strong(expected, rc)) {
          first_msg = msg.empty() ? "bad path" : msg;
          have_msg.store(true);
        }
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(n_threads);
  for (unsigned t = 1; t < n_threads; ++t) {
    try {
      pool.emplace_back(worker, t);
    } catch (const std::system_error&) {   // EAGAIN under a pids limit: the batch goes on with the workers it has
      break;
    }
  }
  worker(0);
  for (auto& t : pool) t.join();
  if (first_rc.load() != R3D_OK) {
    r3d_set_error("%s", have_msg.load() ? first_msg.c_str() : what);
    return first_rc.load();
  }
  return R3D_OK;
}


}  // namespace r3d_host
