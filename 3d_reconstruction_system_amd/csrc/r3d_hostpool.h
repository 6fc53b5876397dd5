// Host thread pool shared by the file decoders (r3d_png.cpp, r3d_jpeg.cpp): n files, one task each, first failure wins.
#pragma once

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "r3d.h"

void r3d_set_error(const char* fmt, ...);

namespace r3d_host {

// runs decode_one(k) for k in [0, n) on a thread pool; first failure wins
template <typename F>
inline int run_batch(int n_files, const char* what, F&& decode_one) {
  unsigned hw = std::thread::hardware_concurrency();
  const unsigned n_threads = std::max(1u, std::min<unsigned>(hw == 0 ? 1 : hw, std::min(32, n_files)));
  std::atomic<int> next{0}, first_rc{R3D_OK};
  std::string first_msg;
  std::atomic<bool> have_msg{false};
  auto worker = [&]() {
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
  for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  if (first_rc.load() != R3D_OK) {
    r3d_set_error("%s", have_msg.load() ? first_msg.c_str() : what);
    return first_rc.load();
  }
  return R3D_OK;
}


}  // namespace r3d_host
