// ThreadSanitizer driver for r3d_host::Crew (csrc/r3d_hostpool.h), the threads behind the host pipeline's staging copies: many
// jobs of odd sizes through crews of 1, 2, 7 and 16 members, every byte checked (the share arithmetic of the copy it replaced
// lost the last bytes of some sizes), no data race reported.
//   g++ -std=c++17 -O2 -fsanitize=thread -Iinclude -I3d_reconstruction_system_amd/csrc tests/c/crew_tsan.cpp -lpthread
#include "r3d_hostpool.h"
#include <cstdio>
#include <cstring>
void r3d_set_error(const char*, ...) {}
int main() {
  std::vector<unsigned char> a(12 << 20), b(12 << 20);
  for (size_t i = 0; i < a.size(); ++i) a[i] = (unsigned char)(i * 2654435761u >> 24);
  for (unsigned n : {1u, 2u, 7u, 16u}) {
    r3d_host::Crew crew(n);
    for (int rep = 0; rep < 40; ++rep) {
      const size_t bytes = (rep % 3 == 0) ? a.size() : (rep % 3 == 1 ? (size_t)65536 * 16 * 7 + 12 : (size_t)(5 << 20) + rep * 4099);
      memset(b.data(), 0, bytes);
      const size_t per = (((bytes + crew.size() - 1) / crew.size()) + 4095) & ~(size_t)4095;
      crew.run([&](unsigned t) { size_t lo = t * per; if (lo < bytes) memcpy(b.data() + lo, a.data() + lo, std::min(per, bytes - lo)); });
      if (memcmp(a.data(), b.data(), bytes)) { printf("mismatch n=%u rep=%d\n", n, rep); return 1; }
    }
  }
  printf("crew ok\n");
}
