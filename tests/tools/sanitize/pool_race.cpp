// The shared host pool under ThreadSanitizer: two callers at once (the FASTA loader works on a background thread
// while the main thread transforms or writes), many short jobs, results checked.   usage: pool_race [rounds]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../../pyani_plus_amd/csrc/host_pool.h"

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  std::atomic<long> wrong{0};
  auto caller = [&](uint32_t workers, int salt) {
    std::vector<long> slots(workers);
    for (int r = 0; r < rounds; ++r) {
      for (auto &s : slots) s = 0;
      HostPool::get().run(workers, [&](uint32_t w, uint32_t n) {
        if (n != workers) ++wrong;
        slots[w] += (long)(w + 1) * (r + salt);
      });
      for (uint32_t w = 0; w < workers; ++w)
        if (slots[w] != (long)(w + 1) * (r + salt)) ++wrong;
    }
  };
  std::thread a(caller, 4u, 1), b(caller, 7u, 1000), c(caller, 2u, 5);
  a.join();
  b.join();
  c.join();
  printf("budget %u threads; %d rounds x 3 callers, wrong results: %ld\n", pa_cpu_budget(), rounds, wrong.load());
  return wrong.load() ? 1 : 0;
}
