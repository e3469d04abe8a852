// ThreadSanitizer / AddressSanitizer unit test of the plan threads (sipnet_amd/csrc/plan_pool.h): a pool that lives as long
// as the process, one job at a time, jobs of any size from several caller threads at once (a node's shards all set up).
// No HIP: builds with g++ (`make -C sipnet_amd/csrc san`), run by tests/test_sanitizers.py.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../sipnet_amd/csrc/plan_pool.h"

#define REQUIRE(c)                                                        \
  do {                                                                    \
    if (!(c)) {                                                           \
      fprintf(stderr, "%s:%d: REQUIRE(%s) failed\n", __FILE__, __LINE__, #c); \
      exit(1);                                                            \
    }                                                                     \
  } while (0)

// one caller: jobs of many shapes; every item is written exactly once (a plain store: an item run twice at the same time,
// or a job that returns before its last item is done, is a data race TSan sees and a wrong sum we see)
static void caller(int id, int rounds) {
  sipnet::PlanPool& pool = sipnet::PlanPool::get();
  for (int r = 0; r < rounds; r++) {
    const int n = (r * 7 + id * 3) % 41;              // 0 .. 40 items
    const int nThreads = 1 + (r + id) % 9;            // 1 .. 9 threads (more than items, sometimes)
    std::vector<int> slot((size_t)n, 0);
    std::atomic<int> calls{0};
    pool.run(n, nThreads, [&](int i) {
      slot[(size_t)i] += i + 1;
      calls.fetch_add(1, std::memory_order_relaxed);
    });
    REQUIRE(calls.load() == n);
    long sum = 0;
    for (int i = 0; i < n; i++) sum += slot[(size_t)i];
    REQUIRE(sum == (long)n * (n + 1) / 2);
  }
}

int main() {
  caller(0, 200);                                      // alone first (the pool grows to 8 workers)
  std::vector<std::thread> ts;
  for (int id = 1; id <= 4; id++) ts.emplace_back(caller, id, 300);   // then four callers at once: they take turns
  for (auto& t : ts) t.join();
  // a job whose items themselves take a while (workers still draining when the caller has run out of items)
  std::atomic<long> total{0};
  sipnet::PlanPool::get().run(64, 6, [&](int i) {
    long x = 0;
    for (int k = 0; k < 20000 + 1000 * (i % 5); k++) x += k % 7;
    total.fetch_add(x > 0 ? 1 : 0);
  });
  REQUIRE(total.load() == 64);
  printf("plan pool: ok\n");
  return 0;
}
