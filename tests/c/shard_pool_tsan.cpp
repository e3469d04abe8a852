// ThreadSanitizer / AddressSanitizer unit test of the node object's host-side threading (sipnet_amd/csrc/shard_pool.h):
// the persistent shard threads, the task hand-over, the barrier and what happens when a shard's task fails.
// No HIP: builds with g++ (`make -C sipnet_amd/csrc san`), run by tests/test_sanitizers.py.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../sipnet_amd/csrc/shard_pool.h"

static thread_local std::string t_err;
#define REQUIRE(c)                                                        \
  do {                                                                    \
    if (!(c)) {                                                           \
      fprintf(stderr, "%s:%d: REQUIRE(%s) failed\n", __FILE__, __LINE__, #c); \
      exit(1);                                                            \
    }                                                                     \
  } while (0)

int main() {
  std::mt19937 rng(20261003);
  for (int n : {1, 2, 3, 8}) {
    sipnet::ShardPool pool;
    std::atomic<int> entered{0};
    pool.start(n, [&](int) { entered++; return true; }, [] { return t_err; }, 100);
    // 1. barriers: every shard adds to a plain (non-atomic) per-phase slot of its own, reads everybody's after the
    // barrier -- a barrier that lets a thread through early is a data race TSan sees and a wrong sum we see
    std::vector<int> slot(n * 4, 0);
    for (int round = 0; round < 300; round++) {
      const int bad = pool.run([&](int k) -> int {
        for (int ph = 0; ph < 4; ph++) {
          slot[ph * n + k] = round * 10 + ph;
          if (!pool.bar.arrive()) return 7;
          long sum = 0;
          for (int j = 0; j < n; j++) sum += slot[ph * n + j];
          if (sum != (long)n * (round * 10 + ph)) return 8;
          if (!pool.bar.arrive()) return 7;   // nobody overwrites a slot another shard is still reading
        }
        return 0;
      });
      REQUIRE(bad == -1);
    }
    REQUIRE(entered.load() >= 300);
    // 2. a failing shard: at a random point of the task shard f returns an error; the others give up at their next
    // arrival instead of waiting for ever, the caller gets f's error -- not a bystander's -- and the pool runs again
    for (int round = 0; round < 200 && n > 1; round++) {
      const int f = (int)(rng() % n), where = (int)(rng() % 3);
      const int bad = pool.run([&](int k) -> int {
        for (int ph = 0; ph < 3; ph++) {
          if (k == f && ph == where) {
            t_err = "shard " + std::to_string(k) + " broke";
            return 42;
          }
          if (!pool.bar.arrive()) {
            t_err = "sipnet_node: another shard failed";
            return 7;
          }
        }
        return 0;
      });
      REQUIRE(bad == f);
      REQUIRE(pool.rc[f] == 42);
      REQUIRE(pool.msg[f] == "shard " + std::to_string(f) + " broke");
      for (int k = 0; k < n; k++) REQUIRE(k == f || pool.rc[k] == 0 || pool.rc[k] == 7);
      const int again = pool.run([&](int) -> int { return pool.bar.arrive() ? 0 : 7; });
      REQUIRE(again == -1);
    }
    // 3. a shard whose device cannot be bound: its task does not run, the others do not wait for it
    if (n > 1) {
      pool.enter = [&](int k) { return k != 1; };
      const int bad = pool.run([&](int) -> int {
        if (!pool.bar.arrive()) {
          t_err = "sipnet_node: another shard failed";
          return 7;
        }
        return 0;
      });
      REQUIRE(bad == 1 && pool.rc[1] == 100);
      pool.enter = [&](int) { return true; };
      REQUIRE(pool.run([&](int) -> int { return pool.bar.arrive() ? 0 : 7; }) == -1);
    }
    pool.stop();
    pool.stop();   // idempotent
  }
  printf("shard_pool: ok\n");
  return 0;
}
