// plan_pool.h -- the plan threads: a pool that lives as long as the process, with no HIP in it (engine.hip uses it for
// the site plans, the light passes before a device-built plan and the copies of a several-sites hand-over), so that it
// builds -- and is raced -- under ThreadSanitizer on a box without a GPU (tests/c/plan_pool_tsan.cpp, `make san`).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sipnet {

// The plan threads live as long as the process (a pool: waking a parked thread costs ~10 us, starting one ~30 -- at 15
// threads per hand-over of a forcing that was 0.4 ms of every setup).  One job at a time: two batches setting up from two
// host threads (a node's shards) take turns, each with all the threads.
class PlanPool {
 public:
  static PlanPool& get() {
    static PlanPool p;
    return p;
  }
  // f(i) for every i in [0, n), on up to nThreads threads (the caller's included); returns when all are done
  void run(int n, int nThreads, const std::function<void(int)>& f) {
    if (nThreads > n) nThreads = n;
    if (nThreads <= 1) {
      for (int i = 0; i < n; i++) f(i);
      return;
    }
    std::lock_guard<std::mutex> oneJob(jobMu);
    {
      std::unique_lock<std::mutex> lk(mu);
      while ((int)workers.size() < nThreads - 1) {
        const int k = (int)workers.size();
        workers.emplace_back([this, k] { loop(k); });
      }
      job = &f;
      nItems = n;
      next.store(0);
      wanted = nThreads - 1;
      active = wanted;
      gen++;
    }
    cvWork.notify_all();
    drain(f);
    std::unique_lock<std::mutex> lk(mu);
    cvDone.wait(lk, [&] { return active == 0; });
    job = nullptr;
  }
  ~PlanPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cvWork.notify_all();
    for (auto& t : workers) t.join();
  }

 private:
  void drain(const std::function<void(int)>& f) {
    for (int i = next.fetch_add(1); i < nItems; i = next.fetch_add(1)) f(i);
  }
  void loop(int k) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* f = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cvWork.wait(lk, [&] { return quit || gen != seen; });
        if (quit) return;
        seen = gen;
        if (k < wanted) f = job;
      }
      if (!f) continue;
      drain(*f);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--active == 0) cvDone.notify_all();
      }
    }
  }
  std::mutex jobMu, mu;
  std::condition_variable cvWork, cvDone;
  std::vector<std::thread> workers;
  const std::function<void(int)>* job = nullptr;
  std::atomic<int> next{0};
  int nItems = 0, wanted = 0, active = 0;
  uint64_t gen = 0;
  bool quit = false;
};

}  // namespace sipnet
