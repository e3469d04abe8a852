// shard_pool.h -- the host side of a node's shards with no HIP in it: the persistent per-shard host threads, the
// hand-over of a task to all of them, and the barrier they meet at.  Kept apart from node.cpp so that it builds --
// and is raced -- under ThreadSanitizer on a box without a GPU (tests/c/shard_pool_tsan.cpp, `make san`).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace sipnet {

// all shard threads meet here (the event-ordered transport: an event must have been recorded by its owner before
// another thread makes its stream wait for it; the RCCL transport: nobody enqueues a collective a failed shard
// would be missing from)
struct HostBarrier {
  std::mutex mu;
  std::condition_variable cv;
  int n = 1, waiting = 0;
  uint64_t phase = 0;
  bool broken = false;   // a shard's task failed: nobody waits for it any more (reset per task)
  // false: a shard has failed, the others give up too
  bool arrive() {
    if (n <= 1) return true;
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const uint64_t p = phase;
    if (++waiting == n) {
      waiting = 0;
      phase++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return phase != p || broken; });
    }
    return !broken;
  }
  void fail() {
    std::lock_guard<std::mutex> lk(mu);
    broken = true;
    cv.notify_all();
  }
  void reset() {
    std::lock_guard<std::mutex> lk(mu);
    broken = false;
    waiting = 0;
  }
};

// n host threads that live as long as the pool (n == 1: none, tasks run on the caller's thread).  run(f) executes f(k)
// on shard k's thread and returns when all have returned; a task that returns non-zero breaks the barrier, so the
// others' next arrive() fails instead of waiting for it.
struct ShardPool {
  HostBarrier bar;
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cvWork, cvDone;
  std::function<int(int)> task;
  uint64_t gen = 0;
  int pending = 0;
  bool quit = false;
  std::vector<int> rc;
  std::vector<std::string> msg;
  // what a shard's thread does before every task (bind its device); false: the task is not run, rc = enterFailedRc
  std::function<bool(int)> enter;
  int enterFailedRc = 1;
  // the error text of the calling thread's last failure (thread-local in the library: carried to the caller's thread)
  std::function<std::string()> lastError;

  int n() const { return (int)rc.size(); }

  void start(int nShards, std::function<bool(int)> enterFn, std::function<std::string()> lastErrorFn, int enterRc) {
    rc.assign(nShards, 0);
    msg.assign(nShards, "");
    bar.n = nShards;
    enter = std::move(enterFn);
    lastError = std::move(lastErrorFn);
    enterFailedRc = enterRc;
    if (nShards > 1)
      for (int k = 0; k < nShards; k++) workers.emplace_back([this, k] { loop(k); });
  }
  void stop() {
    if (workers.empty()) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cvWork.notify_all();
    for (auto& t : workers) t.join();
    workers.clear();
  }
  ~ShardPool() { stop(); }

  void runOne(int k) {
    rc[k] = task(k);
    if (rc[k] != 0) bar.fail();   // the other shards must not wait for this one at a barrier
    msg[k] = rc[k] != 0 ? (lastError ? lastError() : std::string()) : std::string();
  }
  void loop(int k) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cvWork.wait(lk, [&] { return quit || gen != seen; });
        if (quit) return;
        seen = gen;
      }
      if (!enter || enter(k)) {
        runOne(k);
      } else {
        rc[k] = enterFailedRc;
        msg[k] = "binding the shard's device failed";
        bar.fail();
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--pending == 0) cvDone.notify_all();
      }
    }
  }
  // -> the shard whose failure is to be reported (-1: none).  A shard that only gave up at a barrier because another
  // one failed ("another shard failed" in its message) does not hide that one's message.
  int run(std::function<int(int)> f) {
    const int nS = n();
    task = std::move(f);
    bar.reset();
    if (nS == 1) {
      if (enter && !enter(0)) {
        rc[0] = enterFailedRc;
        msg[0] = "binding the shard's device failed";
      } else {
        runOne(0);
      }
    } else {
      std::unique_lock<std::mutex> lk(mu);
      pending = nS;
      gen++;
      cvWork.notify_all();
      cvDone.wait(lk, [&] { return pending == 0; });
    }
    task = nullptr;
    for (int pass = 0; pass < 2; pass++)
      for (int k = 0; k < nS; k++)
        if (rc[k] != 0 && (pass == 1 || msg[k].find("another shard failed") == std::string::npos)) return k;
    return -1;
  }
};

}  // namespace sipnet
