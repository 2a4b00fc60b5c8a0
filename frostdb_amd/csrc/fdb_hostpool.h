// fdb_hostpool.h — a small process-wide pool of host threads (the per-page / per-chunk work of a Parquet row group). Copies of pushed
// records into the pinned slab do NOT go through it: a pooled thread's wake-up costs more than half a megabyte of copying.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace fdb {

// Threads are created once: spawning
// sixty threads per call cost more (≈1.3 ms) than inflating the pages they were spawned for.
class HostPool {
 public:
  static HostPool& get() { static HostPool p; return p; }
  size_t size() const { return workers_.size(); }
  // fn(i) for i < n on the pool's threads and the calling one; the first exception in index order is rethrown.
  template <typename F>
  void parallel_for(size_t n, F&& fn) {
    if (n == 0) return;
    struct Job {
      std::function<void(size_t)> fn;
      size_t n;
      std::atomic<size_t> next{0}, done{0};
      std::vector<std::exception_ptr> errs;
    };
    auto job = std::make_shared<Job>();
    job->fn = std::forward<F>(fn); job->n = n; job->errs.resize(n);
    auto drain = [](const std::shared_ptr<Job>& j) {
      for (;;) {
        const size_t i = j->next.fetch_add(1);
        if (i >= j->n) return;
        try { j->fn(i); } catch (...) { j->errs[i] = std::current_exception(); }
        j->done.fetch_add(1, std::memory_order_release);
      }
    };
    if (n > 1 && !workers_.empty()) {
      const size_t helpers = std::min(n - 1, workers_.size());
      { std::lock_guard<std::mutex> lk(mu_); for (size_t k = 0; k < helpers; k++) queue_.push_back([job, drain] { drain(job); }); }
      cv_.notify_all();
    }
    drain(job);
    while (job->done.load(std::memory_order_acquire) < n) std::this_thread::yield();  // (helpers still inside their last fn)
    for (const std::exception_ptr& e : job->errs) if (e) std::rethrow_exception(e);
  }

 private:
  HostPool() {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n = std::min(hw > 1 ? hw - 1 : 0u, 31u);
    for (unsigned i = 0; i < n; i++) {
      try { workers_.emplace_back([this] { run(); }); } catch (const std::system_error&) { break; }
    }
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  void run() {
    for (;;) {
      std::function<void()> task;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (stop_ && queue_.empty()) return;
        task = std::move(queue_.front());
        queue_.pop_front();
      }
      task();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> queue_;
  std::vector<std::thread> workers_;
  bool stop_ = false;
};

}  // namespace fdb
