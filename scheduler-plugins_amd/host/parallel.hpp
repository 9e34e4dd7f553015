// parallel.hpp — chunked parallel-for over independent rows for the host flatteners.
//
// Round 4: lazily created, process-wide pools of parked worker threads.  Rounds 1-3 spawned std::threads per call ("the
// flatteners own no thread pool"): 16 thread creations cost ~0.4 ms — at 100 000 pods the whole trimaran pod flatten is 0.1 ms of
// work, and the spawn was most of the 0.58 ms the call took (tools/r4/time_host_cycle.py).  A pool holds one job at a time; there
// are up to eight of them (one per 16 hardware threads), so that concurrent callers — the ranks of a multi-device engine each
// flattening their own pod rows, several cgo threads — each get workers; a caller that finds every pool busy (or a nested call)
// runs its rows inline, which is always correct.  Rows are handed out in chunks from an atomic counter, the calling thread works
// too.  After a fork() the child starts pools of its own (pthread_atfork: the parent's threads do not exist there, and the registry's
// lock is not inherited mid-operation).  Never destroyed: the workers sleep on a condition variable until exit — a process that
// dlclose()s libspx.so while they are parked would leave them in unmapped code; the library is meant to stay loaded (as cgo keeps it).
// Round 5 (advisor): the busy flag is an atomic with an RAII release (a std::mutex try_lock'ed by its owner is undefined behaviour), a
// thread running a pool job runs nested calls inline, a failed thread creation shrinks the job instead of escaping the extern "C" caller.
#pragma once

#include <pthread.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <system_error>
#include <thread>
#include <type_traits>
#include <vector>

namespace spx_host {

class RowPool {
 public:
  static constexpr int kMaxPools = 8;
  // the process's pools: [0, *n)
  static RowPool* const* all(int* n) {
    Registry& r = registry();
    std::lock_guard<std::mutex> lk(r.guard);
    if (r.count == 0) {
      const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
      r.count = static_cast<int>(std::min<unsigned>(kMaxPools, std::max(1u, hw / 16)));
      for (int i = 0; i < r.count; ++i) r.pools[i] = new RowPool;
    }
    *n = r.count;
    return r.pools;
  }
  // the job on the first pool that is free; false = all busy (or this thread is itself running a pool job), nothing was run
  static bool run_any(void (*fn)(void*, int64_t, int64_t), void* ctx, int64_t n, int64_t chunk, unsigned threads) {
    if (inside_job()) return false;  // a nested parallel_rows runs inline: the outer job already has the workers
    int count = 0;
    RowPool* const* pools = all(&count);
    for (int i = 0; i < count; ++i)
      if (pools[i]->run(fn, ctx, n, chunk, threads)) return true;
    return false;
  }

  // fn(ctx, begin, end) over [0, n) in chunks of `chunk` rows on up to `threads` threads (the caller is one of them);
  // false = the pool is busy, nothing was run
  bool run(void (*fn)(void*, int64_t, int64_t), void* ctx, int64_t n, int64_t chunk, unsigned threads) {
    if (busy_.exchange(true, std::memory_order_acquire)) return false;
    struct Release {  // also on an exception out of fn (the flatteners are extern "C": nothing may stay locked behind them)
      std::atomic<bool>& b;
      ~Release() { b.store(false, std::memory_order_release); }
    } release{busy_};
    {
      std::lock_guard<std::mutex> lk(mu_);
      while (workers_.size() + 1 < threads) {
        const unsigned id = static_cast<unsigned>(workers_.size());
        try {
          workers_.emplace_back([this, id] { work(id); });
        } catch (const std::system_error&) {  // no more threads to be had: this job runs with the helpers that exist
          threads = static_cast<unsigned>(workers_.size()) + 1;
          break;
        }
        workers_.back().detach();
      }
      fn_ = fn, ctx_ = ctx, n_ = n, chunk_ = chunk;
      next_.store(0, std::memory_order_relaxed);
      helpers_ = threads - 1;
      pending_ = static_cast<int>(threads - 1);
      ++gen_;
    }
    cv_.notify_all();
    struct Join {  // the helpers are inside fn_ with this call's ctx: wait for them whatever happens to the caller's own share
      RowPool* p;
      ~Join() {
        std::unique_lock<std::mutex> lk(p->mu_);
        p->done_.wait(lk, [this] { return p->pending_ == 0; });
      }
    } join{this};
    inside_job() = true;
    struct Leave {
      ~Leave() { inside_job() = false; }
    } leave;
    drain();
    return true;
  }

 private:
  struct Registry {
    std::mutex guard;
    RowPool* pools[kMaxPools] = {};
    int count = 0;
  };
  // fork(): the child must not inherit `guard` locked by a thread that does not exist there, nor pools whose workers are gone —
  // the prepare handler takes the lock (so no other thread holds it across the fork), the child forgets the parent's pools
  // (leaked: their mutexes may be mid-operation) and starts its own on first use
  static Registry& registry() {
    static Registry* r = [] {
      Registry* x = new Registry;
      reg_ptr() = x;
      pthread_atfork([] { reg_ptr()->guard.lock(); }, [] { reg_ptr()->guard.unlock(); },
                     [] {
                       Registry* c = reg_ptr();
                       c->guard.unlock();
                       c->count = 0;
                       for (auto& p : c->pools) p = nullptr;
                       inside_job() = false;
                     });
      return x;
    }();
    return *r;
  }
  static Registry*& reg_ptr() {
    static Registry* p = nullptr;
    return p;
  }
  static bool& inside_job() {
    static thread_local bool in = false;
    return in;
  }
 private:
  void drain() {
    for (;;) {
      const int64_t b = next_.fetch_add(chunk_, std::memory_order_relaxed);
      if (b >= n_) return;
      fn_(ctx_, b, std::min(n_, b + chunk_));
    }
  }
  void work(unsigned id) {
    uint64_t seen = 0;
    inside_job() = true;  // a flattener called from inside a job runs its rows inline
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (id >= helpers_) continue;  // this job wants fewer threads
      }
      drain();
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_.notify_one();
    }
  }

  std::atomic<bool> busy_{false};
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  void (*fn_)(void*, int64_t, int64_t) = nullptr;
  void* ctx_ = nullptr;
  int64_t n_ = 0, chunk_ = 1;
  std::atomic<int64_t> next_{0};
  unsigned helpers_ = 0;
  int pending_ = 0;
  uint64_t gen_ = 0;
};

// fn(begin, end) is called for disjoint ranges covering [0, n), possibly several times per thread (per-call state belongs inside fn)
template <typename Fn>
inline void parallel_rows(int64_t n, Fn&& fn, int64_t min_rows_per_thread = 8192, unsigned max_threads = 16) {
  static const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned want = static_cast<unsigned>(std::min<int64_t>(std::min(hw, max_threads), n / min_rows_per_thread));
  if (want <= 1) {
    fn(static_cast<int64_t>(0), n);
    return;
  }
  // a few chunks per thread: the rows are not equally expensive (pods differ in container counts) and the workers wake at different times
  const int64_t chunk = std::max<int64_t>(64, (n + 4 * want - 1) / (4 * want));
  auto thunk = [](void* c, int64_t b, int64_t e) { (*static_cast<std::remove_reference_t<Fn>*>(c))(b, e); };
  if (!RowPool::run_any(thunk, const_cast<void*>(static_cast<const void*>(&fn)), n, chunk, want)) fn(static_cast<int64_t>(0), n);
}

}  // namespace spx_host
