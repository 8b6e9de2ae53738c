// host_pool.h -- a small persistent pool of host threads shared by the host-side bulk routines
// (pa_ani_host, the JSON and .sig writers, the FASTA loader).
#pragma once

#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// ---- a small persistent pool of host threads --------------------------------------------------------------
// pa_ani_host is called once per column tile; creating 64 threads per call costs more than the pows of a
// sparse tile.  Workers are created on first demand, sleep between jobs and are never joined (the pool lives as
// long as the process; a forked child starts its own).
class HostPool {
 public:
  static inline HostPool &get() {
    static HostPool *pool = nullptr;
    static pid_t owner = 0;
    static std::mutex guard;
    std::lock_guard<std::mutex> lock(guard);
    if (!pool || owner != getpid()) {  // first use, or we are a forked child whose copy has no threads
      pool = new HostPool();
      owner = getpid();
    }
    return *pool;
  }
  // fn(worker, n_workers) on n_workers threads (the caller is worker 0); returns when all are done
  template <typename F>
  void run(uint32_t n_workers, F &&fn) {
    if (n_workers <= 1) { fn(0u, 1u); return; }
    std::exception_ptr first_error;
    std::mutex error_guard;
    auto guarded = [&](uint32_t id, uint32_t n) {
      try {
        fn(id, n);
      } catch (...) {
        std::lock_guard<std::mutex> lock(error_guard);
        if (!first_error) first_error = std::current_exception();
      }
    };
    // One job at a time.  A second caller (the FASTA loader works on a background thread while the main thread
    // formats signatures or matrices) does not queue behind the first: it runs its job on threads of its own.
    std::unique_lock<std::mutex> busy(busy_, std::try_to_lock);
    if (!busy.owns_lock()) {
      std::vector<std::thread> own;
      own.reserve(n_workers - 1);
      for (uint32_t t = 1; t < n_workers; ++t) own.emplace_back([&guarded, t, n_workers] { guarded(t, n_workers); });
      guarded(0u, n_workers);
      for (auto &t : own) t.join();
      if (first_error) std::rethrow_exception(first_error);
      return;
    }
    std::function<void(uint32_t, uint32_t)> job = guarded;
    {
      std::unique_lock<std::mutex> lock(m_);
      while (threads_ < n_workers - 1) {
        const uint32_t id = ++threads_;
        std::thread([this, id] { worker(id); }).detach();
      }
      job_ = &job;
      job_workers_ = n_workers;
      pending_ = n_workers - 1;
      ++generation_;
    }
    wake_.notify_all();
    guarded(0u, n_workers);
    {
      std::unique_lock<std::mutex> lock(m_);
      done_.wait(lock, [this] { return pending_ == 0; });
      job_ = nullptr;
    }
    if (first_error) std::rethrow_exception(first_error);
  }

 private:
  void worker(uint32_t id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(uint32_t, uint32_t)> *job = nullptr;
      uint32_t n = 0;
      {
        std::unique_lock<std::mutex> lock(m_);
        wake_.wait(lock, [&] { return generation_ != seen; });
        seen = generation_;
        if (id < job_workers_) { job = job_; n = job_workers_; }
      }
      if (!job) continue;
      (*job)(id, n);
      std::unique_lock<std::mutex> lock(m_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::mutex m_, busy_;
  std::condition_variable wake_, done_;
  const std::function<void(uint32_t, uint32_t)> *job_ = nullptr;
  uint32_t job_workers_ = 0, pending_ = 0, threads_ = 0;
  uint64_t generation_ = 0;
};


// Host threads that can actually run at once: the CPUs this process may be scheduled on, capped by the cgroup's CPU
// bandwidth quota (cpu.max of cgroup v2, cpu.cfs_quota_us / cpu.cfs_period_us of v1).  A container that sees 256
// CPUs but is allowed 16 CPUs' worth of time per period (the MI355X boxes of this project are set up that way) runs
// 256 busy threads for a sixteenth of each period and is throttled for the rest: more threads than the quota only
// add stalls.
inline uint32_t pa_cpu_budget() {
  static const uint32_t budget = [] {
    uint32_t n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (uint32_t)CPU_COUNT(&set);
    if (n == 0) n = std::max(1u, std::thread::hardware_concurrency());
    double quota = -1.0, period = 0.0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64] = {0};
      if (fscanf(f, "%63s %lf", q, &period) == 2 && q[0] != 'm') quota = atof(q);
      fclose(f);
    } else {
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lf", &quota) != 1) quota = -1.0;
        fclose(g);
      }
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(g, "%lf", &period) != 1) period = 0.0;
        fclose(g);
      }
    }
    if (quota > 0.0 && period > 0.0) {
      const uint32_t q = (uint32_t)((quota + period - 1.0) / period);
      if (q >= 1 && q < n) n = q;
    }
    return n;
  }();
  return budget;
}

// threads worth using for `items` units of work when each thread should get at least `grain` of them
inline uint32_t pa_host_threads(uint64_t items, uint64_t grain, uint32_t requested) {
  uint32_t nt = requested ? requested : std::min<uint32_t>(pa_cpu_budget(), 64u);
  const uint64_t by_work = items / (grain ? grain : 1) + 1;
  if (by_work < nt) nt = (uint32_t)by_work;
  return nt ? nt : 1u;
}


// Body of an extern "C" entry point that runs host-pool jobs or allocates: C++ exceptions never cross the C ABI.
// `set_error` is pa_set_error (declared by the including file); returns the body's status, -3 (PA_E_NOMEM) for
// std::bad_alloc, -1 (PA_E_INVALID) for anything else.
template <typename Body, typename SetError>
inline int pa_host_guard(const char *what, SetError &&set_error, Body &&body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    set_error("%s: out of host memory", what);
    return -3;
  } catch (const std::exception &e) {
    set_error("%s: %s", what, e.what());
    return -1;
  } catch (...) {
    set_error("%s: unknown C++ exception", what);
    return -1;
  }
}
