// host_pool.h -- a small persistent pool of host threads shared by the host-side bulk routines
// (pa_ani_host, the JSON and .sig writers).
#pragma once

#include <unistd.h>

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>

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
    std::function<void(uint32_t, uint32_t)> job = fn;
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
    fn(0u, n_workers);
    std::unique_lock<std::mutex> lock(m_);
    done_.wait(lock, [this] { return pending_ == 0; });
    job_ = nullptr;
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
  std::mutex m_;
  std::condition_variable wake_, done_;
  const std::function<void(uint32_t, uint32_t)> *job_ = nullptr;
  uint32_t job_workers_ = 0, pending_ = 0, threads_ = 0;
  uint64_t generation_ = 0;
};


// threads worth using for `items` units of work when each thread should get at least `grain` of them
inline uint32_t pa_host_threads(uint64_t items, uint64_t grain, uint32_t requested) {
  uint32_t nt = requested ? requested : std::min<uint32_t>(std::max(1u, std::thread::hardware_concurrency()), 64u);
  const uint64_t by_work = items / (grain ? grain : 1) + 1;
  if (by_work < nt) nt = (uint32_t)by_work;
  return nt ? nt : 1u;
}
