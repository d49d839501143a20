// A handful of helper threads that stay: run(n, fn) calls fn(0) ... fn(n-1), fn(0) on the calling thread, the others on helpers that wait for
// work between calls.  `mapDirectly` formats the text of every batch with eight threads, and every sequence upload packs its bases with up to
// thirty-two; started anew per call (std::thread), the slowest of them came in 2-3 times behind the caller's own part — stacks mapped, caches
// cold, now and then 30-50 ms before one got going at all.  No HIP in here (tests/test_task_pool.cpp drives it on the CPU).
#pragma once
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class TaskPool {
 public:
  explicit TaskPool(size_t helpers) { for (size_t i = 0; i < helpers; ++i) th_.emplace_back([this, i] { loop(i); }); }
  ~TaskPool() { { std::lock_guard<std::mutex> lk(m_); stop_ = true; } cv_.notify_all(); for (auto& t : th_) t.join(); }
  TaskPool(const TaskPool&) = delete;
  TaskPool& operator=(const TaskPool&) = delete;
  size_t width() const { return th_.size() + 1; }
  // fn(t) for every t < n, each exactly once; returns when all are done.  Participant p (the caller is 0, helper i is i + 1) takes the tasks
  // p, p + width(), p + 2 width() ...: n may exceed width().  An exception thrown by a task is caught where it is thrown, the round is
  // always waited out (the helpers hold a pointer to the caller's fn), and the first one caught is rethrown here.  One run at a time
  // (the pool belongs to one thread).
  void run(size_t n, const std::function<void(size_t)>& fn) {
    if (n == 0) return;
    const size_t helpers = n - 1 < th_.size() ? n - 1 : th_.size();
    if (helpers) { std::lock_guard<std::mutex> lk(m_); fn_ = &fn; n_ = n; pending_ = helpers; err_ = nullptr; ++round_; }
    if (helpers) cv_.notify_all();
    std::exception_ptr mine;
    try { for (size_t t = 0; t < n; t += width()) fn(t); } catch (...) { mine = std::current_exception(); }
    if (helpers) {
      std::unique_lock<std::mutex> lk(m_);
      done_.wait(lk, [&] { return pending_ == 0; });
      fn_ = nullptr;
      if (!mine) mine = err_;
      err_ = nullptr;
    }
    if (mine) std::rethrow_exception(mine);
  }

 private:
  void loop(size_t i) {                                          // helper i takes task i + 1 of every round that has one
    size_t seen = 0;
    for (;;) {
      const std::function<void(size_t)>* f = nullptr;
      size_t n = 0;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || round_ != seen; });
        if (stop_) return;
        seen = round_;
        n = n_;
        if (i + 1 < n_) f = fn_;
      }
      if (f) {
        std::exception_ptr e;
        try { for (size_t t = i + 1; t < n; t += width()) (*f)(t); } catch (...) { e = std::current_exception(); }
        std::lock_guard<std::mutex> lk(m_);
        if (e && !err_) err_ = e;
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_; std::condition_variable cv_, done_;
  const std::function<void(size_t)>* fn_ = nullptr;
  std::exception_ptr err_;
  size_t n_ = 0, pending_ = 0, round_ = 0; bool stop_ = false;
};
