// rr_parallel.h -- the host-only batch entry points' worker threads: items 0..n-1 handed out one at a time (an atomic
// counter) to `threads` std::threads that live for the call.  A batch is a hundred frames of a few milliseconds each, so
// creating the threads per call costs nothing measurable and leaves no pool to manage across fork / interpreter exit.
#pragma once
#include <atomic>
#include <thread>
#include <vector>

namespace rrpar {

inline int pick_threads(int requested, int n) {
  int t = requested;
  if (t <= 0) {
    t = (int)std::thread::hardware_concurrency();
    if (t <= 0) t = 4;
    if (t > 16) t = 16;                               // (callers that know their CPU quota pass it)
  }
  if (t > n) t = n;
  return t < 1 ? 1 : t;
}

// f(i) for every i in [0, n); f must not throw
template <class F>
void parallel_for(int n, int threads, F f) {
  if (n <= 0) return;
  const int t = pick_threads(threads, n);
  if (t == 1) {
    for (int i = 0; i < n; i++) f(i);
    return;
  }
  std::atomic<int> next(0);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) return;
      f(i);
    }
  };
  std::vector<std::thread> pool;
  try {
    pool.reserve((size_t)t - 1);
    for (int k = 1; k < t; k++) pool.emplace_back(work);
  } catch (...) {                                      // no more threads to be had: the ones that exist (and this one) do the work
  }
  work();
  for (auto& th : pool) th.join();
}

}  // namespace rrpar
