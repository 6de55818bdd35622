// Host-side pieces of the batch path that need no HIP: the restart group's state machine around csrc/lbfgsb.cpp and the worker
// pool of a batch.  Included by pcabo_api.hip (the product) and by host_selftest.cpp (the sanitizer builds of the Makefile:
// `make asan ubsan tsan` compile this header, lbfgsb.cpp and host_entry.cpp with g++ and run them without a GPU).
#pragma once
#include "lbfgsb.h"

#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// One restart group of one run: scipy's L-BFGS-B state machine plus the memoisation of scipy's ScalarFunction (same
// logic as in pcabo_optimize_acqf, which keeps its own copy for the resident-kernel path).
struct RestartGroup {
  Lbfgsb opt;
  int q0 = 0, nq = 0, k = 0, maxiter = 200;
  std::vector<double> x, g, lo, hi, xc, gc, vc;
  double fval = 0.0, fc = 0.0;
  bool have_cache = false, active = true;
  int niter = 0, nfev = 0;
  void init(const double* ics, const double* bounds, int q0_, int nq_, int k_, int maxiter_) {
    q0 = q0_; nq = nq_; k = k_; maxiter = maxiter_;
    const int nv = nq * k;
    x.resize(nv); g.assign(nv, 0.0); lo.resize(nv); hi.resize(nv);
    for (int j = 0; j < nq; ++j)
      for (int c = 0; c < k; ++c) {
        const double l = bounds[c], h = bounds[k + c], v = ics[(size_t)(q0 + j) * k + c];
        lo[j * k + c] = l; hi[j * k + c] = h;
        x[j * k + c] = v < l ? l : (v > h ? h : v);              // columnwise_clamp / np.clip
      }
    opt.init(nv, 10, lo.data(), hi.data(), 1e7, 1e-5, 20);
  }
  void advance() {                    // until the group needs f, g at x (or stops)
    while (active) {
      const int task = opt.step(x.data(), &fval, g.data());
      if (task == LBFGSB_FG) {
        if (have_cache && memcmp(x.data(), xc.data(), x.size() * sizeof(double)) == 0) { fval = fc; g = gc; continue; }
        return;
      }
      if (task == LBFGSB_NEW_X) {
        niter += 1;
        if (niter >= maxiter) opt.stop(LBFGSB_STOP_ITER);
        else if (nfev > 15000) opt.stop(LBFGSB_STOP_FUN);
        continue;
      }
      active = false;
    }
  }
  bool absorb(const double* hVal, const double* hGrad) {      // false: NaN in the gradient
    double fs = 0.0;
    bool nan = false;
    for (int j = 0; j < nq; ++j) fs += hVal[q0 + j];
    for (int t = 0; t < nq * k; ++t) {
      const double gv = -hGrad[(size_t)q0 * k + t];
      if (gv != gv) nan = true;
      g[t] = gv;
    }
    if (nan) return false;
    fval = -fs;
    nfev += 1;
    xc = x; gc = g; fc = fval; have_cache = true;
    vc.assign(hVal + q0, hVal + q0 + nq);
    return true;
  }
};

// Worker pool of a batch: the calling thread is worker 0, n - 1 persistent threads are workers 1..n-1 (sleeping between
// calls, woken per call).  With one worker nothing leaves the calling thread (PCABO_BATCH_THREADS=1: profiler runs).
struct GangPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  std::function<void(int)> fn;
  unsigned epoch = 0;
  int pending = 0;
  bool quit = false;
  void start(int n) {
    const unsigned epoch0 = epoch;                         // (a restarted pool must not take the last call for a new one)
    for (int i = 1; i < n; ++i)
      th.emplace_back([this, i, epoch0] {
        unsigned seen = epoch0;
        for (;;) {
          std::function<void(int)> f;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return quit || epoch != seen; });
            if (quit) return;
            seen = epoch;
            f = fn;
          }
          f(i);
          { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv_done.notify_all(); }
        }
      });
  }
  void run(std::function<void(int)> f) {                   // every worker runs f(worker index); returns when all are done
    {
      std::lock_guard<std::mutex> lk(mu);
      fn = f;
      pending = (int)th.size();
      ++epoch;
    }
    cv.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    for (auto& t : th) if (t.joinable()) t.join();
    th.clear();
    quit = false;
  }
};
