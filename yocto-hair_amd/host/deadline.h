// deadline.h — a bounded wait for a launch. The reference's trace_samples cannot hang (CPU threads over rows,
// pt.cpp:1954-1989); a GPU kernel that never completes would leave yh_trace_samples inside hipStreamSynchronize for ever. The
// library therefore WAITS for a launch only up to a deadline (YHAIR_LAUNCH_TIMEOUT_S, default 1800 s — generous: the longest
// launch of the BASELINE configs takes two seconds) and on expiry returns YH_E_DEVICE and refuses further launches on that context;
// the CLIs print the error and exit(1) like print_fatal (apps/yscenetrace/yscenetrace.cpp:225-226,273). No restart, no re-exec: a
// caller that wants a retry starts a fresh process.
//
// (A launch of the usual length never gets that far: host/trace_launch.cpp first asks hipStreamQuery in a spin for as long as the context's launches have
// been taking, 50 ms at most — nothing blocks, nothing can hang — and only a wait that outlasts it goes through the forms below.)
// Two forms. BoundedCall (what the library uses): the blocking wait itself — hipStreamSynchronize, which notices the end of a launch
// within microseconds — runs on a worker thread of the context and the caller waits for it on a condition variable with the deadline;
// polling hipEventQuery between sleeps (wait_until, the first form of round 5) cost 0.6 ms per 16 ms step in detection latency
// (profiles/r05/bounded_wait_overhead.txt). After an expiry the worker is still inside the call: it is abandoned with its state (the
// process is expected to end). wait_until is kept for callers that have a query but no blocking call.
//
// Header-only and free of HIP so that tests/test_abi.py can compile it with mocked calls on a machine without a GPU.
#ifndef YH_DEADLINE_H_
#define YH_DEADLINE_H_
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace yhh {

enum { WAIT_DONE = 0, WAIT_EXPIRED = 1, WAIT_ERROR = 2 };
enum { QUERY_NOT_READY = 0, QUERY_READY = 1, QUERY_ERROR = -1 };

// Polls `query()` until it reports READY (-> WAIT_DONE) or ERROR (-> WAIT_ERROR), or until `timeout_s` seconds have passed on
// `now()` (seconds, monotonic) (-> WAIT_EXPIRED; the query is asked once more at the deadline so that a launch that finished
// during the last sleep is not reported as hung). Between polls it calls `sleep_us(n)`: 50 us while the wait is young — a 16 ms
// launch is noticed within 0.3 % of its length — then a sixteenth of the time waited so far, at most 2 ms.
template <class Query, class Now, class Sleep>
int wait_until(Query&& query, double timeout_s, Now&& now, Sleep&& sleep_us, double* waited_s = nullptr) {
  const double t0 = now();
  int          rc = WAIT_EXPIRED;
  while (true) {
    const int q = query();
    const double t = now() - t0;
    if (waited_s) *waited_s = t;
    if (q == QUERY_READY) { rc = WAIT_DONE; break; }
    if (q == QUERY_ERROR) { rc = WAIT_ERROR; break; }
    if (t >= timeout_s) { rc = WAIT_EXPIRED; break; }
    double us = t * 1e6 / 16;
    if (us < 50) us = 50;
    if (us > 2000) us = 2000;
    const double left_us = (timeout_s - t) * 1e6;
    if (us > left_us) us = left_us > 1 ? left_us : 1;
    sleep_us((long)us);
  }
  return rc;
}

// A blocking call with a deadline: run(fn, timeout) hands fn to the worker thread (started at the first use) and waits for its
// return value at most `timeout_s` seconds. WAIT_DONE: *result = fn(). WAIT_EXPIRED: fn is still running on the worker — every later
// run() returns WAIT_EXPIRED at once, and the destructor leaves the worker and its state behind instead of joining it.
class BoundedCall {
  struct State {
    std::mutex              m;
    std::condition_variable cv;
    std::function<int()>    fn;
    long                    posted = 0, finished = 0;
    int                     result = 0;
    bool                    quit = false;
  };
  std::shared_ptr<State> st_;
  std::thread            th_;
  bool                   expired_ = false;

 public:
  BoundedCall() = default;
  BoundedCall(const BoundedCall&) = delete;
  BoundedCall& operator=(const BoundedCall&) = delete;
  ~BoundedCall() {
    if (!th_.joinable()) return;
    if (expired_) {  // the worker is inside a call that never returned
      th_.detach();
      return;
    }
    {
      std::lock_guard<std::mutex> lock(st_->m);
      st_->quit = true;
    }
    st_->cv.notify_all();
    th_.join();
  }
  bool expired() const { return expired_; }
  int  run(std::function<int()> fn, double timeout_s, int* result) {
    if (expired_) return WAIT_EXPIRED;
    if (!st_) {
      st_ = std::make_shared<State>();
      std::shared_ptr<State> st = st_;
      th_ = std::thread([st] {
        std::unique_lock<std::mutex> lock(st->m);
        while (true) {
          st->cv.wait(lock, [&] { return st->quit || st->posted > st->finished; });
          if (st->quit) return;
          std::function<int()> fn = st->fn;
          lock.unlock();
          const int r = fn();
          lock.lock();
          st->result = r, st->finished = st->posted;
          st->cv.notify_all();
        }
      });
    }
    std::unique_lock<std::mutex> lock(st_->m);
    st_->fn = std::move(fn), st_->posted++;
    const long mine = st_->posted;
    st_->cv.notify_all();
    const bool ok = st_->cv.wait_for(lock, std::chrono::duration<double>(timeout_s), [&] { return st_->finished >= mine; });
    if (!ok) {
      expired_ = true;
      return WAIT_EXPIRED;
    }
    if (result) *result = st_->result;
    return WAIT_DONE;
  }
};

// YHAIR_LAUNCH_TIMEOUT_S (seconds, fractions allowed; <= 0 or unparsable: the default). Read at every wait: a test sets it per call.
inline double launch_timeout_s() {
  const char* e = getenv("YHAIR_LAUNCH_TIMEOUT_S");
  if (e && *e) {
    char*        end = nullptr;
    const double v   = strtod(e, &end);
    // (clamped: duration<double> -> the condition variable's integer nanoseconds overflows above ~9.2e9 s, and a deadline in the past would
    // poison a healthy context at its first long wait; 1e8 s = three years is "never")
    if (end != e && v > 0) return v < 1e8 ? v : 1e8;
  }
  return 1800.0;
}

}  // namespace yhh
#endif
