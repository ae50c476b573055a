// deadline.h — a bounded wait for a launch. The reference's trace_samples cannot hang (CPU threads over rows,
// pt.cpp:1954-1989); a GPU kernel that never completes would leave yh_trace_samples inside hipStreamSynchronize for ever. The
// library therefore POLLS the launch's end event up to a deadline (YHAIR_LAUNCH_TIMEOUT_S, default 1800 s — generous: the longest
// launch of the BASELINE configs takes two seconds) and on expiry returns YH_E_DEVICE and refuses further launches on that context;
// the CLIs print the error and exit(1) like print_fatal (apps/yscenetrace/yscenetrace.cpp:225-226,273). No restart, no re-exec: a
// caller that wants a retry starts a fresh process.
//
// Header-only and free of HIP so that tests/test_abi.py can compile it with a mocked query on a machine without a GPU.
#ifndef YH_DEADLINE_H_
#define YH_DEADLINE_H_
#include <cstdlib>

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

// YHAIR_LAUNCH_TIMEOUT_S (seconds, fractions allowed; <= 0 or unparsable: the default). Read at every wait: a test sets it per call.
inline double launch_timeout_s() {
  const char* e = getenv("YHAIR_LAUNCH_TIMEOUT_S");
  if (e && *e) {
    char*        end = nullptr;
    const double v   = strtod(e, &end);
    if (end != e && v > 0) return v;
  }
  return 1800.0;
}

}  // namespace yhh
#endif
