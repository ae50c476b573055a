// CPU-side unit test of the launch deadline (yocto-hair_amd/host/deadline.h) with a MOCKED event query and a fake clock:
// compiled and run by tests/test_abi.py::test_launch_deadline_logic (g++ only, no HIP, no GPU).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "deadline.h"

static int fails = 0;
#define CHECK(cond)                                              \
  do {                                                           \
    if (!(cond)) {                                               \
      printf("FAILED line %d: %s\n", __LINE__, #cond);           \
      fails++;                                                   \
    }                                                            \
  } while (0)

int main() {
  using namespace yhh;
  {  // an event that NEVER signals: the wait ends at the deadline, not before, and not much after
    double clock = 100.0, waited = -1;
    long   polls = 0, slept_us = 0;
    int rc = wait_until([&] { polls++; return QUERY_NOT_READY; }, 2.5, [&] { return clock; },
        [&](long us) { slept_us += us, clock += us * 1e-6; }, &waited);
    CHECK(rc == WAIT_EXPIRED);
    CHECK(waited >= 2.5 && waited < 2.5 + 0.003);   // overshoot bounded by one sleep (<= 2 ms)
    CHECK(polls > 100 && polls < 100000);          // it sleeps between polls (no busy spin), and it polls more than a handful of times
    CHECK(slept_us >= 2490000);
  }
  {  // an event that signals at the 7th poll: done, long before the deadline
    double clock = 0, waited = -1;
    int    polls = 0;
    int rc = wait_until([&] { return ++polls >= 7 ? QUERY_READY : QUERY_NOT_READY; }, 1800.0, [&] { return clock; },
        [&](long us) { clock += us * 1e-6; }, &waited);
    CHECK(rc == WAIT_DONE && polls == 7);
    CHECK(waited < 0.01);                           // six sleeps of 50 us while the wait is young
  }
  {  // ready at once: no sleep at all
    long slept = 0;
    int rc = wait_until([] { return QUERY_READY; }, 1.0, [] { return 5.0; }, [&](long us) { slept += us; });
    CHECK(rc == WAIT_DONE && slept == 0);
  }
  {  // a query error is reported as such, at once
    int polls = 0;
    int rc = wait_until([&] { polls++; return QUERY_ERROR; }, 1.0, [] { return 0.0; }, [](long) {});
    CHECK(rc == WAIT_ERROR && polls == 1);
  }
  {  // a launch that finishes during the LAST sleep is not reported as hung: the query is asked again at the deadline
    double clock = 0;
    int rc = wait_until([&] { return clock >= 0.9999 ? QUERY_READY : QUERY_NOT_READY; }, 1.0, [&] { return clock; },
        [&](long us) { clock += us * 1e-6; });
    CHECK(rc == WAIT_DONE);
  }
  {  // sleeps back off with the time waited: 50 us young, 1/16 of the wait later, 2 ms at most
    double clock = 0;
    std::vector<long> sleeps;
    wait_until([] { return QUERY_NOT_READY; }, 10.0, [&] { return clock; }, [&](long us) { sleeps.push_back(us), clock += us * 1e-6; });
    CHECK(sleeps.front() == 50);
    long mx = 0;
    for (long s : sleeps) mx = s > mx ? s : mx;
    CHECK(mx == 2000);
  }
  {  // the environment variable: a positive number of seconds, else the default
    setenv("YHAIR_LAUNCH_TIMEOUT_S", "0.25", 1);
    CHECK(launch_timeout_s() == 0.25);
    setenv("YHAIR_LAUNCH_TIMEOUT_S", "-3", 1);
    CHECK(launch_timeout_s() == 1800.0);
    setenv("YHAIR_LAUNCH_TIMEOUT_S", "soon", 1);
    CHECK(launch_timeout_s() == 1800.0);
    setenv("YHAIR_LAUNCH_TIMEOUT_S", "1e12", 1);  // "never": clamped to what the wait's nanosecond clock can hold
    CHECK(launch_timeout_s() == 1e8);
    {  // ... and a healthy call under that deadline is DONE, not expired at once (an overflowed deadline lies in the past)
      BoundedCall bc;
      int         r = -1;
      CHECK(bc.run([] { std::this_thread::sleep_for(std::chrono::milliseconds(80)); return 3; }, launch_timeout_s(), &r) == WAIT_DONE && r == 3 && !bc.expired());
    }
    setenv("YHAIR_LAUNCH_TIMEOUT_S", "inf", 1);
    CHECK(launch_timeout_s() == 1e8);
    unsetenv("YHAIR_LAUNCH_TIMEOUT_S");
    CHECK(launch_timeout_s() == 1800.0);
  }
  {  // BoundedCall: a blocking call that returns is DONE with its value, again and again on the same worker
    BoundedCall bc;
    int         r = -1, calls = 0;
    CHECK(bc.run([&] { calls++; return 7; }, 5.0, &r) == WAIT_DONE && r == 7);
    CHECK(bc.run([&] { calls++; std::this_thread::sleep_for(std::chrono::milliseconds(20)); return 9; }, 5.0, &r) == WAIT_DONE && r == 9);
    CHECK(calls == 2 && !bc.expired());
  }  // (the destructor joins the idle worker)
  {  // ... one that NEVER returns (a kernel that never completes): EXPIRED at the deadline, every later call at once, and the destructor does not wait for it
    auto t0 = std::chrono::steady_clock::now();
    auto secs = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    {
      BoundedCall bc;
      int         r = -1;
      CHECK(bc.run([] { std::this_thread::sleep_for(std::chrono::hours(1)); return 0; }, 0.2, &r) == WAIT_EXPIRED);
      CHECK(secs() >= 0.2 && secs() < 1.0 && bc.expired() && r == -1);
      const double t1 = secs();
      CHECK(bc.run([] { return 1; }, 5.0, &r) == WAIT_EXPIRED && secs() - t1 < 0.05);
    }
    CHECK(secs() < 1.5);
  }
  {  // the latency of a wait that is already satisfied: two thread hand-overs, far below a millisecond on average
    BoundedCall bc;
    int         r = 0;
    bc.run([] { return 0; }, 1.0, &r);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 200; k++) bc.run([] { return 0; }, 1.0, &r);
    const double per = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 200;
    CHECK(per < 1e-3);
    printf("BoundedCall round trip: %.1f us\n", per * 1e6);
  }
  printf(fails ? "deadline: %d checks FAILED\n" : "deadline: all checks passed\n", fails);
  return fails ? 1 : 0;
}
