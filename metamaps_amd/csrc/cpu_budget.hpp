// How many CPUs this process may keep busy: the smallest of the hardware threads, the scheduler affinity mask and the container's CPU quota
// (cgroup v2 cpu.max, or v1 cpu.cfs_quota_us / cpu.cfs_period_us).  std::thread::hardware_concurrency() answers 256 on a box whose container is
// allowed 16 CPUs' worth of time per 100 ms period: thread pools sized from it burn the period's quota in a few tens of milliseconds and the
// kernel then stops EVERY thread of the process until the next period — found in round 5 as a 35-45 ms stall of all workers, GPU submissions
// included, every 100 ms of `mapDirectly`'s mapping phase (cpu.stat: nr_throttled).  Everything that sizes a pool asks here instead.
// MM_CPU_BUDGET=n overrides (tests; hosts whose limit is set some other way).  No HIP in here.
#pragma once
#include <sched.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace mm {

inline unsigned cpu_budget_uncached() {
  if (const char* e = getenv("MM_CPU_BUDGET")) { const int v = atoi(e); if (v > 0) return (unsigned)v; }
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min<unsigned>(n, (unsigned)c); }
  auto apply = [&](double quota, double period) { if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1.0, quota / period + 0.5)); };
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2: "max 100000" or "<quota> <period>"
    char q[64]; double period = 0;
    if (fscanf(f, "%63s %lf", q, &period) == 2 && q[0] != 'm') apply(atof(q), period);
    fclose(f);
  } else {
    double quota = -1, period = 0;
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &quota) != 1) quota = -1; fclose(g); }
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &period) != 1) period = 0; fclose(g); }
    apply(quota, period);
  }
  return std::max(1u, n);
}
inline unsigned cpu_budget() { static const unsigned n = cpu_budget_uncached(); return n; }

}  // namespace mm
