// CPU harness: metamaps_amd/csrc/host/task_pool.hpp — every task of every round runs exactly once, rounds of any width (also wider than the pool), a throwing task leaves the pool usable, pools of several owners side by side.
#include "../metamaps_amd/csrc/task_pool.hpp"
#include <atomic>
#include <cstdio>

int main() {
  std::atomic<int> bad{0};
  std::vector<std::thread> owners;
  for (int o = 0; o < 4; ++o) owners.emplace_back([o, &bad] {
    TaskPool pool(7);
    if (pool.width() != 8) ++bad;
    for (int round = 0; round < 600; ++round) {
      const size_t n = (size_t)((round * 7 + o) % 21);            // 0 .. 20: also nothing to do, one task, more tasks than the pool is wide
      std::vector<std::atomic<int>> hit(21);
      for (auto& h : hit) h = 0;
      pool.run(n, [&](size_t t) { ++hit[t]; });
      for (size_t t = 0; t < 21; ++t) if (hit[t] != (t < n ? 1 : 0)) ++bad;
      if (round % 50 == 7) {                                      // a task that throws — on the caller (task 0) or on a helper: the round is waited out, the
        const size_t who = (size_t)(round / 50) % 3 == 0 ? 0 : 3; //  exception arrives here, and the next round finds the pool in order
        std::vector<std::atomic<int>> ran(8);
        for (auto& h : ran) h = 0;
        bool caught = false;
        try { pool.run(8, [&](size_t t) { ++ran[t]; if (t == who) throw 42; }); } catch (int v) { caught = v == 42; }
        if (!caught) ++bad;
        for (size_t t = 0; t < 8; ++t) if (ran[t] != 1) ++bad;
      }
    }
  });
  for (auto& t : owners) t.join();
  if (bad) { printf("%d wrong task counts\n", (int)bad); return 1; }
  printf("ok\n");
  return 0;
}
