// CPU harness: metamaps_amd/csrc/host/task_pool.hpp — every task of every round runs exactly once, rounds of any width, pools of several owners side by side.
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
      const size_t n = (size_t)((round * 7 + o) % 10);            // 0 .. 9: also nothing to do, one task, more than the pool is wide
      std::vector<std::atomic<int>> hit(9);
      for (auto& h : hit) h = 0;
      pool.run(n, [&](size_t t) { ++hit[t]; });
      const size_t expect = n > 8 ? 8 : n;
      for (size_t t = 0; t < 9; ++t) if (hit[t] != (t < expect ? 1 : 0)) ++bad;
    }
  });
  for (auto& t : owners) t.join();
  if (bad) { printf("%d wrong task counts\n", (int)bad); return 1; }
  printf("ok\n");
  return 0;
}
