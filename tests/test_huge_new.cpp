// CPU harness: metamaps_amd/csrc/host/huge_new.hpp — big blocks come 2 MiB-aligned from mmap and go back to it, small ones stay with malloc,
// several threads at once, contents survive vector growth.  Prints "ok ..." or the first fault.
#include "../metamaps_amd/csrc/host/huge_new.hpp"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <thread>
#include <vector>

static size_t mappings_of_self() { std::ifstream f("/proc/self/maps"); std::string ln; size_t n = 0; while (std::getline(f, ln)) ++n; return n; }

int main() {
  { char* s = new char[100]; if (((uintptr_t)s & ((2u << 20) - 1)) == 0 && false) return 1; delete[] s; }
  // a big block: aligned, writable over its whole length, returned to the system on delete
  const size_t before = mappings_of_self();
  {
    const size_t n = (size_t)50 << 20;
    char* b = new char[n];
    if ((uintptr_t)b & (((size_t)2 << 20) - 1)) { printf("big block not 2 MiB aligned\n"); return 1; }
    memset(b, 7, n); if (b[n - 1] != 7) return 1;
    delete[] b;
  }
  // (the block waits in the spare list: the next request of its size gets it back, a much larger one does not)
  { char* c = new char[(size_t)50 << 20]; char* d = new char[(size_t)200 << 20];
    if ((uintptr_t)d & (((size_t)2 << 20) - 1)) { printf("second big block not aligned\n"); return 1; }
    delete[] c; delete[] d; }
  if (mappings_of_self() > before + 6) { printf("spare blocks pile up\n"); return 1; }
  // growth keeps contents (vector reallocation crosses the 4 MiB line)
  { std::vector<uint64_t> v; for (uint64_t i = 0; i < 3000000; ++i) v.push_back(i * 2654435761u); for (uint64_t i = 0; i < 3000000; i += 4999) if (v[i] != i * 2654435761u) { printf("contents lost in growth\n"); return 1; } }
  { std::string s; for (int i = 0; i < 2000000; ++i) s += "0123456789"; if (s.size() != 20000000 || s[19999999] != '9') { printf("string growth\n"); return 1; } }
  // threads: mixed sizes, every block filled with its own tag and checked before it goes
  std::vector<std::thread> th; std::atomic<int> bad{0};
  for (int t = 0; t < 8; ++t) th.emplace_back([t, &bad] {
    std::mt19937_64 rng((unsigned)t + 1);
    std::vector<std::pair<unsigned char*, size_t>> live;
    for (int it = 0; it < 400; ++it) {
      if (live.size() < 6 && rng() % 3) { const size_t n = rng() % 4 == 0 ? ((size_t)4 << 20) + rng() % ((size_t)12 << 20) : 1 + rng() % 100000; auto* p = new unsigned char[n]; memset(p, (int)(n & 0xff), n); live.emplace_back(p, n); }
      else if (!live.empty()) { const size_t i = rng() % live.size(); auto pr = live[i]; live[i] = live.back(); live.pop_back();
                                if (pr.first[0] != (unsigned char)(pr.second & 0xff) || pr.first[pr.second - 1] != (unsigned char)(pr.second & 0xff)) ++bad; delete[] pr.first; }
    }
    for (auto& pr : live) delete[] pr.first;
  });
  for (auto& x : th) x.join();
  if (bad) { printf("%d blocks corrupted under threads\n", (int)bad); return 1; }
  if (huge_new_detail::registry().n != 0) { printf("registry not empty at the end: %zu\n", huge_new_detail::registry().n); return 1; }
  printf("ok\n");
  return 0;
}
