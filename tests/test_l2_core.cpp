// CPU unit test of the K5 integer state machine (metamaps_amd/csrc/mm_l2_core.hpp) against the oracle's
// ordered-map sliding window (oracle/orc_map.hpp SlideWindow, restating slidingMap.hpp) on random streams
// with heavy hash duplication.  Built and run by tests/test_l2_core.py with g++ (no GPU needed).
#include "../oracle/orc_map.hpp"
#include "../metamaps_amd/csrc/mm_l2_core.hpp"
#include <random>
#include <algorithm>
#include <set>
#include <vector>
#include <cstdio>

template <typename DT> int run(int rounds);
// l2_bucket (the rank table's bucket of a hash): never falls as the hash rises, stays inside the table, and spreads a sketch
// of window minima so that its longest bucket holds fewer than 32 hashes (the five-step doubling path of l2_classify8).
static int bucket_checks() {
  for (int tshift : {22, 20}) {
    const int nb = 1 << (32 - tshift);
    int prev = -1; long long n = 0;
    auto visit = [&](uint32_t h) -> int {
      const int b = mm::l2_bucket(h, tshift);
      if (b < prev || b < 0 || b >= nb) { printf("l2_bucket(%u, %d) = %d after %d (nb %d)\n", h, tshift, b, prev, nb); return 1; }
      prev = b; ++n; return 0;
    };
    for (uint64_t h = 0; h < (1ull << 32); h += 977) if (visit((uint32_t)h)) return 1;            // a sweep of the whole space
    if (visit(0xffffffffu)) return 1;
    if (mm::l2_bucket(0, tshift) != 0 || prev != nb - 1) { printf("l2_bucket ends: %d .. %d\n", mm::l2_bucket(0, tshift), prev); return 1; }
    for (uint64_t c : {1ull << 24, 1ull << 28, 1ull << 29, 1ull << 30, 1ull << 31, (1ull << 32) - 70000}) {   // every hash around a few places
      prev = mm::l2_bucket((uint32_t)(c - 66000), tshift);
      for (uint64_t h = c - 65536; h < c + 65536 && h < (1ull << 32); ++h) if (visit((uint32_t)h)) return 1;
    }
    std::mt19937_64 rng(99);
    for (int w : {6, 8, 16}) for (int L : {5000, 10000, 50000}) {
      if (tshift == 22 && L > 12500) continue;                  // (the 1 024-bucket table serves sketches up to 3 072 hashes)
      int worst = 0;
      for (int rep = 0; rep < 20; ++rep) {
        std::vector<uint32_t> kh(L); for (auto& x : kh) x = (uint32_t)std::min<uint64_t>(rng() >> 32, rng() >> 32);   // strand minimum
        std::set<uint32_t> sk;
        for (int i = 0; i + w <= L; ++i) sk.insert(*std::min_element(kh.begin() + i, kh.begin() + i + w));
        std::vector<int> cnt(nb, 0);
        for (uint32_t h : sk) worst = std::max(worst, ++cnt[mm::l2_bucket(h, tshift)]);
      }
      if (worst >= 32) { printf("longest bucket %d (w %d, %d bp, tshift %d)\n", worst, w, L, tshift); return 1; }
    }
    printf("l2_bucket tshift %d: %lld hashes in order, inside the table\n", tshift, n);
  }
  return 0;
}
int main(int argc, char** argv) {
  int rounds = argc > 1 ? atoi(argv[1]) : 300;
  if (bucket_checks()) return 1;
  if (run<uint16_t>(rounds)) return 1;
  return run<uint8_t>(rounds);
}
template <typename DT> int run(int rounds) {
  std::mt19937_64 rng(12345);
  long long checked = 0, dups = 0;
  for (int it = 0; it < rounds; ++it) {
    int s = 1 + rng() % 200;
    uint32_t space = s + 50 + rng() % 2000;                      // small hash space => many duplicates
    std::set<uint32_t> qs; while ((int)qs.size() < s) qs.insert(rng() % space);
    orc::Query Q; Q.sketch = s;
    for (uint32_t h : qs) Q.mins.push_back(orc::Mz{h, 0, 0, 1});
    std::vector<uint32_t> Qh(qs.begin(), qs.end());
    int M = 50 + rng() % 1500;
    std::vector<orc::Mz> X(M);
    for (int i = 0; i < M; ++i) X[i] = orc::Mz{(uint32_t)(rng() % space), 0, i, 1};
    std::vector<DT> D(s, 0); std::vector<uint32_t> mt((s + 31) / 32, 0);
    mm::L2StateT<DT> S{Qh.data(), D.data(), mt.data(), s, 0, 0, 0, 0};
    mm::l2_reset(S);
    orc::SlideWindow ref(Q);
    int b = 0, e = 0;                                          // window [b,e)
    auto other_alive = [&](int j) { for (int i = b; i < e; ++i) if (i != j && X[i].hash == X[j].hash) return true; return false; };
    for (int step = 0; step < 3 * M; ++step) {
      bool ins = (e < M) && (b == e || (rng() % 100) < 55);
      if (!ins && b == e) break;
      if (ins) {
        int code = mm::l2_classify(Qh.data(), s, X[e].hash);
        bool dup = other_alive(e);                             // window is [b,e) here: e itself not inside yet
        dups += dup;
        if (!dup) { if (code >= 0) mm::l2_add_matched(S, code); else mm::l2_add_wonly(S, -code - 1); }
        ref.insert(X[e]); ++e;
      } else {
        int code = mm::l2_classify(Qh.data(), s, X[b].hash);
        bool dup = other_alive(b);
        if (!dup) { if (code >= 0) mm::l2_del_matched(S, code); else mm::l2_del_wonly(S, -code - 1); }
        ref.erase(X[b]); ++b;
      }
      ++checked;
      if (S.overflow) { printf("unexpected overflow\n"); return 1; }
      if (S.shared != ref.shared) { printf("MISMATCH it=%d step=%d shared=%d ref=%d s=%d\n", it, step, S.shared, ref.shared, s); return 1; }
    }
  }
  printf("ok %lld events, %lld duplicate inserts (D width %d)\n", checked, dups, (int)sizeof(DT));
  return 0;
}
