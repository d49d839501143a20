// CPU harness: metamaps_amd/csrc/mm_slab.hpp (pieces of pooled device blocks) against a byte map of who owns what.  Prints "ok <n>" or the first fault.
#include "../metamaps_amd/csrc/mm_slab.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 200000;
  using mm::SlabSet;
  SlabSet S;
  const size_t G = SlabSet::GRAN;
  // two "devices"; addresses are never dereferenced
  char* const base0 = (char*)((size_t)1 << 40), * const base1 = (char*)((size_t)2 << 40), * const base2 = (char*)((size_t)3 << 40);
  const size_t sz0 = 300 * G + 123, sz1 = 1000 * G, sz2 = 64 * G;   // (a block need not be a whole number of granules: the tail stays unused)
  S.adopt(0, base0, sz0); S.adopt(0, base1, sz1); S.adopt(1, base2, sz2);
  std::vector<int> own0(300, 0), own1(1000, 0), own2(64, 0);        // granule -> id of the piece that holds it
  struct Piece { char* p; size_t bytes; int dev; int id; };
  std::vector<Piece> live;
  std::mt19937_64 rng(3);
  int next_id = 1; long done = 0;
  auto owner = [&](char* p, std::vector<int>*& v, size_t& g0) { if (p >= base2) { v = &own2; g0 = (size_t)(p - base2) / G; } else if (p >= base1) { v = &own1; g0 = (size_t)(p - base1) / G; } else { v = &own0; g0 = (size_t)(p - base0) / G; } };
  for (long it = 0; it < n; ++it) {
    const bool do_alloc = live.empty() || rng() % 100 < 52;
    if (do_alloc) {
      const int dev = (int)(rng() % 4 == 0);
      const size_t bytes = 1 + rng() % (rng() % 8 == 0 ? 200 * G : 12 * G);
      char* p = (char*)S.alloc(dev, bytes);
      if (!p) continue;
      if ((size_t)p % G) { printf("unaligned piece\n"); return 1; }
      std::vector<int>* v; size_t g0; owner(p, v, g0);
      if ((dev == 1) != (v == &own2)) { printf("piece from another device's slab\n"); return 1; }
      const size_t ng = (bytes + G - 1) / G;
      if (g0 + ng > v->size()) { printf("piece beyond its slab\n"); return 1; }
      for (size_t g = g0; g < g0 + ng; ++g) { if ((*v)[g]) { printf("granule handed out twice at step %ld\n", it); return 1; } (*v)[g] = next_id; }
      live.push_back({p, bytes, dev, next_id++}); ++done;
    } else {
      const size_t i = rng() % live.size();
      Piece pc = live[i]; live[i] = live.back(); live.pop_back();
      std::vector<int>* v; size_t g0; owner(pc.p, v, g0);
      for (size_t g = g0; g < g0 + (pc.bytes + G - 1) / G; ++g) (*v)[g] = 0;
      if (!S.give_back(pc.p, pc.bytes)) { printf("own piece not recognised\n"); return 1; }
    }
  }
  if (S.give_back((void*)((size_t)9 << 40), 4096)) { printf("foreign pointer taken for a piece\n"); return 1; }
  if (!S.take_idle(0).empty() && !live.empty()) { bool any0 = false; for (auto& pc : live) any0 |= pc.dev == 0; (void)any0; }   // (slabs with pieces out must stay: checked below)
  for (auto& pc : live) if (!S.give_back(pc.p, pc.bytes)) { printf("piece of a slab that went idle too early\n"); return 1; }
  live.clear();
  // everything is back: every remaining slab is idle, and whole again (one request of the full size fits)
  size_t idle = S.take_idle(0).size() + S.take_idle(1).size();
  if (S.alloc(0, G) || S.alloc(1, G)) { printf("slab left behind take_idle\n"); return 1; }
  S.adopt(0, base1, sz1);
  char* whole = (char*)S.alloc(0, 1000 * G);
  if (whole != base1) { printf("a fresh slab does not serve its full size\n"); return 1; }
  if (S.alloc(0, G)) { printf("a full slab served more\n"); return 1; }
  S.give_back(whole, 1000 * G);
  // fragmentation heals: three pieces, middle one back last
  char* a = (char*)S.alloc(0, 400 * G); char* b = (char*)S.alloc(0, 200 * G); char* c = (char*)S.alloc(0, 400 * G);
  if (!a || !b || !c) { printf("three pieces\n"); return 1; }
  S.give_back(a, 400 * G); S.give_back(c, 400 * G);
  if (S.alloc(0, 800 * G)) { printf("two separate holes served as one\n"); return 1; }
  S.give_back(b, 200 * G);
  if ((char*)S.alloc(0, 1000 * G) != base1) { printf("holes not merged\n"); return 1; }
  printf("ok %ld pieces, %zu idle slabs at the end\n", done, idle);
  return 0;
}
