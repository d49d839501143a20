// CPU harness: metamaps_amd/csrc/host/id_set.hpp against std::set<std::string> (what the reference keeps, mapWrap.h:56).  Prints "ok <n>" or the first difference.
#include "../metamaps_amd/csrc/host/id_set.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 300000;
  std::mt19937_64 rng(7);
  IdSet fast; std::set<std::string> ref;
  const char alphabet[] = "ACGTacgt0123456789_-/:.@|";
  for (long i = 0; i < n; ++i) {
    std::string id;
    const int kind = (int)(rng() % 4);
    if (kind == 0) id = "read_" + std::to_string(rng() % 50000);                                   // many repeats
    else if (kind == 1) { id = "m54006_160504_020705/" + std::to_string(rng() % 200000) + "/ccs"; }   // long common prefix
    else if (kind == 2) { const int len = (int)(rng() % 40); for (int j = 0; j < len; ++j) id.push_back(alphabet[rng() % (sizeof alphabet - 1)]); }   // incl. the empty ID
    else { id.assign((size_t)(1 + rng() % 3), (char)('a' + rng() % 3)); }                          // a handful of very short IDs
    const bool a = fast.insert(id), b = ref.insert(id).second;
    if (a != b) { printf("MISMATCH at %ld: '%s' fast %d set %d\n", i, id.c_str(), (int)a, (int)b); return 1; }
    if (fast.size() != ref.size()) { printf("size differs at %ld\n", i); return 1; }
  }
  for (const auto& s : ref) if (!fast.contains(s)) { printf("lost '%s'\n", s.c_str()); return 1; }
  if (fast.contains("never inserted, surely")) { printf("phantom member\n"); return 1; }
  // embedded zero bytes and IDs that differ only behind the eighth byte
  const std::string z1("ab\0cd", 5), z2("ab\0ce", 5), p1 = "12345678A", p2 = "12345678B";
  if (!fast.insert(z1) || !fast.insert(z2) || fast.insert(z1) || !fast.insert(p1) || !fast.insert(p2) || fast.insert(p2)) { printf("edge cases\n"); return 1; }
  printf("ok %ld\n", n);
  return 0;
}
