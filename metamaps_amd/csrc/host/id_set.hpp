// The set of read IDs `mapDirectly` has handled (mapWrap.h:71-75, :154-157: a mapping line whose read ID was handled before stops the run).
// The reference keeps a std::set<std::string>; with 10^6 reads per file that set was the slowest thing in the writer thread (a microsecond per
// insert: a tree walk of string compares through cold memory) and with it the bound of the whole mapping phase.  Here: open addressing over
// (64-bit hash, offset into one arena of ID bytes); an ID is compared byte for byte whenever the hashes agree, so the answer is exact.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

class IdSet {
 public:
  IdSet() { slots_.assign(1024, Slot{0, 0, 0}); }
  size_t size() const { return n_; }
  // true when `id` was not in the set (and is now)
  bool insert(const char* id, size_t len) {
    if ((n_ + 1) * 2 > slots_.size()) grow();
    const uint64_t h = hash(id, len) | 1;                         // 0 marks an empty slot
    const size_t mask = slots_.size() - 1;
    for (size_t i = (size_t)(h >> 7) & mask;; i = (i + 1) & mask) {
      Slot& s = slots_[i];
      if (s.h == 0) { s.h = h; s.off = arena_.size(); s.len = (uint32_t)len; arena_.insert(arena_.end(), id, id + len); ++n_; return true; }
      if (s.h == h && s.len == len && memcmp(arena_.data() + s.off, id, len) == 0) return false;
    }
  }
  bool insert(const std::string& id) { return insert(id.data(), id.size()); }
  bool contains(const std::string& id) const {
    const uint64_t h = hash(id.data(), id.size()) | 1;
    const size_t mask = slots_.size() - 1;
    for (size_t i = (size_t)(h >> 7) & mask;; i = (i + 1) & mask) {
      const Slot& s = slots_[i];
      if (s.h == 0) return false;
      if (s.h == h && s.len == id.size() && memcmp(arena_.data() + s.off, id.data(), id.size()) == 0) return true;
    }
  }

 private:
  struct Slot { uint64_t h; uint64_t off; uint32_t len; };
  static uint64_t mix(uint64_t a, uint64_t b) { const __uint128_t p = (__uint128_t)a * b; return (uint64_t)p ^ (uint64_t)(p >> 64); }
  static uint64_t hash(const char* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    while (n >= 8) { uint64_t v; memcpy(&v, p, 8); h = mix(h ^ v, 0xE7037ED1A0B428DBull); p += 8; n -= 8; }
    uint64_t v = 0; memcpy(&v, p, n);
    return mix(h ^ v, 0x8EBC6AF09C88C6E3ull);
  }
  void grow() {
    std::vector<Slot> old; old.swap(slots_);
    slots_.assign(old.size() * 2, Slot{0, 0, 0});
    const size_t mask = slots_.size() - 1;
    for (const Slot& s : old) if (s.h) { size_t i = (size_t)(s.h >> 7) & mask; while (slots_[i].h) i = (i + 1) & mask; slots_[i] = s; }
  }
  std::vector<Slot> slots_;
  std::vector<char> arena_;
  size_t n_ = 0;
};

}  // namespace
