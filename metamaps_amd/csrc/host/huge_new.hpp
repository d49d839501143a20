// operator new / delete of the host program: blocks from 4 MiB on come from mmap, aligned to 2 MiB and marked MADV_HUGEPAGE.
// The program's big buffers (read batches, the text of a mappings file, line tables, output text: hundreds of MB each) are written once and
// read once; with 4 KiB pages their cost was the page faults (a 0.5 GB buffer: 122 000 faults on first touch) and, at exit, the teardown of
// 2.5 GB of page tables (0.26 s of `classify`).  Transparent huge pages in "madvise" mode serve such a block with 512 times fewer faults.
// Everything smaller, and everything when MM_CLI_NO_HUGE is set, goes to malloc as before.  Include in exactly one translation unit.
// (new[] does not promise cleared memory, and a block that comes back from the spare list is not cleared.)
#pragma once
#include <sys/mman.h>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <unordered_map>

namespace huge_new_detail {

constexpr size_t HUGE = (size_t)2 << 20, FROM = (size_t)4 << 20;

struct Registry {                                                // blocks handed out by the mmap path: pointer -> (mapping base, mapping length)
  std::mutex m;
  // (a plain array: the registry must not allocate through operator new while operator new holds its lock)
  struct Entry { void* p; void* base; size_t len; };
  Entry e[4096];
  size_t n = 0;
  bool add(void* p, void* base, size_t len) { std::lock_guard<std::mutex> lk(m); if (n == 4096) return false; e[n++] = Entry{p, base, len}; return true; }
  bool take(void* p, Entry* out) {
    std::lock_guard<std::mutex> lk(m);
    for (size_t i = 0; i < n; ++i) if (e[i].p == p) { *out = e[i]; e[i] = e[--n]; return true; }
    return false;
  }
};
inline Registry& registry() { static Registry r; return r; }
inline bool enabled() { static const bool on = getenv("MM_CLI_NO_HUGE") == nullptr; return on; }

// Blocks given back wait here for the next request they fit (at most twice its size) instead of going back to the system: the program
// allocates and frees buffers of the same few sizes batch after batch (record tables, output text), and a block that is mapped anew is
// faulted in and cleared anew — with every thread of the process queueing on the address-space lock meanwhile.  At most KEEP bytes wait.
struct Spare {
  std::mutex m;
  struct Entry { void* p; void* base; size_t len; };
  static constexpr size_t KEEP = (size_t)4 << 30, SLOTS = 64;
  Entry e[SLOTS]; size_t n = 0, bytes = 0;
  bool take(size_t len, Entry* out) {
    std::lock_guard<std::mutex> lk(m);
    size_t best = SLOTS;
    for (size_t i = 0; i < n; ++i) if (e[i].len >= len && e[i].len <= 2 * len && (best == SLOTS || e[i].len < e[best].len)) best = i;
    if (best == SLOTS) return false;
    *out = e[best]; bytes -= e[best].len; e[best] = e[--n];
    return true;
  }
  bool put(const Entry& en) {                                    // false: no room, the caller unmaps
    std::lock_guard<std::mutex> lk(m);
    if (n == SLOTS || bytes + en.len > KEEP) return false;
    e[n++] = en; bytes += en.len;
    return true;
  }
};
inline Spare& spare() { static Spare s; return s; }

inline void* big_alloc(size_t size) {
  const size_t len = ((size + HUGE - 1) & ~(HUGE - 1)) + HUGE;   // (room to align the start)
  { Spare::Entry sp; if (spare().take(len, &sp)) { if (registry().add(sp.p, sp.base, sp.len)) return sp.p; munmap(sp.base, sp.len); } }
  void* base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (base == MAP_FAILED) return nullptr;
  void* p = (void*)(((uintptr_t)base + HUGE - 1) & ~(uintptr_t)(HUGE - 1));
  (void)madvise(p, len - (size_t)((char*)p - (char*)base), MADV_HUGEPAGE);
  if (!registry().add(p, base, len)) { munmap(base, len); return nullptr; }
  return p;
}
inline void* alloc(size_t size) {
  if (size >= FROM && enabled()) if (void* p = big_alloc(size)) return p;
  if (void* p = malloc(size ? size : 1)) return p;
  throw std::bad_alloc();
}
inline void release(void* p) noexcept {
  if (!p) return;
  if (((uintptr_t)p & (HUGE - 1)) == 0) {
    Registry::Entry en{nullptr, nullptr, 0};
    if (registry().take(p, &en)) { if (!spare().put(Spare::Entry{en.p, en.base, en.len})) munmap(en.base, en.len); return; }
  }
  free(p);
}

}  // namespace huge_new_detail

void* operator new(size_t n) { return huge_new_detail::alloc(n); }
void* operator new[](size_t n) { return huge_new_detail::alloc(n); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { try { return huge_new_detail::alloc(n); } catch (...) { return nullptr; } }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { try { return huge_new_detail::alloc(n); } catch (...) { return nullptr; } }
void operator delete(void* p) noexcept { huge_new_detail::release(p); }
void operator delete[](void* p) noexcept { huge_new_detail::release(p); }
void operator delete(void* p, size_t) noexcept { huge_new_detail::release(p); }
void operator delete[](void* p, size_t) noexcept { huge_new_detail::release(p); }
void operator delete(void* p, const std::nothrow_t&) noexcept { huge_new_detail::release(p); }
void operator delete[](void* p, const std::nothrow_t&) noexcept { huge_new_detail::release(p); }
