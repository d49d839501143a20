// Pieces of pooled device blocks (host-side bookkeeping only: no HIP call in here, tests/test_slab.cpp drives it on the CPU).
#pragma once
#include <cstddef>
#include <iterator>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace mm {

// Slabs: index-scale blocks the device's pool holds (what an index build let go of: ~70 GB of sort buffers behind a 26.8 Gbp build) serve the
// mid-size allocations of the device's contexts instead of lying idle beside them.  Memory the driver has seen freed is cleared when it is
// handed out again (1 ms per 27 MB, MM_ALLOC_TRACE), and a device whose free memory is mostly pooled sends its worker contexts into the
// out-of-memory path, whose hipFree of the pool then stalls every context for seconds (`mapDirectly` with six worker contexts, or with batches
// of 0.5 Gbp: mapping phase 4-5 s instead of 1).  A slab is a pooled block cut by first fit (64 KiB granules, free neighbours merged); a piece
// that comes back through dev_free returns to its slab; a slab all of whose pieces are back returns to the pool when memory runs short.
struct SlabSet {
  struct Slab { char* base; size_t size; int device; std::map<size_t, size_t> free_; size_t used; };
  static constexpr size_t GRAN = (size_t)64 << 10;
  std::mutex m;
  std::vector<Slab> slabs;
  static size_t granules(size_t b) { return (b + GRAN - 1) / GRAN * GRAN; }
  void* alloc(int device, size_t bytes) {                        // nullptr: no slab of this device has room
    const size_t want = granules(bytes);
    std::lock_guard<std::mutex> lk(m);
    for (Slab& sl : slabs) {
      if (sl.device != device || sl.size - sl.used < want) continue;
      for (auto it = sl.free_.begin(); it != sl.free_.end(); ++it) {
        if (it->second < want) continue;
        const size_t off = it->first, len = it->second;
        sl.free_.erase(it);
        if (len > want) sl.free_.emplace(off + want, len - want);
        sl.used += want;
        return sl.base + off;
      }
    }
    return nullptr;
  }
  void adopt(int device, void* base, size_t size) {
    std::lock_guard<std::mutex> lk(m);
    Slab sl{(char*)base, size, device, {}, 0};
    sl.free_.emplace(0, size / GRAN * GRAN);
    slabs.push_back(std::move(sl));
  }
  bool give_back(void* p, size_t bytes) {                        // false: p is not a piece of a slab
    const size_t len = granules(bytes);
    std::lock_guard<std::mutex> lk(m);
    for (Slab& sl : slabs) {
      if ((char*)p < sl.base || (char*)p >= sl.base + sl.size) continue;
      size_t off = (size_t)((char*)p - sl.base), l = len;
      auto nx = sl.free_.lower_bound(off);
      if (nx != sl.free_.end() && off + l == nx->first) { l += nx->second; nx = sl.free_.erase(nx); }
      if (nx != sl.free_.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == off) { off = pv->first; l += pv->second; sl.free_.erase(pv); } }
      sl.free_.emplace(off, l);
      sl.used -= len;
      return true;
    }
    return false;
  }
  bool owns(const void* p) {                                     // p lies inside a slab (trace output only)
    std::lock_guard<std::mutex> lk(m);
    for (const Slab& sl : slabs) if ((const char*)p >= sl.base && (const char*)p < sl.base + sl.size) return true;
    return false;
  }
  // slabs of `device` no piece of which is out: taken off the list, for the caller to hand on (base, size)
  std::vector<std::pair<void*, size_t>> take_idle(int device) {
    std::vector<std::pair<void*, size_t>> out;
    std::lock_guard<std::mutex> lk(m);
    for (size_t i = 0; i < slabs.size();) {
      if (slabs[i].device == device && slabs[i].used == 0) { out.emplace_back(slabs[i].base, slabs[i].size); slabs.erase(slabs.begin() + (long)i); }
      else ++i;
    }
    return out;
  }
};

}  // namespace mm
