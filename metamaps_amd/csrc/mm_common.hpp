// Shared host/device definitions of libmetamaps_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <deque>
#include <vector>
#include <stdexcept>
#include <memory>
#include <chrono>
#include <map>
#include <iterator>
#include <mutex>
#include <atomic>
#include <algorithm>
#include "../../include/metamaps_hip.h"
#include "mm_slab.hpp"
#include "task_pool.hpp"
#include "cpu_budget.hpp"

namespace mm {

// ---- error plumbing: internal code throws, the C ABI layer converts to a status + message ----------
struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& m) : std::runtime_error(m), status(st) {}
};
#define MM_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      throw mm::Error(_e == hipErrorOutOfMemory ? MM_ERR_NOMEM : MM_ERR_DEVICE,              \
                      std::string(#expr) + ": " + hipGetErrorString(_e));                    \
  } while (0)
#define MM_REQUIRE(cond, st, msg) \
  do { if (!(cond)) throw mm::Error((st), (msg)); } while (0)
#define MM_KERNEL_CHECK() MM_HIP(hipGetLastError())

// ---- waiting for a stream ------------------------------------------------------------------------------
// hipStreamSynchronize spins: a host thread per context burns a CPU while its kernels run.  On a host with CPUs to spare that is the lowest
// latency; in a container with a small CPU quota (cpu_budget.hpp) four spinning workers are a quarter of the quota gone, and once the quota of
// a 100 ms period is used up the kernel stops every thread of the process.  So when the budget is small (<= 32 CPUs) a wait records an event
// created with hipEventBlockingSync and sleeps on it instead (an interrupt wakes the thread; 10-30 us later than a spin would have noticed).
// MM_SYNC=spin|block overrides.  Every wait of the library goes through here.
inline bool sync_blocking() {
  static const bool b = [] { const char* e = getenv("MM_SYNC"); if (e && *e) return strcmp(e, "block") == 0; return cpu_budget() <= 32; }();
  return b;
}
// The event a wait sleeps on belongs to the stream: a context registers one with its stream when it is created (mm_ctx_create, aux_ready) and takes
// it back when it goes (mm_ctx_destroy) — no event per host thread (the CLI's worker, pool and on_each threads are created per run and never destroyed
// theirs), no hipGetDevice per wait, and a thread whose current device is another one (the allocator trimming a foreign context's cache) sleeps too
// instead of falling back to the spin.  Streams nobody registered (none in the product) keep the thread-local event.
struct StreamEvents {
  std::mutex mu;
  std::map<hipStream_t, hipEvent_t> ev;
  static StreamEvents& get() { static StreamEvents* s = new StreamEvents; return *s; }   // (never destroyed: contexts may outlive static destruction)
};
inline void stream_event_register(hipStream_t st) {                // (the stream's device is current)
  if (!sync_blocking()) return;
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
  StreamEvents& S = StreamEvents::get();
  std::lock_guard<std::mutex> g(S.mu);
  S.ev[st] = e;
}
inline void stream_event_unregister(hipStream_t st) {
  StreamEvents& S = StreamEvents::get();
  hipEvent_t e = nullptr;
  { std::lock_guard<std::mutex> g(S.mu); auto it = S.ev.find(st); if (it != S.ev.end()) { e = it->second; S.ev.erase(it); } }
  if (e) (void)hipEventDestroy(e);
}
inline hipError_t stream_sync(hipStream_t st) {
  if (!sync_blocking()) return hipStreamSynchronize(st);
  hipEvent_t ev = nullptr;
  { StreamEvents& S = StreamEvents::get(); std::lock_guard<std::mutex> g(S.mu); auto it = S.ev.find(st); if (it != S.ev.end()) ev = it->second; }
  if (!ev) {                                                     // a stream without a registered event: one event per thread and device, as before
    static thread_local hipEvent_t tev = nullptr; static thread_local int ev_dev = -1;
    int dev = 0; (void)hipGetDevice(&dev);
    if (!tev || ev_dev != dev) {
      if (tev) (void)hipEventDestroy(tev);
      tev = nullptr;
      const hipError_t c = hipEventCreateWithFlags(&tev, hipEventBlockingSync | hipEventDisableTiming);
      if (c != hipSuccess) { tev = nullptr; (void)hipGetLastError(); return hipStreamSynchronize(st); }
      ev_dev = dev;
    }
    ev = tev;
  }
  // (two threads never wait for the same context at once: a context is driven by one host thread at a time, include/metamaps_hip.h)
  if (hipEventRecord(ev, st) != hipSuccess) { (void)hipGetLastError(); return hipStreamSynchronize(st); }
  const hipError_t w = hipEventSynchronize(ev);
  if (w != hipSuccess) { (void)hipGetLastError(); return hipStreamSynchronize(st); }
  return hipSuccess;
}

// ---- device memory -----------------------------------------------------------------------------------
// Per-context caching allocator.  A batch needs dozens of temporaries; hipMalloc/hipFree synchronise the
// device, and ROCm 7.2's stream-ordered pool (hipMallocAsync) gave wrong results here when the library ran
// on the system HIP runtime (it only behaved under the older runtime that PyTorch bundles), so blocks are
// recycled by hand: a context owns ONE stream, every kernel and copy is issued on it, and a freed block
// handed to a later allocation is therefore only touched by work that is stream-ordered after its previous
// user.  Index-scale buffers (>= 8 GiB) bypass the cache.
void big_pool_trim(int device);
// Every device block of the library comes from dev_malloc and goes back through dev_free, so that the bytes it holds per device are known.
// MM_DEVICE_BYTES_CAP=<bytes> (a TEST HOOK) makes the library behave as if every device had only that much memory: dev_malloc fails with
// hipErrorOutOfMemory beyond it and dev_mem_info reports it — the CLI's resident / sharded / streamed decision and the allocator's
// out-of-memory paths are then exercised on a small input (tests/test_gpu_cli.py) instead of on a reference larger than 288 GB.
struct DevMeter {
  std::atomic<long long> used[64];
  long long cap;
  DevMeter() { for (auto& u : used) u = 0; const char* e = getenv("MM_DEVICE_BYTES_CAP"); cap = e ? atoll(e) : 0; }
};
inline DevMeter& dev_meter() { static DevMeter m; return m; }
inline int dev_current() { int d = 0; (void)hipGetDevice(&d); return d < 0 || d >= 64 ? 0 : d; }
inline hipError_t dev_malloc(void** p, size_t bytes) {
  DevMeter& m = dev_meter();
  const int d = dev_current();
  if (m.cap > 0 && m.used[d].load() + (long long)bytes > m.cap) { *p = nullptr; return hipErrorOutOfMemory; }
  const hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) m.used[d] += (long long)bytes;
  return e;
}
// (SlabSet: mm_slab.hpp)
inline SlabSet& slab_set() { static SlabSet s; return s; }
inline void dev_free(void* p, size_t bytes) {
  if (!p) return;
  if (slab_set().give_back(p, bytes)) return;                    // (a piece of a pooled block: the block stays the device's)
  // the bytes go off the account of the device the block LIVES on (hipMalloc charged the device current at that time): the thread that frees —
  // a context's destructor on the CLI's main thread, another context trimming this one's cache — may have any device current
  int d = dev_current();
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) == hipSuccess && at.device >= 0 && at.device < 64) d = at.device; else (void)hipGetLastError();
  dev_meter().used[d] -= (long long)bytes; (void)hipFree(p);
}
inline hipError_t dev_mem_info(size_t* fr, size_t* tot) {
  const hipError_t e = hipMemGetInfo(fr, tot);
  DevMeter& m = dev_meter();
  if (e == hipSuccess && m.cap > 0) {
    const long long left = std::max(0LL, m.cap - m.used[dev_current()].load());
    *tot = std::min<size_t>(*tot, (size_t)m.cap); *fr = std::min<size_t>(*fr, (size_t)left);
  }
  return e;
}
inline std::string oom_text(size_t want, hipError_t e) {         // what the device looks like when an allocation fails for good
  size_t fr = 0, tot = 0; (void)dev_mem_info(&fr, &tot);
  return std::string("hipMalloc of ") + std::to_string(want) + " bytes: " + hipGetErrorString(e) + " (device: " + std::to_string(fr >> 20) + " MiB free of " + std::to_string(tot >> 20) + ")";
}
struct DevAlloc;
// Free blocks cached by one context are memory another context of the same device may need (worker contexts beside the one that built
// the indexes): an allocation that fails for lack of memory asks every other context's cache to go back to the driver before it gives up.
void alloc_register(DevAlloc* a, int device);
void alloc_unregister(DevAlloc* a);
void alloc_trim_others(DevAlloc* self, int device);
constexpr size_t SLAB_FROM_BYTES = (size_t)1 << 20;             // smaller requests stay with the driver (they come from its own small pools, quickly)
void* slab_piece(int device, size_t bytes);                      // a piece of a pooled index-scale block of the device, or nullptr (defined behind BigPool)
size_t big_pool_bytes(int device);
void* big_pool_rescue(DevAlloc* self, int device, size_t bytes, size_t* got, bool caches_first);
bool big_pool_trim_until(int device, size_t need);
struct DevAlloc {
  hipStream_t stream = nullptr;
  int device = -1;                           // set by alloc_register
  std::mutex m;                              // the cache: its own context's thread, and any thread that trims it when the device is full
  // set for the duration of a device-filling index build (mm_index.hip): every index-scale allocation first hands the cached blocks back
  // and index-scale blocks go straight to and from the driver, so that the build's memory is returned WHILE it runs.  Returned in one
  // piece afterwards (~100 GB), it came back as a 1 s stall of a mapping step a few seconds later, twice in four bench runs (round 3).
  bool eager = false;
  bool in_build = false;                     // an index build runs on this context: its temporaries do not cut into pooled blocks the build itself is about to ask for
  std::multimap<size_t, void*> cache;        // size -> free block
  size_t cached_bytes = 0;
  static size_t round_up(size_t b) {
    if (b < 4096) return 4096;
    int lg = 63 - __builtin_clzll((unsigned long long)b);
    size_t gran = (size_t)1 << (lg > 3 ? lg - 3 : 0);           // <= 12.5 % slack
    return (b + gran - 1) / gran * gran;
  }
  void trim() {
    std::lock_guard<std::mutex> lk(m);
    if (cache.empty()) return;
    (void)mm::stream_sync(stream);
    for (auto& kv : cache) dev_free(kv.second, kv.first);
    cache.clear(); cached_bytes = 0;
  }
  // hands the largest cached blocks back to the driver until at most `keep` bytes stay cached (after an index build: its temporaries
  // are worth keeping for the next chunk's build, not a hundred gigabytes of them beside the mapping buffers of other contexts)
  void trim_to(size_t keep) {
    std::lock_guard<std::mutex> lk(m);
    if (cached_bytes <= keep) return;
    (void)mm::stream_sync(stream);
    while (cached_bytes > keep && !cache.empty()) { auto it = std::prev(cache.end()); dev_free(it->second, it->first); cached_bytes -= it->first; cache.erase(it); }
  }
  void* get(size_t bytes, size_t* got) {
    const size_t want = round_up(bytes);
    {
    std::lock_guard<std::mutex> lk(m);
    auto it = cache.lower_bound(want);
    // A cached block serves a request it is at most 60 % too large for, and what comes from the driver (from 64 MiB on) is asked for a
    // quarter larger than needed: read batches differ (more or fewer seed hits, candidates, records), and on this runtime memory the
    // driver has seen freed is cleared when it is handed out again — 1 ms per 27 MB, up to seconds when a large region is due
    // (MM_ALLOC_TRACE, round 3: one 738 MB allocation of a bench step took 2.0 s).  With headroom the buffers of the first batches also
    // serve the later ones, and a process in steady state does not go to the driver at all.
    if (it != cache.end() && it->first <= want + want / 4 + (want >= ((size_t)256 << 10) ? want * 7 / 20 : 0)) {
      void* p = it->second; *got = it->first; cached_bytes -= it->first; cache.erase(it); return p;
    }
    }
    size_t ask = want;
    // The last GiB of the device stays with the runtime: a device filled to the brim by hipMalloc lets a later kernel launch fail inside the
    // runtime (its own allocations: HSA_STATUS_ERROR_OUT_OF_RESOURCES, the queue is aborted and the process with it — seen with three worker
    // contexts beside four resident chunk indexes); a request that would take it is treated as one that failed for lack of memory.
    const size_t RUNTIME_RESERVE = dev_meter().cap > 0 ? 0 : (size_t)1 << 30;   // (under the test hook MM_DEVICE_BYTES_CAP the "device" ends at the cap, far below the real one)
    bool refuse = false;
    if (want >= ((size_t)64 << 20)) {                            // (headroom only while a fifth of the device is free: resident chunk indexes can leave less)
      size_t fr = 0, tot = 0;
      if (dev_mem_info(&fr, &tot) == hipSuccess) {
        if (fr > tot / 5) ask = round_up(want + want / 4);
        refuse = fr < want + RUNTIME_RESERVE;
      }
    } else if (want >= ((size_t)256 << 10)) {
      // buffers of 256 KiB .. 64 MiB — per-read and per-candidate arrays — get the same quarter of headroom (no driver query: they cannot fill a device): without it every batch
      // with a few per cent more candidates than its worker context had seen went to the driver for ~50 blocks (round 6, tools/alloc_probe.sh: 141 driver allocations, 1.3 GB,
      // inside the bench's twelve timed steps)
      static const bool mid_headroom = getenv("MM_ALLOC_NO_MID_HEADROOM") == nullptr;
      if (mid_headroom) ask = round_up(want + want / 4);
    }
    void* p = nullptr;
    static const bool trace = getenv("MM_ALLOC_TRACE") != nullptr;     // every block that comes from the driver, with its cost
    static const bool use_slabs = getenv("MM_NO_SLABS") == nullptr;
    if (use_slabs && !eager && !in_build && want >= SLAB_FROM_BYTES) {
      const auto ts0 = std::chrono::steady_clock::now();
      if (void* q = slab_piece(device, ask)) { *got = SlabSet::granules(ask); if (trace) fprintf(stderr, "MM_ALLOC_TRACE slab piece %zu bytes %.3f ms at %.1f ms\n", *got,
                                                                                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count(),
                                                                                               std::chrono::duration<double, std::milli>(std::chrono::system_clock::now().time_since_epoch()).count()); return q; }
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = refuse ? hipErrorOutOfMemory : dev_malloc(&p, ask);
    size_t granted = ask;
    if (trace) fprintf(stderr, "MM_ALLOC_TRACE hipMalloc %zu bytes %.3f ms at %.1f ms\n", ask, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                       std::chrono::duration<double, std::milli>(std::chrono::system_clock::now().time_since_epoch()).count());
    if (e == hipErrorOutOfMemory) {
      if (trace) { int dv0 = 0; (void)hipGetDevice(&dv0); fprintf(stderr, "MM_ALLOC_TRACE out of memory at a request of %zu bytes (%zu bytes pooled)\n", want, big_pool_bytes(dv0)); }
      (void)hipGetLastError(); int dv = 0; (void)hipGetDevice(&dv);
      static const bool rescue = getenv("MM_NO_POOL_RESCUE") == nullptr;
      if (rescue && use_slabs) { size_t g = 0; if (void* q = big_pool_rescue(this, dv, want, &g, false)) { *got = g; if (trace) fprintf(stderr, "MM_ALLOC_TRACE ... served from the pool (%zu bytes)\n", g); return q; } }
      trim(); alloc_trim_others(this, dv);                        // (the caches first: their pieces of pooled blocks go back to the blocks)
      if (!(rescue && big_pool_trim_until(dv, want + RUNTIME_RESERVE))) big_pool_trim(dv);
      granted = want;
      size_t fr = 0, tot = 0;
      if (refuse && dev_mem_info(&fr, &tot) == hipSuccess && fr < want + RUNTIME_RESERVE) e = hipErrorOutOfMemory;   // still not there with every cache given back
      else e = dev_malloc(&p, want);
      if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); big_pool_trim(dv); e = dev_malloc(&p, want); }
    }   // (no headroom when memory is short)
    if (e != hipSuccess) { (void)hipGetLastError(); throw mm::Error(e == hipErrorOutOfMemory ? MM_ERR_NOMEM : MM_ERR_DEVICE, oom_text(want, e)); }
    *got = granted;
    return p;
  }
  void put(void* p, size_t bytes) { std::lock_guard<std::mutex> lk(m); cache.emplace(bytes, p); cached_bytes += bytes; }
  ~DevAlloc() { alloc_unregister(this); trim(); }
};
struct AllocRegistry { std::mutex m; std::vector<DevAlloc*> v; };
inline AllocRegistry& alloc_registry() { static AllocRegistry r; return r; }
inline void alloc_register(DevAlloc* a, int device) { AllocRegistry& r = alloc_registry(); std::lock_guard<std::mutex> lk(r.m); a->device = device; r.v.push_back(a); }
inline void alloc_unregister(DevAlloc* a) {
  AllocRegistry& r = alloc_registry(); std::lock_guard<std::mutex> lk(r.m);
  for (size_t i = 0; i < r.v.size(); ++i) if (r.v[i] == a) { r.v.erase(r.v.begin() + (long)i); break; }
}
inline void alloc_trim_others(DevAlloc* self, int device) {      // (the caller holds no allocator lock)
  AllocRegistry& r = alloc_registry(); std::lock_guard<std::mutex> lk(r.m);
  for (DevAlloc* a : r.v) if (a != self && a->device == device) a->trim();
}
// Index-scale blocks (>= 8 GiB) are recycled per device: on this runtime a freed block of that size is not free for long — one of the
// next allocations stalls for ~6 s (constant, whatever its own size; MM_ALLOC_TRACE) — and an index build, let alone a pass over
// the chunk indexes of a reference larger than HBM (built, mapped, dropped, chunk after chunk), frees and allocates tens of them.  A
// released block waits here for a request it fits (at most an eighth too large: index-scale blocks are what fills the device); everything is handed back to the driver when an
// allocation fails for lack of memory or the last context of the process goes.
struct BigPool {
  std::mutex m;
  std::multimap<size_t, void*> free_;
  size_t bytes = 0;
  void* take(size_t want, size_t* got) {
    std::lock_guard<std::mutex> lk(m);
    auto it = free_.lower_bound(want);
    if (it == free_.end() || it->first > want + want / 8) return nullptr;
    void* p = it->second; *got = it->first; bytes -= it->first; free_.erase(it);
    return p;
  }
  void* take_at_least(size_t want, size_t* got) {               // the smallest pooled block that holds `want` (for a slab)
    std::lock_guard<std::mutex> lk(m);
    auto it = free_.lower_bound(want);
    if (it == free_.end()) return nullptr;
    void* p = it->second; *got = it->first; bytes -= it->first; free_.erase(it);
    return p;
  }
  void give(void* p, size_t sz) { std::lock_guard<std::mutex> lk(m); free_.emplace(sz, p); bytes += sz; }
  void trim() { std::lock_guard<std::mutex> lk(m); for (auto& kv : free_) dev_free(kv.second, kv.first); free_.clear(); bytes = 0; }
  // pooled blocks back to the driver, largest first, only until it can serve `need` bytes (what stays pooled is what the next chunk build takes
  // without a driver call; true: the driver now has the room)
  bool trim_until(size_t need) {
    std::lock_guard<std::mutex> lk(m);
    for (;;) {
      size_t fr = 0, tot = 0;
      if (dev_mem_info(&fr, &tot) == hipSuccess && fr >= need) return true;
      if (free_.empty()) return false;
      auto it = std::prev(free_.end());
      dev_free(it->second, it->first); bytes -= it->first; free_.erase(it);
    }
  }
};
inline BigPool& big_pool(int device) { static BigPool pools[64]; return pools[device < 0 || device >= 64 ? 0 : device]; }
inline size_t big_pool_bytes(int device) { return big_pool(device).bytes; }
inline void big_pool_adopt_idle(int device) {                    // slabs nothing is cut from any more are pooled blocks again
  for (auto& sl : slab_set().take_idle(device)) big_pool(device).give(sl.first, sl.second);
}
inline bool big_pool_trim_until(int device, size_t need) { big_pool_adopt_idle(device); return big_pool(device).trim_until(need); }
inline void big_pool_trim(int device) {                          // (slabs nothing is cut from any more are pooled blocks again, and go with the rest)
  for (auto& sl : slab_set().take_idle(device)) big_pool(device).give(sl.first, sl.second);
  big_pool(device).trim();
}
inline void* slab_piece(int device, size_t bytes) {
  if (void* p = slab_set().alloc(device, bytes)) return p;
  size_t got = 0;
  void* blk = big_pool(device).take_at_least(SlabSet::granules(bytes), &got);
  if (!blk) return nullptr;
  slab_set().adopt(device, blk, got);
  return slab_set().alloc(device, bytes);
}
inline DevAlloc*& current_alloc() { static thread_local DevAlloc* a = nullptr; return a; }
inline hipStream_t& current_stream() { static thread_local hipStream_t s = nullptr; return s; }
// blocks from this size on are "index-scale": pooled per device, not cached per context (MM_INDEX_SCALE_MB: test hook — with a few MB the pool,
// and the slabs cut from it, come into play on a reference of a few Mbp)
inline size_t direct_alloc_bytes() {
  static const size_t v = [] { const char* e = getenv("MM_INDEX_SCALE_MB"); return e && atoll(e) > 0 ? (size_t)atoll(e) << 20 : (size_t)8 << 30; }();
  return v;
}
// A request the driver has refused for lack of memory, served from what the device's pool holds WITHOUT handing the pool back to the driver:
// the caches go back first (their pieces of pooled blocks return to the blocks), blocks nothing is cut from any more are pooled blocks again, and
// then a pooled block of the right size or a piece of a larger one (a slab) is taken; nullptr when the pool has nothing that large.  Until
// round 4's last session every such miss gave the WHOLE pool back (hipFree) and the following allocations came fresh from the driver, which
// clears what it hands out at ~25 GB/s: with the chunk indexes of a reference larger than the device built, mapped and dropped in turn
// (bench.py --config 5, 15 Gbp chunks) that happened once or twice per chunk — 5.6 s of a 7.0 s chunk build (MM_ALLOC_TRACE, tools/alloc_config5_small.sh).
inline void* big_pool_rescue(DevAlloc* self, int device, size_t bytes, size_t* got, bool caches_first) {
  // (the caches only for a device-filling build, which is after the whole blocks the mapping phase has cut its buffers from; given back at every refused
  // mid-size request they come straight back from the driver: config 4's 2.2 Gbp chunk builds beside 250 GB of pooled blocks went from 0.16 to 0.23 s)
  if (caches_first) { if (self) self->trim(); alloc_trim_others(self, device); }
  big_pool_adopt_idle(device);
  if (bytes >= direct_alloc_bytes()) if (void* p = big_pool(device).take(bytes, got)) return p;
  if (bytes >= SLAB_FROM_BYTES) if (void* p = slab_piece(device, bytes)) { *got = SlabSet::granules(bytes); return p; }
  return nullptr;
}
#define DIRECT_ALLOC_BYTES (mm::direct_alloc_bytes())

inline size_t index_scale_class(size_t b) {
  int lg = 63 - __builtin_clzll((unsigned long long)std::max<size_t>(b, 1));
  const size_t gran = std::max<size_t>((size_t)1 << (lg > 6 ? lg - 6 : 0), (size_t)16 << 20);
  return (b + gran - 1) / gran * gran;
}
template <typename T>
struct DBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t block = 0;            // bytes of the underlying block (0 = index-scale block: big_bytes)
  size_t big_bytes = 0; int big_dev = 0;
  DevAlloc* owner = nullptr;
  // A block held jointly by several DBufs (share_from: the read-only minimizers and sketch hashes of a read batch, mapped against one
  // chunk index after the other): the block lives in `shared`, p / n alias it, the last holder's release frees it.
  std::shared_ptr<DBuf<T>> shared;
  DBuf() = default;
  explicit DBuf(size_t count) { alloc(count); }
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : p(o.p), n(o.n), block(o.block), big_bytes(o.big_bytes), big_dev(o.big_dev), owner(o.owner), shared(std::move(o.shared)) { o.p = nullptr; o.n = 0; }
  DBuf& operator=(DBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; block = o.block; big_bytes = o.big_bytes; big_dev = o.big_dev; owner = o.owner; shared = std::move(o.shared); o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DBuf() { release(); }
  void share_from(DBuf& src) {                                   // afterwards both hold the block; neither may write to it
    if (this == &src) return;
    release();
    if (!src.p) return;
    if (!src.shared) {
      auto sp = std::make_shared<DBuf<T>>();
      sp->p = src.p; sp->n = src.n; sp->block = src.block; sp->big_bytes = src.big_bytes; sp->big_dev = src.big_dev; sp->owner = src.owner;
      src.shared = std::move(sp);
    }
    shared = src.shared; p = shared->p; n = shared->n; block = 0; owner = nullptr;
  }
  void alloc(size_t count) {
    release();
    n = count;
    if (!count) return;
    const size_t bytes = count * sizeof(T);
    owner = current_alloc();
    if (bytes < DIRECT_ALLOC_BYTES && !owner) {                  // no context bound to this thread: a plain driver block (not the index-scale pool, whose
      block = 0; big_bytes = 0;                                  // take() only matches requests within an eighth of a block's size: small blocks would pile up there)
      (void)hipGetDevice(&big_dev);
      hipError_t e = dev_malloc((void**)&p, bytes);
      if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); alloc_trim_others(nullptr, big_dev); big_pool_trim(big_dev); e = dev_malloc((void**)&p, bytes); }
      if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; n = 0; throw mm::Error(e == hipErrorOutOfMemory ? MM_ERR_NOMEM : MM_ERR_DEVICE, mm::oom_text(bytes, e)); }
      return;
    }
    if (bytes >= DIRECT_ALLOC_BYTES || !owner) {
      // (the cache is only given up when the device is out of memory: trimming it before every index-scale allocation sent every
      // mid-size temporary of the next index build back to hipMalloc — 1 400 driver allocations per 25 builds, six of which stalled for
      // 6.1 s each on this runtime: MM_ALLOC_TRACE, round 3)
      block = 0;
      static const bool trace = getenv("MM_ALLOC_TRACE") != nullptr;
      static const bool rescue = getenv("MM_NO_POOL_RESCUE") == nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      (void)hipGetDevice(&big_dev);
      BigPool& bp = big_pool(big_dev);
      // Index-scale blocks come in size classes (a 64th of the size's power of two, at least 16 MiB: <= 1.6 % slack): the chunk indexes of a
      // pass differ by a fraction of a percent, and a pooled block a few KB too small for the next chunk's array is a miss
      const size_t count_bytes = bytes;
      const size_t bytes = rescue ? index_scale_class(count_bytes) : count_bytes;
      if (owner && owner->eager) owner->trim();                  // (a device-filling build: nothing stays cached beside it ...)
      big_pool_adopt_idle(big_dev);                              // (blocks the mapping phase had cut its buffers from and has given back)
      p = (T*)bp.take(bytes, &big_bytes);                        // ... but a pooled block of the right size — the previous chunk index of a streaming pass — is taken:
                                                                 // a fresh block from the driver is cleared as it is handed out, 6 s of a 7 s build of a 15 Gbp chunk (round 4)
      if (!p && rescue && owner && owner->eager) {               // (device-filling builds only: the chunk builds of a --maxmemory run live on their context's cached blocks)
        // the mapping phase between two chunk builds cuts its buffers out of pooled blocks (slabs) and keeps them cached: with the caches given
        // back those blocks are whole again — the arrays of the previous chunk's index, which this build is about to ask for
        owner->trim(); alloc_trim_others(owner, big_dev); big_pool_adopt_idle(big_dev);
        p = (T*)bp.take(bytes, &big_bytes);
        // (a piece of a LARGER pooled block before the driver is asked was tried too: the long-lived arrays then sit inside the blocks the next
        // arrays need whole, and the 62 GB occurrence array of a 15 Gbp chunk found neither a block nor room — out of memory with 33 GB free)
        if (p && trace) fprintf(stderr, "MM_ALLOC_TRACE big block of %zu bytes for %zu after the caches went back\n", big_bytes, bytes);
      }
      if (!p) {
        big_bytes = bytes;
        hipError_t e = dev_malloc((void**)&p, bytes);
        if (e == hipErrorOutOfMemory) {
          if (trace) fprintf(stderr, "MM_ALLOC_TRACE out of memory at an index-scale request of %zu bytes (%zu bytes pooled)\n", bytes, bp.bytes);
          (void)hipGetLastError();
          void* q = rescue ? big_pool_rescue(owner, big_dev, bytes, &big_bytes, owner && owner->eager) : nullptr;
          if (!q && rescue && !(owner && owner->eager)) { if (owner) owner->trim(); alloc_trim_others(owner, big_dev); q = big_pool_rescue(owner, big_dev, bytes, &big_bytes, false); }
          if (q) { p = (T*)q; e = hipSuccess; if (trace) fprintf(stderr, "MM_ALLOC_TRACE ... served from the pool (%zu bytes)\n", big_bytes); }
          else {
            big_bytes = bytes;
            if (!rescue) { if (owner) owner->trim(); alloc_trim_others(owner, big_dev); }
            if (!(rescue && big_pool_trim_until(big_dev, bytes))) big_pool_trim(big_dev);
            e = dev_malloc((void**)&p, bytes);
            if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); big_pool_trim(big_dev); e = dev_malloc((void**)&p, bytes); }
          }
        }
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; n = 0; throw mm::Error(e == hipErrorOutOfMemory ? MM_ERR_NOMEM : MM_ERR_DEVICE, mm::oom_text(bytes, e)); }
        if (trace && big_bytes == bytes && !slab_set().owns(p)) fprintf(stderr, "MM_ALLOC_TRACE direct hipMalloc %zu bytes %.3f ms\n", bytes, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      } else if (trace) fprintf(stderr, "MM_ALLOC_TRACE big block of %zu bytes reused for %zu\n", big_bytes, bytes);
    } else p = (T*)owner->get(bytes, &block);
  }
  void release() {
    if (shared) { shared.reset(); p = nullptr; n = 0; block = 0; return; }
    if (p) {
      if (block && owner) owner->put(p, block);
      else {
        int cur = 0; (void)hipGetDevice(&cur);
        if (cur != big_dev) (void)hipSetDevice(big_dev);         // the synchronisation below is for the block's device, whichever the calling thread is on
        (void)hipDeviceSynchronize();
        // Index-scale blocks a build lets go of stay in the device's pool (since round 4; MM_RETURN_INDEX_BLOCKS=1: back to the driver as in round 3).
        // Handing the sort buffers of a 26.8 Gbp build (~70 GB) back with hipFree made the FIRST allocations of the other contexts of the device wait
        // 3.2 s in two runs of three (the worker contexts of `mapDirectly`: mapping phase 3.2 s instead of 0.23 s); pooled, 0 of 6.  What the pool holds
        // is given up when an allocation fails for lack of memory (every allocation path trims it and tries again).
        static const bool keep = getenv("MM_RETURN_INDEX_BLOCKS") == nullptr;
        if ((owner && owner->eager && !keep) || !big_bytes) dev_free(p, big_bytes ? big_bytes : n * sizeof(T));
        else if (!slab_set().give_back(p, big_bytes)) big_pool(big_dev).give(p, big_bytes);   // (nothing on the device still uses it: any context may take it; a piece of a larger pooled block returns to that block)
        if (cur != big_dev) (void)hipSetDevice(cur);
      }
      p = nullptr;
    }
    n = 0; block = 0;
  }
  size_t bytes() const { return n * sizeof(T); }
  void zero(hipStream_t st) { if (n) MM_HIP(hipMemsetAsync(p, 0, bytes(), st)); }
  void upload(const T* h, size_t count, hipStream_t st) { if (count) MM_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st)); }
  void download(T* h, size_t count, hipStream_t st, size_t offset = 0) const {
    if (count) MM_HIP(hipMemcpyAsync(h, p + offset, count * sizeof(T), hipMemcpyDeviceToHost, st));
  }
  std::vector<T> to_host(hipStream_t st, size_t count = (size_t)-1) const {
    if (count == (size_t)-1) count = n;
    std::vector<T> v(count);
    download(v.data(), count, st);
    MM_HIP(mm::stream_sync(st));
    return v;
  }
};

// ---- index / minimizer record ------------------------------------------------------------------------
// One winnowed minimizer = 8 bytes: {hash, pw}.  pw packs window position, strand and two duplicate
// flags the index builder fills in (DESIGN.md §Data layout):
//   bit 0      strand (1 = FWD, 0 = REV)                      base_types.hpp:121-125
//   bit 1      DP: an earlier entry of the same contig carries the same hash
//   bit 2      DN: a later   entry of the same contig carries the same hash
//   bits 3..31 wpos (< 2^29: contigs up to 536 Mbp)
struct Rec { uint32_t hash; uint32_t pw; };
constexpr uint32_t PW_STRAND = 1u, PW_DP = 2u, PW_DN = 4u;
constexpr int PW_SHIFT = 3;
constexpr int64_t MAX_SEQ_LEN = (1LL << 29) - 1;
__host__ __device__ inline int32_t pw_wpos(uint32_t pw) { return (int32_t)(pw >> PW_SHIFT); }
__host__ __device__ inline int32_t pw_strand(uint32_t pw) { return (pw & PW_STRAND) ? 1 : -1; }

// ---- MurmurHash3_x64_128 low 32 bits, seed 42 (murmur3.h:226-303, commonFunc.hpp:33,71-81) -----------
__host__ __device__ inline uint64_t rotl64(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
__host__ __device__ inline uint64_t fmix64(uint64_t v) {
  v ^= v >> 33; v *= 0xff51afd7ed558ccdULL;
  v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL;
  v ^= v >> 33;
  return v;
}
constexpr uint64_t MUR_C1 = 0x87c37b91114253d5ULL, MUR_C2 = 0x4cf5ad432745937fULL;
constexpr uint32_t MUR_SEED = 42u;

// k == 16: exactly one block (lo = bytes 0..7, hi = bytes 8..15, little endian), no tail
__host__ __device__ inline uint32_t murmur16(uint64_t lo, uint64_t hi) {
  uint64_t a = MUR_SEED, b = MUR_SEED;
  lo *= MUR_C1; lo = rotl64(lo, 31); lo *= MUR_C2; a ^= lo;
  a = rotl64(a, 27); a += b; a = a * 5 + 0x52dce729;
  hi *= MUR_C2; hi = rotl64(hi, 33); hi *= MUR_C1; b ^= hi;
  b = rotl64(b, 31); b += a; b = b * 5 + 0x38495ab5;
  a ^= 16; b ^= 16;
  a += b; b += a;
  a = fmix64(a); b = fmix64(b);
  a += b;
  return (uint32_t)a;
}
// general k (bytes need not be aligned); `rev` reads the bytes backwards from p (p points at the LAST byte)
template <bool REV>
__host__ __device__ inline uint32_t murmur_bytes(const uint8_t* p, int k) {
  auto at = [&](int i) -> uint64_t { return REV ? p[-i] : p[i]; };
  uint64_t a = MUR_SEED, b = MUR_SEED;
  int nb = k >> 4;
  for (int blk = 0; blk < nb; ++blk) {
    uint64_t lo = 0, hi = 0;
    for (int j = 0; j < 8; ++j) { lo |= at(16 * blk + j) << (8 * j); hi |= at(16 * blk + 8 + j) << (8 * j); }
    lo *= MUR_C1; lo = rotl64(lo, 31); lo *= MUR_C2; a ^= lo;
    a = rotl64(a, 27); a += b; a = a * 5 + 0x52dce729;
    hi *= MUR_C2; hi = rotl64(hi, 33); hi *= MUR_C1; b ^= hi;
    b = rotl64(b, 31); b += a; b = b * 5 + 0x38495ab5;
  }
  int rem = k & 15, base = nb << 4;
  uint64_t lo = 0, hi = 0;
  for (int j = 8; j < rem; ++j) hi |= at(base + j) << (8 * (j - 8));
  if (rem > 8) { hi *= MUR_C2; hi = rotl64(hi, 33); hi *= MUR_C1; b ^= hi; }
  for (int j = 0; j < rem && j < 8; ++j) lo |= at(base + j) << (8 * j);
  if (rem > 0) { lo *= MUR_C1; lo = rotl64(lo, 31); lo *= MUR_C2; a ^= lo; }
  a ^= (uint64_t)k; b ^= (uint64_t)k;
  a += b; b += a;
  a = fmix64(a); b = fmix64(b);
  a += b;
  return (uint32_t)a;
}

__host__ __device__ inline uint8_t ascii_of_code(uint32_t c) {      // 0,1,2,3 -> A,C,G,T
  return (uint8_t)(0x41 + 2 * (c & 1) + 6 * (c >> 1) + 11 * ((c & 1) & (c >> 1)));
}
__host__ __device__ inline uint8_t complement_ascii(uint8_t c) {    // commonFunc.hpp:38-55
  return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mm

// ---- opaque handle bodies ----------------------------------------------------------------------------
struct mm_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  // A second stream for ONE purpose: the two launches of K5's 10 kb class (four-wave and two-wave workgroups, independent candidates) run side by side,
  // forked from and joined back into `stream` by events — everything else of a context stays on its one stream (the caching allocator relies on that;
  // the buffers the two launches touch are allocated before the fork and live past the join).  Created at first use.
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void aux_ready() {
    if (aux_stream) return;
    MM_HIP(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    mm::stream_event_register(aux_stream);
    MM_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    MM_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  }
  mm::DevAlloc alloc;
  std::string err;
  int cus = 0;
  void* comm = nullptr;          // ncclComm_t
  bool comm_shared = false;      // the communicator belongs to another context of this device (mm_comm_share)
  int comm_rank = 0, comm_size = 1;
  bool em_split = false;         // the resident EM kernel once failed to get its grid onto the device: one launch per phase from then on (mm_post.hip)
  // per-(k, pi) cache of the host statistics thresholds (pure functions of the sketch size), mm_stats.hpp
  std::shared_ptr<void> lut_cache;
  int lut_k = 0; float lut_pi = 0;
  // K5 scratch kept across batches: the per-entry code words of pass A (4 B per streamed entry slot, mm_l2.hpp)
  void raw_alloc(void** p, size_t bytes) {                       // hipMalloc; out of memory: the caches of this context and the device's block pool go first
    hipError_t e = mm::dev_malloc(p, bytes);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); alloc.trim(); mm::big_pool_trim(device); mm::alloc_trim_others(&alloc, device); e = mm::dev_malloc(p, bytes); }
    if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; throw mm::Error(e == hipErrorOutOfMemory ? MM_ERR_NOMEM : MM_ERR_DEVICE, mm::oom_text(bytes, e)); }
  }
  void* l2_codes = nullptr; size_t l2_codes_bytes = 0;
  void* l2_codes_at_least(size_t bytes) {
    if (bytes > l2_codes_bytes) {
      if (l2_codes) { MM_HIP(mm::stream_sync(stream)); mm::dev_free(l2_codes, l2_codes_bytes); }
      l2_codes = nullptr; l2_codes_bytes = 0;
      raw_alloc(&l2_codes, bytes);
      l2_codes_bytes = bytes;
    }
    return l2_codes;
  }
  void* l2_masks = nullptr; size_t l2_masks_bytes = 0;           // class masks of the long-read K5 classes
  void* l2_masks_at_least(size_t bytes) {
    if (bytes > l2_masks_bytes) {
      if (l2_masks) { MM_HIP(mm::stream_sync(stream)); mm::dev_free(l2_masks, l2_masks_bytes); }
      l2_masks = nullptr; l2_masks_bytes = 0;
      raw_alloc(&l2_masks, bytes);
      l2_masks_bytes = bytes;
    }
    return l2_masks;
  }
  // pinned bounce buffer for result downloads into caller-owned (pageable) memory
  void* pinned = nullptr; size_t pinned_bytes = 0;
  // pinned staging buffer of sequence uploads (mm_seq.hip: the 2-bit words are packed straight into it), and the threads that pack
  void* pinned_up = nullptr; size_t pinned_up_bytes = 0;
  std::unique_ptr<TaskPool> pack_pool;
  void* pinned_up_at_least(size_t bytes) {
    if (bytes > pinned_up_bytes) {
      if (pinned_up) { MM_HIP(mm::stream_sync(stream)); (void)hipHostFree(pinned_up); }
      pinned_up = nullptr; pinned_up_bytes = 0;
      const size_t want = bytes + bytes / 8;
      MM_HIP(hipHostMalloc(&pinned_up, want, hipHostMallocDefault));
      pinned_up_bytes = want;
    }
    return pinned_up;
  }
  void* pinned_at_least(size_t bytes) {
    if (bytes > pinned_bytes) {
      if (pinned) (void)hipHostFree(pinned);
      pinned = nullptr; pinned_bytes = 0;
      size_t want = bytes + bytes / 2 + (1 << 20);
      MM_HIP(hipHostMalloc(&pinned, want, hipHostMallocDefault));
      pinned_bytes = want;
    }
    return pinned;
  }
};

struct mm_seqset {
  mm_ctx* ctx = nullptr;
  bool frozen = false;
  // host staging (until upload)
  std::deque<std::string> owned;                                // copies made by mm_seqset_add (stable addresses)
  std::vector<std::pair<const char*, size_t>> staged;           // what upload packs: views into `owned` or into caller memory (mm_seqset_add_view)
  // host-side metadata (always valid after upload / synthesis)
  std::vector<int32_t> len;            // per sequence
  std::vector<uint64_t> base;          // [n+1] first base of sequence i in the packed stream (multiple of 16)
  int64_t total_bases = 0;
  // device
  mm::DBuf<uint32_t> packed;           // 16 bases per word, base b at bits [2b, 2b+2)
  mm::DBuf<uint64_t> d_base;           // [n+1]
  mm::DBuf<int32_t> d_len;             // [n]
  mm::DBuf<uint64_t> exc_start;        // exception runs, sorted by start (stream coordinates)
  mm::DBuf<uint32_t> exc_len;
  mm::DBuf<uint8_t> exc_byte;
  int64_t n_exc = 0;
  int64_t count() const { return (int64_t)len.size(); }
};
