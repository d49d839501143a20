// Per-batch mapping pipeline on the device (replaces Map::mapModule → mapSingleQuerySeq → doL1Mapping /
// doL2Mapping, computeMap.hpp:180-538, for a whole batch of reads; output order == input order, which is
// all the reference's ThreadPool guarantees, ThreadPool.hpp:13-17).
//
//   K1  minimizer sweep of the reads                         mm_minimizer.hpp
//   K2  sketch = sort by hash + unique                       computeMap.hpp:292-298
//   K3  index probe + seed-hit gather                        computeMap.hpp:307-323
//   K4  hit sort + L1 candidate scan                         computeMap.hpp:346-386
//   K5  L2 sliding MinHash window + K6 strand vote           computeMap.hpp:460-538, slidingMap.hpp
//   K7  identity filter: a per-sketch-size integer threshold computed on the host (mm_stats.hpp)
#include "mm_map.hpp"
#include <functional>
#include <memory>
#include <rocprim/rocprim.hpp>
#include "mm_l2_core.hpp"
#include "mm_l2.hpp"
#include "mm_l2z.hpp"
#include "mm_l2_dense.hpp"
#include <cstdlib>
#include <atomic>
#include <thread>
#include "mm_stats.hpp"
#include <algorithm>
#include <numeric>

namespace mm {

// ---------------------------------------------------------------------------------------------------
// workgroup bitonic sort of 64-bit keys (n a power of two), data in LDS or global memory
// ---------------------------------------------------------------------------------------------------
__device__ inline void bitonic_sort_u64(uint64_t* a, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
        int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int p = i | j;
        uint64_t x = a[i], y = a[p];
        bool up = (i & k) == 0;
        if ((x > y) == up) { a[i] = y; a[p] = x; }
      }
      __syncthreads();
    }
  }
}
static inline int pow2_at_least(int64_t n) { int p = 1; while (p < n) p <<= 1; return p; }

// ---------------------------------------------------------------------------------------------------
// K2  sketch: one workgroup per read
// ---------------------------------------------------------------------------------------------------
template <bool IN_LDS>
__global__ void __launch_bounds__(256) sketch_kernel(const Rec* __restrict__ rec, const uint64_t* __restrict__ off,
                                                     const int32_t* __restrict__ read_list, int npow2, uint64_t* __restrict__ gscratch,
                                                     uint32_t* __restrict__ sk_hash, uint8_t* __restrict__ sk_strand,
                                                     int32_t* __restrict__ sk_n, uint8_t* __restrict__ amb) {
  extern __shared__ __align__(16) uint64_t skeys[];
  const int r = read_list[blockIdx.x];
  const uint64_t o = off[r];
  const int n = (int)(off[r + 1] - o);
  uint64_t* a = IN_LDS ? skeys : gscratch + (size_t)blockIdx.x * npow2;
  for (int i = threadIdx.x; i < npow2; i += 256) a[i] = i < n ? (((uint64_t)rec[o + i].hash << 32) | (uint32_t)i) : ~0ull;
  __syncthreads();
  bitonic_sort_u64(a, npow2);
  __shared__ int s_amb;
  if (threadIdx.x == 0) s_amb = 0;
  __syncthreads();
  uint64_t carry = 0;
  for (int base = 0; base < n; base += 256) {
    const int i = base + threadIdx.x;
    bool first = false; uint32_t h = 0; uint32_t st = 0;
    if (i < n) {
      uint64_t key = a[i];
      h = (uint32_t)(key >> 32);
      st = rec[o + (uint32_t)key].pw & PW_STRAND;
      if (i == 0) first = true;
      else {
        uint64_t pk = a[i - 1];
        first = (uint32_t)(pk >> 32) != h;
        if (!first && (rec[o + (uint32_t)pk].pw & PW_STRAND) != st) s_amb = 1;   // same hash, different strands
      }
    }
    uint64_t tot;
    uint64_t ex = block_excl_scan_u64(first ? 1 : 0, &tot);
    if (first) { sk_hash[o + carry + ex] = h; sk_strand[o + carry + ex] = (uint8_t)st; }
    carry += tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) { sk_n[r] = (int32_t)carry; amb[r] = (uint8_t)s_amb; }
}

// The same with an LDS radix sort (stable, so equal hashes stay in winnowing order exactly as with the 64-bit
// (hash, index) keys of the bitonic version): ~4x fewer instructions than the bitonic network, which also pays for the
// padding to a power of two.  IPT = elements per thread; 256 * IPT >= minimizers of the longest read of the class.
template <int IPT>
__global__ void __launch_bounds__(256) sketch_radix_kernel(const Rec* __restrict__ rec, const uint64_t* __restrict__ off,
                                                           const int32_t* __restrict__ read_list, uint32_t* __restrict__ sk_hash,
                                                           uint8_t* __restrict__ sk_strand, int32_t* __restrict__ sk_n, uint8_t* __restrict__ amb) {
  using Sort = rocprim::block_radix_sort<uint32_t, 256, IPT, uint16_t>;
  using Scan = rocprim::block_scan<int, 256>;
  union Tmp { typename Sort::storage_type sort; typename Scan::storage_type scan; };
  extern __shared__ __align__(16) unsigned char sketch_dyn[];    // dynamic: 64 elements per thread need more than 64 KB
  Tmp& tmp = *reinterpret_cast<Tmp*>(sketch_dyn);
  __shared__ uint32_t last_key[256];
  __shared__ uint8_t last_st[256];
  __shared__ int s_amb;
  const int r = read_list[blockIdx.x];
  const uint64_t o = off[r];
  const int n = (int)(off[r + 1] - o);
  const int t = threadIdx.x;
  uint32_t key[IPT]; uint16_t val[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) { const int idx = t * IPT + i; key[i] = idx < n ? rec[o + idx].hash : 0xffffffffu; val[i] = (uint16_t)idx; }
  if (t == 0) s_amb = 0;
  Sort().sort(key, val, tmp.sort);                               // blocked: thread t holds sorted positions t*IPT ..
  uint8_t stv[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) stv[i] = (t * IPT + i < n) ? (uint8_t)(rec[o + val[i]].pw & PW_STRAND) : 0;
  last_key[t] = key[IPT - 1]; last_st[t] = stv[IPT - 1];
  __syncthreads();
  int nfirst = 0; bool first[IPT]; bool ambig = false;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int pos = t * IPT + i;
    const uint32_t pk = i ? key[i - 1] : (t ? last_key[t - 1] : 0u);
    const uint8_t ps = i ? stv[i - 1] : (t ? last_st[t - 1] : 0);
    first[i] = pos < n && (pos == 0 || pk != key[i]);
    if (pos < n && pos > 0 && pk == key[i] && ps != stv[i]) ambig = true;   // same hash, different strands
    nfirst += first[i] ? 1 : 0;
  }
  if (ambig) s_amb = 1;
  int ex = 0, total = 0;
  Scan().exclusive_scan(nfirst, ex, 0, total, tmp.scan);
  const int ex0 = ex;
#pragma unroll
  for (int i = 0; i < IPT; ++i) if (first[i]) { sk_hash[o + ex] = key[i]; sk_strand[o + ex] = stv[i]; ++ex; }
  __threadfence_block();
  __syncthreads();
  // bit 1 of the strand byte: some duplicate of this hash has the other strand, i.e. the strand the reference would keep
  // depends on libstdc++'s sort (resolved on the host, and only if a strand vote ever reads this entry)
  if (s_amb) {
    int ex2 = ex0;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int pos = t * IPT + i;
      if (first[i]) ++ex2;
      const uint32_t pk = i ? key[i - 1] : (t ? last_key[t - 1] : 0u);
      const uint8_t ps = i ? stv[i - 1] : (t ? last_st[t - 1] : 0);
      if (pos < n && pos > 0 && pk == key[i] && ps != stv[i]) sk_strand[o + ex2 - 1] |= 2;
    }
  }
  if (t == 0) { sk_n[r] = total; amb[r] = (uint8_t)(s_amb ? 2 : 0); }   // 2: ambiguous entries are marked
}

// Sketches of more than 16 384 minimizers (reads beyond ~73 kb): (hash << 32 | winnowing index) keys of the listed reads back to back
// in one buffer, one segmented device radix sort, then unique + strand per read from the sorted keys — what sketch_kernel does
// with its bitonic network through global memory (48 ms per 4 000 reads of 75-140 kb) — plus sketch_radix_kernel's per-entry
// ambiguity marks, so that these reads take the lazy strand tie-break too (the bitonic kernel flags the whole read and all its
// minimizer records go to the host up front: a third of 4 000 such reads, 0.5 GB per batch).
__global__ void __launch_bounds__(256) sketch_keys_kernel(const Rec* __restrict__ rec, const uint64_t* __restrict__ off, const int32_t* __restrict__ read_list,
                                                          const uint64_t* __restrict__ koff, uint64_t* __restrict__ keys) {
  const int r = read_list[blockIdx.x];
  const uint64_t o = off[r], k0 = koff[blockIdx.x];
  const uint32_t n = (uint32_t)(off[r + 1] - o);
  for (uint32_t i = threadIdx.x; i < n; i += 256) keys[k0 + i] = ((uint64_t)rec[o + i].hash << 32) | i;
}
__global__ void __launch_bounds__(256) sketch_finish_kernel(const Rec* __restrict__ rec, const uint64_t* __restrict__ off, const int32_t* __restrict__ read_list,
                                                            const uint64_t* __restrict__ koff, const uint64_t* __restrict__ sorted,
                                                            uint32_t* __restrict__ sk_hash, uint8_t* __restrict__ sk_strand, int32_t* __restrict__ sk_n, uint8_t* __restrict__ amb) {
  const int r = read_list[blockIdx.x];
  const uint64_t o = off[r];
  const uint64_t* __restrict__ a = sorted + koff[blockIdx.x];
  const int n = (int)(off[r + 1] - o);
  __shared__ int s_amb;
  if (threadIdx.x == 0) s_amb = 0;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {                          // 0: survivors (first of every run of equal hashes); 1: marks on them
    uint64_t carry = 0;
    for (int base = 0; base < n; base += 256) {
      const int i = base + threadIdx.x;
      bool first = false, differs = false; uint32_t h = 0, stv = 0;
      if (i < n) {
        const uint64_t key = a[i];
        h = (uint32_t)(key >> 32);
        stv = rec[o + (uint32_t)key].pw & PW_STRAND;
        if (i == 0) first = true;
        else {
          const uint64_t pk = a[i - 1];
          first = (uint32_t)(pk >> 32) != h;
          differs = !first && (rec[o + (uint32_t)pk].pw & PW_STRAND) != stv;   // same hash, different strands
        }
      }
      uint64_t tot;
      const uint64_t ex = block_excl_scan_u64(first ? 1 : 0, &tot);
      if (pass == 0) {
        if (first) { sk_hash[o + carry + ex] = h; sk_strand[o + carry + ex] = (uint8_t)stv; }
        if (differs) s_amb = 1;
      } else if (differs) sk_strand[o + carry + ex - 1] |= 2;     // bit 1 on the run's survivor (the last first at or before i): strand unresolved
      carry += tot;
    }
    __threadfence_block();
    __syncthreads();
    if (pass == 0) {
      if (threadIdx.x == 0) { sk_n[r] = (int32_t)carry; amb[r] = (uint8_t)(s_amb ? 2 : 0); }   // 2: ambiguous entries are marked (lazy tie-break)
      if (!s_amb) break;
    }
  }
}

// compact copies for the host-side duplicate-hash tie-break (one workgroup per flagged read)
__global__ void __launch_bounds__(256) gather_amb_kernel(const Rec* __restrict__ rec, const uint64_t* __restrict__ src_off,
                                                         const uint64_t* __restrict__ dst_off, Rec* __restrict__ out) {
  const uint64_t so = src_off[blockIdx.x], d0 = dst_off[blockIdx.x], n = dst_off[blockIdx.x + 1] - d0;
  for (uint64_t i = threadIdx.x; i < n; i += 256) out[d0 + i] = rec[so + i];
}
__global__ void __launch_bounds__(256) scatter_strand_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ src_off,
                                                             const uint64_t* __restrict__ dst_off, const int32_t* __restrict__ cnt,
                                                             uint8_t* __restrict__ sk_strand) {
  const uint64_t d0 = dst_off[blockIdx.x], so = src_off[blockIdx.x];
  const int n = cnt[blockIdx.x];
  for (int i = threadIdx.x; i < n; i += 256) sk_strand[so + i] = in[d0 + i];
}

// ---------------------------------------------------------------------------------------------------
// K3  probe (one workgroup per read) and gather
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) probe_kernel(IndexView I, const uint32_t* __restrict__ sk_hash, const uint64_t* __restrict__ off,
                                                    const int32_t* __restrict__ sk_n, uint32_t* __restrict__ probe_cnt,
                                                    uint64_t* __restrict__ probe_start, const uint8_t* __restrict__ only /* optional: reads to do */) {
  // Four lanes per lookup, each reading one 16-byte slot of the hash's home sector (mm_index.hpp: tab_slot): one 64-byte
  // request resolves nearly every lookup; NP lookups per group in flight.
  const int r = blockIdx.x;
  if (only && !only[r]) return;
  const uint64_t o = off[r];
  const int s = sk_n[r];
  const int grp = threadIdx.x >> 2, sub = threadIdx.x & 3, gshift = (threadIdx.x & 63) & ~3;
  const ulonglong2* __restrict__ tab = reinterpret_cast<const ulonglong2*>(I.tab);
  const uint64_t tslots = (uint64_t)I.tab_buckets << 2;
  constexpr int NP = 4;                                          // lookups in flight per group
  for (int i0 = grp; i0 < s; i0 += 64 * NP) {
    uint32_t hq[NP]; uint64_t sq[NP]; ulonglong2 vq[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) { hq[u] = i0 + 64 * u < s ? sk_hash[o + i0 + 64 * u] : 0u; sq[u] = tab_slot(hq[u], I.tab_buckets); }
#pragma unroll
    for (int u = 0; u < NP; ++u) vq[u] = tab[sq[u] + sub];
    auto resolve = [&](uint32_t h, uint64_t slot, ulonglong2 v, bool active, int i) {
      bool pending = active;
      while (__any(pending)) {                                   // (wave-wide loop: the ballots below need every lane)
        const bool match = pending && v.x != 0 && (uint32_t)v.x == h, empty = pending && v.x == 0;
        const uint32_t gm = (uint32_t)(__ballot(match) >> gshift) & 0xfu, ge = (uint32_t)(__ballot(empty) >> gshift) & 0xfu;
        if (pending && (gm | ge)) {
          // slots are filled in probing order and never emptied: a match is the key's slot, an empty slot without one means absent
          if (match) {
            const uint32_t cnt = (uint32_t)(v.x >> 32);
            const bool keep = (uint64_t)cnt < (uint64_t)(int64_t)I.freq_threshold;   // computeMap.hpp:317
            probe_cnt[o + i] = keep ? cnt : 0u;
            probe_start[o + i] = keep ? v.y : 0ull;
          } else if (!gm && sub == 0) { probe_cnt[o + i] = 0u; probe_start[o + i] = 0ull; }
          pending = false;
        }
        if (pending) { slot = tab_next_sector(slot, tslots); v = tab[slot + sub]; }
      }
    };
#pragma unroll
    for (int u = 0; u < NP; ++u) resolve(hq[u], sq[u], vq[u], i0 + 64 * u < s, i0 + 64 * u);
  }
}

__global__ void __launch_bounds__(256) gather_hits_kernel(IndexView I, const uint64_t* __restrict__ off, const int32_t* __restrict__ sk_n,
                                                          const uint32_t* __restrict__ probe_cnt, const uint64_t* __restrict__ probe_start,
                                                          const uint64_t* __restrict__ hit_off, uint64_t* __restrict__ hits) {
  const int r = blockIdx.x;
  const uint64_t o = off[r];
  const int s = sk_n[r];
  for (int i = threadIdx.x; i < s; i += 256) {
    uint32_t c = probe_cnt[o + i];
    if (!c) continue;
    const uint64_t* src = I.occ + probe_start[o + i];
    uint64_t* dst = hits + hit_off[o + i];
    for (uint32_t j = 0; j < c; ++j) dst[j] = src[j] & ~(uint64_t)(PW_DP | PW_DN);
  }
}

// ---------------------------------------------------------------------------------------------------
// K3c  exact seed-hit pre-filter.  At miniSeq+H density the 32-bit hash space is saturated:
// a read draws ~10^4 chance hits scattered over the whole reference, and only hits that sit in a run of
// `minimumHits` hits of one contig spanning less than the read length can ever produce or shape an L1
// candidate (computeMap.hpp:357-385).  Positions are binned in 8192-base bins of the concatenated reference; a
// run shorter than the read touches at most nb = (len-1)/8192 + 2 consecutive bins, so a hit can be dropped when no
// window of nb consecutive bins around it holds minimumHits hits.  Bins are counted modulo 8192 bins in LDS
// (aliasing and contig borders only over-count, so nothing needed is lost).  Dropping hits that belong to no
// qualifying run leaves every qualifying run intact and cannot create a new one (a run that qualifies after
// dropping also qualifies before, so none of its members was dropped).
// The bin of every index entry is precomputed (occ16[], 2 bytes per entry, same layout as occ[]): both passes
// read a quarter of the list bytes, mostly one 64-byte sector per list, and only survivors touch occ[] itself.
// ---------------------------------------------------------------------------------------------------
// survivors are staged per read (8 B each, capacity 1024 + 2 x sketch size: stage_off); reads with more are re-filtered by the write kernel
// Two slot tables: 8 192 slots counted from the 13-bit codes of occ16[] (reads up to ~32 kb), and 32 768 slots counted from the
// entries of occ[] themselves (slot = position bin + a per-contig offset) for longer reads.  Chance hits grow with the read length and
// so does the window, so with 8 192 slots a 100 kb read (3.8*10^5 seed hits against the bench reference) has 650 hits in every
// window — above minimumHits everywhere, nothing is dropped, K4 sorts 1.5*10^9 hits per 4 000 reads; 32 768 slots keep the
// background a factor of four lower, below the threshold.  Any slot function that keeps neighbouring bins of a contig neighbours is
// a valid (superset) filter; pass 1 and the write pass of a read use the same table.  `cls[r]`: 0 fused kernel, 1 narrow, 2 wide.
template <int SLOT_BITS> struct HitFilterCfg {
  static constexpr int SLOTS = 1 << SLOT_BITS, THREADS = SLOTS / 32;
  static constexpr int EPL = SLOT_BITS == HF_SLOT_BITS_NARROW ? 8 : 2;   // entries per 16-byte load of a lane
  static constexpr int CLS = SLOT_BITS == HF_SLOT_BITS_NARROW ? 1 : 2;
  static constexpr size_t LDS = (size_t)SLOTS * 4 + (size_t)THREADS * 8 + 16;
};
template <bool WRITE, int SLOT_BITS>
__global__ void __launch_bounds__(HitFilterCfg<SLOT_BITS>::THREADS) hit_filter_kernel(IndexView I, const uint64_t* __restrict__ off, const int32_t* __restrict__ sk_n,
                                                         const uint32_t* __restrict__ probe_cnt, const uint64_t* __restrict__ probe_start,
                                                         const int32_t* __restrict__ read_len, const int32_t* __restrict__ min_hits,
                                                         uint32_t* __restrict__ surv_n, const uint64_t* __restrict__ read_hit_off,
                                                         uint64_t* __restrict__ hits, uint64_t* __restrict__ stage, const uint64_t* __restrict__ stage_off, int dbg,
                                                         const uint8_t* __restrict__ cls /* per read: which kernel filters it */,
                                                         uint32_t* __restrict__ raw_hits /* optional (WRITE = false): seed hits of the read before filtering */) {
  using Cfg = HitFilterCfg<SLOT_BITS>;
  constexpr int SLOTS = Cfg::SLOTS, THREADS = Cfg::THREADS, EPL = Cfg::EPL;
  constexpr bool NARROW = SLOT_BITS == HF_SLOT_BITS_NARROW;
  extern __shared__ __align__(16) uint32_t hf_lds[];
  uint32_t* const cnt = hf_lds;                                   // [SLOTS]
  uint32_t* const good = cnt + SLOTS;                             // [THREADS]
  uint32_t* const alive = good + THREADS;                         // [THREADS]
  uint32_t& cursor = alive[THREADS];
  const int r = blockIdx.x;
  const int my_cls = cls[r];
  if (!WRITE && my_cls != Cfg::CLS) return;
  if (WRITE && (my_cls == 2) != (Cfg::CLS == 2)) return;         // (reads of the fused kernel whose stage overflowed are re-filtered by the narrow kernel)
  if (WRITE) {                                                   // staged reads only need a copy
    const uint32_t n_s = surv_n[r];
    if (n_s <= (uint32_t)(stage_off[r + 1] - stage_off[r])) {
      if (dbg == 100 && n_s >= 2u && n_s <= 4096u) return;        // (dbg 100: the LDS radix sort takes these straight from the stage)
      const uint64_t wb = read_hit_off[r];
      for (uint32_t i = threadIdx.x; i < n_s; i += THREADS) hits[wb + i] = stage[stage_off[r] + i];
      return;
    }
  }
  const uint64_t o = off[r];
  const int s = sk_n[r];
  const uint32_t len = (uint32_t)max(read_len[r], 1);
  const int nb = min((int)((len - 1) >> HF_BIN_SHIFT) + 2, SLOTS);
  int m = min_hits[r]; if (m < 1) m = 1;
  for (int i = threadIdx.x; i < SLOTS; i += THREADS) cnt[i] = 0;
  if (threadIdx.x == 0) cursor = 0;
  __syncthreads();
  // One occurrence list per group of 4 lanes, one 16-byte load per lane (8 bin codes, or 2 entries): a list of up to 32 codes is a
  // single request of at most 64 bytes.  Random reads are bound by requests, not bytes (tools/ubench/randread), so the
  // lists of a group are software-pipelined: count/start three lists ahead, data two ahead.
  constexpr int GROUPS = THREADS / 4, PER_REQ = 4 * EPL;
  const int grp = threadIdx.x >> 2, sub = threadIdx.x & 3;
  // fn(c, st0, j0, v): lane `sub` of the group holds entries j0 + EPL*sub .. +EPL-1 of a list of c entries that starts
  // at occ[st0]; called by all lanes of the wave together (c == 0: nothing), so that fn may use wave-wide operations
  auto for_each_chunk = [&](auto&& fn) {
    auto meta = [&](int i, uint32_t& c, uint64_t& st0) { c = 0; st0 = 0; if (i < s) { c = probe_cnt[o + i]; st0 = probe_start[o + i]; } };
    auto issue = [&](uint32_t c, uint64_t st0, uint32_t j0, ulonglong2& v) {   // (clamped into the padded list)
      if (c) {
        const uint64_t e = st0 + min(j0 + (uint32_t)EPL * sub, (c - 1) & ~(uint32_t)(EPL - 1));
        v = NARROW ? *reinterpret_cast<const ulonglong2*>(I.occ16 + e) : *reinterpret_cast<const ulonglong2*>(I.occ + e);
      }
    };
    uint32_t c0, c1, c2, c3; uint64_t s0, s1, s2, s3;
    ulonglong2 v0 = make_ulonglong2(0, 0), v1 = v0, v2 = v0;
    meta(grp, c0, s0); meta(grp + GROUPS, c1, s1); meta(grp + 2 * GROUPS, c2, s2);
    issue(c0, s0, 0, v0); issue(c1, s1, 0, v1);
    for (int ib = 0; ib < s; ib += GROUPS) {                     // (wave-uniform trip count)
      meta(ib + grp + 3 * GROUPS, c3, s3);
      issue(c2, s2, 0, v2);
      fn(c0, s0, 0u, v0);
      for (uint32_t j0 = PER_REQ; __any(j0 < c0); j0 += PER_REQ) {   // long lists: the rest
        const uint32_t cl = j0 < c0 ? c0 : 0u;
        ulonglong2 v = make_ulonglong2(0, 0); issue(cl, s0, j0, v); fn(cl, s0, j0, v);
      }
      c0 = c1; s0 = s1; v0 = v1; c1 = c2; s1 = s2; v1 = v2; c2 = c3; s2 = s3;
    }
  };
  auto code_of = [](const ulonglong2& v, int t) -> uint32_t {
    if (NARROW) return (uint32_t)((t < 4 ? v.x : v.y) >> (16 * (t & 3))) & (uint32_t)(SLOTS - 1);
    const uint64_t e = t ? v.y : v.x;                            // contig << 32 | wpos << 3 | flags
    return (((uint32_t)e >> (3 + HF_BIN_SHIFT)) + (uint32_t)(e >> 32) * 40503u) & (uint32_t)(SLOTS - 1);
  };
  if (dbg == 2) {
    uint32_t a = 0;
    for_each_chunk([&](uint32_t c, uint64_t, uint32_t j0, const ulonglong2& v) { for (int t = 0; t < EPL; ++t) if (j0 + (uint32_t)EPL * sub + t < c) a += code_of(v, t); });
    if (a == 0x12345678u) cnt[0] = 1;
  } else for_each_chunk([&](uint32_t c, uint64_t, uint32_t j0, const ulonglong2& v) {
#pragma unroll
    for (int t = 0; t < EPL; ++t) if (j0 + (uint32_t)EPL * sub + t < c) atomicAdd(&cnt[code_of(v, t)], 1u);
  });
  __syncthreads();
  if (dbg == 1 || dbg == 2) { if (!WRITE && threadIdx.x == 0) surv_n[r] = 0; return; }   // timing aid (MM_HF_DBG): pass 1 only
  {
    // good[b]: the window of nb bins starting at b holds >= m hits (sliding sum over this thread's 32 window starts);
    // alive[b]: some good window contains b, i.e. good dilated by nb positions (all modulo the slot count)
    const int b0 = threadIdx.x * 32;
    uint32_t sum = 0, bits = 0;
    for (int i = 0; i < nb; ++i) sum += cnt[(b0 + i) & (SLOTS - 1)];
    for (int t = 0; t < 32; ++t) {
      bits |= (sum >= (uint32_t)m ? 1u : 0u) << t;
      sum += cnt[(b0 + t + nb) & (SLOTS - 1)] - cnt[(b0 + t) & (SLOTS - 1)];
    }
    good[threadIdx.x] = bits;
    __syncthreads();
    uint32_t al = 0;
    for (int j = 0; j < nb; ++j) {                               // bit b of alive = OR over j < nb of good bit (b - j)
      const int wsh = j >> 5, bsh = j & 31;
      const uint32_t g0 = good[(threadIdx.x - wsh) & (THREADS - 1)], g1 = good[(threadIdx.x - wsh - 1) & (THREADS - 1)];
      al |= bsh ? (g0 << bsh) | (g1 >> (32 - bsh)) : g0;
    }
    alive[threadIdx.x] = al;
    __syncthreads();
  }
  const uint64_t wbase = WRITE ? read_hit_off[r] : 0;
  const uint64_t stage_base = stage_off[r];
  const uint32_t stage_cap = (uint32_t)(stage_off[r + 1] - stage_base);
  uint64_t* const dst = WRITE ? hits + wbase : stage + stage_base;
  const uint32_t dst_cap = WRITE ? 0xffffffffu : stage_cap;
  // second pass: survivors (a few per cent) park the index of their entry, which is then replaced by the entry itself.
  // Per chunk the wave reserves its slots with one atomic (bit mask per lane, prefix sum across the wave) — a branch and
  // an atomic per surviving entry would serialise the wave on LDS round trips.
  const int lane = threadIdx.x & 63;
  for_each_chunk([&](uint32_t c, uint64_t st0, uint32_t j0, const ulonglong2& v) {
    const uint32_t e0 = j0 + (uint32_t)EPL * sub;
    uint32_t mask = 0;
#pragma unroll
    for (int t = 0; t < EPL; ++t) { const uint32_t b = code_of(v, t); mask |= ((e0 + t < c) ? (alive[b >> 5] >> (b & 31)) & 1u : 0u) << t; }
    if (dbg == 4) { if (mask == 0xdeadu) cnt[1] = 1; return; }   // timing aid: reads and bit tests only
    const int mine = __popc(mask);
    const int incl = wave_incl_scan(mine);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return;
    uint32_t base = 0;
    if (lane == 63) base = atomicAdd(&cursor, (uint32_t)total);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
    uint32_t pos = base + (uint32_t)(incl - mine);
    while (mask) {
      const int t = __ffs(mask) - 1; mask &= mask - 1;
      if (pos < dst_cap) dst[pos] = st0 + e0 + t;
      ++pos;
    }
  });
  __syncthreads();
  const uint32_t n_s = min(cursor, dst_cap);
  if (dbg == 3 || dbg == 4) { if (!WRITE && threadIdx.x == 0) surv_n[r] = 0; return; }   // timing aid: without the fetch of the survivors
  for (uint32_t j = threadIdx.x; j < n_s; j += THREADS) dst[j] = I.occ[dst[j]] & ~(uint64_t)(PW_DP | PW_DN);
  if (!WRITE && threadIdx.x == 0) surv_n[r] = cursor;
  if (!WRITE && raw_hits) {                                      // (the bin counters still hold every hit of the read)
    __syncthreads();
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < SLOTS; i += THREADS) acc += cnt[i];
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&raw_hits[r], acc);
  }
}

// ---------------------------------------------------------------------------------------------------
// K3 + K3c fused for reads whose sketch and seed hits fit LDS (the 10 kb class): probe, count, filter in ONE launch.
// hit_filter_kernel above reads every occurrence list twice (count pass, then the pass that tests each entry against the
// surviving bins) and takes its list heads from arrays probe_kernel wrote to global memory.  Random requests, not bytes, are
// what these kernels pay for (tools/ubench/randread), so here every list is requested ONCE: one workgroup of 1024 threads per
// read keeps in LDS
//     the list heads (first occurrence, count) of the sketch          phase 0: table lookups, 4 lanes per hash
//     the 13-bit bin codes of every seed hit, 8 per 16-byte chunk      phase 1: one 16-byte load per lane, all lists of a lane
//                                                                      group in flight together; bins counted as they arrive
// and the second pass (phase 2) is bit tests over LDS; only survivors (a few per cent) touch occ[].  Results, staging and
// overflow protocol are those of hit_filter_kernel<false>: survivors staged per read, surv_n[r] their number.  A read that does
// not fit (sketch > SF_SMAX, more than SF_CHUNKS code chunks, a bin count that would not fit 16 bits, or a full stage) is
// flagged in need_old[] and redone by probe_kernel + hit_filter_kernel, which skip every other read.
// ---------------------------------------------------------------------------------------------------
constexpr int SF_THREADS = 1024, SF_GROUPS = SF_THREADS / 4;
constexpr int SF_LPG = 11;                                      // lists per lane group
constexpr int SF_SMAX = SF_GROUPS * SF_LPG;                     // 2816 sketch hashes (reads up to ~12.5 kb at w = 8)
constexpr int SF_CHUNKS = 6144;                                 // parked code chunks (8 codes, 16 bytes each): 49 152 seed hits incl. padding
constexpr int SF_EXTRA = 1024;                                  // pieces of 32 entries beyond the first of a list (lists longer than 32 entries)
struct SeedFilterLds {
  uint32_t cnt16[HF_SLOTS / 2];                                 // two 16-bit bin counters per word
  uint32_t good[HF_SLOTS / 32], alive[HF_SLOTS / 32];
  uint32_t lstart8[SF_SMAX];                                   // first occurrence of every list / 8: lists start on 64-byte sectors = multiples of 8 entries (padded_counts_kernel), and an index of
                                                                // more than 2^35 padded occurrences (275 GB of occ[] alone) does not fit a device — 11 KB of LDS that a minimizer workgroup of ANOTHER batch fits into beside this kernel
  uint16_t lcnt[SF_SMAX];
  uint16_t coff8[SF_SMAX + 8];                                  // first code chunk of every list (+ total)
  uint32_t extra[SF_EXTRA];                                     // 32-entry pieces beyond a list's first: list << 11 | piece
  uint32_t wsum[SF_THREADS / 64], wsum2[SF_THREADS / 64];
  uint32_t cursor, fallback, total8, hraw, n_extra, tick[2], pad_[1];
  ulonglong2 codes[SF_CHUNKS];
};
__global__ void __launch_bounds__(SF_THREADS) seed_filter_kernel(IndexView I, const uint32_t* __restrict__ sk_hash, const uint64_t* __restrict__ off,
                                                                 const int32_t* __restrict__ sk_n, const int32_t* __restrict__ read_len,
                                                                 const int32_t* __restrict__ min_hits, uint32_t* __restrict__ surv_n,
                                                                 uint64_t* __restrict__ stage, const uint64_t* __restrict__ stage_off,
                                                                 uint8_t* __restrict__ need_old, uint32_t* __restrict__ raw_hits, int dbg /* timing aid (MM_SF_DBG): leave after phase n */) {
  extern __shared__ __align__(16) unsigned char sf_dyn[];
  SeedFilterLds& L = *reinterpret_cast<SeedFilterLds*>(sf_dyn);
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int s = sk_n[r];
  if (need_old[r]) return;                                       // not of this class (set by the host): the two-pass kernels take it
  if (s <= 0) { if (tid == 0) { surv_n[r] = 0; raw_hits[r] = 0; } return; }
  const uint64_t o = off[r];
  const int grp = tid >> 2, sub = tid & 3, gshift = lane & ~3;
  for (int i = tid; i < HF_SLOTS / 2; i += SF_THREADS) L.cnt16[i] = 0;
  if (tid == 0) { L.cursor = 0; L.fallback = 0; }
  __syncthreads();
  // ---- phase 0: table lookups (probe_kernel's scheme: 4 lanes read the four 16-byte slots of the hash's home sector).  All hashes
  // of a lane group are loaded first, then all home sectors requested, then resolved: two memory latencies for the whole sketch.
  {
    const ulonglong2* __restrict__ tab = reinterpret_cast<const ulonglong2*>(I.tab);
    const uint64_t tslots = (uint64_t)I.tab_buckets << 2;
    uint32_t hq[SF_LPG]; ulonglong2 vq[SF_LPG];
#pragma unroll
    for (int u = 0; u < SF_LPG; ++u) { const int i = grp + SF_GROUPS * u; hq[u] = i < s ? sk_hash[o + i] : 0u; }
#pragma unroll
    for (int u = 0; u < SF_LPG; ++u) vq[u] = tab[tab_slot(hq[u], I.tab_buckets) + sub];
#pragma unroll
    for (int u = 0; u < SF_LPG; ++u) {
      const int i = grp + SF_GROUPS * u;
      const uint32_t h = hq[u]; uint64_t slot = tab_slot(h, I.tab_buckets); ulonglong2 v = vq[u];
      bool pending = i < s;
      while (__any(pending)) {
        const bool match = pending && v.x != 0 && (uint32_t)v.x == h, empty = pending && v.x == 0;
        const uint32_t gm = (uint32_t)(__ballot(match) >> gshift) & 0xfu, ge = (uint32_t)(__ballot(empty) >> gshift) & 0xfu;
        if (pending && (gm | ge)) {
          if (match) {
            const uint32_t cnt = (uint32_t)(v.x >> 32);
            const bool keep = (uint64_t)cnt < (uint64_t)(int64_t)I.freq_threshold;   // computeMap.hpp:317
            if (keep && cnt > 0xffffu) L.fallback = 1;              // (a list this long overflows the code area anyway)
            L.lcnt[i] = keep ? (uint16_t)cnt : (uint16_t)0; L.lstart8[i] = keep ? (uint32_t)(v.y >> 3) : 0u;
          } else if (!gm && sub == 0) { L.lcnt[i] = 0; L.lstart8[i] = 0u; }
          pending = false;
        }
        if (pending) { slot = tab_next_sector(slot, tslots); v = tab[slot + sub]; }
      }
    }
  }
  __syncthreads();
  if (dbg == 1) { if (tid == 0) { surv_n[r] = 0; raw_hits[r] = L.lcnt[0]; } return; }
  // ---- code chunk offsets: exclusive scan of ceil(count / 8) over the lists (three lists per thread)
  {
    uint32_t c8[3], hr = 0, mine = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int i = tid * 3 + j; const uint32_t c = i < s ? L.lcnt[i] : 0u; c8[j] = (c + 7) >> 3; mine += c8[j]; hr += c; }
    const uint32_t inc = (uint32_t)wave_incl_scan((int)mine), inc2 = (uint32_t)wave_incl_scan((int)hr);
    if (lane == 63) { L.wsum[wid] = inc; L.wsum2[wid] = inc2; }
    __syncthreads();
    uint32_t basew = 0, tot = 0, tot2 = 0;
#pragma unroll
    for (int q = 0; q < SF_THREADS / 64; ++q) { const uint32_t x = L.wsum[q]; if (q < wid) basew += x; tot += x; tot2 += L.wsum2[q]; }
    uint32_t ex = basew + inc - mine;
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int i = tid * 3 + j; if (i <= s) L.coff8[i] = (uint16_t)min(ex, 0xffffu); ex += c8[j]; }
    if (tid == 0) { L.total8 = tot; L.hraw = tot2; L.n_extra = 0; if (tot > (uint32_t)SF_CHUNKS || tot2 > 65535u) L.fallback = 1; }
  }
  __syncthreads();
  // 32-entry pieces beyond the first of a list get their own table, so that they are requested together, too
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = tid * 3 + j;
    const uint32_t c = i < s ? (uint32_t)L.lcnt[i] : 0u;
    if (c > 32) {
      const uint32_t np = (c - 1) >> 5;                           // pieces 1 .. np
      const uint32_t at = atomicAdd(&L.n_extra, np);
      for (uint32_t p = 0; p < np; ++p) if (at + p < (uint32_t)SF_EXTRA) L.extra[at + p] = ((uint32_t)i << 11) | (p + 1);
    }
  }
  __syncthreads();
  if (L.n_extra > (uint32_t)SF_EXTRA) L.fallback = 1;             // (every thread writes the same value)
  __syncthreads();
  if (L.fallback) { if (tid == 0) { need_old[r] = 1; surv_n[r] = 0; raw_hits[r] = 0; } return; }
  if (dbg == 2) { if (tid == 0) { surv_n[r] = 0; raw_hits[r] = L.hraw; } return; }
  const uint32_t len = (uint32_t)max(read_len[r], 1);
  const int nb = min((int)((len - 1) >> HF_BIN_SHIFT) + 2, HF_SLOTS);
  int m = min_hits[r]; if (m < 1) m = 1;
  // ---- phase 1: every list once.  A lane group owns lists grp, grp + 256, ...; the first 32 entries of all of them are requested
  // before any is used (up to 11 x 16 bytes per lane in flight); the further pieces of long lists follow the same way.
  {
    auto code_of = [](const ulonglong2& v, int t) { return (uint32_t)((t < 4 ? v.x : v.y) >> (16 * (t & 3))) & 0xffffu; };
    // a parked chunk: eight 16-bit slots, bin code in the low 13 bits; the 3 spare bits of the slots together hold the list the chunk
    // belongs to (12 bits) and its number of valid entries - 1 (3 bits), so that the second pass needs no search
    auto take = [&](uint32_t cc, uint32_t li, uint32_t chunk0, uint32_t j0, const ulonglong2& x) {   // lane `sub` holds entries j0 + 8 sub .. + 7 of list li (cc entries)
      const uint32_t e0 = j0 + 8u * sub;
      if (e0 >= cc) return;
      const uint32_t nv = min(8u, cc - e0), meta = li | ((nv - 1u) << 12);
      uint64_t w0 = 0, w1 = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t code = code_of(x, t) & (uint32_t)(HF_SLOTS - 1);
        if ((uint32_t)t < nv) atomicAdd(&L.cnt16[code >> 1], 1u << (16 * (code & 1)));
        const uint64_t slot = code | (((meta >> (3 * t)) & 7u) << 13);
        if (t < 4) w0 |= slot << (16 * t); else w1 |= slot << (16 * (t - 4));
      }
      L.codes[chunk0 + (e0 >> 3)] = make_ulonglong2(w0, w1);
    };
    {
      uint32_t c[SF_LPG]; ulonglong2 v[SF_LPG];
#pragma unroll
      for (int u = 0; u < SF_LPG; ++u) {
        const int i = grp + SF_GROUPS * u;
        c[u] = i < s ? (uint32_t)L.lcnt[i] : 0u;
        v[u] = make_ulonglong2(0, 0);
        if (c[u]) v[u] = *reinterpret_cast<const ulonglong2*>(I.occ16 + ((uint64_t)L.lstart8[i] << 3) + min(8u * sub, (c[u] - 1) & ~7u));
      }
#pragma unroll
      for (int u = 0; u < SF_LPG; ++u) { const int i = grp + SF_GROUPS * u; if (c[u]) take(c[u], (uint32_t)i, (uint32_t)L.coff8[i], 0u, v[u]); }
    }
    {
      constexpr int EPG = SF_EXTRA / SF_GROUPS;                   // extra pieces per lane group
      const uint32_t ne = L.n_extra;
      uint32_t c[EPG], ch0[EPG], j0[EPG], li[EPG]; ulonglong2 v[EPG];
#pragma unroll
      for (int u = 0; u < EPG; ++u) {
        const uint32_t k = (uint32_t)(grp + SF_GROUPS * u);
        c[u] = 0; v[u] = make_ulonglong2(0, 0); ch0[u] = 0; j0[u] = 0; li[u] = 0;
        if (k < ne) {
          const uint32_t e = L.extra[k], i = e >> 11;
          li[u] = i; c[u] = (uint32_t)L.lcnt[i]; ch0[u] = (uint32_t)L.coff8[i]; j0[u] = (e & 0x7ffu) << 5;
          v[u] = *reinterpret_cast<const ulonglong2*>(I.occ16 + ((uint64_t)L.lstart8[i] << 3) + min(j0[u] + 8u * sub, (c[u] - 1) & ~7u));
        }
      }
#pragma unroll
      for (int u = 0; u < EPG; ++u) if (c[u]) take(c[u], li[u], ch0[u], j0[u], v[u]);
    }
  }
  __syncthreads();
  if (dbg == 3) { if (tid == 0) { surv_n[r] = 0; raw_hits[r] = L.cnt16[0]; } return; }
  {
    // good[b]: the window of nb bins starting at b holds >= m hits (eight window starts per thread, one byte of the bit set);
    // alive[b]: some good window contains b, i.e. good dilated by nb positions (hit_filter_kernel)
    const uint16_t* cnt = reinterpret_cast<const uint16_t*>(L.cnt16);
    const int b0 = tid * 8;
    uint32_t sum = 0, bits = 0;
    for (int i = 0; i < nb; ++i) sum += cnt[(b0 + i) & (HF_SLOTS - 1)];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      bits |= (sum >= (uint32_t)m ? 1u : 0u) << t;
      sum += (uint32_t)cnt[(b0 + t + nb) & (HF_SLOTS - 1)] - (uint32_t)cnt[(b0 + t) & (HF_SLOTS - 1)];
    }
    reinterpret_cast<uint8_t*>(L.good)[tid] = (uint8_t)bits;
  }
  __syncthreads();
  if (tid < HF_SLOTS / 32) {
    uint32_t al = 0;
    for (int j = 0; j < nb; ++j) {
      const int wsh = j >> 5, bsh = j & 31;
      const uint32_t g0 = L.good[(tid - wsh) & 255], g1 = L.good[(tid - wsh - 1) & 255];
      al |= bsh ? (g0 << bsh) | (g1 >> (32 - bsh)) : g0;
    }
    L.alive[tid] = al;
  }
  __syncthreads();
  if (dbg == 5) { if (tid == 0) { surv_n[r] = 0; raw_hits[r] = L.alive[0]; } return; }
  // ---- phase 2: bit tests over the parked codes; a survivor is the occurrence (list start + position in the list)
  const uint64_t stage_base = stage_off[r];
  const uint32_t stage_cap = (uint32_t)(stage_off[r + 1] - stage_base);
  uint64_t* const dst = stage + stage_base;
  const uint32_t T8 = L.total8;
  for (uint32_t q0 = 0; q0 < T8; q0 += SF_THREADS) {             // (wave-uniform trip count)
    const uint32_t q = q0 + tid;
    uint32_t mask = 0, meta = 0;
    if (q < T8) {
      const ulonglong2 x = L.codes[q];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t slot = (uint32_t)((t < 4 ? x.x : x.y) >> (16 * (t & 3))) & 0xffffu, code = slot & (uint32_t)(HF_SLOTS - 1);
        meta |= (slot >> 13) << (3 * t);
        mask |= ((L.alive[code >> 5] >> (code & 31)) & 1u) << t;
      }
      mask &= (2u << (meta >> 12 & 7u)) - 1u;                     // valid entries only
    }
    const int mine = __popc(mask);
    const int incl = wave_incl_scan(mine);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) continue;
    uint32_t base = 0;
    if (lane == 63) base = atomicAdd(&L.cursor, (uint32_t)total);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
    uint32_t pos = base + (uint32_t)(incl - mine);
    if (mask) {
      const uint32_t li = meta & 0xfffu;
      const uint64_t first = ((uint64_t)L.lstart8[li] << 3) + (uint64_t)(q - (uint32_t)L.coff8[li]) * 8u;
      while (mask) {
        const int t = __ffs(mask) - 1; mask &= mask - 1;
        if (pos < stage_cap) dst[pos] = first + (uint32_t)t;
        ++pos;
      }
    }
  }
  __syncthreads();
  const uint32_t n_s = L.cursor;
  if (dbg == 4) { if (tid == 0) { surv_n[r] = 0; raw_hits[r] = n_s; } return; }
  if (n_s > stage_cap) { if (tid == 0) { need_old[r] = 1; surv_n[r] = 0; raw_hits[r] = 0; } return; }   // stage too small: the two-pass kernels redo the read
  for (uint32_t j = tid; j < n_s; j += SF_THREADS) dst[j] = I.occ[dst[j]] & ~(uint64_t)(PW_DP | PW_DN);
  if (tid == 0) { surv_n[r] = n_s; raw_hits[r] = L.hraw; }
}

// ---------------------------------------------------------------------------------------------------
// The same filter as a resident workgroup that streams through the reads (one workgroup per CU, reads handed out by a ticket
// counter) and overlaps itself: the table look-ups of read r + 1 are in flight — their answers wait in registers, 11 x 16 bytes per
// lane, so the LDS layout is unchanged — while read r tests its parked codes against the surviving bins, writes its survivor slots
// and fetches its survivors.  seed_filter_kernel above does a read's phases one after the other on a CU that holds one workgroup
// (152 KB of LDS): VALU 40 %, LDS 19 %, waiting on memory 26 % of the cycles (profiles/r02_sq_counters.txt).
// What the form needs to work at all (each found in the ISA, docs/history.md section 4):
//   * the barriers of the loop are LDS-only (s_waitcnt lgkmcnt(0) + s_barrier): nothing may drain the vector memory counter
//     between the issue of the look-ups and their use;
//   * everything a read needs from global memory besides its lists comes through SCALAR loads (class byte, sketch size, offsets,
//     stage bounds, read length, minimumHits): a vector load behind the look-ups waits for them (the counter is in-order);
//   * values derived from the thread index are re-derived per iteration from a value the compiler cannot see through: hoisted out of
//     the loop they are spilled, and a reload from scratch is a vector memory operation;
//   * look-ups that need a second probe (a full home sector) are re-issued together, after all eleven answers have been looked at:
//     one more round trip per read instead of one per list of a lane group (2.4 ms of 15.6 in the first version).
// Results are those of seed_filter_kernel read for read (tests: MM_SF_ONESHOT=1 runs the one-read-per-workgroup form).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// SF_STREAM_WAVES_PER_EU: the register budget of the streaming kernel.  4 (default) = all 512 registers of a SIMD's lane slot go to its four waves, 128
// each.  5 = 96 each, which leaves 128 per SIMD — one wave of another kernel — free beside the resident workgroup; together with the 11 KB of LDS
// that lstart8 freed (20 KB left: a minimizer workgroup of the OTHER worker's batch fits) VALU-bound work could run under this kernel's memory
// waits.  Measured in round 5 (tools/ab.sh, one box, in turns): at 96 registers 43 are spilled, and a reload from scratch waits behind the look-ups in
// flight: this kernel 14.9 -> 16.9 ms alone; the other worker's K1 does get in (its time inside the timed region 17 -> 13 ms), the step does not
// gain: 45.4 / 47.4 ms against 44.5 / 46.9.  Not adopted; the switch stays for the record (tools/ab_build.sh w5 "-DSF_STREAM_WAVES_PER_EU=5").
// (That was the 141 KB layout.  Since the round's last session the kernel keeps 32-bit counters and an anchor table: 158 KB of LDS, nothing fits beside it.)
#ifndef SF_STREAM_WAVES_PER_EU
#define SF_STREAM_WAVES_PER_EU 4
#endif
// The streaming kernel's LDS.  Round 5 (tools/sf_grid_sweep.py): this kernel's time follows the number of CUs at work (64 resident workgroups:
// 55.6 ms, 256: 15.2 ms — 13.9 ms x 4), i.e. it is bound by what a CU executes per read — 2 580 VALU instructions per wave and read, most of them
// in phase 1's count-and-park of the codes (per code: extract, validity test under its own branch, counter address and increment of a packed 16-bit
// pair; per piece: 28 instructions that spread the list number and the valid count over the spare bits) and their undoing in phase 2 — not by
// the memory side, which it loads to 80 % of its random-request ceiling.  So:
//   * the padding entries of occ16[] carry codes of their own (hf_pad_code, mm_index.hpp: 8192 + a number below 64), which land in 64 dummy
//     counters and are never alive: no valid count, no mask, no branch per code;
//   * counters are 32-bit words (address = code * 4, increment 1);
//   * a 16-byte piece is parked as it was loaded; the list a piece belongs to is found from anchor[] (the list of every fourth piece) and a
//     short walk over coff8[];
//   * phase 1 hands the PIECES out to the lanes (piece q to lane q mod 1024), not the lists to groups of four lanes with a second round for what
//     lies beyond a list's first 32 entries: the average list has 17 entries, so a third of the lanes had a piece to count, and an LDS atomic costs
//     what it costs per wave-instruction (6.0 cycles with every third lane active, 7.5 with all: tools/ubench/lds_rates) — 48 of them per wave and
//     read instead of 120, six loads per lane instead of fifteen, one round trip instead of two, and no table of further pieces to build.
struct SeedFilterStreamLds {
  uint32_t cnt[HF_SLOTS + HF_PAD_SLOTS];                        // hits per bin; the last 64: the padding entries' dummies
  uint32_t good[HF_SLOTS / 32], alive[HF_SLOTS / 32 + 4];       // alive[256 ..]: the pad codes' words, zero for the life of the workgroup
  uint32_t lstart8[SF_SMAX];                                    // first occurrence of every list / 8 (lists start on 64-byte sectors = multiples of 8 entries, padded_counts_kernel;
                                                                // an index of more than 2^35 padded occurrences — 275 GB of occ[] alone — does not fit a device)
  uint16_t lcnt[SF_SMAX];
  uint16_t coff8[SF_SMAX + 8];                                  // first code piece of every list (+ total)
  uint16_t anchor[SF_CHUNKS / 4];                               // the list piece 4 a belongs to
  uint32_t wsum[SF_THREADS / 64], wsum2[SF_THREADS / 64];
  uint32_t cursor, fallback, total8, hraw, tick[2], pad_[2];       // pad_[0]: the streaming kernel's group counter of phase 2
  ulonglong2 codes[SF_CHUNKS];                                  // 8 codes per piece, as loaded
};
static_assert(sizeof(SeedFilterStreamLds) <= 160 * 1024, "the streaming seed filter's LDS must fit one CU");
template <bool PROF>
__global__ void __launch_bounds__(SF_THREADS) __attribute__((amdgpu_waves_per_eu(SF_STREAM_WAVES_PER_EU, SF_STREAM_WAVES_PER_EU))) seed_filter_stream_kernel(IndexView I, const uint32_t* __restrict__ sk_hash, const uint64_t* __restrict__ off,
                                                                        const int32_t* __restrict__ sk_n, const int32_t* __restrict__ read_len,
                                                                        const int32_t* __restrict__ min_hits, uint32_t* __restrict__ surv_n,
                                                                        uint64_t* __restrict__ stage, const uint64_t* __restrict__ stage_off,
                                                                        uint8_t* need_old, const uint32_t* __restrict__ cls_words /* = need_old, read-only view */,
                                                                        uint32_t* __restrict__ raw_hits, int n_reads, uint32_t* __restrict__ ticket,
                                                                        unsigned long long* __restrict__ prof /* optional (MM_SF_PROF): cycles per phase, summed over the workgroups */) {
  extern __shared__ __align__(16) unsigned char sf_dyn[];
  SeedFilterStreamLds& L = *reinterpret_cast<SeedFilterStreamLds*>(sf_dyn);
  const ulonglong2* __restrict__ tab = reinterpret_cast<const ulonglong2*>(I.tab);
  const uint64_t tslots = (uint64_t)I.tab_buckets << 2;
  uint32_t hq[SF_LPG]; ulonglong2 vq[SF_LPG];                      // the look-ups in flight: hashes and home-sector slots of the NEXT read
  int r_cur = 0, s_cur = 0; uint64_t o_cur = 0;
  unsigned long long pt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = 0;
  auto lapp = [&](int i) { if (PROF) { const unsigned long long t = __builtin_readcyclecounter(); pt[i] += t - pt0; pt0 = t; } };
  if (PROF) pt0 = __builtin_readcyclecounter();
  
  for (int it = -1; it < 0 || r_cur < n_reads; ++it) {           // it = -1: the prologue (first ticket, first look-ups)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wid = tid >> 6;
    const int grp = tid >> 2, sub = tid & 3, gshift = lane & ~3;
    auto uni64 = [](uint64_t v) { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32; };
    // class, sketch size and offset of read r: three independent scalar loads (cls_words aliases need_old read-only: the class byte of
    // read r is written by the host before the launch and by the workgroup that handles r; nobody else's view of it matters).
    // A read this kernel does not take (another class, or no sketch) comes back with s = 0.
    auto head = [&](int r, int& s, uint64_t& o) {
      const int rr = min(r, n_reads - 1);
      const uint32_t cw = cls_words[rr >> 2]; const int sn = sk_n[rr]; const uint64_t on = off[rr];
      s = 0; o = 0;
      if (r < n_reads && !((cw >> (8 * (rr & 3))) & 0xffu)) {
        s = sn; o = on;
        if (s <= 0) { s = 0; if (tid == 0) { surv_n[r] = 0; raw_hits[r] = 0; } }
      }
    };
    // (unconditional loads off a scalar base with 32-bit lane offsets, the index clamped into the sketch: a load under its own branch,
    // or one whose address registers are recycled, gets a vector-memory wait in front of it — eleven serial round trips; lanes beyond
    // the sketch look a valid hash up again and ignore the answer)
    auto load_hashes = [&](int s_, uint64_t o_) {
      const int su = __builtin_amdgcn_readfirstlane(s_);
      const char* __restrict__ hb = reinterpret_cast<const char*>(sk_hash + uni64(o_));
      const uint32_t last = (uint32_t)max(su - 1, 0);
      if (su > 0) {
#pragma unroll
        for (int u = 0; u < SF_LPG; ++u) hq[u] = *reinterpret_cast<const uint32_t*>(hb + (size_t)(min((uint32_t)(grp + SF_GROUPS * u), last) << 2));
      } else {
#pragma unroll
        for (int u = 0; u < SF_LPG; ++u) hq[u] = 0u;
      }
    };
    auto issue_lookups = [&]() {
#pragma unroll
      for (int u = 0; u < SF_LPG; ++u) asm volatile("" : "+v"(hq[u]));   // (the hashes are first used HERE: keeps the slot arithmetic, and the wait for the hash loads with it, from drifting up to the loads)
      int sub_ = (int)threadIdx.x & 3;
      asm volatile("" : "+v"(sub_));                               // (a value of its own: the lane's table address of the resolve step need not live — in scratch — until here)
#pragma unroll
      for (int u = 0; u < SF_LPG; ++u) vq[u] = tab[tab_slot(hq[u], I.tab_buckets) + sub_];
    };
    if (it < 0) {
      if (tid == 0) L.tick[0] = atomicAdd(ticket, 1u);
      if (tid < 4) L.alive[HF_SLOTS / 32 + tid] = 0;              // (the pad codes' bins: never alive)
      lds_barrier();
      r_cur = __builtin_amdgcn_readfirstlane((int)L.tick[0]);
      head(r_cur, s_cur, o_cur);
      load_hashes(s_cur, o_cur);
      issue_lookups();
      continue;
    }
    if (tid == 0) L.tick[(it + 1) & 1] = atomicAdd(ticket, 1u);   // the read after this one (read by all after the next barrier)
    int r_next = n_reads, s_next = 0; uint64_t o_next = 0;
    bool next_issued = false, next_known = false;
    if (s_cur > 0) {
      const int r = __builtin_amdgcn_readfirstlane(r_cur), s = __builtin_amdgcn_readfirstlane(s_cur);
      // what phase 2 needs of the read, fetched now (scalar loads)
      const uint64_t stage_base = stage_off[r];
      const uint32_t stage_cap = (uint32_t)(stage_off[r + 1] - stage_base);
      const uint32_t len = (uint32_t)max(read_len[r], 1);
      const int nb = min((int)((len - 1) >> HF_BIN_SHIFT) + 2, HF_SLOTS);
      int m = min_hits[r]; if (m < 1) m = 1;
      {
        uint32_t z = 0;
        asm volatile("" : "+v"(z));                                // (a zero made here: hoisted out of the loop, a register pair of zeros is kept in scratch, and its reload waits for the look-ups)
        for (int i = tid; i < (HF_SLOTS + HF_PAD_SLOTS) / 4; i += SF_THREADS) reinterpret_cast<uint4*>(L.cnt)[i] = make_uint4(z, z, z, z);
        if (tid < (int)(sizeof L.lcnt / 16)) reinterpret_cast<uint4*>(L.lcnt)[tid] = make_uint4(z, z, z, z);
        if (tid == 0) { L.cursor = z; L.fallback = z; L.pad_[0] = z; }
      }
      lds_barrier();
      lapp(0);
      // the next read's ticket is visible: its class, sketch size and offset are on their way while this read's look-ups are resolved
      r_next = __builtin_amdgcn_readfirstlane((int)L.tick[(it + 1) & 1]);
      head(r_next, s_next, o_next);
      next_known = true;
      // ---- phase 0: resolve the look-ups issued during the previous read.  Round 0 looks at all eleven answers and re-issues, for the
      // lane groups whose home sector was full without a match, the next sector; round 1 (rarely 2) looks at those.
      // pass 1: every answer looked at once; a lane group whose home sector is full without a match asks for the next sector — all such
      // requests of the lane are in flight together; pass 2 takes them up (and probes on, one sector at a time, in the rare case)
      // (lcnt[] is zero from the top of the iteration: only a hash that is found and kept writes its list; absent or cut by freqThreshold = no list)
      auto settle = [&](int i, uint32_t h, const ulonglong2& v, bool pending) -> bool {   // true: the look-up of this lane group is done
        // slots are filled in probing order and never emptied: a match is the key's slot, an empty slot without one means absent
        const bool match = pending && v.x != 0 && (uint32_t)v.x == h, empty = pending && v.x == 0;
        const uint32_t done = (uint32_t)(__ballot(match || empty) >> gshift) & 0xfu;
        if (match) {
          const uint32_t cnt = (uint32_t)(v.x >> 32);
          if ((uint64_t)cnt < (uint64_t)(int64_t)I.freq_threshold) {   // computeMap.hpp:317
            if (cnt > 0xffffu) L.fallback = 1;                    // (a list this long overflows the code area anyway)
            L.lcnt[i] = (uint16_t)cnt; L.lstart8[i] = (uint32_t)(v.y >> 3);
          }
        }
        return !pending || done != 0;
      };
      uint32_t pmask = 0;
#pragma unroll
      for (int u = 0; u < SF_LPG; ++u) {
        const int i = grp + SF_GROUPS * u;
        if (!settle(i, hq[u], vq[u], i < s)) { pmask |= 1u << u; vq[u] = tab[tab_next_sector(tab_slot(hq[u], I.tab_buckets), tslots) + sub]; }
      }
      lapp(7);                                                    // (first answers looked at, second probes issued)
      if (__any(pmask != 0)) {
#pragma unroll
        for (int u = 0; u < SF_LPG; ++u) {
          const int i = grp + SF_GROUPS * u;
          const uint32_t h = hq[u]; uint64_t slot = tab_next_sector(tab_slot(h, I.tab_buckets), tslots); ulonglong2 v = vq[u];
          bool pending = (pmask >> u) & 1u;
          while (__any(pending)) {
            if (settle(i, h, v, pending)) pending = false;
            if (pending) { slot = tab_next_sector(slot, tslots); v = tab[slot + sub]; }
          }
        }
      }
      lds_barrier();
      lapp(1);
      // ---- code chunk offsets (seed_filter_kernel)
      {
        uint32_t c8[3], hr = 0, mine = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int i = tid * 3 + j; const uint32_t c = i < s ? L.lcnt[i] : 0u; c8[j] = (c + 7) >> 3; mine += c8[j]; hr += c; }
        const uint32_t inc = (uint32_t)wave_incl_scan((int)mine), inc2 = (uint32_t)wave_incl_scan((int)hr);
        if (lane == 63) { L.wsum[wid] = inc; L.wsum2[wid] = inc2; }
        lds_barrier();
        uint32_t basew = 0, tot = 0, tot2 = 0;
#pragma unroll
        for (int q = 0; q < SF_THREADS / 64; ++q) { const uint32_t x = L.wsum[q]; if (q < wid) basew += x; tot += x; tot2 += L.wsum2[q]; }
        uint32_t ex = basew + inc - mine;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int i = tid * 3 + j;
          if (i <= s) L.coff8[i] = (uint16_t)min(ex, 0xffffu);
          for (uint32_t a = (ex + 3) >> 2, a1 = min((ex + c8[j] + 3) >> 2, (uint32_t)(SF_CHUNKS / 4)); a < a1; ++a) L.anchor[a] = (uint16_t)i;   // pieces 4 a of this list
          ex += c8[j];
        }
        if (tid == 0) { L.total8 = tot; L.hraw = tot2; if (tot > (uint32_t)SF_CHUNKS || tot2 > 65535u) L.fallback = 1; }
      }
      lds_barrier();
      lapp(2);
      if (L.fallback) { if (tid == 0) { need_old[r] = 1; surv_n[r] = 0; raw_hits[r] = 0; } }
      else {
        // the next read's hashes: requested here, in front of the pieces (six pieces in flight leave the registers for them): they are there
        // long before the look-ups are issued behind the window sums
        load_hashes(s_next, o_next);
        // ---- phase 1: every 16-byte piece of every list once: piece q to lane q mod 1024, all loads of a lane in flight together, then all
        // eight codes of a piece counted (the pads behind a list's last entry in their dummies) and the piece parked as it is
        {
          constexpr int NP = SF_CHUNKS / SF_THREADS;
          const uint32_t T8u = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.total8);
          ulonglong2 v[NP];
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            v[j] = make_ulonglong2(0, 0);
            if ((uint32_t)(j * SF_THREADS) < T8u) {                // (wave-uniform; lanes beyond the last piece ask for it again and drop the answer)
              const uint32_t q = min((uint32_t)(tid + j * SF_THREADS), T8u - 1);
              uint32_t li = L.anchor[q >> 2];
              while ((uint32_t)L.coff8[li + 1] <= q) ++li;
              v[j] = *reinterpret_cast<const ulonglong2*>(I.occ16 + ((uint64_t)L.lstart8[li] << 3) + (uint64_t)(q - (uint32_t)L.coff8[li]) * 8u);
            }
          }
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            const uint32_t q = (uint32_t)(tid + j * SF_THREADS);
            if (q < T8u) {
              const ulonglong2 x = v[j];
#pragma unroll
              for (int t = 0; t < 8; ++t) atomicAdd(&L.cnt[(uint32_t)((t < 4 ? x.x : x.y) >> (16 * (t & 3))) & 0xffffu], 1u);
              L.codes[q] = x;
            }
          }
          lapp(8);                                                 // (all pieces loaded, counted and parked)
        }
        lds_barrier();
        lapp(3);
        {
          // good[b] = the nb bins from b on hold minimumHits hits.  A lane takes the bins tid, tid + 1024, ...: neighbouring lanes read neighbouring
          // counters (eight consecutive bins per lane — the sliding form — put the 64 lanes of a read on four LDS banks), and a wave's 64 answers are one ballot
          const uint32_t* cnt = L.cnt;
#pragma unroll
          for (int kk = 0; kk < HF_SLOTS / SF_THREADS; ++kk) {
            const int b = tid + kk * SF_THREADS;
            uint32_t sum = 0;
            for (int i = 0; i < nb; ++i) sum += cnt[(b + i) & (HF_SLOTS - 1)];
            const uint64_t gb = __ballot(sum >= (uint32_t)m);
            if (lane == 0) { L.good[(b >> 5)] = (uint32_t)gb; L.good[(b >> 5) + 1] = (uint32_t)(gb >> 32); }
          }
        }
        lds_barrier();
        lapp(9);                                                   // (window sums)
        if (tid < HF_SLOTS / 32) {
          uint32_t al = 0;
          for (int j = 0; j < nb; ++j) {
            const int wsh = j >> 5, bsh = j & 31;
            const uint32_t g0 = L.good[(tid - wsh) & 255], g1 = L.good[(tid - wsh - 1) & 255];
            al |= bsh ? (g0 << bsh) | (g1 >> (32 - bsh)) : g0;
          }
          L.alive[tid] = al;
        }
        lapp(10);                                                  // (alive)
        lds_barrier();
        lapp(4);
        // The next read's home sectors: in flight from here to the top of the next iteration.  Issued BEHIND the barrier that publishes alive[] (round 6; until then in front
        // of it): the look-ups leave a CU at the rate its address path takes them (2 816 sectors, the kernel's bound), and a wave whose eleven are out goes on to
        // the bit tests instead of waiting at the barrier for the last wave's — the step 38.9-39.5 -> 36.4-38.2 ms in turns on one box, this kernel 12.6-13.1 -> 11.9-12.6 ms
        // (profiles/r06_ab_k3_barrier_first.txt).
        issue_lookups();
        lapp(11);                                                  // (hashes arrived, look-ups issued)
        next_issued = true;
        // ---- phase 2: bit tests over the parked codes (seed_filter_kernel, phase 2)
        uint64_t* const dst = stage + stage_base;
        uint16_t* const sv = reinterpret_cast<uint16_t*>(L.cnt);   // (2 x 8 256 slots: stage_cap = 1024 + 2 x sketch size <= 6 656, unless the test hook MM_HF_STAGE_CAP says otherwise)
        const uint32_t sv_cap = min(stage_cap, (uint32_t)(2 * (HF_SLOTS + HF_PAD_SLOTS)));
        const uint32_t T8 = L.total8;
        // Groups of 64 pieces handed out by a counter (round 6; until then piece q0 + tid for q0 = 0, 1 024, ...): the waves reach this phase one after the other — each
        // as its look-ups are out — and a wave that comes early takes more groups instead of waiting at the barrier behind the phase for the wave that comes last
        // (that barrier: 14 % of the kernel's cycles -> 1 %; the kernel 12.45 -> 12.16 ms over three alternations on one box).  The order of the survivor slots was
        // already the order in which the waves reach the cursor.
        uint32_t cv = 0;
        if (lane == 0) cv = atomicAdd(&L.pad_[0], 1u);
        for (uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cv); c * 64u < T8; c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cv)) {
          if (lane == 0) cv = atomicAdd(&L.pad_[0], 1u);         // (the next group: asked for before this one is worked on)
          const uint32_t q = c * 64u + (uint32_t)lane;
          uint32_t mask = 0;
          if (q < T8) {
            const ulonglong2 x = L.codes[q];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const uint32_t code = (uint32_t)((t < 4 ? x.x : x.y) >> (16 * (t & 3))) & 0xffffu;   // (a pad: one of the bins nothing is alive in)
              mask |= ((L.alive[code >> 5] >> (code & 31)) & 1u) << t;
            }
          }
          const int mine = __popc(mask);
          const int incl = wave_incl_scan(mine);
          const int total = __builtin_amdgcn_readlane(incl, 63);
          if (total == 0) continue;
          uint32_t base = 0;
          if (lane == 63) base = atomicAdd(&L.cursor, (uint32_t)total);
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
          uint32_t pos = base + (uint32_t)(incl - mine);
          while (mask) {                                           // a survivor is noted as piece << 3 | entry, 16 bits, where the counters were (they are done with)
            const int t = __ffs(mask) - 1; mask &= mask - 1;
            if (pos < sv_cap) sv[pos] = (uint16_t)(q << 3 | (uint32_t)t);
            ++pos;
          }
        }
        lapp(12);                                                  // (bit tests, survivors noted)
        lds_barrier();
        lapp(5);
        const uint32_t n_s = L.cursor;
        if (n_s > sv_cap) { if (tid == 0) { need_old[r] = 1; surv_n[r] = 0; raw_hits[r] = 0; } }   // stage too small: the two-pass kernels redo the read
        else {
          // the occurrence of every survivor: list start + position in the list (the list of a piece: from the anchor of its group of four,
          // past the lists that end at or before it).  Noted in global memory and read back behind a full barrier, as until round 5, this
          // cost a store, a round trip and a wait for the look-ups in flight before the occurrences could even be asked for.
          for (uint32_t j = tid; j < n_s; j += SF_THREADS) {
            const uint32_t e = sv[j], q = e >> 3;
            uint32_t li = L.anchor[q >> 2];
            while ((uint32_t)L.coff8[li + 1] <= q) ++li;
            dst[j] = I.occ[((uint64_t)L.lstart8[li] << 3) + (uint64_t)(q - (uint32_t)L.coff8[li]) * 8u + (e & 7u)] & ~(uint64_t)(PW_DP | PW_DN);
          }
          if (tid == 0) { surv_n[r] = n_s; raw_hits[r] = L.hraw; }
        }
      }
    }
    if (!next_issued) {                                          // a read that was skipped or fell back: nothing to hide the look-ups behind
      lds_barrier();
      if (!next_known) { r_next = __builtin_amdgcn_readfirstlane((int)L.tick[(it + 1) & 1]); head(r_next, s_next, o_next); }
      load_hashes(s_next, o_next);
      issue_lookups();
    }
    lds_barrier();                                               // the LDS areas are free for the next read
    lapp(6);
    r_cur = r_next; s_cur = s_next; o_cur = o_next;
  }
  if (PROF && threadIdx.x == 0) for (int i = 0; i < 16; ++i) atomicAdd(&prof[i], pt[i]);
}

// range blockIdx.x of src, [sb, se), goes to dst starting at db
__global__ void __launch_bounds__(256) move_ranges_kernel(const uint64_t* __restrict__ src, const uint64_t* __restrict__ sb, const uint64_t* __restrict__ se,
                                                          uint64_t* __restrict__ dst, const uint64_t* __restrict__ db) {
  const uint64_t s0 = sb[blockIdx.x], n = se[blockIdx.x] - s0, d0 = db[blockIdx.x];
  for (uint64_t i = threadIdx.x; i < n; i += 256) dst[d0 + i] = src[s0 + i];
}
// sum of the probe counts (= raw seed hits of the batch); the filter path needs no per-list offsets, only this total
__global__ void __launch_bounds__(256) sum_u32_kernel(const uint32_t* __restrict__ v, int64_t n, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += v[i];
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// debug tap (mm_debug_probed_lists): how long are the occurrence lists the sketches of a batch ask for?  One thread per sketch hash;
// hist[0] = hash not in the index, hist[c] = lists of c entries (c < nb - 2), hist[nb - 2] = longer lists that are kept,
// hist[nb - 1] = lists cut by freqThreshold (computeMap.hpp:317)
__global__ void __launch_bounds__(256) probed_list_hist_kernel(IndexView I, const uint32_t* __restrict__ sk_hash, const uint64_t* __restrict__ off,
                                                               const int32_t* __restrict__ sk_n, int nb, unsigned long long* __restrict__ hist) {
  const int r = blockIdx.x, s = sk_n[r];
  const uint64_t o = off[r];
  for (int i = threadIdx.x; i < s; i += 256) {
    uint32_t cnt = 0; uint64_t start = 0;
    int b = 0;
    if (index_find(I, sk_hash[o + i], &cnt, &start)) b = (uint64_t)cnt < (uint64_t)(int64_t)I.freq_threshold ? (int)min(cnt, (uint32_t)(nb - 2)) : nb - 1;
    atomicAdd(&hist[b], 1ull);
  }
}

__global__ void read_hit_bounds_kernel(const uint64_t* __restrict__ off, const uint64_t* __restrict__ hit_off, int64_t n,
                                       uint64_t* __restrict__ read_hit_off) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r <= n) read_hit_off[r] = hit_off[off[r]];
}

// ---------------------------------------------------------------------------------------------------
// K4a  sort the seed hits of each read by (contig, wpos)          computeMap.hpp:353
// ---------------------------------------------------------------------------------------------------
template <bool IN_LDS>
__global__ void __launch_bounds__(256) sort_hits_kernel(uint64_t* __restrict__ hits, const uint64_t* __restrict__ read_hit_off,
                                                        const int32_t* __restrict__ read_list, int npow2, uint64_t* __restrict__ gscratch) {
  extern __shared__ __align__(16) uint64_t skeys[];
  const int r = read_list[blockIdx.x];
  const uint64_t o = read_hit_off[r];
  const int n = (int)(read_hit_off[r + 1] - o);
  uint64_t* a = IN_LDS ? skeys : gscratch + (size_t)blockIdx.x * npow2;
  for (int i = threadIdx.x; i < npow2; i += 256) a[i] = i < n ? hits[o + i] : ~0ull;
  __syncthreads();
  bitonic_sort_u64(a, npow2);
  for (int i = threadIdx.x; i < n; i += 256) hits[o + i] = a[i];
}

// The same with an LDS radix sort over the significant key bits (contig in the high word, position and strand below it):
// fewer instructions than the bitonic network and no padding to a power of two.  256 * IPT >= hits of the longest read of the class.
template <int IPT>
__global__ void __launch_bounds__(256) sort_hits_radix_kernel(uint64_t* __restrict__ hits, const uint64_t* __restrict__ read_hit_off,
                                                              const int32_t* __restrict__ read_list, int end_bit,
                                                              const uint64_t* __restrict__ stage /* optional: staged survivors of the filter ... */,
                                                              const uint64_t* __restrict__ stage_off /* ... which hold a read's hits whenever they fit its stage */) {
  using Sort = rocprim::block_radix_sort<uint64_t, 256, IPT>;
  extern __shared__ __align__(16) unsigned char sort_dyn[];
  typename Sort::storage_type& tmp = *reinterpret_cast<typename Sort::storage_type*>(sort_dyn);
  const int r = read_list[blockIdx.x];
  const uint64_t o = read_hit_off[r];
  const int n = (int)(read_hit_off[r + 1] - o);
  const uint64_t* __restrict__ src = hits + o;
  if (stage) { const uint64_t sb = stage_off[r]; if ((uint64_t)n <= stage_off[r + 1] - sb) src = stage + sb; }   // (then the filter's write kernel left hits[] alone)
  uint64_t key[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) { const int idx = threadIdx.x * IPT + i; key[i] = idx < n ? src[idx] : ~0ull; }
  Sort().sort(key, tmp, 0, end_bit);                             // blocked: thread t holds sorted positions t*IPT ..  (padding keys sort last)
#pragma unroll
  for (int i = 0; i < IPT; ++i) { const int idx = threadIdx.x * IPT + i; if (idx < n) hits[o + idx] = key[i]; }
}

// ---------------------------------------------------------------------------------------------------
// K4b  L1 candidate scan, one thread per read, the reference's loop verbatim in behaviour
//      (computeL1CandidateRegions, computeMap.hpp:346-386).  WRITE=false counts, WRITE=true writes.
// ---------------------------------------------------------------------------------------------------
template <bool WRITE>
__global__ void l1_scan_kernel(const uint64_t* __restrict__ hits, const uint64_t* __restrict__ read_hit_off, const int32_t* __restrict__ read_len,
                               const int32_t* __restrict__ min_hits, int64_t n_reads, uint32_t* __restrict__ cand_n,
                               const uint64_t* __restrict__ cand_off, int32_t* __restrict__ cand, int32_t* __restrict__ cand_read) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t o = read_hit_off[r];
  const int64_t H = (int64_t)(read_hit_off[r + 1] - o);
  const int len = read_len[r];
  int m = min_hits[r]; if (m < 1) m = 1;                         // :349
  uint32_t nc = 0;
  int lseq = -1, lstart = 0, lend = 0;
  uint64_t wbase = WRITE ? cand_off[r] : 0;
  auto flush = [&]() {
    if (lseq < 0) return;
    if (WRITE) { int32_t* c = cand + 3 * (wbase + nc); c[0] = lseq; c[1] = lstart; c[2] = lend; cand_read[wbase + nc] = (int32_t)r; }
    ++nc;
  };
  for (int64_t i = 0; i + m <= H; ++i) {
    uint64_t a = hits[o + i], b = hits[o + i + m - 1];
    int sa = (int)(a >> 32), sb = (int)(b >> 32);
    int wa = pw_wpos((uint32_t)a), wb = pw_wpos((uint32_t)b);
    if (sa != sb || wb - wa >= len) continue;                    // :365
    int cs = max(0, wb - len + 1), ce = wa;                      // :368
    if (lseq == sa && lend >= cs) lend = max(ce, lend);          // :374-380
    else { flush(); lseq = sa; lstart = cs; lend = ce; }
  }
  flush();
  if (!WRITE) cand_n[r] = nc;
}

// The same loop, one wavefront per read.  Hits are sorted by (contig, position), so the merged region so far ends at the
// position of the latest qualifying hit: hit i opens a new candidate iff the previous qualifying hit lies on another contig
// or before max(0, wpos[i+m-1]-len+1).  That makes every decision local (ballot + one shuffle); a candidate's end is written
// by the last qualifying hit before the next opening one, later chunks of the same candidate simply overwrite it.
template <bool WRITE>
__global__ void __launch_bounds__(256) l1_wave_kernel(const uint64_t* __restrict__ hits, const uint64_t* __restrict__ read_hit_off,
                                                      const int32_t* __restrict__ read_len, const int32_t* __restrict__ min_hits, int64_t n_reads,
                                                      uint32_t* __restrict__ cand_n, const uint64_t* __restrict__ cand_off, int32_t* __restrict__ cand,
                                                      int32_t* __restrict__ cand_read, int32_t* __restrict__ cand_hint /* optional: seed hits inside the candidate */) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n_reads) return;
  const uint64_t o = read_hit_off[r];
  const int64_t H = (int64_t)(read_hit_off[r + 1] - o);
  const int len = read_len[r];
  int m = min_hits[r]; if (m < 1) m = 1;                         // :349
  const uint64_t wbase = WRITE ? cand_off[r] : 0;
  int count = 0, prev_seq = -1, prev_wa = 0;
  int64_t open_i = 0;                                            // the hit that opened the candidate the previous chunk ended in
  for (int64_t base = 0; base + m <= H; base += 64) {
    const int64_t i = base + lane;
    const bool valid = i + m <= H;
    uint64_t a = 0, b = 0;
    if (valid) { a = hits[o + i]; b = hits[o + i + m - 1]; }
    const int sa = (int)(a >> 32), sb = (int)(b >> 32), wa = pw_wpos((uint32_t)a), wb = pw_wpos((uint32_t)b);
    const bool q = valid && sa == sb && wb - wa < len;           // :365
    const int cs = max(0, wb - len + 1);                         // :368
    const uint64_t qm = __ballot(q);
    const uint64_t below = qm & ((1ull << lane) - 1ull);
    const int pl = below ? 63 - __builtin_clzll(below) : 0;
    const int p_seq_l = __shfl(sa, pl, 64), p_wa_l = __shfl(wa, pl, 64);
    const int p_seq = below ? p_seq_l : prev_seq, p_wa = below ? p_wa_l : prev_wa;
    const bool brk = q && !(p_seq == sa && p_wa >= cs);          // :374-380
    const uint64_t bm = __ballot(brk);
    if (WRITE && q) {
      const int k = count + __popcll(bm & ((2ull << lane) - 1ull)) - 1;
      const uint64_t above = lane < 63 ? qm & ~((2ull << lane) - 1ull) : 0ull;
      const bool last = above == 0ull || ((bm >> (__builtin_ctzll(above))) & 1ull);
      int32_t* c = cand + 3 * (wbase + (uint64_t)k);
      if (brk) { c[0] = sa; c[1] = cs; cand_read[wbase + (uint64_t)k] = (int32_t)r; }
      if (last) c[2] = wa;
      if (last && cand_hint) {
        // the seed hits of the candidate: from the hit that opened it to the last hit of the last qualifying run — with --all nearly all of them
        // lie inside ONE read-length window, so this is about what K5 will find as the matched count of its best window (mm_l2z.hpp: the band it predicts)
        const uint64_t opened = bm & ((2ull << lane) - 1ull);
        const int64_t oi = opened ? base + (63 - __builtin_clzll(opened)) : open_i;
        cand_hint[wbase + (uint64_t)k] = (int32_t)min((int64_t)0x7fffffff, i + m - oi);
      }
    }
    if (bm) open_i = base + (63 - __builtin_clzll(bm));
    count += __popcll(bm);
    if (qm) { const int ll = 63 - __builtin_clzll(qm); prev_seq = __shfl(sa, ll, 64); prev_wa = __shfl(wa, ll, 64); }
  }
  if (!WRITE && lane == 0) cand_n[r] = (uint32_t)count;
}

// ---------------------------------------------------------------------------------------------------
// result compaction: accepted candidates -> mapping records, read order preserved
// ---------------------------------------------------------------------------------------------------
// sums of the per-candidate work counters: one atomic per counter per block
// K5 workgroups of the 10 kb class (sketch <= 3072), made on the device: per read, its candidates in groups of four (four-wave
// workgroups); a remainder of one or two goes to a two-wave workgroup.  The same lists came from a host loop before, ~0.85 ms per
// 10^5 reads of branch mispredictions with the device waiting.  Group order across workgroups of this kernel is arbitrary (results are
// indexed by candidate).  ctr: [0] four-wave groups, [1] two-wave groups, [2] reads with candidates left to the host's classes,
// [3] largest sketch among the grouped reads.
__global__ void __launch_bounds__(256) l2_group_kernel(const uint64_t* __restrict__ cand_off, const int32_t* __restrict__ sk_n, const int32_t* __restrict__ read_len,
                                                       int64_t n, int min_len_dense, int dense_from, int no_small,
                                                       int32_t* __restrict__ gA0, int32_t* __restrict__ gAn, int32_t* __restrict__ gS0, int32_t* __restrict__ gSn,
                                                       unsigned int* __restrict__ ctr) {
  __shared__ unsigned int bA, bS, bOther, bMax, baseA, baseS;
  if (threadIdx.x == 0) { bA = 0; bS = 0; bOther = 0; bMax = 0; }
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t c_lo = 0, c_hi = 0; int sr = 0; bool mine = false;
  if (r < n) {
    c_lo = cand_off[r]; c_hi = cand_off[r + 1]; sr = sk_n[r];
    const bool dense = sr >= dense_from && sr < L2_SKETCH_LIMIT && read_len[r] >= min_len_dense;
    mine = c_hi > c_lo && sr <= 3072 && !dense;
    if (c_hi > c_lo && !mine) atomicAdd(&bOther, 1u);
  }
  const unsigned ncr = mine ? (unsigned)(c_hi - c_lo) : 0u, nfull = ncr >> 2, rem = ncr & 3u;
  const bool rem_small = rem != 0 && rem <= 2 && !no_small;
  const unsigned a = nfull + ((rem != 0 && !rem_small) ? 1u : 0u), b = rem_small ? 1u : 0u;
  unsigned la = 0, ls = 0;
  if (a) la = atomicAdd(&bA, a);
  if (b) ls = atomicAdd(&bS, b);
  if (mine) atomicMax(&bMax, (unsigned)sr);
  __syncthreads();
  if (threadIdx.x == 0) {
    baseA = bA ? atomicAdd(&ctr[0], bA) : 0u; baseS = bS ? atomicAdd(&ctr[1], bS) : 0u;
    if (bOther) atomicAdd(&ctr[2], bOther);
    if (bMax) atomicMax(&ctr[3], bMax);
  }
  __syncthreads();
  for (unsigned g = 0; g < a; ++g) { gA0[baseA + la + g] = (int32_t)(c_lo + 4u * g); gAn[baseA + la + g] = (int32_t)min(4u, ncr - 4u * g); }
  if (b) { gS0[baseS + ls] = (int32_t)(c_lo + 4u * nfull); gSn[baseS + ls] = (int32_t)rem; }
}

// K5 workgroups in the order of where their first candidate lies (contig, start): the reads of a sample cover their genomes several times over,
// so workgroups that run at the same time then stream overlapping pieces of pos[] and meet them in L2 / the Infinity Cache (tools/k3_locality.py:
// K5 -5 % with the reads of the bench batch in mapped order; the order of the workgroups is free, results are indexed by candidate).
__global__ void __launch_bounds__(256) l2_group_keys_kernel(const int32_t* __restrict__ g0, const int32_t* __restrict__ gn, const int32_t* __restrict__ cand, int64_t n,
                                                           uint64_t* __restrict__ key, uint64_t* __restrict__ val) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  const int32_t c0 = g0[g];
  key[g] = (uint64_t)(uint32_t)cand[3 * (int64_t)c0] << 32 | (uint32_t)cand[3 * (int64_t)c0 + 1];
  val[g] = (uint64_t)(uint32_t)c0 << 32 | (uint32_t)gn[g];
}
// xcds = 1: the sorted order as it is (the default).  xcds = 8 (MM_L2_XCD_ORDER=1, a measurement switch): the sorted list dealt out so that XCD x — workgroup p
// of a launch goes to XCD p mod 8, every XCD has its own L2 — works through the x-th eighth of it in order (sorted element i -> launch slot
// (i mod n/8) * 8 + i / (n/8)), neighbours in position sharing an L2 and not only the Infinity Cache.  Measured: 15.5 ms against 15.2 for the plain
// sorted order (16.0 unsorted) — eight fronts through the list leave each L2 a smaller share of the in-flight neighbours than one front does.
__global__ void __launch_bounds__(256) l2_group_unpack_kernel(const uint64_t* __restrict__ val, int64_t n, int xcds, int32_t* __restrict__ g0, int32_t* __restrict__ gn) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  const int64_t chunk = n / xcds;
  const int64_t p = (xcds > 1 && g < chunk * xcds) ? (g % chunk) * xcds + g / chunk : g;
  g0[p] = (int32_t)(val[g] >> 32); gn[p] = (int32_t)(uint32_t)val[g];
}

// The streamed range of every candidate (computeMap.hpp:466, :477) — first index entry at or beyond the candidate's start, first at or beyond its end + read length —
// one thread per candidate, both searches interleaved.  The zone kernel's waves did these searches themselves, one behind the other: eight dependent round trips in
// front of every candidate's stream (directory, bucket bounds, two 64-ary probes, twice).  Same lower bounds as contig_lower_bound_wpos (mm_l2.hpp).
__global__ void __launch_bounds__(256) l2_ranges_kernel(IndexView I, const int32_t* __restrict__ cand, const int32_t* __restrict__ cand_read, const int32_t* __restrict__ read_len,
                                                        int64_t n, int64_t* __restrict__ rng) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const int contig = cand[3 * c], rs = cand[3 * c + 1], re = cand[3 * c + 2];
  const int len = read_len[cand_read[c]];
  const int64_t cbeg = (int64_t)I.cstart[contig];
  const uint64_t d0 = I.dir_off[contig], nb = I.dir_off[contig + 1] - d0 - 1;
  const int t0 = rs, t1 = re + len;
  const uint64_t b0 = min((uint64_t)max(t0, 0) >> I.dir_shift, nb - 1), b1 = min((uint64_t)max(t1, 0) >> I.dir_shift, nb - 1);
  int64_t lo0 = cbeg + (int64_t)I.dir[d0 + b0], hi0 = cbeg + (int64_t)I.dir[d0 + b0 + 1];
  int64_t lo1 = cbeg + (int64_t)I.dir[d0 + b1], hi1 = cbeg + (int64_t)I.dir[d0 + b1 + 1];
  while (lo0 < hi0 || lo1 < hi1) {
    const int64_t m0 = lo0 < hi0 ? (lo0 + hi0) >> 1 : lo0, m1 = lo1 < hi1 ? (lo1 + hi1) >> 1 : lo1;
    const uint32_t p0 = I.pos[min(m0, I.N - 1)].pw, p1 = I.pos[min(m1, I.N - 1)].pw;
    if (lo0 < hi0) { if (pw_wpos(p0) < t0) lo0 = m0 + 1; else hi0 = m0; }
    if (lo1 < hi1) { if (pw_wpos(p1) < t1) lo1 = m1 + 1; else hi1 = m1; }
  }
  rng[2 * c] = lo0; rng[2 * c + 1] = max(lo0, lo1);
}

__global__ void __launch_bounds__(256) l2_stats_kernel(const L2Result* __restrict__ l2, int64_t n, unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long acc[5];
  if (threadIdx.x < 5) acc[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long a = 0, b = 0, c = 0, d = 0, e = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { a += l2[i].n_stream; b += l2[i].n_evals; c += l2[i].n_rebuilds; d += l2[i].pad2; e += (unsigned long long)l2[i].pad; }
  atomicAdd(&acc[0], a); atomicAdd(&acc[1], b); atomicAdd(&acc[2], c); atomicAdd(&acc[3], d); atomicAdd(&acc[4], e);
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(&counters[threadIdx.x], acc[threadIdx.x]);
  if (threadIdx.x == 3) atomicAdd(&counters[15], acc[3]);   // slide rounds (diagnostic)
  if (threadIdx.x == 4) atomicAdd(&counters[12], acc[4]);   // zone passes of the zone kernels (diagnostic)
}

__global__ void accept_flags_kernel(const L2Result* __restrict__ l2, int64_t n, uint32_t* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = l2[i].accepted ? 1u : 0u;
}
__global__ void write_records_kernel(const L2Result* __restrict__ l2, const int32_t* __restrict__ cand_read, const int32_t* __restrict__ sk_n,
                                     const uint32_t* __restrict__ flag, const uint64_t* __restrict__ rank, int64_t n,
                                     mm_map_record* __restrict__ rec) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  mm_map_record m;
  m.read = cand_read[i]; m.ref_contig = l2[i].contig; m.ref_start = l2[i].mean_pos; m.shared = l2[i].shared;
  m.sketch = sk_n[m.read]; m.strand = l2[i].strand; m.mapq = 0.0;
  rec[rank[i]] = m;
}
__global__ void read_rec_bounds_kernel(const uint64_t* __restrict__ cand_off, const uint64_t* __restrict__ rank, int64_t n_reads,
                                       uint64_t* __restrict__ rec_off) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r <= n_reads) rec_off[r] = rank[cand_off[r]];
}

// ---------------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------------
namespace {
struct SizeClass { int npow2; std::vector<int32_t> reads; };

// reads grouped by power-of-two capacity; entries beyond lds_cap elements use the global-memory variant
std::vector<SizeClass> make_classes(const std::vector<int64_t>& count, int min_pow2) {
  std::map<int, std::vector<int32_t>> m;
  for (size_t r = 0; r < count.size(); ++r) {
    if (count[r] <= 1) continue;                                // nothing to sort
    m[std::max(min_pow2, pow2_at_least(count[r]))].push_back((int32_t)r);
  }
  std::vector<SizeClass> v;
  for (auto& kv : m) v.push_back(SizeClass{kv.first, std::move(kv.second)});
  return v;
}
constexpr int LDS_SORT_MAX = 16384;     // 128 KiB of 64-bit keys

// MM_HOST_TIMING=1: wall time of the host sections between kernels (stderr)
struct HostLap {
  const bool on = getenv("MM_HOST_TIMING") != nullptr; std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void operator()(const char* what) { if (!on) return; const auto n = std::chrono::steady_clock::now(); fprintf(stderr, "host %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(n - t).count()); t = n; }
};

// Reads grouped by a small class id (n_classes: left out), input order kept inside a class: a counting sort without data-dependent
// branches, because these host loops sit between two kernels of a batch with the device waiting (std::map + push_back, or a branchy
// class function on mixed counts, cost ~0.6 ms per 10^5 reads in mispredictions).
struct ReadBins { std::vector<int32_t> order; std::vector<std::pair<int, size_t>> runs; };   // runs: (class, number of reads), ascending class, back to back in `order`
template <typename F>
ReadBins bin_reads(int64_t n, int n_classes, F&& class_of) {
  ReadBins b;
  std::vector<size_t> start((size_t)n_classes + 2, 0);
  std::vector<uint8_t> cls((size_t)std::max<int64_t>(n, 0));
  for (int64_t r = 0; r < n; ++r) { const unsigned c = (unsigned)class_of(r); cls[(size_t)r] = (uint8_t)c; ++start[(size_t)c + 1]; }
  for (int c = 0; c <= n_classes; ++c) start[(size_t)c + 1] += start[(size_t)c];
  b.order.resize((size_t)std::max<int64_t>(n, 0));               // the left-out reads land behind the classes and are cut off
  std::vector<size_t> at(start.begin(), start.end() - 1);
  for (int64_t r = 0; r < n; ++r) b.order[at[cls[(size_t)r]]++] = (int32_t)r;
  b.order.resize(start[(size_t)n_classes]);
  for (int c = 0; c < n_classes; ++c) if (start[(size_t)c + 1] > start[(size_t)c]) b.runs.emplace_back(c, start[(size_t)c + 1] - start[(size_t)c]);
  return b;
}

struct HostMz { uint32_t hash; int32_t wpos, strand; };
static bool host_less_by_hash(const HostMz& a, const HostMz& b) { return a.hash < b.hash; }   // base_types.hpp:70
static bool host_eq_by_hash(const HostMz& a, const HostMz& b) { return a.hash == b.hash; }    // base_types.hpp:66
}  // namespace

namespace {
// hipEvent pairs on the ctx stream; elapsed times are read once at the end of the batch
struct StageTimer {
  hipStream_t st;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<double*> dst;
  explicit StageTimer(hipStream_t s) : st(s) {}
  ~StageTimer() { for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } }
  size_t begin(double* target) {
    hipEvent_t a, b;
    MM_HIP(hipEventCreate(&a)); MM_HIP(hipEventCreate(&b));
    ev.push_back({a, b}); dst.push_back(target);
    MM_HIP(hipEventRecord(a, st));
    return ev.size() - 1;
  }
  void end(size_t i) { MM_HIP(hipEventRecord(ev[i].second, st)); }
  void collect() {
    MM_HIP(mm::stream_sync(st));
    for (size_t i = 0; i < ev.size(); ++i) { float ms = 0; MM_HIP(hipEventElapsedTime(&ms, ev[i].first, ev[i].second)); *dst[i] += ms; }
  }
};
}  // namespace

// one host-side duplicate-hash tie-break in flight (owned by map_batch's frame, never by its worker thread)
struct AmbState {
  std::thread bg;                          // (the destructor body joins it before any member is destroyed)
  std::vector<Rec> hr; std::vector<uint64_t> dof; std::vector<int32_t> expect; DBuf<uint64_t> d_so, d_do;
  std::vector<uint8_t> sv; std::vector<int32_t> scnt; std::atomic<int> mismatch{0};
  ~AmbState() { if (bg.joinable()) bg.join(); }
};

void map_batch(mm_ctx* ctx, const mm_index* I, const mm_seqset* reads, const mm_map_params& P, mm_mapping* M) {
  hipStream_t st = ctx->stream;
  StageTimer T(st);
  MM_REQUIRE(M->sketch_only || (I && I->k == P.k && I->w == P.w), MM_ERR_ARG, "index was built with different k / window size");   // (mm_sketch_batch: no index)
  const int64_t n = reads->count();
  MM_REQUIRE(n < (1LL << 31), MM_ERR_LIMIT, "more than 2^31 reads in one batch");
  M->ctx = ctx; M->n_reads = n; M->params = P; M->stats = mm_map_stats{};
  M->stats.n_reads = n;
  M->read_len = reads->len;
  M->active.assign((size_t)n, 0);
  for (int64_t r = 0; r < n; ++r) {
    int L = reads->len[(size_t)r];
    bool ok = !(L < P.w || L < P.k || L < P.min_read_len);      // computeMap.hpp:137
    M->active[(size_t)r] = ok;
    if (ok) { M->stats.n_reads_long_enough++; M->stats.bases_long_enough += L; }
  }
  const size_t t_total = T.begin(&M->stats.ms_total);
  // The minimizers and the sketch of a read do not depend on the index: when the same batch is mapped against one index chunk after
  // the other (--maxmemory; the reference runs the whole of mapSingleQuerySeq per chunk, computeMap.hpp:277-298 included), the second
  // and later mappings take them from the first (mm_map_batch_reusing: the two large arrays held jointly, the rest copied).  Strands the donor's tie-break has resolved meanwhile are the
  // strands this mapping would resolve them to (the same library calls on the same records).
  const mm_mapping* const donor = M->sketch_donor;
  // ---- K1
  if (donor) {
    const size_t t = T.begin(&M->stats.ms_minimizer);
    auto dcopy = [&](auto& dst, const auto& src) { dst.alloc(src.n); if (src.n) MM_HIP(hipMemcpyAsync(dst.p, src.p, src.bytes(), hipMemcpyDeviceToDevice, st)); };
    // the two big read-only arrays — minimizer records (8 B per minimizer) and sketch hashes (4 B) — are held jointly with the donor, not
    // copied (2.2 GB per chunk mapping of a 0.8 Gbp batch); the strand bytes and the per-read arrays are this mapping's own (its tie-break
    // writes to them)
    mm_mapping* const dn = const_cast<mm_mapping*>(donor);       // (only the ownership record of the two blocks changes)
    const bool same_ctx = donor->ctx == ctx;                     // (a block shared across contexts could go back to one context's cache while the other's stream still reads it)
    if (same_ctx) M->mz.rec.share_from(dn->mz.rec); else dcopy(M->mz.rec, donor->mz.rec);
    dcopy(M->mz.off, donor->mz.off);
    M->mz.h_off = donor->mz.h_off; M->mz.total = donor->mz.total;
    if (same_ctx) M->sk_hash.share_from(dn->sk_hash); else dcopy(M->sk_hash, donor->sk_hash);
    dcopy(M->sk_strand, donor->sk_strand); dcopy(M->sk_n, donor->sk_n); dcopy(M->amb, donor->amb);
    T.end(t);
  } else { size_t t = T.begin(&M->stats.ms_minimizer); run_minimizers(ctx, reads, P.k, P.w, M->active, false, M->mz); T.end(t); }
  const int64_t total_mz = M->mz.total;
  const std::vector<uint64_t>& hoff = M->mz.h_off;
  if (!donor) {
    M->sk_hash.alloc((size_t)std::max<int64_t>(total_mz, 1));
    M->sk_strand.alloc((size_t)std::max<int64_t>(total_mz, 1));
    M->sk_n.alloc((size_t)std::max<int64_t>(n, 1)); M->sk_n.zero(st);
    M->amb.alloc((size_t)std::max<int64_t>(n, 1)); M->amb.zero(st);
  }
  // ---- K2
  if (!donor) {
    size_t t_sk = T.begin(&M->stats.ms_sketch);
    // up to 16384 minimizers: radix sort in LDS, 4 ... 64 elements per thread.  The sort's cost follows the elements per thread, so the
    // reads are grouped by the capacity they really need (a 10 kb read has ~2 200 minimizers: 10 per thread instead of 16).
    // Single-element lists are "sorted" already but still need their sketch written.
    static const int ipts[] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64};
    uint8_t cls_of_need[66];                                      // elements per thread needed -> index into ipts; [0]: empty, [65]: beyond the LDS sort
    for (int need = 0, i = 0; need <= 64; ++need) { while (ipts[i] < need) ++i; cls_of_need[need] = (uint8_t)i; }
    cls_of_need[0] = 12; cls_of_need[65] = 12;
    uint64_t big_seen = 0;
    HostLap hl;
    const ReadBins RB = bin_reads(n, 12, [&](int64_t r) -> int {
      const uint64_t c = hoff[(size_t)r + 1] - hoff[(size_t)r];
      big_seen |= (uint64_t)(c > 16384);
      return cls_of_need[std::min<uint64_t>((c + 255) / 256, 65)];
    });
    const bool any_big = big_seen != 0;
    {
      hl("K2 bin");
      DBuf<int32_t> list(std::max<size_t>(RB.order.size(), 1));
      list.upload(RB.order.data(), RB.order.size(), st);
      hl("K2 list upload");
      size_t at = 0;
      for (auto& run : RB.runs) {
        const unsigned nb = (unsigned)run.second;
        const int32_t* lp = list.p + at;
        auto launch = [&](auto ipt_tag) {
          constexpr int IPT = decltype(ipt_tag)::value;
          using SortT = rocprim::block_radix_sort<uint32_t, 256, IPT, uint16_t>;
          using ScanT = rocprim::block_scan<int, 256>;
          const size_t lds = std::max(sizeof(typename SortT::storage_type), sizeof(typename ScanT::storage_type)) + 16;
          if (lds > 48 * 1024) MM_HIP(hipFuncSetAttribute((const void*)sketch_radix_kernel<IPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          sketch_radix_kernel<IPT><<<dim3(nb), dim3(256), lds, st>>>(M->mz.rec.p, M->mz.off.p, lp, M->sk_hash.p, M->sk_strand.p, M->sk_n.p, M->amb.p);
        };
        switch (ipts[run.first]) {
          case 4: launch(std::integral_constant<int, 4>{}); break;
          case 6: launch(std::integral_constant<int, 6>{}); break;
          case 8: launch(std::integral_constant<int, 8>{}); break;
          case 10: launch(std::integral_constant<int, 10>{}); break;
          case 12: launch(std::integral_constant<int, 12>{}); break;
          case 16: launch(std::integral_constant<int, 16>{}); break;
          case 20: launch(std::integral_constant<int, 20>{}); break;
          case 24: launch(std::integral_constant<int, 24>{}); break;
          case 32: launch(std::integral_constant<int, 32>{}); break;
          case 40: launch(std::integral_constant<int, 40>{}); break;
          case 48: launch(std::integral_constant<int, 48>{}); break;
          default: launch(std::integral_constant<int, 64>{}); break;
        }
        MM_KERNEL_CHECK();
        at += run.second;
      }
      hl("K2 launches");
      MM_HIP(mm::stream_sync(st));                          // RB.order is the source of the async upload
      hl("K2 sync (kernels)");
    }
    std::vector<int64_t> cnt;                                     // longer lists (reads beyond ~73 kb)
    if (any_big && !getenv("MM_SKETCH_BITONIC")) {                 // one segmented device sort (MM_SKETCH_BITONIC=1: the bitonic network below, cross-check)
      std::vector<int32_t> big; std::vector<uint64_t> koff{0};
      for (int64_t r = 0; r < n; ++r) { const uint64_t c = hoff[(size_t)r + 1] - hoff[(size_t)r]; if (c > 16384) { big.push_back((int32_t)r); koff.push_back(koff.back() + c); } }
      const size_t nb = big.size(); const uint64_t nk = koff.back();
      MM_REQUIRE(nk < ((uint64_t)1 << 32), MM_ERR_LIMIT, "more than 2^32 minimizers of reads beyond 16384 minimizers in one batch");
      DBuf<int32_t> d_big(nb); d_big.upload(big.data(), nb, st);
      DBuf<uint64_t> d_koff(nb + 1); d_koff.upload(koff.data(), nb + 1, st);
      DBuf<uint64_t> keys((size_t)nk), sorted((size_t)nk);
      sketch_keys_kernel<<<dim3((unsigned)nb), dim3(256), 0, st>>>(M->mz.rec.p, M->mz.off.p, d_big.p, d_koff.p, keys.p);
      MM_KERNEL_CHECK();
      size_t tmp_bytes = 0;
      MM_HIP(rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, keys.p, sorted.p, (unsigned int)nk, (unsigned int)nb, d_koff.p, d_koff.p + 1, 0, 64, st));
      DBuf<uint8_t> tmp(std::max<size_t>(tmp_bytes, 16));
      MM_HIP(rocprim::segmented_radix_sort_keys((void*)tmp.p, tmp_bytes, keys.p, sorted.p, (unsigned int)nk, (unsigned int)nb, d_koff.p, d_koff.p + 1, 0, 64, st));
      sketch_finish_kernel<<<dim3((unsigned)nb), dim3(256), 0, st>>>(M->mz.rec.p, M->mz.off.p, d_big.p, d_koff.p, sorted.p, M->sk_hash.p, M->sk_strand.p, M->sk_n.p, M->amb.p);
      MM_KERNEL_CHECK();
      MM_HIP(mm::stream_sync(st));                          // big / koff are upload sources
    } else if (any_big) { cnt.assign((size_t)n, 0); for (int64_t r = 0; r < n; ++r) { const int64_t c = (int64_t)(hoff[(size_t)r + 1] - hoff[(size_t)r]); if (c > 16384) cnt[(size_t)r] = c; } }
    for (auto& cls : make_classes(cnt, 256)) {
      DBuf<int32_t> list(cls.reads.size());
      list.upload(cls.reads.data(), cls.reads.size(), st);
      if (cls.npow2 <= LDS_SORT_MAX) {
        size_t lds = (size_t)cls.npow2 * 8;
        if (lds > 64 * 1024) MM_HIP(hipFuncSetAttribute((const void*)sketch_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        sketch_kernel<true><<<dim3((unsigned)cls.reads.size()), dim3(256), lds, st>>>(M->mz.rec.p, M->mz.off.p, list.p, cls.npow2, nullptr,
                                                                                     M->sk_hash.p, M->sk_strand.p, M->sk_n.p, M->amb.p);
        MM_KERNEL_CHECK();
      } else {
        DBuf<uint64_t> scratch((size_t)cls.npow2 * cls.reads.size());
        sketch_kernel<false><<<dim3((unsigned)cls.reads.size()), dim3(256), 0, st>>>(M->mz.rec.p, M->mz.off.p, list.p, cls.npow2, scratch.p,
                                                                                    M->sk_hash.p, M->sk_strand.p, M->sk_n.p, M->amb.p);
        MM_KERNEL_CHECK();
      }
      MM_HIP(mm::stream_sync(st));                          // cls.reads is the source of the async upload
    }
    T.end(t_sk);
  }
  HostLap hl;
  M->h_sk_n = M->sk_n.to_host(st, (size_t)n);
  if (M->sketch_only) {                                           // mm_sketch_batch: K1 + K2 alone, the donor of mm_map_batch_reusing
    for (int64_t r = 0; r < n; ++r) M->stats.sum_sketch += M->h_sk_n[(size_t)r];
    M->h_rec_off.assign((size_t)n + 1, 0); M->n_rec = 0;          // an empty mapping for every call that reads records
    T.end(t_total);
    T.collect();
    return;
  }
  // mm_map_batch_phased, stage 1: K1 + K2 are complete (the download above waited for them), nothing of the seed stage is enqueued yet
  if (M->at_stage) M->at_stage(M->at_stage_user, 1);
  std::vector<uint8_t> h_amb = M->amb.to_host(st, (size_t)n);
  hl("post-K2 downloads");
  {
    // Reads whose sketch has >= 32768 hashes (~145 kb at w = 8) are beyond the LDS-resident window state of the K5 classes:
    // their candidates go through l2_giant_kernel (state in global memory); counted for the caller's information only.
    int64_t giant = 0;
    for (int64_t r = 0; r < n; ++r) if (M->h_sk_n[(size_t)r] >= L2_SKETCH_LIMIT) ++giant;
    M->stats.n_reads_giant = giant;
  }
  std::function<void()> amb_finish;
  // ---- duplicate-hash strand tie-break (computeMap.hpp:292-295: std::sort is not stable, std::unique keeps
  //      whichever equal-hash element introsort left first).  Only the strand of the survivor is observable
  //      (slidingMap.hpp:247), so it is resolved here with the same library calls on the same input order.
  //      Entries whose strand depends on that are marked by K2 (bit 1 of the strand byte); the library sort is only run for
  //      reads whose strand vote actually read such an entry (found out by K6, amb_used[]), and those few candidates are
  //      then redone.  Reads sorted by the bitonic kernel (> 16 384 minimizers) carry no marks and are resolved up front.
  std::vector<int64_t> eager_reads, lazy_reads;
  {
    const bool all_eager = getenv("MM_EAGER_TIEBREAK") != nullptr;   // tests that compare every sketch strand with the oracle
    // (MM_L2_NO_DENSE=1 sends sketches of >= 32768 hashes to l2_giant_kernel, which has no "vote read an unresolved strand" feedback)
    const bool giant_eager = getenv("MM_L2_NO_DENSE") != nullptr;
    for (int64_t r = 0; r < n; ++r) if (h_amb[(size_t)r])
      ((h_amb[(size_t)r] == 2 && !all_eager && !(giant_eager && M->h_sk_n[(size_t)r] >= L2_SKETCH_LIMIT)) ? lazy_reads : eager_reads).push_back(r);
    M->stats.n_ambiguous_sketch_reads = (int64_t)(eager_reads.size() + lazy_reads.size());
  }
  // The tie-break states live in this frame: whatever way map_batch is left (return, MM_REQUIRE, a failed allocation), their
  // destructors join the worker first and release the device buffers on this thread.  The worker only sees a raw pointer and
  // its own copies of the host data it reads.
  std::vector<std::unique_ptr<AmbState>> amb_states;
  // starts the host work for `amb_reads` on background threads and returns the closure that joins it and patches the strands
  auto start_tiebreak = [&](const std::vector<int64_t>& amb_reads) -> std::function<void()> {
    const size_t na = amb_reads.size();
    std::vector<uint64_t> so(na), dof(na + 1, 0);
    std::vector<int32_t> expect(na);
    for (size_t i = 0; i < na; ++i) {
      int64_t r = amb_reads[i];
      so[i] = hoff[(size_t)r];
      dof[i + 1] = dof[i] + (hoff[(size_t)r + 1] - hoff[(size_t)r]);
      expect[i] = M->h_sk_n[(size_t)r];
    }
    DBuf<uint64_t> d_so(na), d_do(na + 1);
    d_so.upload(so.data(), na, st); d_do.upload(dof.data(), na + 1, st);
    DBuf<Rec> comp((size_t)dof[na]);
    gather_amb_kernel<<<dim3((unsigned)na), dim3(256), 0, st>>>(M->mz.rec.p, d_so.p, d_do.p, comp.p);
    MM_KERNEL_CHECK();
    // The library sort of ~1 % of the reads is the only per-read host work of a batch.  Only the L2 strand vote needs its
    // result, so it runs on host threads while the device goes through K3 and K4.
    amb_states.push_back(std::make_unique<AmbState>());
    AmbState* const A = amb_states.back().get();
    A->hr = comp.to_host(st);
    A->dof = dof; A->expect = std::move(expect); A->d_so = std::move(d_so); A->d_do = std::move(d_do);
    A->sv.assign((size_t)dof[na], 0);
    A->scnt.assign(na, 0);
    // host part (no device calls): starts now on its own threads; joined right before the L2 launch
    A->bg = std::thread([A, na]() {
      std::atomic<size_t> next{0};
      auto worker = [&]() {
        std::vector<HostMz> v;
        for (size_t i = next.fetch_add(1); i < na; i = next.fetch_add(1)) {
          const size_t cntr = (size_t)(A->dof[i + 1] - A->dof[i]);
          v.resize(cntr);
          for (size_t j = 0; j < cntr; ++j) { const Rec& x = A->hr[(size_t)A->dof[i] + j]; v[j] = HostMz{x.hash, pw_wpos(x.pw), pw_strand(x.pw)}; }
          std::sort(v.begin(), v.end(), host_less_by_hash);
          auto ue = std::unique(v.begin(), v.end(), host_eq_by_hash);
          const size_t sN = (size_t)(ue - v.begin());
          if ((int64_t)sN != A->expect[i]) A->mismatch = 1;
          for (size_t j = 0; j < sN; ++j) A->sv[(size_t)A->dof[i] + j] = v[j].strand == 1 ? 1 : 0;
          A->scnt[i] = (int32_t)sN;
        }
      };
      const unsigned nthr = std::max(1u, std::min(32u, std::min<unsigned>(mm::cpu_budget(), (unsigned)((na + 15) / 16))));
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(worker);
      worker();
      for (auto& t : pool) t.join();
    });
    return [M, A, na, st]() {
      if (A->bg.joinable()) A->bg.join();
      MM_REQUIRE(A->mismatch == 0, MM_ERR_DEVICE, "sketch size disagrees between device and host tie-break");
      DBuf<uint8_t> d_sv(A->sv.size()); d_sv.upload(A->sv.data(), A->sv.size(), st);
      DBuf<int32_t> d_cnt(na); d_cnt.upload(A->scnt.data(), na, st);
      scatter_strand_kernel<<<dim3((unsigned)na), dim3(256), 0, st>>>(d_sv.p, A->d_so.p, A->d_do.p, d_cnt.p, M->sk_strand.p);
      MM_KERNEL_CHECK();
      MM_HIP(mm::stream_sync(st));                          // host vectors above are the H2D sources
    };
  };
  hl("post-K2 amb lists");
  if (!eager_reads.empty()) amb_finish = start_tiebreak(eager_reads);
  hl("post-K2 tiebreak start");
  // ---- K7 host thresholds per distinct sketch size
  {
    if (!ctx->lut_cache || ctx->lut_k != P.k || ctx->lut_pi != P.perc_identity) {
      ctx->lut_cache = stats::LutCache::for_params(P.k, P.perc_identity);
      ctx->lut_k = P.k; ctx->lut_pi = P.perc_identity;
    }
    stats::LutCache& lut = *static_cast<stats::LutCache*>(ctx->lut_cache.get());
    std::vector<int32_t> mh((size_t)n, 0), am((size_t)n, 0);
    int smax = 0, s_hi = 0;
    for (int64_t r = 0; r < n; ++r) s_hi = std::max(s_hi, (int)M->h_sk_n[(size_t)r]);
    std::vector<int32_t> slot((size_t)s_hi + 1, -1);             // sketch size -> place in `sizes`
    std::vector<int> sizes;
    for (int64_t r = 0; r < n; ++r) { const int s = M->h_sk_n[(size_t)r]; if (s > 0 && slot[(size_t)s] < 0) { slot[(size_t)s] = (int32_t)sizes.size(); sizes.push_back(s); } }
    const std::vector<stats::SketchLut> luts = lut.get_many(sizes);
    for (int64_t r = 0; r < n; ++r) {
      int s = M->h_sk_n[(size_t)r];
      if (s <= 0) continue;
      const stats::SketchLut L = luts[(size_t)slot[(size_t)s]];
      mh[(size_t)r] = L.min_hits; am[(size_t)r] = L.accept_min;
      if (s < L2_SKETCH_LIMIT) smax = std::max(smax, s);         // (LDS sizing of the K5 classes; larger sketches never enter them)
      M->stats.sum_sketch += s;
    }
    M->smax = smax;
    M->min_hits.alloc((size_t)std::max<int64_t>(n, 1)); M->min_hits.upload(mh.data(), (size_t)n, st);
    M->accept_min.alloc((size_t)std::max<int64_t>(n, 1)); M->accept_min.upload(am.data(), (size_t)n, st);
    M->h_min_hits = mh;
    MM_HIP(mm::stream_sync(st));
    hl("K7 thresholds + uploads");
  }
  M->d_read_len.alloc((size_t)std::max<int64_t>(n, 1));
  M->d_read_len.upload(reads->len.data(), (size_t)n, st);
  IndexView IV = make_view(I);
  // ---- K3
  DBuf<uint32_t> probe_cnt((size_t)total_mz + 1);
  DBuf<uint64_t> probe_start((size_t)total_mz + 1);
  DBuf<uint64_t> hit_off, scan_tmp;
  const char* nf_env = getenv("MM_NO_HIT_FILTER");               // parity tests of the raw hit list
  const bool use_filter = !(nf_env && nf_env[0] == '1');
  // the fused probe + filter kernel takes the reads whose sketch fits its LDS layout (MM_NO_FUSED_FILTER=1: cross-check switch)
  const bool use_fused = use_filter && !getenv("MM_NO_FUSED_FILTER");
  DBuf<uint8_t> need_old; DBuf<uint32_t> raw_per_read;
  int64_t n_fused = 0, n_wide = 0;
  if (use_filter && n > 0) { raw_per_read.alloc((size_t)n); raw_per_read.zero(st); }
  else if (total_mz > 0) probe_cnt.zero(st);                     // (unfiltered path: the offsets come from a scan over every slot)
  if (use_filter && n > 0) {
    // which kernel filters a read: 0 the fused kernel, 1 the two-pass kernels with 8 192 slots, 2 with 32 768 slots (sketches beyond
    // MM_HF_WIDE_FROM hashes, default 13000 = reads from ~58 kb on; 0 switches the wide table off).  Measured on the bench reference:
    // 45-58 kb reads 88.8 ms narrow / 91.2 ms wide per batch of 8 000 (the wide kernel reads 8-byte entries, three requests per list
    // instead of one, on one workgroup per CU), 60-73 kb reads 873 / 125 ms per batch of 6 000, 75-140 kb 1 388 / 376 ms per 4 000.
    const int wide_env = getenv("MM_HF_WIDE_FROM") ? atoi(getenv("MM_HF_WIDE_FROM")) : 13000;
    const int wide_from = wide_env > 0 ? wide_env : INT_MAX;
    std::vector<uint8_t> h_need((size_t)n, 0);
    for (int64_t r = 0; r < n; ++r) {
      const int sr = M->h_sk_n[(size_t)r];
      const uint8_t c = sr > wide_from ? 2 : ((use_fused && sr <= SF_SMAX) ? 0 : 1);
      h_need[(size_t)r] = c; n_fused += c == 0; n_wide += c == 2;
    }
    need_old.alloc((size_t)n + 4); need_old.upload(h_need.data(), (size_t)n, st);   // (+4: the streaming seed filter reads the class bytes as whole words)
    MM_HIP(mm::stream_sync(st));                            // h_need is the source of the async upload
  }
  hl("K3 prep (need_old etc.)");
  const size_t t_pg = T.begin(&M->stats.ms_probe_gather);
  M->read_hit_off.alloc((size_t)n + 1);
  uint64_t raw_hits = 0;
  DBuf<unsigned long long> raw_sum(1);
  DBuf<uint32_t> surv;
  DBuf<uint64_t> stage, stage_off;
  if (use_filter && n > 0) {
    surv.alloc((size_t)n + 1); surv.zero(st);
    std::vector<uint64_t> h_stage_off((size_t)n + 1, 0);
    const char* cap_env = getenv("MM_HF_STAGE_CAP");              // tests: a tiny capacity forces the re-filtering write path
    for (int64_t r = 0; r < n; ++r)
      h_stage_off[(size_t)r + 1] = h_stage_off[(size_t)r] + (M->h_sk_n[(size_t)r] > 0 ? (cap_env ? (uint64_t)atoi(cap_env) : 1024 + 2 * (uint64_t)M->h_sk_n[(size_t)r]) : 0);
    stage_off.alloc((size_t)n + 1); stage_off.upload(h_stage_off.data(), h_stage_off.size(), st);
    stage.alloc((size_t)std::max<uint64_t>(h_stage_off[(size_t)n], 1));
    MM_HIP(mm::stream_sync(st));                            // h_stage_off is the source of the async upload
    hl("K3 stage_off loop + upload");
    if (use_fused && n_fused > 0) {
      const size_t lds1 = sizeof(SeedFilterLds), lds = sizeof(SeedFilterStreamLds);
      MM_HIP(hipFuncSetAttribute((const void*)seed_filter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
      MM_HIP(hipFuncSetAttribute((const void*)seed_filter_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      MM_HIP(hipFuncSetAttribute((const void*)seed_filter_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      // default: the streaming form (one resident workgroup per CU, look-ups of the next read under the LDS phases of this one);
      // MM_SF_ONESHOT=1 / MM_SF_DBG: one workgroup per read, the form the phase timings of docs/history.md were taken on
      const bool oneshot = getenv("MM_SF_ONESHOT") || getenv("MM_SF_DBG");
      DBuf<uint32_t> sf_ticket(1);
      if (!oneshot) sf_ticket.zero(st);
      DBuf<unsigned long long> sf_prof;                            // MM_SF_PROF=1: cycles per phase of the streaming kernel, printed per batch
      if (getenv("MM_SF_PROF")) { sf_prof.alloc(16); sf_prof.zero(st); }
      int sf_grid = (int)std::min<int64_t>(n, std::max(ctx->cus, 1));
      if (const char* e = getenv("MM_SF_GRID")) sf_grid = std::max(1, std::min(sf_grid, atoi(e)));   // (measurement aid: fewer resident workgroups = fewer CUs at work)
      const size_t t_sf = T.begin(&M->stats.ms_hit_filter);
      if (oneshot)
        seed_filter_kernel<<<dim3((unsigned)n), dim3(SF_THREADS), lds1, st>>>(IV, M->sk_hash.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->min_hits.p, surv.p,
                                                                           stage.p, stage_off.p, need_old.p, raw_per_read.p, getenv("MM_SF_DBG") ? atoi(getenv("MM_SF_DBG")) : 0);
      else if (sf_prof.p)
        seed_filter_stream_kernel<true><<<dim3((unsigned)sf_grid), dim3(SF_THREADS), lds, st>>>(IV, M->sk_hash.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->min_hits.p, surv.p, stage.p, stage_off.p,
                                                                           need_old.p, reinterpret_cast<const uint32_t*>(need_old.p), raw_per_read.p, (int)n, sf_ticket.p, sf_prof.p);
      else
        seed_filter_stream_kernel<false><<<dim3((unsigned)sf_grid), dim3(SF_THREADS), lds, st>>>(IV, M->sk_hash.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->min_hits.p, surv.p, stage.p, stage_off.p,
                                                                           need_old.p, reinterpret_cast<const uint32_t*>(need_old.p), raw_per_read.p, (int)n, sf_ticket.p, nullptr);
      MM_KERNEL_CHECK();
      T.end(t_sf);
      if (sf_prof.p && !oneshot) {
        auto h = sf_prof.to_host(st);
        const double tot = (double)std::accumulate(h.begin(), h.end(), 0ull);
        fprintf(stderr, "MM_SF_PROF share of cycles: zero+top %.3f | next head + resolve %.3f | scan %.3f | lists+count %.3f | window sums+alive+issue %.3f | phase 2 %.3f | survivors+end %.3f | total %.3g cycles over %d workgroups\n",
                h[0] / tot, (h[1] + h[7]) / tot, h[2] / tot, (h[3] + h[8]) / tot, (h[4] + h[9] + h[10] + h[11]) / tot, (h[5] + h[12]) / tot, h[6] / tot, tot, sf_grid);
        fprintf(stderr, "MM_SF_PROF in detail: top %.3f | first answers + second probes issued %.3f, their answers %.3f | scan %.3f | pieces loaded, counted, parked %.3f, next hashes asked for + barrier %.3f | "
                        "window sums %.3f, alive %.3f, barrier %.3f, hashes there + look-ups issued %.3f | bit tests + slots %.3f, barrier %.3f | survivors+end %.3f\n",
                h[0] / tot, h[7] / tot, h[1] / tot, h[2] / tot, h[8] / tot, h[3] / tot, h[9] / tot, h[10] / tot, h[4] / tot, h[11] / tot, h[12] / tot, h[5] / tot, h[6] / tot);
      }
    }
  }
  const uint8_t* const only = (use_filter && n > 0) ? need_old.p : nullptr;
  if (n > 0 && total_mz > 0) {                                     // (reads the fused kernel flagged are only known on the device: the two-pass kernels always run and skip the rest)
    probe_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>(IV, M->sk_hash.p, M->mz.off.p, M->sk_n.p, probe_cnt.p, probe_start.p, only);
    MM_KERNEL_CHECK();
  }
  if (!use_filter || n == 0) {
    hit_off.alloc((size_t)total_mz + 2);
    exclusive_scan_u32_u64(probe_cnt.p, total_mz, hit_off.p, scan_tmp, st);
    MM_HIP(hipMemcpyAsync(&raw_hits, hit_off.p + total_mz, sizeof raw_hits, hipMemcpyDeviceToHost, st));
  }
  if (use_filter && n > 0) {
    const bool time_old = !(use_fused && n_fused > 0);            // ms_hit_filter: the kernel that handles the bulk of the reads
    const size_t t_hf = time_old ? T.begin(&M->stats.ms_hit_filter) : 0;
    using HfN = HitFilterCfg<HF_SLOT_BITS_NARROW>; using HfW = HitFilterCfg<HF_SLOT_BITS_WIDE>;
    const int hf_dbg = getenv("MM_HF_DBG") ? atoi(getenv("MM_HF_DBG")) : 0;
    hit_filter_kernel<false, HF_SLOT_BITS_NARROW><<<dim3((unsigned)n), dim3(HfN::THREADS), HfN::LDS, st>>>(IV, M->mz.off.p, M->sk_n.p, probe_cnt.p, probe_start.p, M->d_read_len.p,
                                                                   M->min_hits.p, surv.p, nullptr, nullptr, stage.p, stage_off.p, hf_dbg, only, raw_per_read.p);
    MM_KERNEL_CHECK();
    if (n_wide > 0) {
      MM_HIP(hipFuncSetAttribute((const void*)hit_filter_kernel<false, HF_SLOT_BITS_WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HfW::LDS));
      hit_filter_kernel<false, HF_SLOT_BITS_WIDE><<<dim3((unsigned)n), dim3(HfW::THREADS), HfW::LDS, st>>>(IV, M->mz.off.p, M->sk_n.p, probe_cnt.p, probe_start.p, M->d_read_len.p,
                                                                   M->min_hits.p, surv.p, nullptr, nullptr, stage.p, stage_off.p, hf_dbg, only, raw_per_read.p);
      MM_KERNEL_CHECK();
    }
    if (time_old) T.end(t_hf);
    raw_sum.zero(st);                                            // only the total of the raw seed hits is needed
    sum_u32_kernel<<<dim3(256), dim3(256), 0, st>>>(raw_per_read.p, n, raw_sum.p);
    MM_KERNEL_CHECK();
    MM_HIP(hipMemcpyAsync(&raw_hits, raw_sum.p, sizeof raw_hits, hipMemcpyDeviceToHost, st));
    exclusive_scan_u32_u64(surv.p, n, M->read_hit_off.p, scan_tmp, st);
  } else {
    read_hit_bounds_kernel<<<dim3((unsigned)ceil_div(n + 1, 256)), dim3(256), 0, st>>>(M->mz.off.p, hit_off.p, n, M->read_hit_off.p);
    MM_KERNEL_CHECK();
  }
  hl("K3 launches");
  M->h_read_hit_off = M->read_hit_off.to_host(st);
  hl("K3 wait + hit_off download");
  const int64_t total_hits = (int64_t)M->h_read_hit_off[(size_t)n];
  M->stats.sum_hits = (int64_t)raw_hits;
  M->stats.sum_hits_kept = total_hits;
  M->hits.alloc((size_t)std::max<int64_t>(total_hits, 1));
  if (total_hits > 0) {
    if (use_filter) {
      using HfN = HitFilterCfg<HF_SLOT_BITS_NARROW>; using HfW = HitFilterCfg<HF_SLOT_BITS_WIDE>;
      const int wdbg = getenv("MM_HITS_BITONIC") ? 0 : 100;
      hit_filter_kernel<true, HF_SLOT_BITS_NARROW><<<dim3((unsigned)n), dim3(HfN::THREADS), HfN::LDS, st>>>(IV, M->mz.off.p, M->sk_n.p, probe_cnt.p, probe_start.p, M->d_read_len.p,
                                                                    M->min_hits.p, surv.p, M->read_hit_off.p, M->hits.p, stage.p, stage_off.p, wdbg, only, nullptr);
      if (n_wide > 0) {
        MM_KERNEL_CHECK();
        MM_HIP(hipFuncSetAttribute((const void*)hit_filter_kernel<true, HF_SLOT_BITS_WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HfW::LDS));
        hit_filter_kernel<true, HF_SLOT_BITS_WIDE><<<dim3((unsigned)n), dim3(HfW::THREADS), HfW::LDS, st>>>(IV, M->mz.off.p, M->sk_n.p, probe_cnt.p, probe_start.p, M->d_read_len.p,
                                                                    M->min_hits.p, surv.p, M->read_hit_off.p, M->hits.p, stage.p, stage_off.p, wdbg, only, nullptr);
      }
    } else
      gather_hits_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>(IV, M->mz.off.p, M->sk_n.p, probe_cnt.p, probe_start.p, hit_off.p, M->hits.p);
    MM_KERNEL_CHECK();
  }
  T.end(t_pg);
  if (total_hits > 0) {
    const size_t t_sh = T.begin(&M->stats.ms_sort_hits);
    // ---- K4a
    hl("K4 hits alloc + emit launch");
    std::vector<int64_t> hc;                                      // hit counts of the reads the LDS radix sort does not take
    auto hits_of = [&](int64_t r) -> int64_t { return (int64_t)(M->h_read_hit_off[(size_t)r + 1] - M->h_read_hit_off[(size_t)r]); };
    // beyond 4096 hits the device's segmented radix sort is faster (50 kb reads: 7.9 -> 7.0 ms); below, the LDS network (10 kb: 2.2 vs 3.9 ms)
    const char* ss_env = getenv("MM_SEGSORT_FROM");
    const int segsort_from = std::min(ss_env ? atoi(ss_env) : 4096, LDS_SORT_MAX);
    const bool seg_ok = total_hits < (int64_t)0xffffffffll && !getenv("MM_HITS_BITONIC");
    std::vector<int32_t> seg_reads;                               // reads of every class handled by the segmented sort: one call for all
    // up to 4096 hits per read: LDS radix sort, the reads grouped by the elements per thread they need
    int key_bits = 32; while (key_bits < 64 && ((int64_t)1 << (key_bits - 32)) < I->n_contigs) ++key_bits;
    const bool use_radix = !getenv("MM_HITS_BITONIC");
    bool any_left = !use_radix;
    if (use_radix) {
      static const int ipts[] = {1, 2, 3, 4, 6, 8, 12, 16};
      uint8_t cls_of_need[18];                                    // elements per thread needed -> index into ipts; [17]: beyond 4096 hits
      for (int need = 0, i = 0; need <= 16; ++need) { while (ipts[i] < need) ++i; cls_of_need[need] = (uint8_t)i; }
      cls_of_need[17] = 8;
      uint64_t left_seen = 0;
      const ReadBins RB = bin_reads(n, 8, [&](int64_t r) -> int {
        const uint64_t c = (uint64_t)hits_of(r);
        left_seen |= (uint64_t)(c > 4096);
        const unsigned k = cls_of_need[std::min<uint64_t>((c + 255) / 256, 17)];
        return c <= 1 ? 8 : (int)k;                              // zero or one hit: nothing to sort
      });
      any_left = left_seen != 0;
      hl("K4 bin");
      DBuf<int32_t> list(std::max<size_t>(RB.order.size(), 1));
      list.upload(RB.order.data(), RB.order.size(), st);
      hl("K4 list upload");
      size_t at = 0;
      for (auto& run : RB.runs) {
        const int32_t* lp = list.p + at;
        auto launch = [&](auto tag) {
          constexpr int IPT = decltype(tag)::value;
          using SortT = rocprim::block_radix_sort<uint64_t, 256, IPT>;
          const size_t lds = sizeof(typename SortT::storage_type) + 16;
          if (lds > 48 * 1024) MM_HIP(hipFuncSetAttribute((const void*)sort_hits_radix_kernel<IPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          sort_hits_radix_kernel<IPT><<<dim3((unsigned)run.second), dim3(256), lds, st>>>(M->hits.p, M->read_hit_off.p, lp, key_bits, use_filter ? stage.p : nullptr, use_filter ? stage_off.p : nullptr);
        };
        switch (ipts[run.first]) {
          case 1: launch(std::integral_constant<int, 1>{}); break;
          case 2: launch(std::integral_constant<int, 2>{}); break;
          case 3: launch(std::integral_constant<int, 3>{}); break;
          case 4: launch(std::integral_constant<int, 4>{}); break;
          case 6: launch(std::integral_constant<int, 6>{}); break;
          case 8: launch(std::integral_constant<int, 8>{}); break;
          case 12: launch(std::integral_constant<int, 12>{}); break;
          default: launch(std::integral_constant<int, 16>{}); break;
        }
        MM_KERNEL_CHECK();
        at += run.second;
      }
      MM_HIP(mm::stream_sync(st));                          // RB.order is the source of the async upload
    }
    if (any_left) {                                              // the loops below only see the longer lists
      hc.assign((size_t)n, 0);
      for (int64_t r = 0; r < n; ++r) { const int64_t c = hits_of(r); if (!use_radix || c > 4096) hc[(size_t)r] = c; }
    }
    for (auto& cls : make_classes(hc, 256)) {
      if (cls.npow2 > segsort_from && seg_ok) { seg_reads.insert(seg_reads.end(), cls.reads.begin(), cls.reads.end()); continue; }
      DBuf<int32_t> list(cls.reads.size());
      list.upload(cls.reads.data(), cls.reads.size(), st);
      if (cls.npow2 <= LDS_SORT_MAX) {
        size_t lds = (size_t)cls.npow2 * 8;
        if (lds > 64 * 1024) MM_HIP(hipFuncSetAttribute((const void*)sort_hits_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        sort_hits_kernel<true><<<dim3((unsigned)cls.reads.size()), dim3(256), lds, st>>>(M->hits.p, M->read_hit_off.p, list.p, cls.npow2, nullptr);
        MM_KERNEL_CHECK();
        MM_HIP(mm::stream_sync(st));
      } else {
        // (fallback) a few reads at a time through a global scratch buffer
        const size_t per = (size_t)cls.npow2;
        const size_t group = std::max<size_t>(1, std::min<size_t>(cls.reads.size(), ((size_t)1 << 28) / per));
        DBuf<uint64_t> scratch(per * group);
        for (size_t g0 = 0; g0 < cls.reads.size(); g0 += group) {
          size_t g = std::min(group, cls.reads.size() - g0);
          sort_hits_kernel<false><<<dim3((unsigned)g), dim3(256), 0, st>>>(M->hits.p, M->read_hit_off.p, list.p + g0, cls.npow2, scratch.p);
          MM_KERNEL_CHECK();
        }
        MM_HIP(mm::stream_sync(st));
      }
    }
    if (!seg_reads.empty()) {
      // large segments (reads beyond ~30 kb): the device's segmented radix sort over exactly these reads' ranges (the bitonic
      // network through global memory took 0.5 s for a few hundred such reads).  When they are a minority of the batch their
      // ranges are gathered into a compact buffer first, so that the scratch is twice their hits instead of a copy of all hits.
      std::sort(seg_reads.begin(), seg_reads.end());
      const size_t ns = seg_reads.size();
      std::vector<uint64_t> hb(ns), he(ns), cb(ns), ce(ns);
      uint64_t run = 0;
      for (size_t i = 0; i < ns; ++i) {
        const int32_t r = seg_reads[i];
        hb[i] = M->h_read_hit_off[(size_t)r]; he[i] = M->h_read_hit_off[(size_t)r + 1];
        cb[i] = run; run += he[i] - hb[i]; ce[i] = run;
      }
      const bool compact = 2 * run <= (uint64_t)total_hits;
      DBuf<uint64_t> d_hb(ns), d_he(ns), d_cb(ns), d_ce(ns);
      d_hb.upload(hb.data(), ns, st); d_he.upload(he.data(), ns, st);
      auto seg_sort = [&](uint64_t* in, uint64_t* out, uint64_t count, uint64_t* begins, uint64_t* ends) {
        size_t tmp_bytes = 0;
        MM_HIP(rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, in, out, (unsigned int)count, (unsigned int)ns, begins, ends, 0, key_bits, st));
        DBuf<uint8_t> tmp(std::max<size_t>(tmp_bytes, 16));
        MM_HIP(rocprim::segmented_radix_sort_keys((void*)tmp.p, tmp_bytes, in, out, (unsigned int)count, (unsigned int)ns, begins, ends, 0, key_bits, st));
      };
      if (compact) {
        d_cb.upload(cb.data(), ns, st); d_ce.upload(ce.data(), ns, st);
        DBuf<uint64_t> packed((size_t)run), sorted((size_t)run);
        move_ranges_kernel<<<dim3((unsigned)ns), dim3(256), 0, st>>>(M->hits.p, d_hb.p, d_he.p, packed.p, d_cb.p);
        MM_KERNEL_CHECK();
        seg_sort(packed.p, sorted.p, run, d_cb.p, d_ce.p);
        move_ranges_kernel<<<dim3((unsigned)ns), dim3(256), 0, st>>>(sorted.p, d_cb.p, d_ce.p, M->hits.p, d_hb.p);
        MM_KERNEL_CHECK();
        MM_HIP(mm::stream_sync(st));
      } else {
        DBuf<uint64_t> sorted((size_t)total_hits);
        seg_sort(M->hits.p, sorted.p, (uint64_t)total_hits, d_hb.p, d_he.p);
        move_ranges_kernel<<<dim3((unsigned)ns), dim3(256), 0, st>>>(sorted.p, d_hb.p, d_he.p, M->hits.p, d_hb.p);
        MM_KERNEL_CHECK();
        MM_HIP(mm::stream_sync(st));
      }
    }
    T.end(t_sh);
  }
  hl("K4 launches + sync");
  // ---- K4b
  const size_t t_l1 = T.begin(&M->stats.ms_l1_scan);
  const bool l1_serial = getenv("MM_L1_SERIAL") != nullptr;       // cross-check switch: the one-thread-per-read loop
  DBuf<uint32_t> cand_n((size_t)n + 1); cand_n.zero(st);
  M->cand_off.alloc((size_t)n + 2);
  const unsigned rblk = (unsigned)ceil_div(std::max<int64_t>(n, 1), 128);
  if (n > 0) {
    if (l1_serial) l1_scan_kernel<false><<<dim3(rblk), dim3(128), 0, st>>>(M->hits.p, M->read_hit_off.p, M->d_read_len.p, M->min_hits.p, n, cand_n.p, nullptr, nullptr, nullptr);
    else l1_wave_kernel<false><<<dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st>>>(M->hits.p, M->read_hit_off.p, M->d_read_len.p, M->min_hits.p, n, cand_n.p, nullptr, nullptr, nullptr, nullptr);
    MM_KERNEL_CHECK();
  }
  exclusive_scan_u32_u64(cand_n.p, n, M->cand_off.p, scan_tmp, st);
  M->h_cand_off = M->cand_off.to_host(st, (size_t)n + 1);
  hl("L1 count + cand_off download");
  const int64_t ncand = (int64_t)M->h_cand_off[(size_t)n];
  M->n_cand = ncand;
  M->stats.n_candidates = ncand;
  M->cand.alloc((size_t)std::max<int64_t>(3 * ncand, 1));
  M->cand_read.alloc((size_t)std::max<int64_t>(ncand, 1));
  M->l2.alloc((size_t)std::max<int64_t>(ncand, 1));
  DBuf<int32_t> cand_hint((size_t)std::max<int64_t>(ncand, 1));   // seed hits inside each candidate (l1_wave_kernel): the zone kernel's prediction of its band
  DBuf<int64_t> cand_rng;                                        // [first, behind-last) index entry of each candidate's stream (l2_ranges_kernel)
  const bool no_hint = l1_serial || getenv("MM_L2_NO_FUSE");
  if (no_hint) cand_hint.zero(st);                               // (0: no prediction, the masks of the band come from a second pass over the stream)
  M->rec_off.alloc((size_t)n + 1);
  if (ncand > 0) {
    if (l1_serial) l1_scan_kernel<true><<<dim3(rblk), dim3(128), 0, st>>>(M->hits.p, M->read_hit_off.p, M->d_read_len.p, M->min_hits.p, n, nullptr, M->cand_off.p, M->cand.p, M->cand_read.p);
    else l1_wave_kernel<true><<<dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st>>>(M->hits.p, M->read_hit_off.p, M->d_read_len.p, M->min_hits.p, n, nullptr, M->cand_off.p, M->cand.p, M->cand_read.p, no_hint ? nullptr : cand_hint.p);
    MM_KERNEL_CHECK();
    T.end(t_l1);
    // ---- K5/K6
    if (!getenv("MM_L2_NO_RANGES")) {                            // (MM_L2_NO_RANGES=1: the zone kernel's waves search their ranges themselves, as until round 6)
      cand_rng.alloc(2 * (size_t)ncand);
      l2_ranges_kernel<<<dim3((unsigned)ceil_div(ncand, 256)), dim3(256), 0, st>>>(IV, M->cand.p, M->cand_read.p, M->d_read_len.p, ncand, cand_rng.p);
      MM_KERNEL_CHECK();
    }
    MM_REQUIRE(ncand < (1LL << 31), MM_ERR_LIMIT, "more than 2^31 L1 candidates in one batch");
    const int smax = M->smax;
    const char* full_env = getenv("MM_L2_FULL");                 // cross-check switch: evaluate every window
    const bool skip = !(full_env && full_env[0] == '1');
    // sketches from this size on take the dense path (MM_L2_DENSE_FROM: experiments; MM_L2_NO_DENSE=1: the LDS classes / literal automaton)
    const bool use_dense = !getenv("MM_L2_NO_DENSE");
    // (from ~58 kb reads on the streamed range of a candidate outgrows the 32 768-entry masks of the LDS classes' exact skip-ahead, which
    //  then evaluate every window with a rebuild per zone exit: 6 000 reads of 60-73 kb: 171 ms there, 83 ms here)
    const int dense_from = getenv("MM_L2_DENSE_FROM") ? atoi(getenv("MM_L2_DENSE_FROM")) : 13000;
    const bool no_small_groups = getenv("MM_L2_NO_SMALL_GROUPS") != nullptr;   // cross-check / timing switch
    // the workgroups of the 10 kb class are put together on the device while the L1 kernel above still runs (l2_group_kernel);
    // MM_L2_HOST_GROUPS=1: the host loop makes them as well (cross-check)
    const bool dev_groups = skip && !getenv("MM_L2_HOST_GROUPS");
    DBuf<int32_t> d_gA0, d_gAn, d_gS0, d_gSn;
    DBuf<unsigned int> grp_ctr(4);
    if (dev_groups) {
      d_gA0.alloc((size_t)ncand); d_gAn.alloc((size_t)ncand); d_gS0.alloc((size_t)ncand); d_gSn.alloc((size_t)ncand);
      grp_ctr.zero(st);
      l2_group_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st>>>(M->cand_off.p, M->sk_n.p, M->d_read_len.p, n, P.w + P.k + 1, use_dense ? dense_from : INT_MAX,
                                                                            no_small_groups ? 1 : 0, d_gA0.p, d_gAn.p, d_gS0.p, d_gSn.p, grp_ctr.p);
      MM_KERNEL_CHECK();
    }
    const size_t lds_wide = l2_lds_bytes<uint16_t>(smax, skip, 1, 8);
    // Sketches of >= 32768 hashes (L2_SKETCH_LIMIT): the rebuild's 1024-bucket histogram would be as coarse as the 64-rank pivot
    // zone, and the window state of the full slide no longer fits LDS either -> l2_giant_kernel, state in global memory.
    std::vector<int32_t> listG; int smG = 0;
    for (int64_t r = 0; r < n && M->stats.n_reads_giant > 0; ++r) {
      const int sr = M->h_sk_n[(size_t)r];
      if (sr < L2_SKETCH_LIMIT) continue;
      smG = std::max(smG, sr);
      for (uint64_t c0 = M->h_cand_off[(size_t)r]; c0 < M->h_cand_off[(size_t)r + 1]; ++c0) listG.push_back((int32_t)c0);
    }
    MM_REQUIRE(lds_wide <= 160 * 1024, MM_ERR_LIMIT, "L2 window state does not fit LDS");
    auto set_lds = [&](const void* fn, size_t bytes) { if (bytes > 64 * 1024) MM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)); };
    DBuf<unsigned long long> counters(16); counters.zero(st);
    if (getenv("MM_L2_STOP") || getenv("MM_L2_PHASES") || getenv("MM_FORCE_AMB_REDO") || getenv("MM_L2Z_DBG")) {
      const char* ds = getenv("MM_L2_STOP"); unsigned long long v = (unsigned long long)((ds ? atoi(ds) & 0xff : 0) | (getenv("MM_L2_PHASES") ? 0x100 : 0) | (getenv("MM_FORCE_AMB_REDO") ? 0x200 : 0) | (getenv("MM_L2Z_DBG") ? (atoi(getenv("MM_L2Z_DBG")) == 2 ? 0xc00 : 0x400) : 0)); MM_HIP(hipMemcpyAsync(counters.p + 11, &v, sizeof v, hipMemcpyHostToDevice, st)); MM_HIP(mm::stream_sync(st)); }   // timing aid: leave the kernel after phase n (results are then meaningless)
    DBuf<int32_t> ovf((size_t)ncand);
    DBuf<unsigned int> ovf_n(1); ovf_n.zero(st);
    DBuf<uint8_t> amb_used;
    if (!lazy_reads.empty()) { amb_used.alloc((size_t)n); amb_used.zero(st); }
    uint8_t* const amb_used_p = lazy_reads.empty() ? nullptr : amb_used.p;
    if (amb_finish) { amb_finish(); amb_finish = nullptr; }        // strands of ambiguous sketches: needed by the vote only
    const size_t t_l2 = T.begin(&M->stats.ms_l2);
    // Long reads: the window state in global memory, one wave per candidate (mm_l2_dense.hpp).  `list`: candidates, those of a read
    // consecutive; smax_l: largest sketch among them.
    const int force_amb = getenv("MM_FORCE_AMB_REDO") ? 1 : 0;
    auto run_dense = [&](const std::vector<int32_t>& list, int smax_l, uint8_t* amb_ptr) {
      if (list.empty()) return;
      const size_t nl = list.size();
      DBuf<int32_t> d_list(nl); d_list.upload(list.data(), nl, st);
      DBuf<L2Range> d_rng(nl);
      l2_range_kernel<<<dim3((unsigned)ceil_div((int64_t)nl, 4)), dim3(256), 0, st>>>(IV, M->cand.p, M->cand_read.p, M->d_read_len.p, d_list.p, (int)nl, d_rng.p);
      MM_KERNEL_CHECK();
      std::vector<L2Range> rng = d_rng.to_host(st);               // (also keeps `list` alive until its upload is done)
      std::vector<uint64_t> coff(nl + 1, 0);
      for (size_t i = 0; i < nl; ++i) coff[i + 1] = coff[i] + (uint64_t)std::max(rng[i].m, 0);
      std::vector<int32_t> cr = M->cand_read.to_host(st, (size_t)ncand);
      std::vector<int32_t> gfirst;                               // one classification workgroup per read
      for (size_t i = 0; i < nl; ++i) if (i == 0 || cr[(size_t)list[i]] != cr[(size_t)list[i - 1]]) gfirst.push_back((int32_t)i);
      gfirst.push_back((int32_t)nl);
      DBuf<uint64_t> d_coff(nl + 1); d_coff.upload(coff.data(), nl + 1, st);
      DBuf<int32_t> d_gf(gfirst.size()); d_gf.upload(gfirst.data(), gfirst.size(), st);
      DBuf<uint32_t> codes((size_t)std::max<uint64_t>(coff[nl], 1));
      const int q_in_lds = smax_l <= LD_Q_LDS_MAX ? 1 : 0;
      const size_t lds = ((size_t)((LD_TSIZE + 3) & ~3) + (q_in_lds ? (size_t)smax_l + 4 : 0)) * 4;
      set_lds((const void*)l2_codes_kernel, lds);
      l2_codes_kernel<<<dim3((unsigned)(gfirst.size() - 1)), dim3(256), lds, st>>>(IV, M->cand_read.p, M->sk_hash.p, M->mz.off.p, M->sk_n.p, d_list.p, d_gf.p,
                                                                                  d_rng.p, d_coff.p, codes.p, q_in_lds);
      MM_KERNEL_CHECK();
      const unsigned slots = (unsigned)std::min<size_t>(nl, (size_t)ctx->cus * 12);
      DBuf<uint32_t> scratch((size_t)slots * l2_dense_slot_words(smax_l));
      DBuf<unsigned int> next(1); next.zero(st);
      l2_dense_kernel<<<dim3(slots), dim3(64), 0, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_strand.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p,
                                                      P.k, P.w, smax_l, M->l2.p, d_list.p, (int)nl, d_rng.p, d_coff.p, codes.p, scratch.p, next.p, amb_ptr, force_amb,
                                                      getenv("MM_L2_DENSE_NO_STOP") ? 0 : 1);
      MM_KERNEL_CHECK();
      MM_HIP(mm::stream_sync(st));                          // host vectors above are upload sources; the buffers die with this scope
    };
    DBuf<int32_t> d_listG(listG.size());
    DBuf<uint32_t> giant_scratch;
    if (!listG.empty() && use_dense) run_dense(listG, smG, amb_used_p);
    else if (!listG.empty()) {
      d_listG.upload(listG.data(), listG.size(), st);
      const unsigned slots = (unsigned)std::min<size_t>(listG.size(), (size_t)ctx->cus * 8);
      giant_scratch.alloc((size_t)slots * l2_giant_slot_words(smG));
      l2_giant_kernel<<<dim3(slots), dim3(64), 0, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p,
                                                      M->accept_min.p, P.k, P.w, smG, M->l2.p, d_listG.p, (int)listG.size(), giant_scratch.p);
      MM_KERNEL_CHECK();
    }
    if (!skip) {
      std::vector<int32_t> listF;
      for (int64_t r = 0; r < n; ++r) if (M->h_sk_n[(size_t)r] < L2_SKETCH_LIMIT)
        for (uint64_t c0 = M->h_cand_off[(size_t)r]; c0 < M->h_cand_off[(size_t)r + 1]; ++c0) listF.push_back((int32_t)c0);
      DBuf<int32_t> d_listF(listF.size());
      if (!listF.empty()) {
        d_listF.upload(listF.data(), listF.size(), st);
        set_lds((const void*)l2_kernel<false, uint16_t, 1, 8>, lds_wide);
        l2_kernel<false, uint16_t, 1, 8><<<dim3((unsigned)listF.size()), dim3(64), lds_wide, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
            M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smax, M->l2.p, counters.p, nullptr, nullptr, d_listF.p, nullptr, nullptr, amb_used_p, nullptr, nullptr, nullptr, 0);
        MM_KERNEL_CHECK();
      }
      MM_HIP(mm::stream_sync(st));                          // listF is the source of the async upload
    } else {
      // Reads are grouped by sketch size so that one long read does not size the LDS state (and the occupancy) of all:
      //   A  s <= 3072   (reads up to ~14 kb at w=8)  compact: 4 candidates of a read per workgroup share the sketch,
      //                                                8-bit gap counters, masks for 8 192 streamed entries
      //   B  s <= 7168   (~32 kb)                      the same with masks for 32 768 entries, kept in global memory
      //   D  s <= 16384  (~74 kb)                      as B, launched separately so that B keeps its smaller sketch area
      //   C  larger                                    one wave per workgroup, 16-bit counters, 32 768 entries
      // Reads shorter than w+k are handed back by these kernels and go through the literal full slide.
      // per-entry code words of pass A: one slot range per wave of a launch (the launches of a batch run one after the other
      // on the stream, so they share the buffer); classes whose ranks do not fit 16 bits (C) search the sketch instead
      // scratch slots of the skip kernels (mm_l2.hpp): a slot per RESIDENT wave (the hardware keeps at most 32 per CU), taken and given
      // back by the waves through one flag word each; MM_L2_NO_SLOTS=1: one slot per wave of the launch, no flags (cross-check switch)
      // (MM_L2_SLOTS=n: fewer slots than resident waves — they wait for each other's; tests of the hand-over)
      // twice as many slots as waves can be resident (the 10 kb class keeps 24 per CU, the long-read classes 8-12): a wave finds a free one at
      // its first or second try
      const bool no_slots = getenv("MM_L2_NO_SLOTS") != nullptr;
      const size_t env_slots = getenv("MM_L2_SLOTS") ? ((size_t)std::max(atoi(getenv("MM_L2_SLOTS")), 1) + 7) / 8 * 8 : 0;
      auto max_slots = [&](int nwq) -> size_t { return no_slots ? (size_t)1 << 40 : env_slots ? env_slots : (size_t)ctx->cus * (nwq == 2 ? 64 : 32); };
      DBuf<unsigned int> slot_flags(std::max((size_t)ctx->cus * 64, env_slots)); slot_flags.zero(st);   // (MM_L2_SLOTS may ask for more slots than cus * 64)
      unsigned int* const slot_flags_p = no_slots ? nullptr : slot_flags.p;
      auto slots_of = [&](size_t n_waves, int nwq = 8) -> size_t { return std::min((std::max<size_t>(n_waves, 1) + 7) / 8 * 8, max_slots(nwq)); };   // (a multiple of 8: one share per XCD)
      auto masks_for = [&](size_t n_waves, int nwq = 8) -> uint8_t* { return (uint8_t*)ctx->l2_masks_at_least(slots_of(n_waves, nwq) * l2_skip_bytes(nwq)); };
      auto codes_for = [&](size_t n_waves, int nwq) -> void* {
        if (getenv("MM_L2_NO_CODES")) return nullptr;              // cross-check switch
        return ctx->l2_codes_at_least(slots_of(n_waves, nwq) * (size_t)(64 * 64 * nwq) * (nwq == 2 ? sizeof(uint16_t) : sizeof(uint32_t)));
      };
      // the zone kernels (mm_l2z.hpp, the default; MM_L2_V1=1: l2_kernel for every class): matched list + masks per slot
      const bool v2 = !getenv("MM_L2_V1");
      const bool v2_long = v2 && !getenv("MM_L2_V1_LONG");           // (MM_L2_V1_LONG=1: the long-read classes, sketches of 3 073 .. 13 000 hashes, through l2_kernel)
      const int32_t* const cand_hint_p = cand_hint.p;
      auto lists_for = [&](size_t n_waves, int nwq) -> void* { return ctx->l2_codes_at_least(slots_of(n_waves, nwq) * l2z_list_bytes(nwq)); };
      auto zmasks_for = [&](size_t n_waves, int nwq) -> uint8_t* { return (uint8_t*)ctx->l2_masks_at_least(slots_of(n_waves, nwq) * l2z_mask_bytes(nwq)); };
      DBuf<int32_t> big((size_t)(v2 ? ncand : 1));                // candidates with more streamed entries than the zone kernel's masks hold: l2_kernel's widest class
      DBuf<unsigned int> big_n(1); big_n.zero(st);
      std::vector<int32_t> gA0, gAn, gB0, gBn, gD0, gDn, listC, gS0, gSn;   // gS: groups of one or two candidates of the 10 kb class (two-wave workgroups)
      int smA = 0, smB = 0, smC = 0, smD = 0;
      std::vector<int32_t> listL; int smL = 0;                    // long reads below the giant class that take the dense path
      hl("K5 prep before grouping");
      std::vector<unsigned int> gctr(4, 0);
      if (dev_groups) gctr = grp_ctr.to_host(st);                 // (waits for the L1 kernel and the grouping kernel)
      smA = (int)gctr[3];
      for (int64_t r = 0; r < n && (!dev_groups || gctr[2] > 0); ++r) {   // the host's classes: everything the device did not group
        const uint64_t c_lo = M->h_cand_off[(size_t)r], c_hi = M->h_cand_off[(size_t)r + 1];
        if (c_lo == c_hi) continue;
        const int sr = M->h_sk_n[(size_t)r];
        if (dev_groups && sr <= 3072 && !(use_dense && sr >= dense_from && sr < L2_SKETCH_LIMIT && M->read_len[(size_t)r] >= P.w + P.k + 1)) continue;
        if (use_dense && sr >= dense_from && sr < L2_SKETCH_LIMIT && M->read_len[(size_t)r] >= P.w + P.k + 1) {
          smL = std::max(smL, sr);
          for (uint64_t c0 = c_lo; c0 < c_hi; ++c0) listL.push_back((int32_t)c0);
          continue;
        }
        if (sr <= 7168) {
          auto& g0 = sr <= 3072 ? gA0 : gB0; auto& gn = sr <= 3072 ? gAn : gBn;
          (sr <= 3072 ? smA : smB) = std::max(sr <= 3072 ? smA : smB, sr);
          for (uint64_t c0 = c_lo; c0 < c_hi; c0 += 4) {
            const int32_t cnt = (int32_t)std::min<uint64_t>(4, c_hi - c0);
            // a workgroup holds the read's sketch once: four-wave workgroups with one or two candidates leave half of their waves'
            // LDS share idle (species of 1-12 strains: every candidate count occurs), those go to two-wave workgroups
            if (sr <= 3072 && cnt <= 2 && !no_small_groups) { gS0.push_back((int32_t)c0); gSn.push_back(cnt); }
            else { g0.push_back((int32_t)c0); gn.push_back(cnt); }
          }
        } else if (sr <= 16384) {
          smD = std::max(smD, sr);
          for (uint64_t c0 = c_lo; c0 < c_hi; c0 += 4) { gD0.push_back((int32_t)c0); gDn.push_back((int32_t)std::min<uint64_t>(4, c_hi - c0)); }
        } else if (sr < L2_SKETCH_LIMIT) {
          smC = std::max(smC, sr);
          for (uint64_t c0 = c_lo; c0 < c_hi; ++c0) listC.push_back((int32_t)c0);
        }                                                        // (larger: listG above)
      }
      hl("K5 grouping");
      run_dense(listL, smL, amb_used_p);
      DBuf<int32_t> d_gB0(gB0.size()), d_gBn(gBn.size()), d_listC(listC.size());
      size_t nA = gctr[0], nS = gctr[1];
      if (!dev_groups) {
        nA = gA0.size(); nS = gS0.size();
        d_gA0.alloc(std::max<size_t>(nA, 1)); d_gAn.alloc(std::max<size_t>(nA, 1)); d_gS0.alloc(std::max<size_t>(nS, 1)); d_gSn.alloc(std::max<size_t>(nS, 1));
        d_gA0.upload(gA0.data(), nA, st); d_gAn.upload(gAn.data(), nA, st); d_gS0.upload(gS0.data(), nS, st); d_gSn.upload(gSn.data(), nS, st);
      }
      const bool sort_groups = getenv("MM_L2_NO_GROUP_SORT") == nullptr;
      const size_t sort_from = getenv("MM_L2_GROUP_SORT_MIN") ? (size_t)std::max(atoi(getenv("MM_L2_GROUP_SORT_MIN")), 1) : 2048;   // (test hook: small batches take the sort too)
      auto sort_by_position = [&](DBuf<int32_t>& g0, DBuf<int32_t>& gn, size_t ng) {
        if (!sort_groups || ng < sort_from || I->n_contigs <= 0) return;                // (small batches: three launches and a sort cost more than the order gives)
        DBuf<uint64_t> key(ng), val(ng), key2(ng), val2(ng);
        l2_group_keys_kernel<<<dim3((unsigned)ceil_div((int64_t)ng, 256)), dim3(256), 0, st>>>(g0.p, gn.p, M->cand.p, (int64_t)ng, key.p, val.p);
        int cbits = 1; while (cbits < 31 && ((int64_t)1 << cbits) < I->n_contigs) ++cbits;
        size_t tmp_bytes = 0;                                     // (positions at 4 kb granularity: bits 12 .. 32 + contig bits)
        MM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key.p, key2.p, val.p, val2.p, ng, 12u, (unsigned)(32 + cbits), st));
        DBuf<uint8_t> tmp(std::max<size_t>(tmp_bytes, 1));
        MM_HIP(rocprim::radix_sort_pairs((void*)tmp.p, tmp_bytes, key.p, key2.p, val.p, val2.p, ng, 12u, (unsigned)(32 + cbits), st));
        l2_group_unpack_kernel<<<dim3((unsigned)ceil_div((int64_t)ng, 256)), dim3(256), 0, st>>>(val2.p, (int64_t)ng, getenv("MM_L2_XCD_ORDER") ? 8 : 1, g0.p, gn.p);
        MM_KERNEL_CHECK();
      };
      sort_by_position(d_gA0, d_gAn, nA);
      sort_by_position(d_gS0, d_gSn, nS);
      // The 10 kb class runs as two launches — groups of three or four candidates of a read in four-wave workgroups, groups of one or two in two-wave
      // workgroups — over disjoint candidates.  One behind the other on one stream each of them ends with a tail of a few long candidates on an otherwise
      // idle device (the two-wave launch keeps the VALU 55 % busy against 87 %; at an eighth of the batch the tails are a third of K5's time).  Side by
      // side — the second launch on the context's auxiliary stream, forked from and joined into the main one by events — each covers the other's tail.
      // Both take their scratch slots from ONE pool with ONE split by XCD (a slot's traffic stays in one L2, mm_l2.hpp), sized for the larger launch.
      // MM_L2_ONE_STREAM=1: one behind the other as until round 5 (cross-check and A/B).
      {
        // side by side both launches draw on the pool at the same time: it holds a slot for every wave of both (as far as they can be resident — slots_of
        // caps it) so that small batches do not queue for each other's slots; one behind the other the larger launch sizes it
        const bool side_by_side = nA && nS && !no_slots && !getenv("MM_L2_ONE_STREAM");   // (without slots the scratch is indexed by wave number of the launch: one launch at a time)
        const size_t n_waves = side_by_side ? nA * 4 + nS * 2 : std::max(nA * 4, nS * 2);
        void* const codes = !(nA || nS) ? nullptr : v2 ? lists_for(n_waves, 2) : codes_for(n_waves, 2);
        uint8_t* const masks = !(nA || nS) ? nullptr : v2 ? zmasks_for(n_waves, 2) : masks_for(n_waves, 2);
        const int n_slots = (int)slots_of(n_waves, 2);
        hipStream_t st_small = st;
        if (side_by_side) {
          ctx->aux_ready();
          st_small = ctx->aux_stream;
          MM_HIP(hipEventRecord(ctx->ev_fork, st));
          MM_HIP(hipStreamWaitEvent(st_small, ctx->ev_fork, 0));
        }
        if (v2) {
          const int bbl = l2z_bloom_log2(smA);
          if (nA) {
            const size_t lds = l2z_lds_bytes(smA, 2, true, bbl, 4);
            set_lds((const void*)l2z_kernel<4, 2, true>, lds);
            l2z_kernel<4, 2, true><<<dim3((unsigned)nA), dim3(256), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p,
                M->accept_min.p, P.k, P.w, smA, bbl, M->l2.p, counters.p, d_gA0.p, d_gAn.p, ovf.p, ovf_n.p, big.p, big_n.p, amb_used_p, (uint32_t*)codes, masks, slot_flags_p, n_slots, cand_hint_p, cand_rng.p);
            MM_KERNEL_CHECK();
          }
          if (nS) {
            // groups of one or two candidates: a two-wave workgroup with the sketch in LDS holds 20 KB for two waves (16 waves per CU); with the sketch left
            // in global memory (the default; MM_L2_SMALL_QLDS=1: in LDS) it holds 10 KB and the CU its 24 waves
            if (getenv("MM_L2_SMALL_QLDS")) {
              const size_t lds = l2z_lds_bytes(smA, 2, true, bbl, 2);
              set_lds((const void*)l2z_kernel<2, 2, true>, lds);
              l2z_kernel<2, 2, true><<<dim3((unsigned)nS), dim3(128), lds, st_small>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p,
                  M->accept_min.p, P.k, P.w, smA, bbl, M->l2.p, counters.p, d_gS0.p, d_gSn.p, ovf.p, ovf_n.p, big.p, big_n.p, amb_used_p, (uint32_t*)codes, masks, slot_flags_p, n_slots, cand_hint_p, cand_rng.p);
            } else {
              const size_t lds = l2z_lds_bytes(smA, 2, false, bbl, 2);
              set_lds((const void*)l2z_kernel<2, 2, false>, lds);
              l2z_kernel<2, 2, false><<<dim3((unsigned)nS), dim3(128), lds, st_small>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p,
                  M->accept_min.p, P.k, P.w, smA, bbl, M->l2.p, counters.p, d_gS0.p, d_gSn.p, ovf.p, ovf_n.p, big.p, big_n.p, amb_used_p, (uint32_t*)codes, masks, slot_flags_p, n_slots, cand_hint_p, cand_rng.p);
            }
            MM_KERNEL_CHECK();
          }
        } else {
        if (nA) {
            const size_t lds = l2_lds_bytes<uint8_t>(smA, true, 4, 2);
            set_lds((const void*)l2_kernel<true, uint8_t, 4, 2>, lds);
            l2_kernel<true, uint8_t, 4, 2><<<dim3((unsigned)nA), dim3(256), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
                M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smA, M->l2.p, counters.p, d_gA0.p, d_gAn.p, nullptr, ovf.p, ovf_n.p, amb_used_p, codes, masks, slot_flags_p, n_slots);
            MM_KERNEL_CHECK();
          }
          if (nS) {
            const size_t lds = l2_lds_bytes<uint8_t>(smA, true, 2, 2);
            set_lds((const void*)l2_kernel<true, uint8_t, 2, 2>, lds);
            l2_kernel<true, uint8_t, 2, 2><<<dim3((unsigned)nS), dim3(128), lds, st_small>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
                M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smA, M->l2.p, counters.p, d_gS0.p, d_gSn.p, nullptr, ovf.p, ovf_n.p, amb_used_p, codes, masks, slot_flags_p, n_slots);
            MM_KERNEL_CHECK();
          }
        }
        if (side_by_side) {
          MM_HIP(hipEventRecord(ctx->ev_join, st_small));
          MM_HIP(hipStreamWaitEvent(st, ctx->ev_join, 0));
        }
      }
      if (!gB0.empty()) {
        d_gB0.upload(gB0.data(), gB0.size(), st); d_gBn.upload(gBn.data(), gBn.size(), st);
        sort_by_position(d_gB0, d_gBn, gB0.size());
        if (v2_long) {
          const int bbl = l2z_bloom_log2(smB);
          const size_t lds = l2z_lds_bytes(smB, 8, false, bbl, 4);
          set_lds((const void*)l2z_kernel<4, 8, false>, lds);
          l2z_kernel<4, 8, false><<<dim3((unsigned)gB0.size()), dim3(256), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p,
              M->accept_min.p, P.k, P.w, smB, bbl, M->l2.p, counters.p, d_gB0.p, d_gBn.p, ovf.p, ovf_n.p, big.p, big_n.p, amb_used_p, (uint32_t*)lists_for(gB0.size() * 4, 8), zmasks_for(gB0.size() * 4, 8), slot_flags_p, (int)slots_of(gB0.size() * 4), cand_hint_p, cand_rng.p);
          MM_KERNEL_CHECK();
        } else {
          const size_t lds = l2_lds_bytes<uint8_t>(smB, true, 4, 8);
          set_lds((const void*)l2_kernel<true, uint8_t, 4, 8>, lds);
          l2_kernel<true, uint8_t, 4, 8><<<dim3((unsigned)gB0.size()), dim3(256), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
              M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smB, M->l2.p, counters.p, d_gB0.p, d_gBn.p, nullptr, ovf.p, ovf_n.p, amb_used_p, codes_for(gB0.size() * 4, 8), masks_for(gB0.size() * 4), slot_flags_p, (int)slots_of(gB0.size() * 4));
          MM_KERNEL_CHECK();
        }
      }
      DBuf<int32_t> d_gD0(gD0.size()), d_gDn(gDn.size());
      if (!gD0.empty()) {
        d_gD0.upload(gD0.data(), gD0.size(), st); d_gDn.upload(gDn.data(), gDn.size(), st);
        sort_by_position(d_gD0, d_gDn, gD0.size());
        if (v2_long) {
          const int bbl = l2z_bloom_log2(smD);
          const size_t lds = l2z_lds_bytes(smD, 8, false, bbl, 4);
          set_lds((const void*)l2z_kernel<4, 8, false>, lds);
          l2z_kernel<4, 8, false><<<dim3((unsigned)gD0.size()), dim3(256), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p, M->mz.off.p, M->sk_n.p, M->d_read_len.p,
              M->accept_min.p, P.k, P.w, smD, bbl, M->l2.p, counters.p, d_gD0.p, d_gDn.p, ovf.p, ovf_n.p, big.p, big_n.p, amb_used_p, (uint32_t*)lists_for(gD0.size() * 4, 8), zmasks_for(gD0.size() * 4, 8), slot_flags_p, (int)slots_of(gD0.size() * 4), cand_hint_p, cand_rng.p);
          MM_KERNEL_CHECK();
        } else {
          const size_t lds = l2_lds_bytes<uint8_t>(smD, true, 4, 8);
          set_lds((const void*)l2_kernel<true, uint8_t, 4, 8>, lds);
          l2_kernel<true, uint8_t, 4, 8><<<dim3((unsigned)gD0.size()), dim3(256), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
              M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smD, M->l2.p, counters.p, d_gD0.p, d_gDn.p, nullptr, ovf.p, ovf_n.p, amb_used_p, codes_for(gD0.size() * 4, 8), masks_for(gD0.size() * 4), slot_flags_p, (int)slots_of(gD0.size() * 4));
          MM_KERNEL_CHECK();
        }
      }
      if (!listC.empty()) {
        d_listC.upload(listC.data(), listC.size(), st);
        const size_t lds = l2_lds_bytes<uint16_t>(smC, true, 1, 8);
        set_lds((const void*)l2_kernel<true, uint16_t, 1, 8>, lds);
        l2_kernel<true, uint16_t, 1, 8><<<dim3((unsigned)listC.size()), dim3(64), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
            M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smC, M->l2.p, counters.p, nullptr, nullptr, d_listC.p, ovf.p, ovf_n.p, amb_used_p, nullptr, masks_for(listC.size()), slot_flags_p, (int)slots_of(listC.size()));
        MM_KERNEL_CHECK();
      }
      hl("K5 uploads + launches");
      if (M->at_stage) M->at_stage(M->at_stage_user, 2);          // mm_map_batch_phased, stage 2: the last big kernel is enqueued
      // candidates the skip kernels hand back (reads shorter than w+k): the literal full slide
      int64_t n_fallback = 0;
      auto run_fallback = [&](uint8_t* amb_ptr) {
        unsigned int h_ovf = 0;
        MM_HIP(hipMemcpyAsync(&h_ovf, ovf_n.p, sizeof h_ovf, hipMemcpyDeviceToHost, st));
        MM_HIP(mm::stream_sync(st));                        // also keeps the host lists alive until the uploads are done
        if (!h_ovf) return;
        const size_t lds = l2_lds_bytes<uint16_t>(smax, false, 1, 8);
        set_lds((const void*)l2_kernel<false, uint16_t, 1, 8>, lds);
        l2_kernel<false, uint16_t, 1, 8><<<dim3(h_ovf), dim3(64), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
            M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smax, M->l2.p, counters.p, nullptr, nullptr, ovf.p, nullptr, nullptr, amb_ptr, nullptr, nullptr, nullptr, 0);
        MM_KERNEL_CHECK();
        ovf_n.zero(st);
        n_fallback += h_ovf;
      };
      int64_t n_big = 0;
      if (v2) {                                                    // what the zone kernels handed back for its size: one wave per candidate, masks for 32 768 entries (beyond: every window)
        unsigned int h_big = 0;
        MM_HIP(hipMemcpyAsync(&h_big, big_n.p, sizeof h_big, hipMemcpyDeviceToHost, st));
        MM_HIP(mm::stream_sync(st));
        if (h_big) {
          const int smW = std::max(std::max(smA, smB), smD);
          const size_t lds = l2_lds_bytes<uint16_t>(smW, true, 1, 8);
          set_lds((const void*)l2_kernel<true, uint16_t, 1, 8>, lds);
          l2_kernel<true, uint16_t, 1, 8><<<dim3(h_big), dim3(64), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
              M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smW, M->l2.p, counters.p, nullptr, nullptr, big.p, ovf.p, ovf_n.p, amb_used_p, nullptr, masks_for(h_big), slot_flags_p, (int)slots_of(h_big));
          MM_KERNEL_CHECK();
          n_big = h_big;
        }
      }
      run_fallback(amb_used_p);
      int64_t n_redo = 0;
      if (!lazy_reads.empty()) {                                 // votes that read an unresolved strand: resolve those reads, redo their candidates
        std::vector<uint8_t> used = amb_used.to_host(st, (size_t)n);
        std::vector<int64_t> fix;
        for (int64_t r : lazy_reads) if (used[(size_t)r]) fix.push_back(r);
        if (!fix.empty()) {
          start_tiebreak(fix)();
          std::vector<int32_t> redo, redoL; int smR = 0, smRL = 0;   // redoL: reads of the dense path (long sketches) go through it again
          for (int64_t r : fix) {
            const int sr = M->h_sk_n[(size_t)r];
            const bool dense_r = use_dense && (sr >= L2_SKETCH_LIMIT || (sr >= dense_from && M->read_len[(size_t)r] >= P.w + P.k + 1));
            (dense_r ? smRL : smR) = std::max(dense_r ? smRL : smR, sr);
            for (uint64_t c0 = M->h_cand_off[(size_t)r]; c0 < M->h_cand_off[(size_t)r + 1]; ++c0) (dense_r ? redoL : redo).push_back((int32_t)c0);
          }
          run_dense(redoL, smRL, nullptr);
          n_redo += (int64_t)redoL.size();
          DBuf<int32_t> d_redo(std::max<size_t>(redo.size(), 1)); d_redo.upload(redo.data(), redo.size(), st);
          if (!redo.empty()) {
            const size_t lds = l2_lds_bytes<uint16_t>(smR, true, 1, 8);
            set_lds((const void*)l2_kernel<true, uint16_t, 1, 8>, lds);
            l2_kernel<true, uint16_t, 1, 8><<<dim3((unsigned)redo.size()), dim3(64), lds, st>>>(IV, M->cand.p, M->cand_read.p, M->sk_hash.p, M->sk_strand.p,
                M->mz.off.p, M->sk_n.p, M->d_read_len.p, M->accept_min.p, P.k, P.w, smR, M->l2.p, counters.p, nullptr, nullptr, d_redo.p, ovf.p, ovf_n.p, nullptr, nullptr, masks_for(redo.size()), slot_flags_p, (int)slots_of(redo.size()));
            MM_KERNEL_CHECK();
            run_fallback(nullptr);
          }
          MM_HIP(mm::stream_sync(st));
          n_redo += (int64_t)redo.size();
        }
      }
      M->stats.n_l2_wide_redo = n_redo + n_fallback + n_big;
    }
    l2_stats_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(ncand, 256), 1024)), dim3(256), 0, st>>>(M->l2.p, ncand, counters.p);
    MM_KERNEL_CHECK();
    T.end(t_l2);
    hl("K5 wait");
    auto hc = counters.to_host(st);
    M->stats.sum_l2_stream_entries = (int64_t)hc[0];
    M->stats.sum_l2_evals = (int64_t)hc[1];
    M->stats.n_l2_rebuilds = (int64_t)hc[2];
    if (getenv("MM_L2_PHASES")) { fprintf(stderr, "l2 rounds %llu zone passes %llu; ", hc[15], hc[12]); fprintf(stderr, "l2 phase clocks [setup passA bounds rebuild slide passB vote]:"); for (int i = 0; i < 7; ++i) fprintf(stderr, " %.3g", (double)hc[3 + i]); fprintf(stderr, "\n"); }
    // ---- compaction
    if (amb_finish) { amb_finish(); amb_finish = nullptr; }
    const size_t t_cp = T.begin(&M->stats.ms_compact);
    DBuf<uint32_t> flag((size_t)ncand);
    DBuf<uint64_t> rank((size_t)ncand + 1);
    accept_flags_kernel<<<dim3((unsigned)ceil_div(ncand, 256)), dim3(256), 0, st>>>(M->l2.p, ncand, flag.p);
    MM_KERNEL_CHECK();
    exclusive_scan_u32_u64(flag.p, ncand, rank.p, scan_tmp, st);
    uint64_t nrec = 0;
    MM_HIP(hipMemcpyAsync(&nrec, rank.p + ncand, sizeof nrec, hipMemcpyDeviceToHost, st));
    MM_HIP(mm::stream_sync(st));
    M->n_rec = (int64_t)nrec;
    M->rec.alloc((size_t)std::max<uint64_t>(nrec, 1));
    write_records_kernel<<<dim3((unsigned)ceil_div(ncand, 256)), dim3(256), 0, st>>>(M->l2.p, M->cand_read.p, M->sk_n.p, flag.p, rank.p, ncand, M->rec.p);
    MM_KERNEL_CHECK();
    read_rec_bounds_kernel<<<dim3((unsigned)ceil_div(n + 1, 256)), dim3(256), 0, st>>>(M->cand_off.p, rank.p, n, M->rec_off.p);
    MM_KERNEL_CHECK();
    T.end(t_cp);
    MM_HIP(mm::stream_sync(st));
  } else {
    T.end(t_l1);
    if (amb_finish) { amb_finish(); amb_finish = nullptr; }
    M->n_rec = 0;
    M->rec.alloc(1);
    M->rec_off.zero(st);
    MM_HIP(mm::stream_sync(st));
  }
  M->h_rec_off = M->rec_off.to_host(st, (size_t)n + 1);
  T.end(t_total);
  T.collect();
  M->stats.n_mappings = M->n_rec;
  for (int64_t r = 0; r < n; ++r) if (M->h_rec_off[(size_t)r + 1] > M->h_rec_off[(size_t)r]) M->stats.n_reads_mapped++;
}

void probed_list_hist(mm_ctx* ctx, const mm_index* I, const mm_mapping* M, int nb, int64_t* hist) {
  hipStream_t st = ctx->stream;
  DBuf<unsigned long long> d((size_t)nb);
  d.zero(st);
  if (M->n_reads > 0)
    probed_list_hist_kernel<<<dim3((unsigned)M->n_reads), dim3(256), 0, st>>>(make_view(I), M->sk_hash.p, M->mz.off.p, M->sk_n.p, nb, d.p);
  MM_KERNEL_CHECK();
  auto h = d.to_host(st);
  for (int i = 0; i < nb; ++i) hist[i] = (int64_t)h[(size_t)i];
}

}  // namespace mm
