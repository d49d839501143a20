// Index construction on the device (replaces Sketch::build_and_store_index + computeFreqHist for one chunk,
// winSketch.hpp:180-365, :452-494).  K1 sweep over the contigs, one global radix sort by hash (rocPRIM —
// a plain library sort, run once per index, outside the per-read hot path), CSR + bucket table, duplicate
// flags for the L2 sliding window, occurrence histogram.
#include "mm_index.hpp"
#include "mm_minimizer.hpp"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdlib>

namespace mm {

__global__ void split_records_kernel(const Rec* __restrict__ rec, const uint32_t* __restrict__ rec_seq, int64_t n,
                                     uint32_t* __restrict__ key, uint64_t* __restrict__ val) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    Rec r = rec[i]; key[i] = r.hash; val[i] = ((uint64_t)rec_seq[i] << 32) | r.pw;
  }
}

// ---- partitioned sort for indexes beyond the library sort's 32-bit element range -------------------------
// 65536-bin histogram over the top 16 hash bits picks hash ranges of at most PART_MAX entries; each range is
// selected in input order (stable: tile counts -> scan -> ordered tile writes), sorted by the library, and
// lands at its final offset.
constexpr int64_t PART_MAX_DEFAULT = 1500000000LL;
constexpr int SEL_TILE = 2048;
__global__ void __launch_bounds__(256) top16_hist_kernel(const Rec* __restrict__ rec, int64_t n, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int lh[65536 / 4];                          // four passes of 16384 bins keep LDS at 64 KiB
  for (int pass = 0; pass < 4; ++pass) {
    for (int i = threadIdx.x; i < 16384; i += 256) lh[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      uint32_t b = rec[i].hash >> 16;
      if ((int)(b >> 14) == pass) atomicAdd(&lh[b & 16383], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16384; i += 256) if (lh[i]) atomicAdd(&hist[pass * 16384 + i], (unsigned long long)lh[i]);
    __syncthreads();
  }
}
template <bool WRITE>
__global__ void __launch_bounds__(256) select_range_kernel(const Rec* __restrict__ rec, const uint32_t* __restrict__ rec_seq, int64_t n,
                                                           uint32_t lo16, uint32_t hi16, uint32_t* __restrict__ tile_cnt,
                                                           const uint64_t* __restrict__ tile_off, uint32_t* __restrict__ key, uint64_t* __restrict__ val) {
  const int64_t base = (int64_t)blockIdx.x * SEL_TILE + (int64_t)threadIdx.x * (SEL_TILE / 256);
  uint32_t mask = 0;
#pragma unroll
  for (int i = 0; i < SEL_TILE / 256; ++i) {
    int64_t j = base + i;
    if (j < n) { uint32_t b = rec[j].hash >> 16; if (b >= lo16 && b < hi16) mask |= 1u << i; }
  }
  uint64_t tot;
  uint64_t ex = block_excl_scan_u64(__popc(mask), &tot);
  if (!WRITE) { if (threadIdx.x == 0) tile_cnt[blockIdx.x] = (uint32_t)tot; return; }
  uint64_t o = tile_off[blockIdx.x] + ex;
#pragma unroll
  for (int i = 0; i < SEL_TILE / 256; ++i)
    if (mask & (1u << i)) { Rec r = rec[base + i]; key[o] = r.hash; val[o] = ((uint64_t)rec_seq[base + i] << 32) | r.pw; ++o; }
}

// CSR over the unique hashes of the hash-sorted keys without per-entry arrays: a "head" is an entry whose key differs from its predecessor's.
// Heads are counted per tile of SCAN_TILE entries, the tile counts are scanned (mm_scan.hpp, a few MB), and the fill pass finds the heads of
// its tile again and ranks them with a block scan.  (Rounds 1-2 wrote a flag word and a 64-bit rank per entry — 71 GB for the miniSeq+H
// index, allocated in the middle of the build: the driver clears recycled memory when it hands it out, and those two allocations alone
// cost 2-3 s of the build's 9, MM_ALLOC_TRACE.)  Thread t of a tile owns SCAN_ITEMS consecutive entries, so ranks follow the entry order.
__global__ void __launch_bounds__(SCAN_THREADS) head_count_kernel(const uint32_t* __restrict__ key, int64_t n, uint32_t* __restrict__ tile_cnt) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t c = 0;
  uint32_t prev = base > 0 && base <= n ? key[base - 1] : 0u;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const int64_t j = base + i;
    if (j < n) { const uint32_t kj = key[j]; c += (j == 0 || kj != prev) ? 1u : 0u; prev = kj; }
  }
  uint64_t tot;
  block_excl_scan_u64(c, &tot);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = (uint32_t)tot;
}
__global__ void __launch_bounds__(SCAN_THREADS) csr_fill_tiles_kernel(const uint32_t* __restrict__ key, int64_t n, const uint64_t* __restrict__ tile_off, int64_t ntiles,
                                                                      uint32_t* __restrict__ uh, uint64_t* __restrict__ ustart) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t kk[SCAN_ITEMS]; bool head[SCAN_ITEMS];
  uint32_t c = 0;
  uint32_t prev = base > 0 && base <= n ? key[base - 1] : 0u;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const int64_t j = base + i;
    kk[i] = j < n ? key[j] : 0u;
    head[i] = j < n && (j == 0 || kk[i] != prev);
    c += head[i] ? 1u : 0u; prev = kk[i];
  }
  uint64_t u = block_excl_scan_u64(c, nullptr) + tile_off[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) if (head[i]) { uh[u] = kk[i]; ustart[u] = (uint64_t)(base + i); ++u; }
  if (blockIdx.x == 0 && threadIdx.x == 0) ustart[tile_off[ntiles]] = (uint64_t)n;   // tile_off[ntiles] = U (scan total)
}

// paranoia for > 2^32-element library sorts: number of adjacent inversions must be zero
__global__ void count_inversions_kernel(const uint32_t* __restrict__ key, int64_t n, unsigned long long* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += (int64_t)gridDim.x * blockDim.x)
    if (key[i] > key[i + 1]) atomicAdd(bad, 1ull);
}

// Occurrence lists are laid out on 64-byte sector boundaries: the seed-hit filter reads every list of every read twice and
// pays per 64-byte sector touched (docs/history.md K3c); a list that starts mid-sector touches one sector more than it needs.
// Cost: each list is padded to a multiple of 8 entries (~3.5 entries per unique hash).
__global__ void padded_counts_kernel(const uint64_t* __restrict__ ustart, int64_t U, uint32_t* __restrict__ pc) {
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t c = ustart[u + 1] - ustart[u];
    pc[u] = (uint32_t)((c + 7) & ~7ull);                         // (a list of >= 2^32 entries cannot exist: N < 2^32 per contig set is checked by the caller's limits)
  }
}
// eight lanes per list: entry j of list u moves from occ[ustart[u] + j] to out[pstart[u] + j]
// (also fills occ16[]: the filter's position bin of every entry, cbase[c] = first base of contig c in the concatenated reference)
__global__ void __launch_bounds__(256) pad_lists_kernel(const uint64_t* __restrict__ occ, const uint64_t* __restrict__ ustart,
                                                        const uint64_t* __restrict__ pstart, int64_t U, const uint64_t* __restrict__ cbase,
                                                        uint64_t* __restrict__ out, uint16_t* __restrict__ out16) {
  const int sub = threadIdx.x & 7;
  for (int64_t u = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; u < U; u += ((int64_t)gridDim.x * blockDim.x) >> 3) {
    const uint64_t a = ustart[u], c = ustart[u + 1] - a, b = pstart[u], cpad = (c + 7) & ~7ull;
    for (uint64_t j = sub; j < cpad; j += 8) {
      if (j < c) {
        const uint64_t e = occ[a + j];
        out[b + j] = e;
        out16[b + j] = (uint16_t)(((cbase[e >> 32] + (uint64_t)pw_wpos((uint32_t)e)) >> HF_BIN_SHIFT) & (HF_SLOTS - 1));
      } else { out[b + j] = 0; out16[b + j] = hf_pad_code(b, j); }   // padding: no occurrence, a bin of its own (mm_index.hpp)
    }
  }
}

// open-addressing insert of every unique hash: slot word 0 = count<<32 | hash (non-zero: count >= 1), word 1 = start
__global__ void table_insert_kernel(const uint32_t* __restrict__ uh, const uint64_t* __restrict__ ustart, const uint64_t* __restrict__ pstart,
                                    int64_t U, uint32_t buckets, unsigned long long* __restrict__ tab) {
  const uint64_t slots = (uint64_t)buckets << 2;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = uh[u];
    const uint64_t st = pstart[u], cnt = ustart[u + 1] - ustart[u];
    const unsigned long long w0 = ((unsigned long long)(cnt > 0xFFFFFFFFull ? 0xFFFFFFFFull : cnt) << 32) | h;
    uint64_t slot = tab_slot(h, buckets);
    while (atomicCAS(&tab[2 * slot], 0ull, w0) != 0ull) slot = tab_next_slot(slot, slots);
    tab[2 * slot + 1] = st;
  }
}

// Duplicate flags: two entries of one contig with the same hash.  The hash-sorted table is stable, so such
// entries are adjacent there.  DN on the earlier, DP on the later (slidingMap.hpp:139-214 needs "is another
// occurrence of this hash inside the window?", which only same-contig neighbours can answer yes to).
// Entry number of (contig c, wpos p) in pos[]: one directory read bounds it to a bucket of ~128 entries.
__device__ inline int64_t entry_ordinal(const Rec* __restrict__ pos, const uint64_t* __restrict__ cstart, const uint32_t* __restrict__ dir,
                                        const uint64_t* __restrict__ dir_off, int dir_shift, int32_t c, int32_t p) {
  const int64_t cbeg = (int64_t)cstart[c];
  const uint64_t d0 = dir_off[c], nb = dir_off[c + 1] - d0 - 1;
  const uint64_t bk = min((uint64_t)max(p, 0) >> dir_shift, nb - 1);
  int64_t lo = cbeg + (int64_t)dir[d0 + bk], hi = cbeg + (int64_t)dir[d0 + bk + 1];
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (pw_wpos(pos[mid].pw) < p) lo = mid + 1; else hi = mid; }
  return lo;
}
// DIST=false: the flags.  DIST=true (second sweep, once the flags of all entries are known and ranked): the distance between the
// two entries of every pair, into the slot of the later one as "previous" and of the earlier one as "next" (mm_index.hpp).
template <bool DIST>
__global__ void dup_pairs_kernel(const uint32_t* __restrict__ key, const uint64_t* __restrict__ val, int64_t n,
                                 const uint64_t* __restrict__ cstart, const uint32_t* __restrict__ dir, const uint64_t* __restrict__ dir_off, int dir_shift,
                                 Rec* __restrict__ pos, unsigned long long* __restrict__ ndup,
                                 const uint64_t* __restrict__ dup_bits, const uint64_t* __restrict__ dup_rank, uint16_t* __restrict__ dist16, int sat) {
  unsigned long long mine = 0;                                   // pairs this thread found (one add per wave at the end: 3*10^8 adds to one word serialise)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (key[i] != key[i + 1]) continue;
    const uint64_t a = val[i], b = val[i + 1];
    if ((a >> 32) != (b >> 32)) continue;
    const int32_t c = (int32_t)(a >> 32);
    const int64_t oa = entry_ordinal(pos, cstart, dir, dir_off, dir_shift, c, pw_wpos((uint32_t)a));
    const int64_t ob = entry_ordinal(pos, cstart, dir, dir_off, dir_shift, c, pw_wpos((uint32_t)b));
    if (!DIST) {
      atomicOr(&pos[oa].pw, PW_DN);
      atomicOr(&pos[ob].pw, PW_DP);
      ++mine;
    } else {
      const uint16_t d = (uint16_t)min<int64_t>(ob - oa, (int64_t)sat);
      auto slot = [&](int64_t j) -> uint64_t { return dup_rank[j >> 6] + (uint64_t)__popcll(dup_bits[j >> 6] & ((1ull << (j & 63)) - 1ull)); };
      dist16[2 * slot(oa) + 1] = d;                               // an entry has at most one pair on each side: no two threads write one half
      dist16[2 * slot(ob)] = d;
    }
  }
  if (!DIST) {
    for (int dlt = 32; dlt > 0; dlt >>= 1) mine += __shfl_xor(mine, dlt, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(ndup, mine);
  }
}
// one wave per block of 64 entries: which of them are flagged
__global__ void __launch_bounds__(256) dup_bits_kernel(const Rec* __restrict__ pos, int64_t N, uint64_t* __restrict__ bits, uint32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t nblk = (N + 63) >> 6, stride = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t blk = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6; blk < nblk; blk += stride) {
    const int64_t j = (blk << 6) + lane;
    const uint64_t m = __ballot(j < N && (pos[j].pw & (PW_DP | PW_DN)) != 0u);
    if (lane == 0) { bits[blk] = m; cnt[blk] = (uint32_t)__popcll(m); }
  }
}

constexpr int HIST_BINS = 4096;
// occurrence-count histogram; counts >= HIST_BINS go to an overflow list (count per hash)
__global__ void __launch_bounds__(256) count_hist_kernel(const uint64_t* __restrict__ ustart, int64_t U, unsigned long long* __restrict__ bins,
                                                         unsigned long long* __restrict__ big, unsigned long long* __restrict__ nbig, int64_t big_cap) {
  __shared__ unsigned int lb[HIST_BINS];
  for (int i = threadIdx.x; i < HIST_BINS; i += 256) lb[i] = 0;
  __syncthreads();
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < U; u += (int64_t)gridDim.x * 256) {
    uint64_t c = ustart[u + 1] - ustart[u];
    if (c < HIST_BINS) atomicAdd(&lb[c], 1u);
    else { unsigned long long s = atomicAdd(nbig, 1ull); if ((int64_t)s < big_cap) big[s] = c; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HIST_BINS; i += 256) if (lb[i]) atomicAdd(&bins[i], (unsigned long long)lb[i]);
}

// directory entry g: contig by binary search over dir_off, then the lower bound of (b << shift) among the contig's positions
__global__ void __launch_bounds__(256) dir_build_kernel(const Rec* __restrict__ pos, const uint64_t* __restrict__ cstart, const uint64_t* __restrict__ dir_off,
                                                        int64_t n_contigs, int shift, uint64_t total, uint32_t* __restrict__ dir) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  int64_t lo = 0, hi = n_contigs;                                // last c with dir_off[c] <= g
  while (hi - lo > 1) { const int64_t m = (lo + hi) >> 1; if (dir_off[m] <= g) lo = m; else hi = m; }
  const int64_t c = lo;
  const int64_t target = (int64_t)(g - dir_off[c]) << shift;
  int64_t a = (int64_t)cstart[c], b = (int64_t)cstart[c + 1];
  const int64_t base = a;
  while (a < b) { const int64_t m = (a + b) >> 1; if ((int64_t)pw_wpos(pos[m].pw) < target) a = m + 1; else b = m; }
  dir[g] = (uint32_t)(a - base);
}

static void build_directory(mm_index* I, hipStream_t st) {
  // about 128 entries per directory bucket at the expected density 2 / (w + 1)
  int shift = 4; while ((128.0 * (I->w + 1) / 2.0) >= (double)((int64_t)2 << shift) && shift < 20) ++shift;
  I->dir_shift = shift;
  std::vector<uint64_t> off((size_t)I->n_contigs + 1, 0);
  for (int64_t c = 0; c < I->n_contigs; ++c) off[(size_t)c + 1] = off[(size_t)c] + ((uint64_t)std::max(I->contig_len[(size_t)c], 0) >> shift) + 2;
  const uint64_t total = off.back();
  I->dir_off.alloc(off.size()); I->dir_off.upload(off.data(), off.size(), st);
  I->dir.alloc(std::max<size_t>((size_t)total, 1));
  if (total && I->n_contigs) {
    dir_build_kernel<<<dim3((unsigned)ceil_div((int64_t)total, 256)), dim3(256), 0, st>>>(I->pos.p, I->cstart.p, I->dir_off.p, I->n_contigs, shift, total, I->dir.p);
    MM_KERNEL_CHECK();
  }
  MM_HIP(mm::stream_sync(st));                              // `off` is the upload source
}

void index_build(mm_ctx* ctx, const mm_seqset* contigs, int k, int w, mm_index* I) {
  hipStream_t st = ctx->stream;
  struct BuildClock {                                            // MM_HOST_TIMING=1: wall time of the build's sections (stderr; each lap drains the stream)
    hipStream_t st; bool on = getenv("MM_HOST_TIMING") != nullptr; std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
      if (!on) return;
      (void)mm::stream_sync(st);
      const auto n = std::chrono::steady_clock::now();
      fprintf(stderr, "INFO, index build: %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count()); t = n;
    }
  } clk{st};
  // Two allocation regimes (mm_common.hpp).  The index of a chunk of a --maxmemory run, one of many of its size: the cached blocks of
  // earlier work stay and serve this build (0.15 s per 13 GB chunk instead of 1-6 s through the driver), an allocation that fails for
  // lack of memory trims the caches itself.  An index that takes a good part of the device: memory goes back to the driver as the build
  // proceeds (DevAlloc::eager), nothing is left cached beside it.
  struct EagerGuard { DevAlloc& a; bool was, was_build; ~EagerGuard() { a.eager = was; a.in_build = was_build; } } eager_guard{ctx->alloc, ctx->alloc.eager, ctx->alloc.in_build};
  ctx->alloc.in_build = true;
  { size_t fr = 0, tot = 0;
    if (dev_mem_info(&fr, &tot) == hipSuccess && (double)contigs->total_bases * 5.5 > (double)tot / 4) {
      // (until round 4's last session the device's pool went back to the driver here, before every device-filling build: with the chunk indexes of a
      // reference larger than the device built in turn that meant ~150 GB fresh from the driver per 15 Gbp chunk, cleared as it is handed out — 5.6 s of a
      // 7.0 s build.  Now the pool stays: what the build asks for and the pool has is taken, what the driver cannot give is freed block by block.  MM_INDEX_PRETRIM=1: as before)
      ctx->alloc.eager = true;
      if (!getenv("MM_INDEX_NO_PRETRIM")) { ctx->alloc.trim(); if (getenv("MM_INDEX_PRETRIM") || getenv("MM_NO_POOL_RESCUE")) big_pool_trim(ctx->device); else big_pool_adopt_idle(ctx->device); }
    } }
  clk.lap("pool handling");
  I->ctx = ctx; I->k = k; I->w = w;
  I->n_contigs = contigs->count();
  I->contig_len = contigs->len;
  I->d_contig_len.alloc(std::max<size_t>(I->contig_len.size(), 1));
  I->d_contig_len.upload(I->contig_len.data(), I->contig_len.size(), st);

  // K1 over every contig; contigs shorter than w or k contribute metadata only (winSketch.hpp:258-264)
  MinimizerSet ms;
  run_minimizers(ctx, contigs, k, w, {}, /*want_rec_seq=*/true, ms);
  clk.lap("minimizers");
  const int64_t N = ms.total;
  I->N = N;
  I->pos = std::move(ms.rec);
  I->cstart = std::move(ms.off);
  I->h_cstart = ms.h_off;
  I->U = 0; I->n_dup = 0; I->hist.clear();
  build_directory(I, st);
  clk.lap("position directory");
  if (N == 0) {
    I->tab_buckets = 64;
    I->tab.alloc((size_t)I->tab_buckets * 8); I->tab.zero(st);
    I->occ.alloc(1); I->occ16.alloc(16);
    I->dup_bits.alloc(1); I->dup_rank.alloc(2); I->dup_dist.alloc(1);
    I->dup_bits.zero(st); I->dup_rank.zero(st); I->dup_dist.zero(st);
    MM_HIP(mm::stream_sync(st));
    return;
  }
  // HIP caps gridDim.x*blockDim.x below 2^32 threads: per-element kernels are grid-stride loops on a capped grid
  const unsigned nblk = (unsigned)std::min<int64_t>(ceil_div(N, 256), 1 << 22);
  // sort (hash -> contig<<32|pw) by hash, stable
  DBuf<uint32_t> key_in, key_out((size_t)N);
  DBuf<uint64_t> val_in;
  I->occ.alloc((size_t)N + 2);                                  // +2: the seed-hit filter reads lists in aligned 16-byte pieces
  DBuf<uint8_t> sort_tmp;                                        // the library's double buffers (12 B per element), kept across the partitions: every
  auto library_sort = [&](uint32_t* kin, uint64_t* vin, uint32_t* kout, uint64_t* vout, size_t cnt) {   // fresh 18 GB block is recycled memory the driver clears first
    size_t tmp_bytes = 0;
    MM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, cnt, 0, 32, st));
    if (sort_tmp.bytes() < tmp_bytes) sort_tmp.alloc(tmp_bytes);
    MM_HIP(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, kin, kout, vin, vout, cnt, 0, 32, st));
    MM_HIP(mm::stream_sync(st));
  };
  const char* pm_env = getenv("MM_INDEX_PART_MAX");              // tests force the partitioned path on small inputs
  const int64_t PART_MAX = pm_env ? std::max<int64_t>(atoll(pm_env), 1) : PART_MAX_DEFAULT;
  if (N <= PART_MAX) {
    key_in.alloc((size_t)N); val_in.alloc((size_t)N);
    split_records_kernel<<<dim3(nblk), dim3(256), 0, st>>>(I->pos.p, ms.rec_seq.p, N, key_in.p, val_in.p);
    MM_KERNEL_CHECK();
    ms.rec_seq.release();
    library_sort(key_in.p, val_in.p, key_out.p, I->occ.p, (size_t)N);
  } else {
    DBuf<unsigned long long> d_hist(65536); d_hist.zero(st);
    top16_hist_kernel<<<dim3(2048), dim3(256), 0, st>>>(I->pos.p, N, d_hist.p);
    MM_KERNEL_CHECK();
    auto hist = d_hist.to_host(st);
    std::vector<std::pair<uint32_t, uint32_t>> parts;            // [lo16, hi16)
    std::vector<int64_t> part_cnt;
    {
      uint32_t lo = 0; int64_t acc = 0;
      for (uint32_t b = 0; b < 65536; ++b) {
        MM_REQUIRE((int64_t)hist[b] <= PART_MAX, MM_ERR_LIMIT, "one 16-bit hash prefix holds more than 1.5e9 index entries");
        if (acc + (int64_t)hist[b] > PART_MAX) { parts.push_back({lo, b}); part_cnt.push_back(acc); lo = b; acc = 0; }
        acc += (int64_t)hist[b];
      }
      parts.push_back({lo, 65536u}); part_cnt.push_back(acc);
    }
    const int64_t ntile = ceil_div(N, SEL_TILE);
    DBuf<uint32_t> tcnt((size_t)ntile);
    DBuf<uint64_t> toff((size_t)ntile + 1), scan_tmp2;
    int64_t maxp = 0; for (auto c : part_cnt) maxp = std::max(maxp, c);
    key_in.alloc((size_t)maxp); val_in.alloc((size_t)maxp);
    { size_t tb = 0;                                             // (sized for the largest partition once)
      MM_HIP(rocprim::radix_sort_pairs(nullptr, tb, key_in.p, key_out.p, val_in.p, I->occ.p, (size_t)maxp, 0, 32, st));
      sort_tmp.alloc(tb + (tb >> 6)); }
    int64_t done = 0;
    for (size_t p = 0; p < parts.size(); ++p) {
      if (part_cnt[p] == 0) continue;
      select_range_kernel<false><<<dim3((unsigned)ntile), dim3(256), 0, st>>>(I->pos.p, ms.rec_seq.p, N, parts[p].first, parts[p].second, tcnt.p, nullptr, nullptr, nullptr);
      MM_KERNEL_CHECK();
      exclusive_scan_u32_u64(tcnt.p, ntile, toff.p, scan_tmp2, st);
      select_range_kernel<true><<<dim3((unsigned)ntile), dim3(256), 0, st>>>(I->pos.p, ms.rec_seq.p, N, parts[p].first, parts[p].second, nullptr, toff.p, key_in.p, val_in.p);
      MM_KERNEL_CHECK();
      library_sort(key_in.p, val_in.p, key_out.p + done, I->occ.p + done, (size_t)part_cnt[p]);
      done += part_cnt[p];
    }
    MM_REQUIRE(done == N, MM_ERR_DEVICE, "partitioned sort lost index entries");
    ms.rec_seq.release();
  }
  {
    DBuf<unsigned long long> bad(1); bad.zero(st);
    count_inversions_kernel<<<dim3(nblk), dim3(256), 0, st>>>(key_out.p, N, bad.p);
    MM_KERNEL_CHECK();
    auto hb = bad.to_host(st);
    MM_REQUIRE(hb[0] == 0, MM_ERR_DEVICE, "radix sort of the index left the hash keys unsorted");
  }
  clk.lap("sort by hash");
  key_in.release(); val_in.release(); sort_tmp.release();
  // CSR over unique hashes
  const int64_t ntiles_csr = ceil_div(N, SCAN_TILE);
  DBuf<uint32_t> tile_heads((size_t)ntiles_csr);
  DBuf<uint64_t> tile_rank((size_t)ntiles_csr + 1), scan_tmp;
  head_count_kernel<<<dim3((unsigned)ntiles_csr), dim3(SCAN_THREADS), 0, st>>>(key_out.p, N, tile_heads.p);
  MM_KERNEL_CHECK();
  exclusive_scan_u32_u64(tile_heads.p, ntiles_csr, tile_rank.p, scan_tmp, st);
  uint64_t U = 0;
  MM_HIP(hipMemcpyAsync(&U, tile_rank.p + ntiles_csr, sizeof U, hipMemcpyDeviceToHost, st));
  MM_HIP(mm::stream_sync(st));
  I->U = (int64_t)U;
  I->uh.alloc((size_t)U); I->ustart.alloc((size_t)U + 1);
  csr_fill_tiles_kernel<<<dim3((unsigned)ntiles_csr), dim3(SCAN_THREADS), 0, st>>>(key_out.p, N, tile_rank.p, ntiles_csr, I->uh.p, I->ustart.p);
  MM_KERNEL_CHECK();
  tile_heads.release(); tile_rank.release();
  clk.lap("distinct hashes");
  // duplicate flags into pos[]
  DBuf<unsigned long long> ndup(1); ndup.zero(st);
  dup_pairs_kernel<false><<<dim3(nblk), dim3(256), 0, st>>>(key_out.p, I->occ.p, N, I->cstart.p, I->dir.p, I->dir_off.p, I->dir_shift, I->pos.p, ndup.p,
                                                            nullptr, nullptr, nullptr, 0);
  MM_KERNEL_CHECK();
  {                                                              // ... and how far the flagged entries' same-hash neighbours are (mm_index.hpp)
    const int64_t nb64 = (N + 63) >> 6;
    if (const char* e = getenv("MM_DUP_SAT")) I->dup_sat = std::min(std::max(atoi(e), 1), 65535);
    I->dup_bits.alloc((size_t)nb64); I->dup_rank.alloc((size_t)nb64 + 1);
    DBuf<uint32_t> bcnt((size_t)nb64);
    DBuf<uint64_t> scan_tmp4;
    dup_bits_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(nb64, 4), 1 << 20)), dim3(256), 0, st>>>(I->pos.p, N, I->dup_bits.p, bcnt.p);
    MM_KERNEL_CHECK();
    exclusive_scan_u32_u64(bcnt.p, nb64, I->dup_rank.p, scan_tmp4, st);
    uint64_t nflag = 0;
    MM_HIP(hipMemcpyAsync(&nflag, I->dup_rank.p + nb64, sizeof nflag, hipMemcpyDeviceToHost, st));
    MM_HIP(mm::stream_sync(st));
    I->dup_dist.alloc(std::max<size_t>((size_t)nflag, 1)); I->dup_dist.zero(st);
    if (nflag) {
      dup_pairs_kernel<true><<<dim3(nblk), dim3(256), 0, st>>>(key_out.p, I->occ.p, N, I->cstart.p, I->dir.p, I->dir_off.p, I->dir_shift, I->pos.p, nullptr,
                                                               I->dup_bits.p, I->dup_rank.p, (uint16_t*)I->dup_dist.p, I->dup_sat);
      MM_KERNEL_CHECK();
    }
  }
  clk.lap("duplicate flags");
  // occurrence histogram (winSketch.hpp:456-459)
  const int64_t big_cap = 1 << 20;
  DBuf<unsigned long long> bins(HIST_BINS), big((size_t)big_cap), nbig(1);
  bins.zero(st); nbig.zero(st);
  count_hist_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div((int64_t)U, 256), 2048)), dim3(256), 0, st>>>(I->ustart.p, (int64_t)U, bins.p, big.p, nbig.p, big_cap);
  MM_KERNEL_CHECK();
  auto hb = bins.to_host(st);
  auto hn = nbig.to_host(st);
  auto hd = ndup.to_host(st);
  I->n_dup = (int64_t)hd[0];
  MM_REQUIRE((int64_t)hn[0] <= big_cap, MM_ERR_LIMIT, "more than 2^20 hashes occur >= 4096 times in one index chunk");
  for (int i = 0; i < HIST_BINS; ++i) if (hb[i]) I->hist[i] += (int64_t)hb[i];
  if (hn[0]) { auto hbig = big.to_host(st, (size_t)hn[0]); for (auto c : hbig) I->hist[(int64_t)c] += 1; }
  key_out.release();
  // sector-aligned occurrence lists (see padded_counts_kernel)
  clk.lap("occurrence histogram");
  DBuf<uint64_t> pstart((size_t)U + 1);
  {
    DBuf<uint32_t> pc((size_t)U + 1); pc.zero(st);
    padded_counts_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div((int64_t)U, 256), 1 << 20)), dim3(256), 0, st>>>(I->ustart.p, (int64_t)U, pc.p);
    MM_KERNEL_CHECK();
    DBuf<uint64_t> scan_tmp3;
    exclusive_scan_u32_u64(pc.p, (int64_t)U, pstart.p, scan_tmp3, st);
    uint64_t P = 0;
    MM_HIP(hipMemcpyAsync(&P, pstart.p + U, sizeof P, hipMemcpyDeviceToHost, st));
    MM_HIP(mm::stream_sync(st));
    MM_REQUIRE(P < ((uint64_t)1 << 35), MM_ERR_LIMIT, "more than 2^35 padded occurrences in one index chunk (the seed filter keeps list starts / 8 in 32 bits)");
    DBuf<uint64_t> padded((size_t)P + 2);
    I->occ16.alloc((size_t)P + 16);
    std::vector<uint64_t> h_cbase((size_t)I->n_contigs + 1, 0);
    for (int64_t c = 0; c < I->n_contigs; ++c) h_cbase[(size_t)c + 1] = h_cbase[(size_t)c] + (uint64_t)I->contig_len[(size_t)c];
    DBuf<uint64_t> d_cbase(h_cbase.size()); d_cbase.upload(h_cbase.data(), h_cbase.size(), st);
    pad_lists_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div((int64_t)U * 8, 256), 1 << 20)), dim3(256), 0, st>>>(I->occ.p, I->ustart.p, pstart.p, (int64_t)U, d_cbase.p,
                                                                                                               padded.p, I->occ16.p);
    MM_KERNEL_CHECK();
    MM_HIP(mm::stream_sync(st));
    I->occ = std::move(padded);
  }
  clk.lap("padded lists + bin codes");
  // lookup table (load factor <= 0.55, any number of 4-slot buckets), then the CSR arrays are no longer needed
  I->tab_buckets = tab_buckets_for((int64_t)U);
  I->tab.alloc((size_t)I->tab_buckets * 8);
  I->tab.zero(st);
  table_insert_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div((int64_t)U, 256), 1 << 20)), dim3(256), 0, st>>>(I->uh.p, I->ustart.p, pstart.p, (int64_t)U, I->tab_buckets,
                                                                                                              (unsigned long long*)I->tab.p);
  MM_KERNEL_CHECK();
  MM_HIP(mm::stream_sync(st));
  I->uh.release(); I->ustart.release();
  clk.lap("lookup table");
}


// ---------------------------------------------------------------------------------------------------
// --maxmemory chunk rule (winSketch.hpp:180-365, memory model :165-178) evaluated on the whole-reference index.
//
// The reference streams contigs and, before adding contig c, asks whether
//   memory( hashes so far in this chunk + hashes of c not yet in the chunk's table , minimizers so far + |c| )
// exceeds the limit; if so it flushes the chunk and c opens the next one.  With c0 = first contig of the current
// chunk, "hashes of c not yet in the table" = #{h : the first occurrence of h at or after contig c0 lies in c}.
// Every hash group in occ[] is sorted by contig, so one pass over the table finds that first occurrence by a
// lower bound inside the group — one pass per chunk, no per-contig set arithmetic.
__global__ void __launch_bounds__(256) novel_hashes_kernel(const uint64_t* __restrict__ tab, int64_t slots, const uint64_t* __restrict__ occ,
                                                           uint32_t c0, unsigned int* __restrict__ novel) {
  for (int64_t sl = (int64_t)blockIdx.x * 256 + threadIdx.x; sl < slots; sl += (int64_t)gridDim.x * 256) {
    const uint64_t w0 = tab[2 * sl];
    if (w0 == 0) continue;
    const uint64_t start = tab[2 * sl + 1];
    int64_t lo = 0, hi = (int64_t)(w0 >> 32);
    const uint64_t key = (uint64_t)c0 << 32;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (occ[start + mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < (int64_t)(w0 >> 32)) atomicAdd(&novel[(uint32_t)(occ[start + lo] >> 32)], 1u);
  }
}

static size_t chunk_memory(size_t hashes, size_t mins) {           // Sketch::estimateMemory, winSketch.hpp:165-178 (LP64 sizes)
  size_t buckets = hashes / 10;
  size_t table = buckets * (8 + 8) + hashes * 8 + hashes * 24 + mins * 12;
  table *= 1.2;                                                    // size_t *= double, as there
  size_t vec = 24 + mins * 16;
  return table + vec;
}

void index_plan_chunks(mm_ctx* ctx, const mm_index* I, uint64_t max_memory, std::vector<int32_t>& first_contig) {
  hipStream_t st = ctx->stream;
  first_contig.assign(1, 0);
  if (max_memory == 0 || I->n_contigs == 0) return;
  const int64_t C = I->n_contigs, slots = (int64_t)I->tab_buckets * 4;
  DBuf<unsigned int> novel((size_t)C);
  std::vector<unsigned int> h((size_t)C);
  int64_t c0 = 0;
  for (;;) {
    novel.zero(st);
    novel_hashes_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(slots, 256), 1 << 20)), dim3(256), 0, st>>>(I->tab.p, slots, I->occ.p, (uint32_t)c0, novel.p);
    MM_KERNEL_CHECK();
    novel.download(h.data(), (size_t)C, st);
    MM_HIP(mm::stream_sync(st));
    size_t runH = 0, runM = 0;
    int64_t c = c0;
    for (; c < C; ++c) {
      const int len = I->contig_len[(size_t)c];
      if (len < I->w || len < I->k) continue;                      // metadata only (:258-264)
      const size_t addM = (size_t)(I->h_cstart[(size_t)c + 1] - I->h_cstart[(size_t)c]);
      const size_t mem = chunk_memory(runH + h[(size_t)c], runM + addM);
      if (mem > max_memory) {
        // nothing in the chunk yet: the contig alone is over the limit (:318-322)
        MM_REQUIRE(runH != 0 || runM != 0, MM_ERR_LIMIT, "Can't index the reference within current memory limits - a contig is too large");
        break;
      }
      runH += h[(size_t)c]; runM += addM;
    }
    if (c >= C) break;
    first_contig.push_back((int32_t)c);
    c0 = c;
  }
}

}  // namespace mm

// ---------------------------------------------------------------------------------------------------
// Persistent device index (what createIndex / the archive loads of mapAgainstIndex are to the reference,
// mapWrap.h:358-405, :443-554, winSketch.hpp:73-83): the arrays of mm_index as they lie in HBM, so that a load is file -> pinned
// staging -> device, and the only kernel is the one that sums every array up again.  Little endian, versioned; every array is preceded by
// its element count and a checksum of its bytes as they lay in HBM (array_sum_kernel: a position-dependent 64-bit sum, taken on the
// device on both sides, so a load also vouches for its own copies), and the file ends with a closing word (a truncated file is
// refused).  Counts are held against the header (N entries, U hashes, the contig table) before anything is allocated.
//   "MMINDEX1"  u32 version=3  i32 k  i32 w  i32 dir_shift  i32 dup_sat  i32 freq_threshold  u32 tab_buckets  u32 0
//   i64 n_contigs  i64 N  i64 U  i64 n_dup  i64 n_hist
//   i64 hist[n_hist][2]  i32 contig_len[n_contigs]  u64 h_cstart[n_contigs + 1]
//   { u64 count, u64 checksum, bytes }  for pos, cstart, occ, occ16, tab, dir, dir_off, dup_bits, dup_rank, dup_dist
//   u64 0x58444e4958444e49
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t IDX_VERSION = 3;                             // (3: the padding entries of occ16[] carry codes of their own, hf_pad_code)
// checksum of an array as it lies in HBM: sum over its 8-byte words of mix(word + C * index) (+ the same over the bytes of a last partial
// word); a sum, so the order of the threads does not matter
__device__ __forceinline__ uint64_t sum_mix(uint64_t x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; return x; }
__global__ void __launch_bounds__(256) array_sum_kernel(const uint64_t* __restrict__ w, uint64_t n_words, const uint8_t* __restrict__ tail, int n_tail,
                                                        unsigned long long* __restrict__ out) {
  uint64_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256) acc += sum_mix(w[i] + 0xD6E8FEB86659FD93ull * (i + 1));
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int t = 0; t < n_tail; ++t) acc += sum_mix((uint64_t)tail[t] + 0xD6E8FEB86659FD93ull * (n_words + 1 + (uint64_t)t));
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}
template <typename T> uint64_t array_sum(const mm::DBuf<T>& a, size_t count, hipStream_t st) {
  const size_t bytes = count * sizeof(T);
  if (!bytes) return 0;
  mm::DBuf<unsigned long long> d(1); d.zero(st);
  const uint64_t nw = bytes >> 3;
  array_sum_kernel<<<dim3((unsigned)std::min<uint64_t>(std::max<uint64_t>((nw + 255) / 256, 1), 4096)), dim3(256), 0, st>>>((const uint64_t*)a.p, nw, (const uint8_t*)a.p + (nw << 3), (int)(bytes & 7), d.p);
  MM_KERNEL_CHECK();
  return (uint64_t)d.to_host(st)[0];
}
constexpr uint64_t IDX_TAIL = 0x58444e4958444e49ull;
constexpr size_t IDX_STAGE = (size_t)64 << 20;
struct IdxFile {
  FILE* f = nullptr;
  ~IdxFile() { if (f) fclose(f); }
  int close() { const int rc = f ? fclose(f) : 0; f = nullptr; return rc; }   // (a deferred write error — quota, a network file system — shows up here)
};
struct Pinned2 {                                                   // two staging buffers: the device copies into / out of one while the file has the other
  char* b[2] = {nullptr, nullptr};
  Pinned2() { for (auto& p : b) MM_HIP(hipHostMalloc((void**)&p, IDX_STAGE, hipHostMallocDefault)); }
  ~Pinned2() { for (auto p : b) if (p) (void)hipHostFree(p); }
};
void put(FILE* f, const void* p, size_t bytes, const char* what) {
  MM_REQUIRE(bytes == 0 || fwrite(p, 1, bytes, f) == bytes, MM_ERR_ARG, std::string("short write (") + what + ")");
}
void get(FILE* f, void* p, size_t bytes, const char* what) {
  MM_REQUIRE(bytes == 0 || fread(p, 1, bytes, f) == bytes, MM_ERR_ARG, std::string("truncated index file (") + what + ")");
}
template <typename T> void put_array(FILE* f, const mm::DBuf<T>& a, size_t count, Pinned2& pin, hipStream_t st, const char* what) {
  const uint64_t c64 = count, sum = array_sum(a, count, st); put(f, &c64, 8, what); put(f, &sum, 8, what);
  const size_t bytes = count * sizeof(T);
  const char* src = (const char*)a.p;
  size_t off = 0; int cur = 0;
  if (bytes) MM_HIP(hipMemcpyAsync(pin.b[0], src, std::min(bytes, IDX_STAGE), hipMemcpyDeviceToHost, st));
  while (off < bytes) {
    const size_t n = std::min(bytes - off, IDX_STAGE);
    MM_HIP(mm::stream_sync(st));
    if (off + n < bytes) MM_HIP(hipMemcpyAsync(pin.b[cur ^ 1], src + off + n, std::min(bytes - off - n, IDX_STAGE), hipMemcpyDeviceToHost, st));
    put(f, pin.b[cur], n, what);
    off += n; cur ^= 1;
  }
}
// the element count must lie in [lo, hi] (what the header's N, U and contig table allow for this array) before anything is allocated
template <typename T> void get_array(FILE* f, mm::DBuf<T>& a, size_t min_alloc, Pinned2& pin, hipStream_t st, uint64_t lo, uint64_t hi, const char* what) {
  uint64_t c64 = 0, sum = 0; get(f, &c64, 8, what); get(f, &sum, 8, what);
  MM_REQUIRE(c64 >= lo && c64 <= hi, MM_ERR_ARG, std::string("index file is inconsistent (") + what + ")");
  MM_REQUIRE(c64 < ((uint64_t)1 << 40), MM_ERR_ARG, std::string("corrupt index file (") + what + ")");
  a.alloc(std::max<size_t>((size_t)c64, min_alloc));
  const size_t bytes = (size_t)c64 * sizeof(T);
  char* dst = (char*)a.p;
  size_t off = 0; int cur = 0;
  while (off < bytes) {                                            // the read of block i+1 runs beside the copy of block i
    const size_t n = std::min(bytes - off, IDX_STAGE);
    get(f, pin.b[cur], n, what);
    MM_HIP(mm::stream_sync(st));                              // (the other buffer's copy: done before that buffer is read into again)
    MM_HIP(hipMemcpyAsync(dst + off, pin.b[cur], n, hipMemcpyHostToDevice, st));
    off += n; cur ^= 1;
  }
  MM_HIP(mm::stream_sync(st));
  MM_REQUIRE(array_sum(a, (size_t)c64, st) == sum, MM_ERR_ARG, std::string("index file is damaged: checksum of the ") + what + " differs");
}
}  // namespace

namespace mm {

void index_save(const mm_index* I, const char* path) {
  hipStream_t st = I->ctx->stream;
  IdxFile fc; fc.f = fopen(path, "wb");
  MM_REQUIRE(fc.f != nullptr, MM_ERR_ARG, std::string("cannot open ") + path + " for writing");
  setvbuf(fc.f, nullptr, _IONBF, 0);
  Pinned2 pin;
  const uint32_t head[8] = {IDX_VERSION, (uint32_t)I->k, (uint32_t)I->w, (uint32_t)I->dir_shift, (uint32_t)I->dup_sat, (uint32_t)I->freq_threshold, I->tab_buckets, 0u};
  const int64_t dims[5] = {I->n_contigs, I->N, I->U, I->n_dup, (int64_t)I->hist.size()};
  put(fc.f, "MMINDEX1", 8, "magic"); put(fc.f, head, sizeof head, "header"); put(fc.f, dims, sizeof dims, "header");
  std::vector<int64_t> hh; for (auto& kv : I->hist) { hh.push_back(kv.first); hh.push_back(kv.second); }
  put(fc.f, hh.data(), hh.size() * 8, "histogram");
  MM_REQUIRE((int64_t)I->contig_len.size() == I->n_contigs && (int64_t)I->h_cstart.size() == I->n_contigs + 1, MM_ERR_STATE, "index without its contig table");
  put(fc.f, I->contig_len.data(), I->contig_len.size() * 4, "contig lengths");
  put(fc.f, I->h_cstart.data(), I->h_cstart.size() * 8, "contig entry ranges");
  put_array(fc.f, I->pos, I->pos.n, pin, st, "entries");
  put_array(fc.f, I->cstart, I->cstart.n, pin, st, "contig starts");
  put_array(fc.f, I->occ, I->occ.n, pin, st, "occurrences");
  put_array(fc.f, I->occ16, I->occ16.n, pin, st, "occurrence bins");
  put_array(fc.f, I->tab, I->tab.n, pin, st, "hash table");
  put_array(fc.f, I->dir, I->dir.n, pin, st, "position directory");
  put_array(fc.f, I->dir_off, I->dir_off.n, pin, st, "directory offsets");
  put_array(fc.f, I->dup_bits, I->dup_bits.n, pin, st, "duplicate bits");
  put_array(fc.f, I->dup_rank, I->dup_rank.n, pin, st, "duplicate ranks");
  put_array(fc.f, I->dup_dist, I->dup_dist.n, pin, st, "duplicate distances");
  put(fc.f, &IDX_TAIL, 8, "closing word");
  MM_REQUIRE(fflush(fc.f) == 0, MM_ERR_ARG, std::string("write to ") + path + " failed");
  MM_REQUIRE(fc.close() == 0, MM_ERR_ARG, std::string("closing ") + path + " failed: the index file is incomplete");
}

void index_load(mm_ctx* ctx, const char* path, mm_index* I) {
  hipStream_t st = ctx->stream;
  IdxFile fc; fc.f = fopen(path, "rb");
  MM_REQUIRE(fc.f != nullptr, MM_ERR_ARG, std::string("cannot open ") + path);
  setvbuf(fc.f, nullptr, _IONBF, 0);
  char magic[8]; uint32_t head[8]; int64_t dims[5];
  get(fc.f, magic, 8, "magic"); get(fc.f, head, sizeof head, "header"); get(fc.f, dims, sizeof dims, "header");
  MM_REQUIRE(memcmp(magic, "MMINDEX1", 8) == 0 && head[0] == IDX_VERSION, MM_ERR_ARG, std::string(path) + " is not an index file of this version");
  I->ctx = ctx;
  I->k = (int)head[1]; I->w = (int)head[2]; I->dir_shift = (int)head[3]; I->dup_sat = (int)head[4]; I->freq_threshold = (int)head[5]; I->tab_buckets = head[6];
  I->n_contigs = dims[0]; I->N = dims[1]; I->U = dims[2]; I->n_dup = dims[3];
  MM_REQUIRE(I->k >= 1 && I->k <= 64 && I->w >= 1 && I->w <= 4096 && I->n_contigs >= 0 && I->n_contigs < (1LL << 31) && I->N >= 0 && I->N < (1LL << 38) && I->U >= 0 && I->U <= I->N &&
             I->n_dup >= 0 && I->n_dup <= I->N && dims[4] >= 0 && dims[4] <= std::min<int64_t>(I->N + 1, 1 << 21) && I->dir_shift >= 1 && I->dir_shift < 30 &&
             I->dup_sat >= 1 && I->dup_sat <= 65535 && I->tab_buckets >= 1 && (I->N == 0 || I->tab_buckets == mm::tab_buckets_for(I->U)), MM_ERR_ARG, "corrupt index header");
  std::vector<int64_t> hh((size_t)dims[4] * 2);
  get(fc.f, hh.data(), hh.size() * 8, "histogram");
  I->hist.clear(); for (size_t i = 0; i + 1 < hh.size(); i += 2) I->hist[hh[i]] = hh[i + 1];
  I->contig_len.resize((size_t)I->n_contigs); I->h_cstart.resize((size_t)I->n_contigs + 1);
  get(fc.f, I->contig_len.data(), I->contig_len.size() * 4, "contig lengths");
  get(fc.f, I->h_cstart.data(), I->h_cstart.size() * 8, "contig entry ranges");
  MM_REQUIRE(I->h_cstart.front() == 0 && (int64_t)I->h_cstart.back() == I->N && std::is_sorted(I->h_cstart.begin(), I->h_cstart.end()), MM_ERR_ARG,
             "index file is inconsistent (contig entry ranges)");
  uint64_t dir_total = 0;                                          // what build_directory lays out for these contigs
  for (int32_t len : I->contig_len) { MM_REQUIRE(len >= 0, MM_ERR_ARG, "index file is inconsistent (contig lengths)"); dir_total += ((uint64_t)len >> I->dir_shift) + 2; }
  Pinned2 pin;
  // element counts as index_build leaves them: a few spare elements behind N entries / P <= N + 7 U padded occurrences, never fewer than the kernels read
  const uint64_t N = (uint64_t)I->N, U = (uint64_t)I->U, C = (uint64_t)I->n_contigs, nb64 = (N + 63) >> 6, SP = 64;
  get_array(fc.f, I->pos, 1, pin, st, N, N + SP, "entries");
  get_array(fc.f, I->cstart, 1, pin, st, C + 1, C + 1 + SP, "contig starts");
  get_array(fc.f, I->occ, 1, pin, st, N ? N : 1, N + 7 * U + SP, "occurrences");
  get_array(fc.f, I->occ16, 1, pin, st, I->occ.n, (uint64_t)I->occ.n + SP, "occurrence bins");
  get_array(fc.f, I->tab, 1, pin, st, (uint64_t)I->tab_buckets * 8, (uint64_t)I->tab_buckets * 8, "hash table");
  get_array(fc.f, I->dir, 1, pin, st, std::max<uint64_t>(dir_total, 1), std::max<uint64_t>(dir_total, 1), "position directory");
  get_array(fc.f, I->dir_off, 1, pin, st, C + 1, C + 1, "directory offsets");
  get_array(fc.f, I->dup_bits, 1, pin, st, std::max<uint64_t>(nb64, 1), nb64 + SP, "duplicate bits");
  get_array(fc.f, I->dup_rank, 1, pin, st, (uint64_t)I->dup_bits.n + (N ? 1 : 0), (uint64_t)I->dup_bits.n + SP, "duplicate ranks");
  get_array(fc.f, I->dup_dist, 1, pin, st, 1, N + SP, "duplicate distances");
  uint64_t tail = 0; get(fc.f, &tail, 8, "closing word");
  MM_REQUIRE(tail == IDX_TAIL, MM_ERR_ARG, "index file is inconsistent (closing word)");
  I->d_contig_len.alloc(std::max<size_t>(I->contig_len.size(), 1));
  I->d_contig_len.upload(I->contig_len.data(), I->contig_len.size(), st);
  MM_HIP(mm::stream_sync(st));
}

}  // namespace mm
