// K5 + K6 — L2 sliding MinHash window and strand vote, one wavefront per L1 candidate
// (replaces Map::computeL2MappedRegions + the statistics part of doL2Mapping, computeMap.hpp:396-538;
//  SlideMapper, slidingMap.hpp; MIIteratorL2::next, MIIteratorL2.hpp:74-96).
//
// Window sequence.  A window is the index range [b,e) with `sw_pos`; e is always the first entry with
// wpos > sw_pos+cnt-1 and wpos[b] <= sw_pos < wpos[b+1].  Whenever b has just advanced, sw_pos == wpos[b]
// and e == e_min(b) := lower_bound(wpos >= wpos[b]+cnt): every b is a re-entry point of the sequence.
//
// Exact skip-ahead (SKIP=true).  The reference evaluates ~1.16 windows per streamed index entry, each an
// ordered-map update.  Only (max shared, first and last window reaching it, the first optimal window) are
// outputs, so a window whose shared count provably stays below the best count found so far — or below the
// acceptance threshold accept_min(s), since a candidate whose maximum is smaller is dropped by the identity
// filter (computeMap.hpp:415) — cannot influence anything.  Upper bounds, per block of 64 consecutive b:
//     shared(W) <= m_all(W)  = matched entries in W (query hash present, with multiplicity)
//     if r0 + a(W) >= s  then pivot rank R(W) <= r0 and shared(W) <= m_lo(W) = matched entries of rank < r0,
//         where a(W) = window-only entries below Q[r0] that are first occurrences in their contig (DP clear),
//         a lower bound of the DISTINCT window-only hashes below Q[r0]  (mm_l2_core.hpp: R = min r with
//         r + #distinct window-only hashes below Q[r] >= s).
// All three are prefix-sum differences over per-entry class bits (ballot masks + word prefixes);
// m is taken over the largest window of the block, a over the smallest, so the bound holds for every window
// of the block.  r0 = expected pivot rank of the most promising block + 2.5 sigma (no probe; validity is checked per
// block).  Blocks that pass are evaluated exactly, re-entering at (b, e_min(b)) with the state rebuilt in parallel
// (LDS atomics; D and mt are order independent); the sweep starts near the largest bound and its trackers compare
// positions, so the visiting order is free.  Everything skipped is provably below the final maximum, so the result is
// bit-identical to the full slide (SKIP=false: the literal serial automaton, kept for reads shorter than w+k and as the
// cross-check in tests).
//
// Exact evaluation, 64 windows per round.  The window sequence is the merge of two sorted time lists (leave / enter);
// step indices come from cross-ranking two 64-entry chunks, window j is the state after j steps, and its shared count
// follows from prefix sums over per-entry indicators plus a per-lane binary search of the pivot inside a 64-rank
// "pivot zone" held in registers (block_slide below).  Pass A parks the rank code of every streamed entry in global
// memory (4 B per entry) for the rebuilds, the rounds and the vote.
//
// Launch shapes (mm_map.hip): reads are grouped by sketch size; the 10 kb class runs 4 candidates of a read per
// workgroup sharing the sketch in LDS, 8-bit gap counters, 80 VGPRs (24 waves per CU); longer reads get masks for
// 32 768 streamed entries and blocks of 2^j words.
#pragma once
#include "mm_index.hpp"
#include "mm_map.hpp"
#include "mm_l2_core.hpp"
#include <type_traits>

namespace mm {

// The skip path keeps one mask word per 64 streamed entries ("word"); NWQ = words / 64 is a template parameter
// (2: up to 8 192 streamed entries, reads up to ~18 kb at w=8; 8: 32 768 entries).  Bounds are taken per block of BW words,
// BW the smallest power of two with at most 128 blocks (two per lane).
constexpr int L2_NBLK_MAX = 128;
// bucket table over the top hash bits of the sketch: 1 024 buckets for the 10 kb class (sketches up to 3 072 hashes: at most four halving steps
// inside a bucket; 2 048 buckets were measured in round 2 and cost in LDS what they saved), 4 096 for the long-read classes (sketches up to
// 32 768: with 1 024 buckets their searches needed five steps and the generic loop, tools/l2_long_phases.py)
#ifndef L2_WAVES_10K
#define L2_WAVES_10K 6          // waves per SIMD the 10 kb class is compiled for: 6 = 80 registers, 15 of them spilled (round 5 measured 5 = 96 registers: tools/ab_build.sh)
#endif
#ifndef L2_TBITS_10K
#define L2_TBITS_10K 10
#endif
__host__ __device__ constexpr int l2_tbits(int nwq) { return nwq == 2 ? L2_TBITS_10K : 12; }
__host__ __device__ constexpr int l2_tsize(int nwq) { return (1 << l2_tbits(nwq)) + 1; }

// does [lo, hi) of pos[] hold hash h?  512 entries per step, the eight loads of a step in flight together: one load per step
// made every duplicate-flagged entry of a 50 kb window cost ~200 dependent memory latencies (two thirds of the rebuild time).
__device__ inline bool wave_has_hash(const Rec* __restrict__ pos, int64_t lo, int64_t hi, uint32_t h, int lane) {
  for (int64_t base = lo; base < hi; base += 512) {              // `base` is wave-uniform
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int64_t j = base + lane + 64 * i; x[i] = pos[j < hi ? j : lo].hash; }
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) hit |= (base + lane + 64 * i < hi) && x[i] == h;
    if (__ballot(hit) != 0ull) return true;
  }
  return false;
}
// The same question answered from the index's same-hash neighbour distances (mm_index.hpp: dup_bits / dup_rank / dup_dist) for an entry
// that carries the flag: g = entry number in pos[], the window bounds relative to it.  1 yes, 0 no, -1 when the stored distance is
// saturated (I.dup_sat = 65535; tests lower it) and the window reaches at least that far (the caller then scans).  Any lane may ask about its own entry.
__device__ inline uint32_t dup_word(const IndexView& I, int64_t g) {
  const uint64_t bits = I.dup_bits[g >> 6], r = I.dup_rank[g >> 6];
  return I.dup_dist[r + (uint64_t)__popcll(bits & ((1ull << (g & 63)) - 1ull))];
}
__device__ inline int dup_before(const IndexView& I, int64_t g, int64_t back /* entries of the window before g */) {
  const int64_t d = (int64_t)(dup_word(I, g) & 0xffffu);
  if (d < I.dup_sat) return d <= back ? 1 : 0;
  return back < I.dup_sat ? 0 : -1;
}
__device__ inline int dup_after(const IndexView& I, int64_t g, int64_t ahead /* entries of the window after g */) {
  const int64_t d = (int64_t)(dup_word(I, g) >> 16);
  if (d < I.dup_sat) return d <= ahead ? 1 : 0;
  return ahead < I.dup_sat ? 0 : -1;
}
// wave-uniform forms for the serial automata (x, lo, hi uniform): does [lo, x) / (x, hi) of pos[] hold the hash of the flagged entry x?
__device__ inline bool wave_dup_before(const IndexView& I, const Rec* __restrict__ pos, int64_t g0, int lo, int x, uint32_t h, int lane) {
  const int r = dup_before(I, g0 + x, (int64_t)x - lo);
  return r >= 0 ? r != 0 : wave_has_hash(pos, lo, x, h, lane);
}
__device__ inline bool wave_dup_after(const IndexView& I, const Rec* __restrict__ pos, int64_t g0, int x, int hi, uint32_t h, int lane) {
  const int r = dup_after(I, g0 + x, (int64_t)hi - 1 - x);
  return r >= 0 ? r != 0 : wave_has_hash(pos, x + 1, hi, h, lane);
}
// the results of the wave reductions are uniform; readfirstlane tells the compiler so (SGPRs, scalar branches)
__device__ inline int wave_sum(int v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64); return __builtin_amdgcn_readfirstlane(v); }
__device__ inline int wave_min(int v) { for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, 64)); return __builtin_amdgcn_readfirstlane(v); }
__device__ inline int wave_max(int v) { for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64)); return __builtin_amdgcn_readfirstlane(v); }
// wave64 prefix sum with DPP row shifts and row broadcasts: six adds, no LDS traffic (the __shfl_up form is 6 x (ds_bpermute
// + compare + add))
__device__ inline int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
  return v;
}
__device__ inline int wave_excl_scan(int v, int lane) { (void)lane; return wave_incl_scan(v) - v; }

// Wave-cooperative lower_bound over wpos in pos[lo,hi): 64 pivots per step (one dependent memory round trip narrows
// the range 64-fold) instead of one pivot per step.  All arguments wave-uniform; returns the first index whose
// wpos >= target (hi if none).
__device__ inline int64_t wave_lower_bound_wpos(const Rec* __restrict__ pos, int64_t lo, int64_t hi, int target, int lane) {
  while (hi - lo > 64) {
    const int64_t step = (hi - lo + 63) / 64;
    const int64_t idx = lo + (int64_t)lane * step;                // lane 0 probes lo itself
    const bool less = idx < hi && pw_wpos(pos[idx].pw) < target;
    const int p = __popcll(__ballot(less));                      // `less` is monotone over lanes: p leading lanes are below target
    if (p == 0) return lo;
    const int64_t nlo = lo + (int64_t)(p - 1) * step + 1;
    const int64_t nhi = min(hi, lo + (int64_t)p * step);
    lo = nlo; hi = nhi;
  }
  const int64_t idx = lo + lane;
  const bool less = idx < hi && pw_wpos(pos[idx].pw) < target;
  return lo + __popcll(__ballot(less));
}

// The same over a whole contig, entered through the index's position directory (mm_index.hpp): one directory read bounds the
// answer to one bucket (~128 entries, part of the range that is streamed afterwards anyway).
__device__ inline int64_t contig_lower_bound_wpos(const IndexView& I, int contig, int target, int lane) {
  const int64_t cbeg = (int64_t)I.cstart[contig];
  const uint64_t d0 = I.dir_off[contig], nb = I.dir_off[contig + 1] - d0 - 1;   // buckets 0 .. nb-1, entry nb = the contig's entry count
  const uint64_t b = min((uint64_t)max(target, 0) >> I.dir_shift, nb - 1);
  const int64_t lo = cbeg + (int64_t)I.dir[d0 + b], hi = cbeg + (int64_t)I.dir[d0 + b + 1];
  // entries before lo lie below b << shift <= target; the entry at hi (if any) lies at or beyond (b + 1) << shift > target, or
  // b is the last bucket and hi is the contig's end
  return wave_lower_bound_wpos(I.pos, lo, hi, target, lane);
}

// (l2_bucket, the bucket of a hash in the rank table T: mm_l2_core.hpp, where the host tests reach it)
// Rank of a hash in the sorted sketch Q (lower bound).  T[b] = first rank whose hash >= b << tshift, so the answer lies
// in a run of fewer than 2^steps elements starting at T[b]; because all of Q is sorted, the branch-free doubling search
// below needs no upper limit (elements behind the bucket are larger than h anyway; Q is padded with 16 x 0xffffffff).
// Four independent searches are interleaved so that one step costs one LDS latency for four entries.
constexpr int L2_QPAD = 16;
__device__ inline void l2_classify4(const uint32_t* __restrict__ Q, const uint16_t* __restrict__ T, int tshift, int steps, int s,
                                    const uint32_t (&h)[4], int (&code)[4]) {
  int lo0 = T[l2_bucket(h[0], tshift)], lo1 = T[l2_bucket(h[1], tshift)], lo2 = T[l2_bucket(h[2], tshift)], lo3 = T[l2_bucket(h[3], tshift)];
  if (steps <= 5) {
#define MM_L2_STEP(ST)                                                                                              \
    { const uint32_t v0 = Q[lo0 + ST - 1], v1 = Q[lo1 + ST - 1], v2 = Q[lo2 + ST - 1], v3 = Q[lo3 + ST - 1];         \
      lo0 += v0 < h[0] ? ST : 0; lo1 += v1 < h[1] ? ST : 0; lo2 += v2 < h[2] ? ST : 0; lo3 += v3 < h[3] ? ST : 0; }
    if (steps > 4) MM_L2_STEP(16)
    if (steps > 3) MM_L2_STEP(8)
    if (steps > 2) MM_L2_STEP(4)
    if (steps > 1) MM_L2_STEP(2)
    if (steps > 0) MM_L2_STEP(1)
#undef MM_L2_STEP
  } else {
    int hi0 = T[l2_bucket(h[0], tshift) + 1], hi1 = T[l2_bucket(h[1], tshift) + 1], hi2 = T[l2_bucket(h[2], tshift) + 1], hi3 = T[l2_bucket(h[3], tshift) + 1];
    for (int it = 0; it < steps; ++it) {
      const int m0 = min((lo0 + hi0) >> 1, s - 1), m1 = min((lo1 + hi1) >> 1, s - 1), m2 = min((lo2 + hi2) >> 1, s - 1), m3 = min((lo3 + hi3) >> 1, s - 1);
      const uint32_t v0 = Q[m0], v1 = Q[m1], v2 = Q[m2], v3 = Q[m3];
      if (lo0 < hi0) { if (v0 < h[0]) lo0 = m0 + 1; else hi0 = m0; }
      if (lo1 < hi1) { if (v1 < h[1]) lo1 = m1 + 1; else hi1 = m1; }
      if (lo2 < hi2) { if (v2 < h[2]) lo2 = m2 + 1; else hi2 = m2; }
      if (lo3 < hi3) { if (v3 < h[3]) lo3 = m3 + 1; else hi3 = m3; }
    }
  }
  const uint32_t e0 = Q[lo0], e1 = Q[lo1], e2 = Q[lo2], e3 = Q[lo3];   // lo <= s: the padding is readable
  code[0] = (lo0 < s && e0 == h[0]) ? lo0 : -(lo0 + 1);
  code[1] = (lo1 < s && e1 == h[1]) ? lo1 : -(lo1 + 1);
  code[2] = (lo2 < s && e2 == h[2]) ? lo2 : -(lo2 + 1);
  code[3] = (lo3 < s && e3 == h[3]) ? lo3 : -(lo3 + 1);
}
// Eight at a time (the streaming passes hold eight chunks in registers): twice the LDS reads in flight per step.
__device__ inline void l2_classify8(const uint32_t* __restrict__ Q, const uint16_t* __restrict__ T, int tshift, int steps, int s,
                                    const uint32_t (&h)[8], int (&code)[8]) {
  if (steps > 5) {
    uint32_t a[4], b[4]; int ca[4], cb[4];
    for (int i = 0; i < 4; ++i) { a[i] = h[i]; b[i] = h[4 + i]; }
    l2_classify4(Q, T, tshift, steps, s, a, ca); l2_classify4(Q, T, tshift, steps, s, b, cb);
    for (int i = 0; i < 4; ++i) { code[i] = ca[i]; code[4 + i] = cb[i]; }
    return;
  }
  int lo[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) lo[i] = T[l2_bucket(h[i], tshift)];
#define MM_L2_STEP8(ST)                                                                   \
  { uint32_t v[8];                                                                        \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) v[i] = Q[lo[i] + ST - 1];               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) lo[i] += v[i] < h[i] ? ST : 0; }
  if (steps > 4) MM_L2_STEP8(16)
  if (steps > 3) MM_L2_STEP8(8)
  if (steps > 2) MM_L2_STEP8(4)
  if (steps > 1) MM_L2_STEP8(2)
  if (steps > 0) MM_L2_STEP8(1)
#undef MM_L2_STEP8
  uint32_t ev[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ev[i] = Q[lo[i]];
#pragma unroll
  for (int i = 0; i < 8; ++i) code[i] = (lo[i] < s && ev[i] == h[i]) ? lo[i] : -(lo[i] + 1);
}
__device__ inline int l2_classify1(const uint32_t* __restrict__ Q, const uint16_t* __restrict__ T, int tshift, int steps, int s, uint32_t h) {
  int lo = T[l2_bucket(h, tshift)];
  if (steps <= 5) {
    if (steps > 4) lo += Q[lo + 15] < h ? 16 : 0;
    if (steps > 3) lo += Q[lo + 7] < h ? 8 : 0;
    if (steps > 2) lo += Q[lo + 3] < h ? 4 : 0;
    if (steps > 1) lo += Q[lo + 1] < h ? 2 : 0;
    if (steps > 0) lo += Q[lo] < h ? 1 : 0;
  } else {
    int hi = T[l2_bucket(h, tshift) + 1];
    for (int it = 0; it < steps; ++it) {
      const int m = min((lo + hi) >> 1, s - 1);
      const uint32_t v = Q[m];
      if (lo < hi) { if (v < h) lo = m + 1; else hi = m; }
    }
  }
  return (lo < s && Q[lo] == h) ? lo : -(lo + 1);
}

// LDS layout: Q[smax] (shared by the waves of a workgroup) | per wave: D[smax] | mt | skip-ahead class arrays | slide scratch
constexpr int L2_SCRATCH_BYTES = 64 * 4 + 64 + 64;             // step times, step-has-deletion, step-has-addition
__host__ __device__ inline size_t l2_skip_bytes(int nwq) { return (((size_t)(64 * nwq + 1) * (3 * 8 + 3 * 2)) + 15) & ~(size_t)15; }
constexpr int L2_HBUCKETS = 1024;                              // coarse gap histogram of a rebuild (16-bit counters)
constexpr int L2_SKETCH_LIMIT = 32 * L2_HBUCKETS;              // sketch sizes below this keep a histogram bucket narrower than the 64-rank zone
template <typename DT>
__host__ __device__ inline size_t l2_wave_bytes(int smax, bool skip, int nwq) {
  if (skip && nwq > 2) return ((size_t)L2_HBUCKETS * 2 + L2_SCRATCH_BYTES + 15) & ~(size_t)15;   // long-read classes: histogram | slide scratch
  if (skip) {                                                    // 10 kb class: D[smax] | mt | slide scratch
    size_t b = (((size_t)smax * sizeof(DT) + 3) & ~(size_t)3) + (size_t)((smax + 31) / 32) * 4;
    return (((b + 15) & ~(size_t)15) + L2_SCRATCH_BYTES + 15) & ~(size_t)15;
  }
  size_t b = (((size_t)smax * sizeof(DT) + 3) & ~(size_t)3) + (size_t)((smax + 31) / 32) * 4;   // full slide: D[smax] | mt
  return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t l2_qpart_bytes(int smax) { return ((size_t)(smax + L2_QPAD) * 4 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t l2_tpart_bytes(int nwq) { return ((size_t)l2_tsize(nwq) * 2 + 4 + 15) & ~(size_t)15; }
// sketch | search table | strand bits of the sketch (two 64-bit words per 64 ranks: strand, unresolved duplicate)
__host__ __device__ inline size_t l2_q_bytes(int smax, int nwq) { return l2_qpart_bytes(smax) + l2_tpart_bytes(nwq) + (size_t)((smax + 63) / 64) * 16; }
template <typename DT>
inline size_t l2_lds_bytes(int smax, bool skip, int waves, int nwq) { return l2_q_bytes(smax, nwq) + (size_t)waves * l2_wave_bytes<DT>(smax, skip, nwq); }

// visibility of a wave's own LDS writes to its other lanes (workgroups may hold several independent waves)
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// SKIP: exact skip-ahead on/off.  DT: counter width of D (uint8_t compact / uint16_t wide).  WAVES: candidates per
// workgroup; with WAVES > 1 the waves of a workgroup map candidates of ONE read and share its sketch Q in LDS.
template <bool SKIP, typename DT, int WAVES, int NWQ>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(NWQ == 2 ? L2_WAVES_10K : 2))) l2_kernel(IndexView I, const int32_t* __restrict__ cand, const int32_t* __restrict__ cand_read,
                                                const uint32_t* __restrict__ sk_hash, const uint8_t* __restrict__ sk_strand,
                                                const uint64_t* __restrict__ mz_off, const int32_t* __restrict__ sk_n,
                                                const int32_t* __restrict__ read_len, const int32_t* __restrict__ accept_min,
                                                int k, int w, int smax, L2Result* __restrict__ out,
                                                unsigned long long* __restrict__ counters /* [0] streamed entries, [1] evaluated windows, [2] rebuilds */,
                                                const int32_t* __restrict__ grp_cand0 /* WAVES>1: first candidate of the group */,
                                                const int32_t* __restrict__ grp_n /* WAVES>1: candidates in the group */,
                                                const int32_t* __restrict__ cand_list /* WAVES==1: optional indirection (fallback runs) */,
                                                int32_t* __restrict__ ovf_list, unsigned int* __restrict__ ovf_n,
                                                uint8_t* __restrict__ amb_used /* optional: set per read when a vote read an unresolved strand */,
                                                void* __restrict__ code_buf /* optional: 64*64*NWQ code words (2 bytes each for NWQ == 2, else 4) per scratch slot */,
                                                uint8_t* __restrict__ mask_buf /* NWQ > 2: l2_skip_bytes(NWQ) per scratch slot */,
                                                unsigned int* __restrict__ slot_flags /* n_slots (a multiple of 8) words, all 0 between launches: a wave takes a free slot of its XCD's share for its lifetime (null: slot = wave number in the launch) */,
                                                int n_slots) {
  extern __shared__ __align__(16) uint32_t lds[];
  uint32_t* Q = lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* wbase = (uint8_t*)lds + l2_q_bytes(smax, NWQ) + (size_t)wave * l2_wave_bytes<DT>(smax, SKIP, NWQ);
  DT* D = (DT*)wbase;                                            // (full slide only; the skip kernels keep a histogram here)
  uint32_t* mt = (uint32_t*)(wbase + (((size_t)smax * sizeof(DT) + 3) & ~(size_t)3));
  const int64_t c0 = WAVES > 1 ? (int64_t)grp_cand0[blockIdx.x] : (cand_list ? (int64_t)cand_list[blockIdx.x] : (int64_t)blockIdx.x);
  const int r = cand_read[c0];                                   // every wave of the workgroup serves this read
  const int s = sk_n[r];
  const uint64_t qo = mz_off[r];
  const int len = read_len[r];
  uint16_t* T = (uint16_t*)((uint8_t*)lds + l2_qpart_bytes(smax));
  constexpr int TBITS = l2_tbits(NWQ), tshift = 32 - TBITS;
  int* tmaxp = (int*)(T + ((l2_tsize(NWQ) + 1) & ~1));
  const int dbg_early = (int)counters[11] & 0xff;
  auto early_out = [&]() { if (!(WAVES > 1 && wave >= grp_n[blockIdx.x]) && lane == 0) { L2Result z{}; out[c0 + (WAVES > 1 ? wave : 0)] = z; } };
  if (dbg_early == 6) { early_out(); return; }
  for (int i = threadIdx.x; i < s; i += 64 * WAVES) Q[i] = sk_hash[qo + i];
  if (dbg_early == 7) { __syncthreads(); early_out(); return; }
  // strand byte of every sketch entry (bit 0 strand, bit 1 unresolved duplicate: mm_map.hip, K2) as two bit sets: the vote
  // looks them up per matched entry, and a global load there is a second dependent memory latency in every step
  uint64_t* const SB = (uint64_t*)((uint8_t*)lds + l2_qpart_bytes(smax) + l2_tpart_bytes(NWQ));
  for (int i0 = 64 * wave; i0 < s; i0 += 64 * WAVES) {
    const uint8_t sbyte = i0 + lane < s ? sk_strand[qo + i0 + lane] : (uint8_t)0;
    const uint64_t b0 = __ballot(sbyte & 1), b1 = __ballot(sbyte & 2);
    if (lane == 0) { SB[2 * (i0 >> 6)] = b0; SB[2 * (i0 >> 6) + 1] = b1; }
  }
  if (threadIdx.x < L2_QPAD) Q[s + threadIdx.x] = 0xffffffffu;
  if (threadIdx.x == 0) *tmaxp = 0;
  __syncthreads();
  // T[b] = first rank whose bucket (l2_bucket) is >= b: element i is the answer for the buckets after Q[i-1]'s up to its own
  // (i == s closes the table), so every T entry is written exactly once, without searching
  for (int i = threadIdx.x; i <= s; i += 64 * WAVES) {
    const int lo = i ? l2_bucket(Q[i - 1], tshift) + 1 : 0;
    const int hi = i < s ? l2_bucket(Q[i], tshift) : (1 << TBITS);
    for (int bb = lo; bb <= hi; ++bb) T[bb] = (uint16_t)i;
  }
  __syncthreads();
  {                                                              // longest bucket: wave maximum first, one LDS atomic per wave
    int tm = 0;
    for (int bkt = threadIdx.x; bkt < (1 << TBITS); bkt += 64 * WAVES) tm = max(tm, (int)T[bkt + 1] - (int)T[bkt]);
    tm = wave_max(tm);
    if ((threadIdx.x & 63) == 0) atomicMax(tmaxp, tm);
  }
  __syncthreads();
  const int tsteps = *tmaxp ? 32 - __clz(*tmaxp) : 0;
  const int dbg_flags = (int)counters[11];                       // timing aids: low byte MM_L2_STOP, bit 8 MM_L2_PHASES; 0 in normal operation
  const int dbg_stop = dbg_flags & 0xff;
  if (WAVES > 1 && wave >= grp_n[blockIdx.x]) return;
  const int64_t c = c0 + (WAVES > 1 ? wave : 0);
  if (dbg_stop) { if (lane == 0) { L2Result z{}; out[c] = z; } if (dbg_stop == 1) return; }
  constexpr int DPER = 4 / (int)sizeof(DT);                      // counters per 32-bit word
  // Pass A classifies every streamed entry once; its result (rank / gap code in the low 16 bits, strand and duplicate
  // flags above) is parked in global memory, 4 bytes per entry, and read back by the rebuilds, the slide rounds and the
  // vote instead of searching the sketch again.
  // 10 kb class (NWQ == 2, sketch <= 3072): 13-bit code + 3 flag bits in 16 bits; other classes: 16-bit code + flags in 32 bits
  using CW = std::conditional_t<NWQ == 2, uint16_t, uint32_t>;
  // The code words and class masks of a candidate live in a scratch slot for as long as its wave does.  A launch has hundreds of
  // thousands of waves and a few thousand of them are resident at a time: the slots are taken and given back (one flag word each),
  // so the scratch of a launch is what its resident waves need — a few hundred MB that stay in the last-level cache between the pass
  // that writes the words and the passes that read them — instead of 16-128 KB for every wave of the launch (7-20 GB per read batch).
  // The L2s of the XCDs are not coherent with each other (a line one holder left dirty in its XCD's L2 could be written back over
  // the next holder's data), so the slots are split by XCD — a wave takes one from the share of the XCD it runs on
  // (HW_REG_XCC_ID) and all traffic of a slot goes through ONE L2 for the whole launch.  Within an XCD the hand-over needs: the
  // holder's stores landed in L2 before the flag falls (s_waitcnt vmcnt(0): the L1 is write-through), and nothing from the new
  // holder — it writes every word before it reads it, and a CU's L1 follows its own stores.
  CW* cw = nullptr;
  int slot = -1;
  auto acquire_slot = [&]() {
    unsigned int sidx = WAVES > 1 ? blockIdx.x * WAVES + wave : blockIdx.x;
    if (slot_flags) {
      const unsigned int per = (unsigned int)n_slots >> 3;        // slots of one XCD (n_slots is a multiple of 8)
      const unsigned int lo = ((unsigned int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) * per;   // hwreg(HW_REG_XCC_ID, 0, 4)
      unsigned int t = (sidx * 2654435761u >> 7) % per;            // (scattered start: workgroup b runs on XCD b % 8, the plain remainder would use an eighth of the share's starts)
      if (lane == 0) while (atomicCAS(&slot_flags[lo + t], 0u, 1u) != 0u) t = t + 1u == per ? 0u : t + 1u;
      sidx = lo + (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    }
    slot = (int)sidx;
    if (code_buf) cw = (CW*)code_buf + (size_t)sidx * (size_t)(64 * 64 * NWQ);
  };
  auto release_slot = [&]() {                                    // (every read of the slot has returned: the wave used the values)
    if (slot_flags && slot >= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) atomicExch(&slot_flags[slot], 0u);
    }
    slot = -1;
  };

  bool have_codes = false;                                       // set once pass A has run
  auto cw_make = [](int code, uint32_t flags) -> CW {
    if constexpr (NWQ == 2) return (CW)(((uint32_t)code & 0x1fffu) | (flags << 13));
    else return (CW)((uint32_t)(uint16_t)(int16_t)code | (flags << 16));
  };
  auto cw_code = [](CW ew) -> int {
    if constexpr (NWQ == 2) return ((int)((uint32_t)ew << 19)) >> 19;
    else return (int)(int16_t)(uint16_t)((uint32_t)ew & 0xffffu);
  };
  auto cw_flags = [](CW ew) -> uint32_t { if constexpr (NWQ == 2) return (uint32_t)ew >> 13; else return (uint32_t)ew >> 16; };

  const int contig = cand[3 * c], rs = cand[3 * c + 1], re = cand[3 * c + 2];
  const int cnt = len - (w - 1) - (k - 1);                       // computeMap.hpp:470
  if (SKIP && cnt < 2) {                                         // reads shorter than w+k: left to the literal full slide (host launches it on this list)
    if (lane == 0) { L2Result z{}; out[c] = z; ovf_list[atomicAdd(ovf_n, 1u)] = (int32_t)c; }
    return;
  }
  // all window arithmetic below is 32-bit and relative to the first streamed entry of this candidate
  const int64_t first0 = contig_lower_bound_wpos(I, contig, rs, lane);                // searchIndex, :466
  const int64_t last0 = max(first0, contig_lower_bound_wpos(I, contig, re + len, lane));   // :477
  const Rec* __restrict__ pos = I.pos + first0;
  const int first = 0;
  const int last_end = (int)(last0 - first0);
  const int nmax = (int)min((int64_t)0x7fffffff, I.N - 1 - first0);
  int amin = accept_min[r]; if (amin < 1) amin = 1;
  if (dbg_stop == 3) return;                                     // (timing aid: the two range searches alone)

  L2StateT<DT> S{Q, D, mt, s, 0, 0, 0, 0};
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};                // phase clocks: setup, passA, bounds, rebuild, slide, passB, vote, -
  long long tmark = clock64();
  auto lap = [&](int ph) { long long now = clock64(); tph[ph] += now - tmark; tmark = now; };

  // e_min(b): first entry at or after b whose wpos >= wpos[b] + cnt (never beyond last_end).  Wave-uniform argument.
  auto e_min = [&](int bb) -> int {
    const int target = pw_wpos(pos[bb].pw) + cnt;
    return (int)wave_lower_bound_wpos(pos, bb, last_end, target, lane);
  };
  // ---- register-resident chunks of 64 consecutive entries at both window ends -------------------------
  int baseB = first, baseE = first;
  Rec rb = pos[min(baseB + lane, nmax)], rE = rb;
  int codeB = l2_classify1(Q, T, tshift, tsteps, s, rb.hash), codeE = codeB;
  auto loadB = [&](int nb) { baseB = nb; rb = pos[min(nb + lane, nmax)]; codeB = l2_classify1(Q, T, tshift, tsteps, s, rb.hash); };
  auto loadE = [&](int ne) { baseE = ne; rE = pos[min(ne + lane, nmax)]; codeE = l2_classify1(Q, T, tshift, tsteps, s, rE.hash); };

  int b = first, e = first;
  int sw_pos = 0;
  auto add_entry = [&](int x) {                              // slidingMap.hpp:139-160
    if (x - baseE >= 64 || x < baseE) loadE(x);
    const int ln = (int)(x - baseE);
    const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)rE.hash, ln);
    const uint32_t pw = (uint32_t)__builtin_amdgcn_readlane((int)rE.pw, ln);
    const int code = __builtin_amdgcn_readlane(codeE, ln);
    if (code == -(s + 1)) return;                                // above every query hash: never counted
    if ((pw & PW_DP) && wave_dup_before(I, pos, first0, b, x, h, lane)) return;   // REV: hash already in the window
    if (code >= 0) l2_add_matched(S, code); else l2_add_wonly(S, -code - 1);
  };
  auto del_entry = [&](int x, int wend) {                // slidingMap.hpp:170-214
    const int ln = (int)(x - baseB);
    const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)rb.hash, ln);
    const uint32_t pw = (uint32_t)__builtin_amdgcn_readlane((int)rb.pw, ln);
    const int code = __builtin_amdgcn_readlane(codeB, ln);
    if (code == -(s + 1)) return;
    if ((pw & PW_DN) && wave_dup_after(I, pos, first0, x, wend, h, lane)) return;   // NOOP: a later occurrence stays
    if (code >= 0) l2_del_matched(S, code); else l2_del_wonly(S, -code - 1);
  };

  // ---- state of window [nb,ne) from scratch, all lanes (the arrays are order independent) ---------------
  unsigned long long rebuilds = 0, rounds = 0;
  // Pivot zone (SKIP path).  The serial slide keeps no LDS state at all: lane l owns rank r = z0 + l and holds
  //   fz = r + D[z0] + ... + D[r]                  (VGPR; huge for r >= s so that lane r == s acts as the "R = s" sentinel)
  // next to the wave-uniform scalars cbase = D[0] + ... + D[z0-1], sb = matched ranks below z0 present in the window and
  // pm = presence mask of the zone's ranks.  Then R = z0 + ctz(ballot(fz >= s - cbase)) and
  // shared = sb + popcount(pm below R): one compare + scalar bit operations per window instead of LDS round trips.
  // An event outside the zone only touches a scalar; inside it is one predicated vector add.  When the pivot reaches a
  // zone edge the state is rebuilt from the window's entries, re-centred on the new pivot.
  int z0 = 0, cbase = 0, sb = 0, fz = 0;
  uint64_t pm = 0;
  // 10 kb class (NWQ == 2): one gap counter per rank in LDS (D, packed 8-bit) and the matched bitmap, filled with LDS atomics
  int overflow = 0;
  constexpr int DBITS = 8 * (int)sizeof(DT);
  auto rebuild_dense = [&](int nb, int ne) __attribute__((always_inline)) {
    ++rebuilds;
    uint32_t* Dw = (uint32_t*)D;
    for (int i = lane; i < (s + DPER - 1) / DPER; i += 64) Dw[i] = 0;
    for (int i = lane; i < (s + 31) / 32; i += 64) mt[i] = 0;
    wave_sync();
    // packed counter += 1, fire and forget.  A saturated counter carries into its neighbour, which lowers the sum of
    // all counters by DMAX per carry: comparing that sum with the number of increments detects it exactly.
    int n_inc = 0;
    auto d_inc = [&](int g) {
      atomicAdd(&Dw[g / DPER], 1u << (DBITS * (g % DPER)));
      ++n_inc;
    };
    for (int base = nb; base < ne; base += 512) {                // eight loads in flight per wait
      int cd[8]; uint32_t fl[8];
      if (have_codes) {
        const CW* __restrict__ pc = cw + (base - first) + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const CW ew = pc[min(64 * i, 64 * 64 * NWQ - 1 - (base - first) - lane)]; cd[i] = cw_code(ew); fl[i] = cw_flags(ew); }
      } else {
        Rec x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int j = base + lane + 64 * i; x[i] = pos[min(j, nmax)]; }
        uint32_t hh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { hh[i] = x[i].hash; fl[i] = x[i].pw & 7u; }
        l2_classify8(Q, T, tshift, tsteps, s, hh, cd);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = base + lane + 64 * i;
        if (base + 64 * i >= ne) continue;
        const int code = cd[i];
        const int g = -code - 1;
        const bool in = j < ne;
        if (in && code >= 0) atomicOr(&mt[code >> 5], 1u << (code & 31));
        const bool wonly = in && code < 0 && g < s;
        const bool flagged = wonly && (fl[i] & PW_DP);            // an earlier occurrence exists in the contig: inside the window?
        const int dres = flagged ? dup_before(I, first0 + j, (int64_t)j - nb) : 0;
        if (wonly && dres == 0) d_inc(g);
        uint64_t fm = __ballot(dres < 0);                        // (windows of 65535+ entries only: a wave-wide scan each)
        while (fm) {
          const int l = __ffsll((unsigned long long)fm) - 1;
          fm &= fm - 1;
          const uint32_t hj = pos[base + l + 64 * i].hash;
          const bool dup = wave_has_hash(pos, nb, base + l + 64 * i, hj, lane);
          if (!dup && lane == l) d_inc(g);
        }
      }
    }
    wave_sync();
    // pivot: R = min r with r + C(r) >= s.  Lane l sums the counters of its share of D word-wise; the first lane whose
    // last rank satisfies the condition holds the pivot, and its share is resolved rank by rank with one wave scan.
    const int NW = (s + DPER - 1) / DPER, wch = (NW + 63) / 64, rch = wch * DPER;
    const int r_lo = min(lane * rch, s), r_hi = min(r_lo + rch, s);
    int local = 0;
    for (int wi = 0; wi < wch; ++wi) {
      const int wd = lane * wch + wi;
      const uint32_t v = wd < NW ? Dw[wd] : 0u;
      local += sizeof(DT) == 1 ? (int)__builtin_amdgcn_sad_u8(v, 0u, 0u) : (int)((v & 0xffffu) + (v >> 16));
    }
    const int basec = wave_excl_scan(local, lane);
    const int total = __builtin_amdgcn_readlane(basec, 63) + __builtin_amdgcn_readlane(local, 63);
    if (wave_sum(n_inc) != total) overflow = 1;
    const uint64_t hm = __ballot(r_lo < s && r_hi - 1 + basec + local >= s);
    int R = s, cb = total;
    if (hm) {
      const int P = __builtin_ctzll(hm), rP = P * rch;
      int acc = __builtin_amdgcn_readlane(basec, P);
      for (int t0 = 0; t0 < rch; t0 += 64) {
        const int r = rP + t0 + lane;
        const bool in = r < s && t0 + lane < rch;
        const int d = in ? (int)D[r] : 0;
        const int ex = wave_excl_scan(d, lane);
        const uint64_t m = __ballot(in && r + acc + ex + d >= s);
        if (m) { const int l = __builtin_ctzll(m); R = rP + t0 + l; cb = acc + __builtin_amdgcn_readlane(ex, l); break; }
        acc += __builtin_amdgcn_readlane(ex, 63) + __builtin_amdgcn_readlane(d, 63);
      }
    }
    z0 = max(0, min(R - 32, s - 63));
    const int rz = z0 + lane;
    const int dz = rz < s ? (int)D[rz] : 0;
    const int exz = wave_excl_scan(dz, lane);
    fz = rz < s ? rz + exz + dz : (1 << 29);
    cbase = cb - __builtin_amdgcn_readlane(exz, R - z0);          // D[0..z0) = D[0..R) - D[z0..R)
    int sbl = 0;
    for (int wd = lane; wd * 32 < z0; wd += 64) {
      uint32_t m = mt[wd];
      const int rem = z0 - wd * 32;
      if (rem < 32) m &= (1u << rem) - 1u;
      sbl += __popc(m);
    }
    sb = wave_sum(sbl);
    pm = __ballot(rz < s && ((mt[rz >> 5] >> (rz & 31)) & 1u));
  };
  // State of window [nb, ne) from scratch, in two passes over its entries (their rank codes come from pass A's parked
  // words, or from a search where pass A did not run):
  //   1. a coarse histogram of the distinct window-only hashes by gap (L2_HBUCKETS 16-bit LDS counters, fire and
  //      forget); its prefix sums locate the bucket of the pivot R = min r with r + C(r) >= s;
  //   2. with the zone centred on that bucket: window-only hashes between the bucket-aligned zone start and z0 (-> cbase),
  //      matched ranks below z0 (-> sb), and the zone's own entries (-> fz, pm).
  // Duplicates inside the window (entries flagged DP) are resolved one by one as everywhere else.
  int bsh = 0;
  while ((s >> bsh) >= L2_HBUCKETS) ++bsh;
  auto rebuild_hist = [&](int nb, int ne) __attribute__((always_inline)) {
    ++rebuilds;
    uint32_t* Hw = (uint32_t*)wbase;                             // two 16-bit counters per word
    for (int i = lane; i < L2_HBUCKETS / 2; i += 64) Hw[i] = 0;
    wave_sync();
    auto fetch8 = [&](int base, int (&cd)[8], uint32_t (&fl)[8]) {
      if (have_codes) {
        const CW* __restrict__ pc = cw + (base - first) + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const CW ew = pc[min(64 * i, 64 * 64 * NWQ - 1 - (base - first) - lane)]; cd[i] = cw_code(ew); fl[i] = cw_flags(ew); }
      } else {
        Rec x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int j = base + lane + 64 * i; x[i] = pos[min(j, nmax)]; }
        uint32_t hh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { hh[i] = x[i].hash; fl[i] = x[i].pw & 7u; }
        l2_classify8(Q, T, tshift, tsteps, s, hh, cd);
      }
    };
    // an entry flagged DP counts only if no earlier occurrence of its hash lies inside the window
    auto first_in_window = [&](int j) -> bool { return !wave_has_hash(pos, nb, j, pos[j].hash, lane); };
    // the parked code words of [nb, ne), 512 entries per step; the loads of the next step are issued before the current one is
    // processed (the long-read classes keep only 8-12 waves per CU: nothing else would cover the latency of each step)
    auto stream_codes = [&](auto&& body) __attribute__((always_inline)) {
      if (have_codes) {
        auto load_raw = [&](int base, CW (&raw)[8]) {
          const CW* __restrict__ pc = cw + (base - first) + lane;
#pragma unroll
          for (int i = 0; i < 8; ++i) raw[i] = pc[min(64 * i, 64 * 64 * NWQ - 1 - (base - first) - lane)];
        };
        CW cur[8], nxt[8];
        load_raw(nb, cur);
        for (int base = nb; base < ne; base += 512) {
          if (base + 512 < ne) load_raw(base + 512, nxt);
          int cd[8]; uint32_t fl[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) { cd[i] = cw_code(cur[i]); fl[i] = cw_flags(cur[i]); }
          body(base, cd, fl);
#pragma unroll
          for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
        }
      } else {
        for (int base = nb; base < ne; base += 512) { int cd[8]; uint32_t fl[8]; fetch8(base, cd, fl); body(base, cd, fl); }
      }
    };
    stream_codes([&](int base, int (&cd)[8], uint32_t (&fl)[8]) __attribute__((always_inline)) {   // pass 1
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = base + lane + 64 * i;
        if (base + 64 * i >= ne) continue;
        const int g = -cd[i] - 1;
        const bool wonly = j < ne && cd[i] < 0 && g < s;
        const bool flagged = wonly && (fl[i] & PW_DP);
        const int dres = flagged ? dup_before(I, first0 + j, (int64_t)j - nb) : 0;
        bool count_it = wonly && dres == 0;
        uint64_t fm = __ballot(dres < 0);
        while (fm) {
          const int l = __builtin_ctzll(fm); fm &= fm - 1;
          const bool ok = first_in_window(base + l + 64 * i);
          if (ok && lane == l) count_it = true;
        }
        if (count_it) { const int bq = g >> bsh; atomicAdd(&Hw[bq >> 1], 1u << (16 * (bq & 1))); }
      }
    });
    wave_sync();
    // prefix sums over the buckets (16 per lane); the first bucket whose last rank satisfies r + C(r) >= s holds the pivot
    constexpr int BPL = L2_HBUCKETS / 64;
    int hv[BPL], hsum = 0;
#pragma unroll
    for (int t = 0; t < BPL / 2; ++t) { const uint32_t v = Hw[lane * (BPL / 2) + t]; hv[2 * t] = (int)(v & 0xffffu); hv[2 * t + 1] = (int)(v >> 16); hsum += hv[2 * t] + hv[2 * t + 1]; }
    int run = wave_excl_scan(hsum, lane);
    int myJ = 1 << 30;
    wave_sync();                                                 // all lanes have read their counters: the words now take the exclusive prefixes
#pragma unroll
    for (int t = 0; t < BPL; ++t) {
      const int bq = lane * BPL + t;
      const int rend = min(((bq + 1) << bsh) - 1, s - 1);        // last rank of the bucket
      if (t & 1) Hw[lane * (BPL / 2) + (t >> 1)] = (uint32_t)(uint16_t)(run - hv[t - 1]) | ((uint32_t)(uint16_t)run << 16);
      run += hv[t];
      if (myJ == (1 << 30) && (bq << bsh) < s && rend + run >= s) myJ = bq;
    }
    const int jst = wave_min(myJ);
    wave_sync();
    const int r_est = jst == (1 << 30) ? s : min(s, (jst << bsh) + ((1 << bsh) >> 1));
    z0 = max(0, min(r_est - 32, s - 63));
    const int zb = (z0 >> bsh) << bsh;                           // bucket-aligned rank at or below z0
    const uint32_t pw_ = Hw[(z0 >> bsh) >> 1];
    cbase = (int)(((z0 >> bsh) & 1) ? (pw_ >> 16) : (pw_ & 0xffffu));   // distinct window-only hashes in the buckets below
    const int rz = z0 + lane;
    fz = rz < s ? rz : (1 << 29);
    sb = 0; pm = 0;
    stream_codes([&](int base, int (&cd)[8], uint32_t (&fl)[8]) __attribute__((always_inline)) {   // pass 2
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = base + lane + 64 * i;
        if (base + 64 * i >= ne) continue;
        const int code = cd[i], g = -code - 1;
        const bool in = j < ne;
        const bool wonly = in && code < 0 && g < s;
        bool part_w = wonly && g >= zb && g < z0, zone_w = wonly && g >= z0 && g < z0 + 64;
        bool low_m = in && code >= 0 && code < z0, zone_m = in && code >= z0 && code < z0 + 64;
        const int dres = ((part_w || zone_w || low_m) && (fl[i] & PW_DP)) ? dup_before(I, first0 + j, (int64_t)j - nb) : 0;
        if (dres > 0) { part_w = false; zone_w = false; low_m = false; }
        uint64_t fm = __ballot(dres < 0);
        while (fm) {
          const int l = __builtin_ctzll(fm); fm &= fm - 1;
          const bool ok = first_in_window(base + l + 64 * i);
          if (!ok && lane == l) { part_w = false; zone_w = false; low_m = false; }
        }
        cbase += __popcll(__ballot(part_w));
        sb += __popcll(__ballot(low_m));
        uint64_t zm = __ballot(zone_m);
        while (zm) { const int l = __builtin_ctzll(zm); zm &= zm - 1; pm |= 1ull << (__builtin_amdgcn_readlane(code, l) - z0); }
        uint64_t zw = __ballot(zone_w);
        while (zw) { const int l = __builtin_ctzll(zw); zw &= zw - 1; const int gg = __builtin_amdgcn_readlane(g, l) - z0; fz += (lane >= gg) ? 1 : 0; }
      }
    });
  };

  auto rebuild_state = [&](int nb, int ne) __attribute__((always_inline)) {
    if constexpr (NWQ == 2) rebuild_dense(nb, ne); else rebuild_hist(nb, ne);
  };

  // ---- the reference's loop body (computeMap.hpp:496-533): evaluate [b,e), then MIIteratorL2::next ------
  int best = 0, bestR = 0, beg_pos = 0, last_pos = 0;
  int opt_b = 0, opt_e = 0, last_b = 0;
  unsigned long long evals = 0;
  // slides while e < last_end and b < b_stop; TRACK=false only records the maximum (for the bound), it does
  // not touch the reference-visible trackers
  // ---- 64 consecutive windows per round, one lane per window (SKIP path) ---------------------------------
  // The window sequence is the merge of two sorted time lists: entry b+i leaves when sw_pos reaches A_i = wpos[b+i+1],
  // entry e+i enters at B_i = wpos[e+i]-(cnt-1); equal times are one step (deletion, then addition).  Step indices by
  // cross-ranking the two lists (wave-wide binary searches through ds_bpermute), window j = state after the first j steps
  // = [b+d_j, e+a_j).  Its pivot-zone state follows from prefix sums over the per-entry indicators: cbase_j and sb_j by
  // one packed scan per side, then R_j by a per-lane binary search over the zone's fz (rank-indexed lanes, read with
  // ds_bpermute) and shared_j = sb_j + popcount(pm_j below R_j).  Events inside the zone are rare and applied one by one.
  // Only times up to min(A_63, B_63) are certain (later entries of the other list could interleave), the rest of the
  // chunk is redone by the next round.
  int* tst = (int*)(wbase + (NWQ > 2 ? (size_t)L2_HBUCKETS * 2 : (((((size_t)smax * sizeof(DT) + 3) & ~(size_t)3) + (size_t)((smax + 31) / 32) * 4 + 15) & ~(size_t)15)));
  uint8_t* fdel = (uint8_t*)(tst + 64);
  uint8_t* fadd = fdel + 64;
  auto rank_search = [&](int arr, int v) -> int {                // number of leading lanes whose (ascending) arr < v
    int lo = 0;
    for (int st = 32; st >= 1; st >>= 1) { const int x = __shfl(arr, lo + st - 1, 64); if (x < v) lo += st; }
    const int x = __shfl(arr, lo, 64);
    return lo + (x < v ? 1 : 0);
  };
  bool pending_rebuild = false;
  // sweep state (phase 1): the slide runs on across block boundaries as long as the next block's bound still passes
  int ub2[2] = {-1, -1};
  int bk = 0, j0 = 0, stage = 0, blk_end = 0x7fffffff, nblk_s = 0, bspan = 64;
  bool run_stop = false;
  auto lane2 = [&](const int (&v)[2], int k) -> int { return __builtin_amdgcn_readlane(k < 64 ? v[0] : v[1], k & 63); };
  auto block_slide = [&](int b_stop) __attribute__((always_inline)) {
    constexpr int INF = 0x7fffffff;
    while (e < last_end && b < b_stop) {
      if (pending_rebuild) { if (dbg_flags & 0x100) lap(4); rebuild_state(b, e); pending_rebuild = false; if (dbg_flags & 0x100) lap(3); if (dbg_stop == 8) break; }   // the only instance of the rebuild code
      if (dbg_stop == 9 && rounds >= 1) break;
      ++rounds;
      const Rec xb = pos[min(b + lane, nmax)];
      const Rec xe = pos[min(e + lane, nmax)];
      const int w64 = pw_wpos(pos[min(b + 64, nmax)].pw);
      int cB, cE;
      if (have_codes) { cB = cw_code(cw[min(b - first + lane, 64 * 64 * NWQ - 1)]); cE = cw_code(cw[min(e - first + lane, 64 * 64 * NWQ - 1)]); }
      else { cB = l2_classify1(Q, T, tshift, tsteps, s, xb.hash); cE = l2_classify1(Q, T, tshift, tsteps, s, xe.hash); }
      const int wpb = pw_wpos(xb.pw);
      int nextw = __shfl_down(wpb, 1, 64);
      if (lane == 63) nextw = w64;
      const int tA = (b + lane + 1 < last_end) ? nextw : INF;
      const int tB = (e + lane < last_end) ? pw_wpos(xe.pw) - (cnt - 1) : INF;
      // cross ranks and ties
      const int nB = rank_search(tB, tA), nA = rank_search(tA, tB);
      const int tB_at = __shfl(tB, min(nB, 63), 64);               // (shuffles are executed by all lanes, never under a branch)
      const bool tie = tA != INF && tB_at == tA && nB < 64;
      const int tie_ex = wave_excl_scan(tie ? 1 : 0, lane);
      const int ties_all = __builtin_amdgcn_readlane(tie_ex, 63) + (__builtin_amdgcn_readlane((int)tie, 63) ? 1 : 0);
      const int tie_at = __shfl(tie_ex, min(nA, 63), 64);
      const int t_lim = min(__builtin_amdgcn_readlane(tA, 63), __builtin_amdgcn_readlane(tB, 63));
      const bool okA = tA != INF && tA <= t_lim, okB = tB != INF && tB <= t_lim;
      const int kA = okA ? lane + nB - tie_ex : (1 << 20);
      const int kB = okB ? lane + nA - (nA < 64 ? tie_at : ties_all) : (1 << 20);
      const int ksteps = wave_max(max(okA ? kA + 1 : 0, okB ? kB + 1 : 0));
      // step table: which steps delete / add, and their times
      fdel[lane] = 0; fadd[lane] = 0;
      wave_sync();
      if (kA < 64) { fdel[kA] = 1; tst[kA] = tA; }
      if (kB < 64) { fadd[kB] = 1; tst[kB] = tB; }
      wave_sync();
      const int hasDel = fdel[lane], hasAdd = fadd[lane];
      const int dj = wave_excl_scan(hasDel, lane), aj = wave_excl_scan(hasAdd, lane);   // lane j: window j = [b+dj, e+aj)
      const bool cond = lane < ksteps && (e + aj < last_end) && (b + dj < b_stop);
      const uint64_t cm = __ballot(cond);
      int n_eval = (~cm == 0ull) ? 64 : __builtin_ctzll(~cm);    // windows 0 .. n_eval-1 are evaluated (n_eval >= 1)
      // events that count: not "above every query hash", and distinct inside their window (flagged entries are rare)
      bool vE = cE != -(s + 1) && kB < n_eval, vB = cB != -(s + 1) && kA < n_eval;
      {
        const bool fE = vE && (xe.pw & PW_DP), fB = vB && (xb.pw & PW_DN);
        if (__ballot(fE || fB) != 0ull) {                         // (flagged entries are rare outside repeats)
          // REV: the hash of the entering entry e + lane is already inside [b', x), b' = window start at its step (after the step's deletion)
          const int kEc = min(kB, 63), kBc = min(kA, 63);
          const int hbL = b + __shfl(dj, kEc, 64) + __shfl(hasDel, kEc, 64), weL = e + __shfl(aj, kBc, 64);
          const int rE = fE ? dup_before(I, first0 + e + lane, (int64_t)(e + lane) - hbL) : 0;
          if (rE > 0) vE = false;
          uint64_t fm = __ballot(rE < 0);                        // (windows of 65535+ entries only)
          while (fm) {
            const int l = __builtin_ctzll(fm); fm &= fm - 1;
            const int kk = __builtin_amdgcn_readlane(kB, l);
            const int hb = b + __builtin_amdgcn_readlane(dj, kk) + __builtin_amdgcn_readlane(hasDel, kk);
            const bool dup = wave_has_hash(pos, hb, e + l, (uint32_t)__builtin_amdgcn_readlane((int)xe.hash, l), lane);
            if (dup && lane == l) vE = false;
          }
          // NOOP: a later occurrence of the leaving entry b + lane stays inside (x, e'), e' = window end at its step (before the step's addition)
          const int rB = fB ? dup_after(I, first0 + b + lane, (int64_t)weL - 1 - (b + lane)) : 0;
          if (rB > 0) vB = false;
          fm = __ballot(rB < 0);
          while (fm) {
            const int l = __builtin_ctzll(fm); fm &= fm - 1;
            const int kk = __builtin_amdgcn_readlane(kA, l);
            const int we = e + __builtin_amdgcn_readlane(aj, kk);
            const bool stays = wave_has_hash(pos, b + l + 1, we, (uint32_t)__builtin_amdgcn_readlane((int)xb.hash, l), lane);
            if (stays && lane == l) vB = false;
          }
        }
      }
      // below-zone indicators, packed (window-only | matched << 16), inclusive prefix per side
      const int gE = -cE - 1, gB = -cB - 1;
      const int indE = vE ? ((cE < 0 && gE < z0 ? 1 : 0) | (cE >= 0 && cE < z0 ? 1 << 16 : 0)) : 0;
      const int indB = vB ? ((cB < 0 && gB < z0 ? 1 : 0) | (cB >= 0 && cB < z0 ? 1 << 16 : 0)) : 0;
      const int pE = wave_excl_scan(indE, lane) + indE, pB = wave_excl_scan(indB, lane) + indB;
      const int gE_ = __shfl(pE, max(aj - 1, 0), 64), gB_ = __shfl(pB, max(dj - 1, 0), 64);
      const int accE = aj > 0 ? gE_ : 0, accB = dj > 0 ? gB_ : 0;
      const int cbase_j = cbase + (accE & 0xffff) - (accB & 0xffff);
      const int sb_j = sb + (accE >> 16) - (accB >> 16);
      const int thr = s - cbase_j;
      // pivot per window; zone events in step order (each one changes the windows after its step)
      auto pivot_of = [&]() -> int { return rank_search(fz, thr); };
      int pj = pivot_of();
      uint64_t pm_j = pm;
      uint64_t zE = __ballot(vE && ((cE >= 0) ? (cE >= z0 && cE < z0 + 64) : (gE >= z0 && gE < z0 + 64)));
      uint64_t zB = __ballot(vB && ((cB >= 0) ? (cB >= z0 && cB < z0 + 64) : (gB >= z0 && gB < z0 + 64)));
      while (zE | zB) {
        const int lE = zE ? __builtin_ctzll(zE) : 0, lB = zB ? __builtin_ctzll(zB) : 0;
        const int kE_ = zE ? __builtin_amdgcn_readlane(kB, lE) : INF, kB_ = zB ? __builtin_amdgcn_readlane(kA, lB) : INF;
        const bool takeB = kB_ <= kE_;                           // the deletion of a step comes first
        const int code = takeB ? __builtin_amdgcn_readlane(cB, lB) : __builtin_amdgcn_readlane(cE, lE);
        const int kk = takeB ? kB_ : kE_;
        const int sign = takeB ? -1 : 1;
        if (takeB) zB &= zB - 1; else zE &= zE - 1;
        if (code >= 0) {
          const uint64_t bit = 1ull << (code - z0);
          pm ^= bit;
          if (lane > kk) pm_j ^= bit;
        } else {
          const int g = -code - 1;
          fz += (lane >= g - z0) ? sign : 0;
          const int p2 = pivot_of();
          if (lane > kk) pj = p2;
        }
      }
      // pivot at a zone edge in some window: evaluate the windows before it, then re-centre there
      const uint64_t xm = __ballot(lane < n_eval && (pj >= 64 || (pj == 0 && z0 > 0)));
      const bool zone_exit = xm != 0ull;
      if (zone_exit) n_eval = __builtin_ctzll(xm);
      const int sh_j = lane < n_eval ? sb_j + __popcll(pm_j & ((1ull << (pj & 63)) - 1ull)) : -1;
      if (n_eval > 0) {
        const int m = wave_max(sh_j);
        const uint64_t at = __ballot(sh_j == m);
        const int j1 = __builtin_ctzll(at), jl = 63 - __builtin_clzll(at);
        // trackers by position, not by evaluation order (the sweep visits the most promising run first):
        // the first window reaching the maximum (:510-518) and the last one equal to it (:520-524)
        auto set_first = [&]() {
          bestR = z0 + __builtin_amdgcn_readlane(pj, j1);
          const int d1 = __builtin_amdgcn_readlane(dj, j1);
          opt_b = b + d1; opt_e = e + __builtin_amdgcn_readlane(aj, j1);
          beg_pos = __builtin_amdgcn_readlane(wpb, d1);
        };
        auto set_last = [&]() {
          const int dl = __builtin_amdgcn_readlane(dj, jl);
          last_b = b + dl; last_pos = __builtin_amdgcn_readlane(wpb, dl);
        };
        if (m > best) { best = m; set_first(); set_last(); }
        else if (m == best && best > 0) {
          if (b + __builtin_amdgcn_readlane(dj, j1) < opt_b) set_first();
          if (b + __builtin_amdgcn_readlane(dj, jl) > last_b) set_last();
        }
        evals += (unsigned long long)n_eval;
      }
      // state of window n_eval
      int dn, an;
      if (n_eval < 64) { dn = __builtin_amdgcn_readlane(dj, n_eval); an = __builtin_amdgcn_readlane(aj, n_eval); }
      else { dn = __builtin_amdgcn_readlane(dj, 63) + __builtin_amdgcn_readlane(hasDel, 63); an = __builtin_amdgcn_readlane(aj, 63) + __builtin_amdgcn_readlane(hasAdd, 63); }
      if (n_eval > 0) sw_pos = tst[n_eval - 1];
      b += dn; e += an;
      while (b >= blk_end) {                                     // entered the next block: does its bound still pass?
        ++bk;
        if (bk >= nblk_s || (stage == 1 && bk == j0) || lane2(ub2, bk) < max(best, amin)) { run_stop = true; break; }
        blk_end += bspan;
      }
      if (run_stop) break;
      if (zone_exit) pending_rebuild = true;
      else {
        const int fE = an > 0 ? __builtin_amdgcn_readlane(pE, an - 1) : 0, fB = dn > 0 ? __builtin_amdgcn_readlane(pB, dn - 1) : 0;
        cbase += (fE & 0xffff) - (fB & 0xffff);
        sb += (fE >> 16) - (fB >> 16);
        const uint64_t ge = __ballot(fz >= s - cbase);           // the pivot of the next round's first window must be inside too
        if (ge == 0ull || ((ge & 1ull) && z0 > 0)) pending_rebuild = true;
      }
    }
  };
  auto slide = [&](int b_stop) __attribute__((always_inline)) {
    while (e < last_end && b < b_stop) {
      if (b + 1 - baseB >= 64 || b < baseB) loadB(b);
      if (e - baseE >= 64 || e < baseE) loadE(e);
      const int cur_wb = pw_wpos((uint32_t)__builtin_amdgcn_readlane((int)rb.pw, (int)(b - baseB)));
      if (S.shared > best) { best = S.shared; bestR = S.R; opt_b = b; opt_e = e; beg_pos = last_pos = cur_wb; }   // :510-518
      else if (S.shared == best) last_pos = cur_wb;              // :520-524
      ++evals;
      const int wb1 = pw_wpos((uint32_t)__builtin_amdgcn_readlane((int)rb.pw, (int)(b + 1 - baseB)));
      const int we = pw_wpos((uint32_t)__builtin_amdgcn_readlane((int)rE.pw, (int)(e - baseE)));
      const int d_beg = wb1 - sw_pos, d_end = we - (sw_pos + cnt - 1);
      const int adv = min(d_beg, d_end);                         // MIIteratorL2.hpp:83
      sw_pos += adv;
      if (adv == d_beg) { del_entry(b, e); ++b; }
      if (adv == d_end) { add_entry(e); ++e; }
    }
  };

  const int M = last_end - first;
  // cnt < 2 (reads shorter than w+k): the literal serial automaton; the parallel window sequence assumes that an entry
  // enters before it leaves
  const bool classic = !SKIP || cnt < 2;
  if (!classic) {
    // phase 0: bounds first (becomes 1: sweep over the blocks whose bound passes), 2: every window
    constexpr int NWORDS_MAX = 64 * NWQ;
    int phase = (M <= 64 * NWORDS_MAX && M > 192) ? 0 : 2;
    bool finished = false;
    // class masks and their prefix counts: LDS for the 10 kb class; for the long-read classes (NWQ > 2) they are written once
    // and read a few times per block, so they live in global memory and the LDS they would take buys resident waves instead
    acquire_slot();
    uint64_t* mAll = (uint64_t*)(mask_buf + (size_t)slot * l2_skip_bytes(NWQ));
    uint64_t* mLo = mAll + (NWORDS_MAX + 1);
    uint64_t* mA = mLo + (NWORDS_MAX + 1);
    uint16_t* pAll = (uint16_t*)(mA + (NWORDS_MAX + 1));
    uint16_t* pLo = pAll + (NWORDS_MAX + 1);
    uint16_t* pA = pLo + (NWORDS_MAX + 1);
    const int nwords = (int)((M + 63) >> 6);
    int bwl = 0;                                                 // log2 of the words per block
    while (((nwords + (1 << bwl) - 1) >> bwl) > L2_NBLK_MAX) ++bwl;
    const int nblk = (nwords + (1 << bwl) - 1) >> bwl;
    bspan = 64 << bwl;                                           // entries per block
    auto pfx = [&](const uint64_t* m, const uint16_t* p, int j) -> int {   // set bits among entries [first, j)
      const int o = (int)(j - first), wd = o >> 6, bit = o & 63;
      return (int)p[wd] + __popcll(m[wd] & ((1ull << bit) - 1ull));
    };
    // The kernel is bound by instruction issue (one wave instruction per cycle and CU), not by HBM, so the streaming
    // passes keep per-word work minimal: eight 512-byte loads off one address, masks handled as scalar bit sets
    // (ballots combined with s_and/s_andn2), per-word results parked in lane (word & 63) of register (word >> 6) and
    // written to LDS once at the end, prefix counts by wave scans afterwards.
    auto load8 = [&](Rec (&x)[8], int base) {
      if (base + 512 <= last_end) {
        const Rec* __restrict__ pp = pos + base + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = pp[64 * i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = pos[min(base + lane + 64 * i, nmax)];
      }
    };
    auto valid_mask = [&](int chunk_base) -> uint64_t {           // lanes of a 64-entry word that lie below last_end
      const int nv = last_end - chunk_base;
      return nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : (1ull << nv) - 1ull);
    };
    auto store_masks = [&](uint64_t* m, uint16_t* pf, const uint64_t (&reg)[NWQ]) {   // lane l holds words l, l+64, ...
      int carry = 0;
#pragma unroll
      for (int q = 0; q < NWQ; ++q) {
        const int c = __popcll(reg[q]);
        const int ex = carry + wave_excl_scan(c, lane);
        m[lane + 64 * q] = reg[q]; pf[lane + 64 * q] = (uint16_t)ex;
        carry = __builtin_amdgcn_readlane(ex, 63) + __builtin_amdgcn_readlane(c, 63);
      }
      if (lane == 0) { m[NWORDS_MAX] = 0; pf[NWORDS_MAX] = (uint16_t)carry; }
    };
    // a group of eight words lives in one register set q = word >> 6 (groups never straddle): run `body` with q and
    // "group entirely below last_end" as compile-time constants
    auto dispatch = [&](int q, bool full, auto&& body) {
#define MM_L2_CASE(QV)                                                                                                  \
      if constexpr (NWQ > QV) if (q == QV) { if (full) body(std::integral_constant<int, QV>{}, std::true_type{});        \
                                             else body(std::integral_constant<int, QV>{}, std::false_type{}); return; }
      MM_L2_CASE(0) MM_L2_CASE(1) MM_L2_CASE(2) MM_L2_CASE(3) MM_L2_CASE(4) MM_L2_CASE(5) MM_L2_CASE(6) MM_L2_CASE(7)
#undef MM_L2_CASE
    };
    static_assert(NWQ <= 8, "dispatch covers eight register sets");
    // pass A: which entries carry a query hash; lane l keeps the masks of words l, l+64, ... and the position of their
    // first entry.  e_min of every block start (first entry with wpos >= wpos[block start] + cnt) afterwards, all blocks at
    // once: the word by ranking the target among the word starts, the entry by a binary search inside that word.
    int w0r[NWQ], eLo[2] = {last_end, last_end};
    uint64_t rAll[NWQ];
#pragma unroll
    for (int q = 0; q < NWQ; ++q) { w0r[q] = 0x7fffffff; rAll[q] = 0; }
    auto pass_matched = [&]() {
      Rec nx[8];
      load8(nx, first);
      for (int base = first; base < last_end; base += 512) {
        Rec x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = nx[i];
        if (base + 512 < last_end) load8(nx, base + 512);
        int cd[8];
        {
          uint32_t hh[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) hh[i] = x[i].hash;
          l2_classify8(Q, T, tshift, tsteps, s, hh, cd);
        }
        if (cw) {
          CW* __restrict__ pc = cw + (base - first) + lane;
#pragma unroll
          for (int i = 0; i < 8; ++i) if (base + 64 * i < last_end) pc[64 * i] = cw_make(cd[i], x[i].pw & 7u);
        }
        const int wd0 = (int)((base - first) >> 6);
        dispatch(wd0 >> 6, base + 512 <= last_end, [&](auto qtag, auto fulltag) {
          constexpr int QH = decltype(qtag)::value;
          constexpr bool FULL = decltype(fulltag)::value;
#pragma unroll
          for (int i = 0; i < 8; ++i) {                          // (fully unrolled: x[] and cd[] must stay in registers)
            const int wd = wd0 + i;
            if (!FULL && wd >= nwords) continue;
            uint64_t m = __ballot(cd[i] >= 0);
            if (!FULL) m &= valid_mask(base + 64 * i);
            const bool mine = lane == (wd & 63);
            rAll[QH] = mine ? m : rAll[QH];
            const int wfirst = pw_wpos((uint32_t)__builtin_amdgcn_readfirstlane((int)x[i].pw));
            w0r[QH] = mine ? wfirst : w0r[QH];
          }
        });
      }
      store_masks(mAll, pAll, rAll);
      {
        // lane l owns blocks l and l+64; their first words are (l << bwl) and ((l + 64) << bwl)
        int tg[2], lo[2], hi[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int wd = (lane + 64 * qq) << bwl;
          int v = 0x7fffffff;
#pragma unroll
          for (int q = 0; q < NWQ; ++q) { const int t = __shfl(w0r[q], wd & 63, 64); if ((wd >> 6) == q) v = t; }
          tg[qq] = v == 0x7fffffff ? v : v + cnt;                // (cnt >= 2 on this path; unused blocks hold INT_MAX and are masked below)
          int c = -1;
#pragma unroll
          for (int q = 0; q < NWQ; ++q) c += rank_search(w0r[q], tg[qq]);
          lo[qq] = first + 64 * max(c, 0); hi[qq] = min(lo[qq] + 64, last_end);
        }
        for (int it = 0; it < 7; ++it) {                         // two independent searches per lane, steps interleaved
          const int m0 = min((lo[0] + hi[0]) >> 1, nmax), m1 = min((lo[1] + hi[1]) >> 1, nmax);
          const int p0 = pw_wpos(pos[m0].pw), p1 = pw_wpos(pos[m1].pw);
          if (lo[0] < hi[0]) { if (p0 < tg[0]) lo[0] = m0 + 1; else hi[0] = m0; }
          if (lo[1] < hi[1]) { if (p1 < tg[1]) lo[1] = m1 + 1; else hi[1] = m1; }
        }
        eLo[0] = lane < nblk ? lo[0] : last_end;
        eLo[1] = lane + 64 < nblk ? lo[1] : last_end;
      }
      wave_sync();
    };
    // pass B: rank below r0  <=>  hash below Q[r0]; no search needed once the matched bits are known
    auto pass_low = [&](int r0) {
      const bool every = r0 >= s;
      const uint32_t qr0 = every ? 0xffffffffu : Q[r0];
      uint64_t rLo[NWQ], rA[NWQ];
#pragma unroll
      for (int q = 0; q < NWQ; ++q) { rLo[q] = 0; rA[q] = 0; }
      for (int wd0 = 0; wd0 < nwords; wd0 += 8) {
        // (reads the entries, not the parked code words: half the bytes, but this pass is also what pulls the stream into
        // L2/MALL for the sweep, whose scattered entry reads otherwise cost 1.3 ms more than the 0.3 ms saved here)
        Rec x[8];
        load8(x, first + wd0 * 64);
        dispatch(wd0 >> 6, first + wd0 * 64 + 512 <= last_end, [&](auto qtag, auto fulltag) {
          constexpr int QH = decltype(qtag)::value;
          constexpr bool FULL = decltype(fulltag)::value;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int wd = wd0 + i;
            if (!FULL && wd >= nwords) continue;
            const int l = wd & 63;
            const uint64_t mk = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rAll[QH] >> 32), l) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rAll[QH], l);
            uint64_t below = __ballot(every || x[i].hash < qr0);
            if (!FULL) below &= valid_mask(first + wd * 64);
            const uint64_t first_occ = __ballot(!(x[i].pw & PW_DP));
            const bool mine = lane == l;
            rLo[QH] = mine ? (below & mk) : rLo[QH];
            rA[QH] = mine ? (below & ~mk & first_occ) : rA[QH];
          }
        });
      }
      store_masks(mLo, pLo, rLo);
      store_masks(mA, pA, rA);
      wave_sync();
    };
    int eHi[2] = {0, 0}, ub_all[2] = {-1, -1};
    int bkmax = 0;
    nblk_s = nblk;
    if (phase == 0) {
      lap(0);
      pass_matched();
      have_codes = cw != nullptr;
      lap(1);
      if (dbg_stop == 2) { release_slot(); return; }
      // per block of `bspan` b's: largest window [bF, eHi), smallest window [bL, eLo)   (lane l owns blocks l and l+64)
      {
        const int up0 = __shfl_down(eLo[0], 1, 64), up1 = __shfl_down(eLo[1], 1, 64);   // e_min of the next block start
        const int e64 = __builtin_amdgcn_readlane(eLo[1], 0);
        eHi[0] = lane == 63 ? e64 : up0;
        eHi[1] = lane == 63 ? last_end : up1;
      }
      for (int q = 0; q < 2; ++q) {
        const int bq = lane + 64 * q;
        ub_all[q] = -1;
        if (bq < nblk) {
          const int bF = first + bq * bspan, bL = min(bF + bspan - 1, last_end - 1);
          if (!(bL + 1 < last_end)) eHi[q] = last_end;
          if (eLo[q] < last_end) ub_all[q] = pfx(mAll, pAll, eHi[q]) - pfx(mAll, pAll, bF);
        } else eLo[q] = eHi[q] = last_end;
      }
      const int ubmax = wave_max(max(ub_all[0], ub_all[1]));
      lap(2);
      if (ubmax < amin) finished = true;                         // no window can reach the acceptance threshold
      else {
        // r0 without a probe: the pivot rank of a window is the number of query hashes among the s smallest of
        // query + window-only hashes.  For the most promising block (fewest window-only hashes, so the largest
        // pivot) that is hypergeometric with mean s*s/(s+wo); r0 = mean + 2.5 sigma (tuned on the bench workload).
        // A wrong guess costs only tightness: validity (r0 + a >= s) is checked per block below.
        const int key = max(ub_all[0], ub_all[1]) == ubmax ? ((ub_all[0] == ubmax) ? lane : lane + 64) : 1 << 20;
        const int bkb = wave_min(key);
        const int bLb = min(first + bkb * bspan + bspan - 1, last_end - 1);
        const int wo = max(lane2(eLo, bkb) - bLb - ubmax, 0);
        const float pq = (float)s / (float)(s + wo);
        const float sigma = sqrtf((float)s * pq * (1.0f - pq) * (1.0f - pq));
        const int r0 = min(s, (int)((float)s * pq) + max(8, (int)(2.5f * sigma) + 4));
        pass_low(r0);
        lap(5);
        if (dbg_stop == 4) { release_slot(); return; }
        for (int q = 0; q < 2; ++q) {
          const int bq = lane + 64 * q;
          int u = -1;
          if (bq < nblk && eLo[q] < last_end) {
            const int bF = first + bq * bspan, bL = min(bF + bspan - 1, last_end - 1);
            const int a = eLo[q] > bL ? pfx(mA, pA, eLo[q]) - pfx(mA, pA, bL) : 0;
            u = (r0 + a >= s) ? pfx(mLo, pLo, eHi[q]) - pfx(mLo, pLo, bF) : ub_all[q];
          }
          ub2[q] = u;
        }
        // the sweep starts a little before the block with the largest bound, so that the maximum is known early and
        // the rest (left flank afterwards, right flank on the way) is pruned against it
        const int u2max = wave_max(max(ub2[0], ub2[1]));
        const int key2 = max(ub2[0], ub2[1]) == u2max ? ((ub2[0] == u2max) ? lane : lane + 64) : 1 << 20;
        bkmax = wave_min(key2);
        j0 = bkmax;
        while (j0 > 0 && bkmax - j0 < 3 && 100 * lane2(ub2, j0 - 1) >= 95 * u2max) --j0;
        phase = 1;
      }
    }
    int done_hi = nblk;
    bk = j0;
    bool live = false;                                           // sweep: the state stands at the first b of block bk
    while (!finished) {
      int nb = first, ne = first;
      bool need_rb = true;
      if (phase == 1) {
        // the next block whose bound reaches max(best so far, amin); everything else is provably below the maximum.
        // stage 0: from j0 to the first failing block behind bkmax; stage 1: all other blocks in index order.
        bool found = false;
        for (;;) {
          if (stage == 1 && bk == j0) { bk = done_hi; live = false; }
          if (bk >= nblk) { if (stage == 0) { done_hi = nblk; stage = 1; bk = 0; live = false; continue; } break; }
          if (lane2(ub2, bk) >= max(best, amin)) { found = true; break; }
          live = false;
          if (stage == 0 && bk >= bkmax) { done_hi = bk + 1; stage = 1; bk = 0; continue; }
          ++bk;
        }
        if (!found) break;
        nb = first + bk * bspan; blk_end = nb + bspan; run_stop = false;
        need_rb = !(live && b == nb);
        if (need_rb) ne = lane2(eLo, bk);
      } else ne = e_min(first);                                  // :473, :489, MIIteratorL2.hpp:62
      if (need_rb) { b = nb; e = ne; sw_pos = pw_wpos(pos[nb].pw); pending_rebuild = true; }
      lap(2);
      block_slide(last_end);                                     // the only instance of the slide code
      lap(4);
      if (dbg_stop == 8 || dbg_stop == 9) { release_slot(); return; }
      if (phase != 1) break;
      // the slide stopped inside block bk, whose bound failed (or at the right end of the candidate)
      live = false;
      if (e >= last_end || bk >= nblk) {
        if (stage == 1) break;
        done_hi = nblk; stage = 1; bk = 0;
      }
    }
  } else {                                                       // full slide, exactly the reference's order
    for (int i = lane; i < (s + DPER - 1) / DPER; i += 64) ((uint32_t*)D)[i] = 0;
    for (int i = lane; i < (s + 31) / 32; i += 64) mt[i] = 0;
    wave_sync();
    l2_reset(S);
    const int first_end = e_min(first);                          // :473
    b = first; e = first;
    loadB(first); loadE(first);
    for (; e < first_end; ++e) add_entry(e);                     // first super-window, :489
    sw_pos = pw_wpos(pos[first].pw);                             // MIIteratorL2.hpp:62
    slide(last_end);
  }

  // K6 strand vote over the first optimal window (computeMap.hpp:424-433, slidingMap.hpp:232-254):
  // sum over query ranks below the pivot that are present in the window of strandQ * strandR, where
  // strandR comes from the LAST occurrence of the hash in the window (insert_ref overwrites, :155-156).
  lap(4);
  if (dbg_stop == 5) { release_slot(); return; }
  int strand = -1, accepted = 0;
  if (best >= amin) {
    accepted = 1;
    int votes = 0, amb_votes = 0;                                 // votes of resolved strands / number of votes whose query strand is unresolved
    for (int base = opt_b; base < opt_e; base += 512) {
      int cd[8]; uint32_t fl[8];
      if (have_codes) {
        const CW* __restrict__ pc = cw + (base - first) + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const CW ew = pc[min(64 * i, 64 * 64 * NWQ - 1 - (base - first) - lane)]; cd[i] = cw_code(ew); fl[i] = cw_flags(ew); }
      } else {
        Rec x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int j = base + lane + 64 * i; x[i] = pos[min(j, nmax)]; }
        uint32_t hh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { hh[i] = x[i].hash; fl[i] = x[i].pw & 7u; }
        l2_classify8(Q, T, tshift, tsteps, s, hh, cd);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = base + lane + 64 * i;
        if (base + 64 * i >= opt_e) continue;
        const int code = cd[i];
        const bool cnt_it = j < opt_e && code >= 0 && code < bestR;
        const int cq = cnt_it ? code : 0;
        const uint32_t sq = cnt_it ? (uint32_t)((SB[2 * (cq >> 6)] >> (cq & 63)) & 1ull) | (uint32_t)(((SB[2 * (cq >> 6) + 1] >> (cq & 63)) & 1ull) << 1) : 0u;
        const bool unres = (sq & 2) && amb_used != nullptr;       // (after the host resolved the read, amb_used is null and bit 1 is gone)
        const int contrib = cnt_it ? ((sq & 1) ? 1 : -1) * pw_strand(fl[i]) : 0;
        const bool flagged = cnt_it && (fl[i] & PW_DN);           // a later occurrence exists in the contig: inside the window?
        const int dres = flagged ? dup_after(I, first0 + j, (int64_t)opt_e - 1 - j) : 0;
        if (cnt_it && dres == 0) { if (unres) ++amb_votes; else votes += contrib; }
        uint64_t fm = __ballot(dres < 0);
        while (fm) {
          const int l = __ffsll((unsigned long long)fm) - 1;
          fm &= fm - 1;
          const uint32_t hj = pos[base + l + 64 * i].hash;
          const bool later = wave_has_hash(pos, base + l + 64 * i + 1, opt_e, hj, lane);
          if (!later && lane == l) { if (unres) ++amb_votes; else votes += contrib; }
        }
      }
    }
    votes = wave_sum(votes);
    amb_votes = wave_sum(amb_votes);
    // each unresolved vote is +1 or -1: the sign of the total is already decided unless the resolved votes are that close
    if (amb_votes > 0 && ((votes - amb_votes <= 0 && votes + amb_votes > 0) || (dbg_flags & 0x200)) && lane == 0) amb_used[r] = 1;   // (0x200: tests force the resolution path)
    strand = votes > 0 ? 1 : -1;
  }
  release_slot();
  if (NWQ == 2 && __ballot(overflow) != 0ull) {                   // a packed 8-bit counter saturated: hand the candidate to the full slide
    if (lane == 0) ovf_list[atomicAdd(ovf_n, 1u)] = (int32_t)c;
    accepted = 0; best = 0;
  }
  if (lane == 0) {
    L2Result o;
    o.contig = contig; o.mean_pos = (beg_pos + last_pos) / 2;    // :537
    o.shared = best; o.strand = strand; o.accepted = accepted; o.pad = 0;
    o.opt_beg = first0 + opt_b; o.opt_end = first0 + opt_e;
    // work counters travel with the result: hundreds of thousands of waves adding to the same few words would
    // serialise in one L2 channel (measured: half of the kernel's time)
    o.n_stream = (uint32_t)(last_end - first); o.n_evals = (uint32_t)evals; o.n_rebuilds = (uint32_t)rebuilds; o.pad2 = (uint32_t)rounds;
    out[c] = o;
    lap(6);
    if (dbg_flags & 0x100) for (int i = 0; i < 8; ++i) atomicAdd(&counters[3 + i], (unsigned long long)tph[i]);   // MM_L2_PHASES only
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Reads whose sketch does not fit the LDS-resident classes above (>= L2_SKETCH_LIMIT hashes: ~145 kb at w = 8, shorter at
// smaller w).  The reference sizes its map from the read (computeMap.hpp:228-263, slidingMap.hpp:114-131) and knows no
// limit, so these take the literal serial automaton (the SKIP=false path above: add_entry / del_entry / slide, exactly the
// reference's order) with every array in global memory: Q is the read's sketch where K2 left it, D (32-bit gap counters) and
// the matched bitmap live in a per-wave scratch slot, rank codes come from a plain binary search over Q.  One wave per
// candidate, a fixed number of resident waves looping over the candidate list.  Slow (one L2 round trip per window step) but
// exact; such reads are rare.  Their sketches come from the bitonic K2 kernel, so duplicate-hash strands were resolved on the
// host before this runs (mm_map.hip: eager tie-break) and the vote needs no "unresolved" bookkeeping.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t l2_giant_slot_words(int smax) { return (size_t)smax + (size_t)((smax + 31) / 32) + 16; }

__global__ void __launch_bounds__(64) l2_giant_kernel(IndexView I, const int32_t* __restrict__ cand, const int32_t* __restrict__ cand_read,
                                                      const uint32_t* __restrict__ sk_hash, const uint8_t* __restrict__ sk_strand,
                                                      const uint64_t* __restrict__ mz_off, const int32_t* __restrict__ sk_n,
                                                      const int32_t* __restrict__ read_len, const int32_t* __restrict__ accept_min,
                                                      int k, int w, int smax, L2Result* __restrict__ out,
                                                      const int32_t* __restrict__ cand_list, int n_list, uint32_t* __restrict__ scratch) {
  const int lane = threadIdx.x & 63;
  uint32_t* const D = scratch + (size_t)blockIdx.x * l2_giant_slot_words(smax);
  uint32_t* const mt = D + smax;
  for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
    const int64_t c = cand_list[li];
    const int r = cand_read[c];
    const int s = sk_n[r];
    const uint64_t qo = mz_off[r];
    const uint32_t* __restrict__ Q = sk_hash + qo;
    const int len = read_len[r];
    const int contig = cand[3 * c], rs = cand[3 * c + 1], re = cand[3 * c + 2];
    const int cnt = len - (w - 1) - (k - 1);                     // computeMap.hpp:470
      const int64_t first0 = contig_lower_bound_wpos(I, contig, rs, lane);                // searchIndex, :466
    const int64_t last0 = max(first0, contig_lower_bound_wpos(I, contig, re + len, lane));   // :477
    const Rec* __restrict__ pos = I.pos + first0;
    const int last_end = (int)(last0 - first0);
    const int nmax = (int)min((int64_t)0x7fffffff, I.N - 1 - first0);
    int amin = accept_min[r]; if (amin < 1) amin = 1;
    for (int i = lane; i < s; i += 64) D[i] = 0;
    for (int i = lane; i < (s + 31) / 32; i += 64) mt[i] = 0;
    wave_sync();
    L2StateT<uint32_t> S{Q, D, mt, s, 0, 0, 0, 0};
    l2_reset(S);
    int baseB = 0, baseE = 0;
    Rec rb = pos[min(lane, nmax)], rE = rb;
    int codeB = l2_classify(Q, s, rb.hash), codeE = codeB;
    auto loadB = [&](int nb) { baseB = nb; rb = pos[min(nb + lane, nmax)]; codeB = l2_classify(Q, s, rb.hash); };
    auto loadE = [&](int ne) { baseE = ne; rE = pos[min(ne + lane, nmax)]; codeE = l2_classify(Q, s, rE.hash); };
    int b = 0, e = 0, sw_pos = 0;
    auto add_entry = [&](int x) {                                // slidingMap.hpp:139-160
      if (x - baseE >= 64 || x < baseE) loadE(x);
      const int ln = x - baseE;
      const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)rE.hash, ln);
      const uint32_t pw = (uint32_t)__builtin_amdgcn_readlane((int)rE.pw, ln);
      const int code = __builtin_amdgcn_readlane(codeE, ln);
      if (code == -(s + 1)) return;                              // above every query hash: never counted
      if ((pw & PW_DP) && wave_dup_before(I, pos, first0, b, x, h, lane)) return;   // REV: hash already in the window
      if (code >= 0) l2_add_matched(S, code); else l2_add_wonly(S, -code - 1);
    };
    auto del_entry = [&](int x, int wend) {                      // slidingMap.hpp:170-214
      const int ln = x - baseB;
      const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)rb.hash, ln);
      const uint32_t pw = (uint32_t)__builtin_amdgcn_readlane((int)rb.pw, ln);
      const int code = __builtin_amdgcn_readlane(codeB, ln);
      if (code == -(s + 1)) return;
      if ((pw & PW_DN) && wave_dup_after(I, pos, first0, x, wend, h, lane)) return;   // NOOP: a later occurrence stays
      if (code >= 0) l2_del_matched(S, code); else l2_del_wonly(S, -code - 1);
    };
    int best = 0, bestR = 0, beg_pos = 0, last_pos = 0, opt_b = 0, opt_e = 0;
    unsigned long long evals = 0;
    {
      const int first_end = (int)wave_lower_bound_wpos(pos, 0, last_end, pw_wpos(pos[0].pw) + cnt, lane);   // :473
      loadB(0); loadE(0);
      for (; e < first_end; ++e) add_entry(e);                   // first super-window, :489
      sw_pos = pw_wpos(pos[0].pw);                               // MIIteratorL2.hpp:62
      while (e < last_end) {                                     // computeMap.hpp:496-533 + MIIteratorL2::next
        if (b + 1 - baseB >= 64 || b < baseB) loadB(b);
        if (e - baseE >= 64 || e < baseE) loadE(e);
        const int cur_wb = pw_wpos((uint32_t)__builtin_amdgcn_readlane((int)rb.pw, b - baseB));
        if (S.shared > best) { best = S.shared; bestR = S.R; opt_b = b; opt_e = e; beg_pos = last_pos = cur_wb; }   // :510-518
        else if (S.shared == best) last_pos = cur_wb;            // :520-524
        ++evals;
        const int wb1 = pw_wpos((uint32_t)__builtin_amdgcn_readlane((int)rb.pw, b + 1 - baseB));
        const int we = pw_wpos((uint32_t)__builtin_amdgcn_readlane((int)rE.pw, e - baseE));
        const int d_beg = wb1 - sw_pos, d_end = we - (sw_pos + cnt - 1);
        const int adv = min(d_beg, d_end);                       // MIIteratorL2.hpp:83
        sw_pos += adv;
        if (adv == d_beg) { del_entry(b, e); ++b; }
        if (adv == d_end) { add_entry(e); ++e; }
      }
    }
    // K6 strand vote over the first optimal window (computeMap.hpp:424-433, slidingMap.hpp:232-254)
    int strand = -1, accepted = 0;
    if (best >= amin) {
      accepted = 1;
      int votes = 0;
      for (int base = opt_b; base < opt_e; base += 64) {
        const int j = base + lane;
        const Rec x = pos[min(j, nmax)];
        const int code = l2_classify(Q, s, x.hash);
        const bool cnt_it = j < opt_e && code >= 0 && code < bestR;
        const int contrib = cnt_it ? ((sk_strand[qo + code] & 1) ? 1 : -1) * pw_strand(x.pw) : 0;
        const bool flagged = cnt_it && (x.pw & PW_DN);           // a later occurrence exists in the contig: inside the window?
        const int dres = flagged ? dup_after(I, first0 + j, (int64_t)opt_e - 1 - j) : 0;
        if (cnt_it && dres == 0) votes += contrib;
        uint64_t fm = __ballot(dres < 0);
        while (fm) {
          const int l = __ffsll((unsigned long long)fm) - 1;
          fm &= fm - 1;
          const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)x.hash, l);
          const bool later = wave_has_hash(pos, base + l + 1, opt_e, hj, lane);
          if (!later && lane == l) votes += contrib;
        }
      }
      votes = wave_sum(votes);
      strand = votes > 0 ? 1 : -1;
    }
    if (lane == 0) {
      L2Result o;
      o.contig = contig; o.mean_pos = (beg_pos + last_pos) / 2;  // :537
      o.shared = best; o.strand = strand; o.accepted = accepted; o.pad = 0;
      o.opt_beg = first0 + opt_b; o.opt_end = first0 + opt_e;
      o.n_stream = (uint32_t)last_end; o.n_evals = (uint32_t)evals; o.n_rebuilds = 0; o.pad2 = 0;
      out[c] = o;
    }
    wave_sync();
  }
}

}  // namespace mm
