// K5 + K6, second form — the "zone" kernel: one wavefront per L1 candidate, everything but set membership in HASH space
// (replaces Map::computeL2MappedRegions + the statistics part of doL2Mapping, computeMap.hpp:396-538;
//  SlideMapper, slidingMap.hpp:26-318; MIIteratorL2::next, MIIteratorL2.hpp:74-96 — like mm_l2.hpp, whose window sequence,
//  skip-ahead bounds and 64-windows-per-round evaluation it keeps; mm_l2_core.hpp has the counting argument).
//
// Why a second form.  l2_kernel (mm_l2.hpp) ranks EVERY streamed index entry in the read's sorted sketch Q (bucket table + 4-5 search
// steps in LDS, 48 wave instructions per 64 entries), parks a 2-byte rank code per entry and rebuilds the window state from those codes
// with one LDS counter per rank.  It is bound by instruction issue (2.8 VALU wave instructions per streamed entry, 87 % VALU busy), and
// its long-read classes by LDS: a sketch of 7-16 k hashes plus per-wave state leaves 1-3 workgroups per CU (VALU 20 % busy).
// What the sliding MinHash needs of an entry is far less than its rank:
//   * is its hash in Q ("matched")?                                                — set membership
//   * is its hash at or below two thresholds tau_lo < tau_hi, the hashes that bound a ZONE of 64 consecutive ranks
//     [z0, z0 + 64) of Q around the expected pivot?                                — two integer compares
//   * its rank only if it falls INSIDE the zone (3 % of the entries), and there it is the number of the zone's 64 hashes
//     below it: one compare against a register that holds Q[z0 + lane] and a population count.
// Pivot rank R(W) = min r with r + #{distinct window-only hashes of W below Q[r]} >= s and shared(W) = #{matched ranks < R} (mm_l2_core.hpp)
// become, for a window W whose pivot lies in the zone,
//   cbase(W) = distinct window-only entries with hash <= tau_lo,   sb(W) = distinct matched entries with hash <= tau_lo      (prefix sums over two bit masks)
//   fz[l]    = z0 + l + #{distinct window-only entries of W in zone gaps z0 .. z0 + l}                                        (lane l, from the zone's few entries)
//   R = z0 + #{l : fz[l] < s - cbase},   shared = sb + #{matched zone ranks of W below R}.
// Membership is a Bloom-type bit table of the sketch in LDS (one ds_read per entry, 6-12 % false positives); the entries that pass
// it (matched 6 % + false positives) are compacted through an LDS ring and only they are searched in Q, 64 at a time with all lanes
// busy.  The matched ones form the candidate's MATCHED LIST (entry, rank: 4 bytes) — all the strand vote reads — and a bit mask.
// Nothing is parked per entry, no LDS counter per rank exists, and the sketch itself need not be in LDS at all (long-read classes
// search it in global memory, through the bucket table): LDS per workgroup is the bit table + the bucket table + ~2 KB per wave.
//
// Band and zone.  Pass B computes, for one BAND of 128 consecutive ranks [zb, zb + 128) around the expected pivot of the most promising
// block (hypergeometric mean of the pivot; measured on the bench batch the best window's pivot lies at that estimate - 11 +- 19 ranks,
// tools/l2z_pivot_hist.py), prefix masks at a reference rank r_ref inside the band (matched / window-only entries at or below Q[r_ref]:
// the skip-ahead bound and the anchor of every window state), the mask of the band's own entries (12 % of the stream) and the mask of the
// entries that have an earlier occurrence in their contig.  A window state is built from the prefix sums plus the band entries of the
// window (a word per lane, their band rank by a 7-step search over the band's 128 hashes in LDS): gap counts of the band -> pivot ->
// the 64-rank ZONE [z0, z0 + 64) is centred on the ACTUAL pivot, as l2_kernel does.  The slide itself only compares hashes against the
// zone's two thresholds, so the zone moves freely inside the band (a zone exit re-centres it with another such rebuild).  A window whose
// pivot lies outside the band is not scored in that pass; its block is flagged, and after the sweep the flagged blocks are swept again
// with the next band up (then down) — every window is scored in a pass whose band holds its pivot, the trackers compare positions, so
// the order is free and the result is bit-identical to the full slide.
//
// Kept from mm_l2.hpp: scratch slots per resident wave split by XCD, candidate groups of a read per workgroup, position-ordered launch,
// skip-ahead bounds per block of 64 b's (m_all, and m_lo with the validity test (zt - 1) + a >= s, zt = zone top), the sweep order, the
// merge of leave/enter times by cross-ranking, trackers by position, the duplicate-hash handling through DP/DN flags and
// dup_before / dup_after (slidingMap.hpp:148-157, 186-209).
#pragma once
#include "mm_l2.hpp"

namespace mm {

#ifndef L2Z_RING
#define L2Z_RING 128
#endif
__host__ __device__ constexpr int l2z_qcap(bool qlds) { return qlds ? L2Z_RING : 2 * L2Z_RING; }   // (sketch searched in global memory: two searches per lane at a time, see dense2)
constexpr int L2Z_QCAP_ = L2Z_RING;                                   // ring of compacted (hash, entry) pairs waiting for the search (4 + 2 bytes each): eight words are added at a time, 64 taken
#ifndef L2Z_WAVES_10K
#define L2Z_WAVES_10K 5                                         // waves per SIMD the 10 kb class is compiled for (96 registers).  6 (80 registers) was the better choice while the band loop
                                                                // kept hoisted values in scratch (12.99 against 13.35 ms); without them 5 is: 9.87 + 1.87 -> 9.70 + 1.74 ms, the step - 0.3 ms
                                                                // (4: 11.15 + 1.83, 7: 10.75 + 2.20; profiles/r06_ab_k5_waves.txt)
#endif
#ifndef L2Z_WAVES_LONG
#define L2Z_WAVES_LONG 4
#endif
// per-slot global scratch besides the matched list (4 bytes x 4096 NWQ): five bit masks and three 16-bit prefix arrays over the 64 NWQ (+1) words
// of the stream
__host__ __device__ inline size_t l2z_mask_bytes(int nwq) { return (((size_t)(64 * nwq + 1) * (5 * 8 + 3 * 2)) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t l2z_list_bytes(int nwq) { return (size_t)(64 * 64 * nwq) * 4; }
// bits of the membership table: at least 8 per sketch hash, at least 2^15
__host__ __device__ inline int l2z_bloom_log2(int smax) { int b = 15; while (((size_t)1 << b) < (size_t)8 * (size_t)smax && b < 19) ++b; return b; }
// ranks of a band: 128 for the 10 kb class (+-3.4 standard deviations of the best window's pivot around its expectation there), 512 for the long-read
// classes (the deviation grows with the square root of the sketch: ~47 ranks at 13 000 hashes)
__host__ __device__ constexpr int l2z_band(int nwq) { return nwq == 2 ? 128 : 512; }
// Where the pivot of the best window lies against the two estimates of it, measured on the bench batch (tools/l2z_pivot_hist.py): against the one from
// pass A's matched counts -11 +- 19 ranks, against the one predicted from L1's seed hits +7 +- 19.  The band is centred there; the reference rank of
// the prefix masks and the bound sits ~2.3 sigma above the centre (lower: tighter bound but more blocks whose bound is not valid; 10 / 20 / 32 above
// the first estimate measured 381 / 281 / 236 million windows scored per bench step).
// In units of the hypergeometric deviation sigma = sqrt(s p (1 - p)^2), p = s / (s + window-only hashes) (15.7 ranks for the bench's 10 kb reads): centre at
// the estimate - 0.7 sigma (from the matched counts) / + 0.43 sigma (predicted), reference rank 3.5 sigma above the centre (2.0 / 2.75 / 3.5: 215 / 149 / 122 million windows scored on 20 000 reads of 20-50 kb, 276 / 234 / 236 on the 10 kb bench batch).
#ifndef L2Z_CENTRE_SIG
#define L2Z_CENTRE_SIG (-0.70f)
#endif
#ifndef L2Z_CENTRE_SIG_PRED
#define L2Z_CENTRE_SIG_PRED 0.43f
#endif
#ifndef L2Z_REF_SIG
#define L2Z_REF_SIG (NWQ == 2 ? 3.0f : 4.5f)                     // (measured, ms of K5 on the 10 kb bench batch / on 20 000 reads of 20-50 kb: 3.0 11.58 / 11.59, 3.5 11.58 / 10.88, 4.5 11.76 / 10.27)
#endif
// per-wave LDS: the first entry's position of every word (pass A writes, the e_min searches read) | a region used by pass A as
// {ring of (hash, entry) pairs, matched bits of the current group of 64 words} and afterwards as {band gap counters / prefixes, the band's
// hashes, band presence bits, slide scratch}
__host__ __device__ constexpr int l2z_x_bytes(int nwq, bool qlds) {
  const int xa = l2z_qcap(qlds) * 6 + 64 * 8, xb = l2z_band(nwq) * 4 * 2 + l2z_band(nwq) / 8 + L2_SCRATCH_BYTES;
  return (xa > xb ? xa : xb) + 64;                               // (+ eight phase clocks at its end)
}
__host__ __device__ inline size_t l2z_wave_bytes(int nwq, bool qlds) { return ((((size_t)(64 * nwq + 1) * 4) + 15) & ~(size_t)15) + (size_t)l2z_x_bytes(nwq, qlds); }
__host__ __device__ inline size_t l2z_shared_bytes(int smax, int nwq, bool qlds, int bbl) {
  return ((size_t)1 << (bbl - 3)) + l2_tpart_bytes(nwq) + (qlds ? l2_qpart_bytes(smax) : 0);
}
__host__ __device__ inline size_t l2z_lds_bytes(int smax, int nwq, bool qlds, int bbl, int waves) { return l2z_shared_bytes(smax, nwq, qlds, bbl) + (size_t)waves * l2z_wave_bytes(nwq, qlds); }

// number of leading lanes whose (ascending, unsigned) arr lies below v
__device__ inline int rank_search_u(uint32_t arr, uint32_t v) {
  int lo = 0;
  for (int st = 32; st >= 1; st >>= 1) { const uint32_t x = (uint32_t)__shfl((int)arr, lo + st - 1, 64); if (x < v) lo += st; }
  const uint32_t x = (uint32_t)__shfl((int)arr, lo, 64);
  return lo + (x < v ? 1 : 0);
}
__device__ inline uint64_t readlane_u64(uint64_t v, int l) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
// lane l of three register pairs := three wave-uniform 64-bit values, M0 set up once
__device__ inline void park64x3(uint64_t& r0, uint64_t v0, uint64_t& r1, uint64_t v1, uint64_t& r2, uint64_t v2, int l) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t a0 = (uint32_t)r0, a1 = (uint32_t)(r0 >> 32), b0 = (uint32_t)r1, b1 = (uint32_t)(r1 >> 32), c0 = (uint32_t)r2, c1 = (uint32_t)(r2 >> 32);
  const int ls = __builtin_amdgcn_readfirstlane(l);
  uint32_t keep;
  asm("s_mov_b32 %6, m0\n\ts_mov_b32 m0, %13\n\tv_writelane_b32 %0, %7, m0\n\tv_writelane_b32 %1, %8, m0\n\tv_writelane_b32 %2, %9, m0\n\tv_writelane_b32 %3, %10, m0\n\t"
      "v_writelane_b32 %4, %11, m0\n\tv_writelane_b32 %5, %12, m0\n\ts_mov_b32 m0, %6"
      : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "=&s"(keep)
      : "s"((uint32_t)v0), "s"((uint32_t)(v0 >> 32)), "s"((uint32_t)v1), "s"((uint32_t)(v1 >> 32)), "s"((uint32_t)v2), "s"((uint32_t)(v2 >> 32)), "s"(ls));
  r0 = (uint64_t)a0 | ((uint64_t)a1 << 32); r1 = (uint64_t)b0 | ((uint64_t)b1 << 32); r2 = (uint64_t)c0 | ((uint64_t)c1 << 32);
#endif
}
// lane l of the register pair `reg` := val (a wave-uniform 64-bit value): two v_writelane, no compare, no select
__device__ inline uint64_t park64(uint64_t reg, uint64_t val, int l) {
  uint32_t lo = (uint32_t)reg, hi = (uint32_t)(reg >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t vlo = (uint32_t)val, vhi = (uint32_t)(val >> 32);
  const int ls = __builtin_amdgcn_readfirstlane(l);
  // (gfx9 takes one SGPR per VALU instruction over the constant bus: the lane select goes through M0)
  uint32_t keep;                                                 // (M0 is the compiler's: put back what it held)
  asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %5\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\ts_mov_b32 m0, %2"
      : "+v"(lo), "+v"(hi), "=&s"(keep) : "s"(vlo), "s"(vhi), "s"(ls));
#endif
  return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ inline int mbcnt64(uint64_t m, int base = 0) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)base)); }   // base + set bits of m below this lane

// Pass B as a pass of its own — the masks of a band from a second reading of the stream.  Rare (a band that was not predicted, or the next band up /
// down: 0.3 % of the bench's candidates), so it is a function the kernel CALLS: inlined, its 40 registers of stream words in flight were live ranges the
// allocator made room for by spilling ~47 of the kernel's registers around the band loop of EVERY candidate (25 KB of scratch traffic each).
struct L2ZPassB { const Rec* pos; int M, nmax, nwords; uint32_t tau_bh, tau_ref, tau_bl; int has_bl; const uint64_t* mAll; uint64_t* mLo; uint64_t* mA; uint64_t* mZ; uint64_t* mX; uint16_t* pLo; uint16_t* pA; };
__device__ __attribute__((noinline)) void l2z_pass_b(const L2ZPassB& a) {
  const int lane = threadIdx.x & 63;
  const int last_end = __builtin_amdgcn_readfirstlane(a.M);
  auto load8 = [&](Rec (&x)[8], int base) {
    if (base + 512 <= last_end) {
      const Rec* __restrict__ pp = a.pos + base + lane;
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = pp[64 * i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = a.pos[min(base + lane + 64 * i, a.nmax)];
    }
  };
  auto valid_mask = [&](int chunk_base) -> uint64_t {
    const int nv = last_end - chunk_base;
    return nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : (1ull << nv) - 1ull);
  };
  // (the arguments are wave-uniform; loaded through a pointer the compiler cannot know that)
  const int nwords = __builtin_amdgcn_readfirstlane(a.nwords);
  const uint32_t tau_bh = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tau_bh), tau_ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tau_ref), tau_bl = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tau_bl);
  const bool has_bl = __builtin_amdgcn_readfirstlane(a.has_bl) != 0;
  int carryLo = 0, carryA = 0;
  // per word four ballots, each parked in lane (word & 63) of a register pair; the masks are combined 64 words at a time
  uint64_t rRef = 0, rBH = 0, rBL = 0, rNF = 0;                  // at or below Q[r_ref] | at or below the band's top | at or below Q[zb - 1] | an earlier occurrence exists (rare)
  Rec nx[8];
  load8(nx, 0);
  for (int wd0 = 0; wd0 < nwords; wd0 += 8) {
    Rec x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = nx[i];
    if (wd0 + 8 < nwords) load8(nx, (wd0 + 8) * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int wd = wd0 + i;
      if (wd >= nwords) continue;
      uint64_t le_bh = __ballot(x[i].hash <= tau_bh);
      if (wd0 + 8 >= nwords) le_bh &= valid_mask(wd * 64);
      const uint64_t le_ref = __ballot(x[i].hash <= tau_ref);
      const uint64_t nf = __ballot((x[i].pw & PW_DP) != 0u);
      park64x3(rBH, le_bh, rRef, le_ref, rBL, has_bl ? __ballot(x[i].hash <= tau_bl) : 0ull, wd & 63);
      if (nf) rNF = park64(rNF, nf, wd & 63);
    }
    if (((wd0 + 8) & 63) == 0 || wd0 + 8 >= nwords) {
      const int g0 = wd0 & ~63;
      const uint64_t regAll = a.mAll[g0 + lane];
      rRef &= rBH;
      const uint64_t rLo = rRef & regAll, rA = rRef & ~regAll & ~rNF;
      const int cl = __popcll(rLo), ca = __popcll(rA);
      const int exl = carryLo + wave_excl_scan(cl, lane), exa = carryA + wave_excl_scan(ca, lane);
      a.mLo[g0 + lane] = rLo; a.pLo[g0 + lane] = (uint16_t)exl;
      a.mA[g0 + lane] = rA; a.pA[g0 + lane] = (uint16_t)exa;
      a.mZ[g0 + lane] = rBH & ~rBL; a.mX[g0 + lane] = rBH & rNF;
      carryLo = __builtin_amdgcn_readlane(exl, 63) + __builtin_amdgcn_readlane(cl, 63);
      carryA = __builtin_amdgcn_readlane(exa, 63) + __builtin_amdgcn_readlane(ca, 63);
      rRef = rBH = rBL = rNF = 0;
    }
  }
  if ((nwords & 63) == 0 && lane == 0) { a.mLo[nwords] = 0; a.pLo[nwords] = (uint16_t)carryLo; a.mA[nwords] = 0; a.pA[nwords] = (uint16_t)carryA; }
}

// WAVES: candidates of ONE read per workgroup (they share the bit table, the bucket table and — QLDS — the sketch).  NWQ: 64 NWQ mask words,
// i.e. candidates of up to 4096 NWQ streamed entries (larger ones go to `big_list`: the host runs them through l2_kernel).
// QLDS: the sorted sketch lives in LDS (10 kb class); otherwise the few entries that pass the bit table search it in global memory.
template <int WAVES, int NWQ, bool QLDS>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(NWQ == 2 ? L2Z_WAVES_10K : L2Z_WAVES_LONG)))
l2z_kernel(IndexView I, const int32_t* __restrict__ cand, const int32_t* __restrict__ cand_read,
           const uint32_t* __restrict__ sk_hash, const uint8_t* __restrict__ sk_strand,
           const uint64_t* __restrict__ mz_off, const int32_t* __restrict__ sk_n,
           const int32_t* __restrict__ read_len, const int32_t* __restrict__ accept_min,
           int k, int w, int smax, int bbl /* log2 of the bits of the membership table */, L2Result* __restrict__ out,
           unsigned long long* __restrict__ counters /* [11] debug flags, [3..9] phase clocks (MM_L2_PHASES) */,
           const int32_t* __restrict__ grp_cand0, const int32_t* __restrict__ grp_n,
           int32_t* __restrict__ ovf_list, unsigned int* __restrict__ ovf_n /* reads shorter than w + k: the literal full slide */,
           int32_t* __restrict__ big_list, unsigned int* __restrict__ big_n /* more than 4096 NWQ streamed entries */,
           uint8_t* __restrict__ amb_used, uint32_t* __restrict__ ml_buf /* 4096 NWQ words per slot */, uint8_t* __restrict__ mask_buf /* l2z_mask_bytes(NWQ) per slot */,
           unsigned int* __restrict__ slot_flags, int n_slots,
           const int32_t* __restrict__ cand_hint /* seed hits inside each candidate (l1_wave_kernel; 0: none): what the best window's matched count will be */,
           const int64_t* __restrict__ cand_rng /* optional: [first, behind-last) index entry of each candidate's stream (l2_ranges_kernel) */) {
  extern __shared__ __align__(16) uint32_t lds[];
  constexpr int NW = 64 * NWQ, CAP = 64 * NW, NW1 = NW + 1;
  constexpr int QCAP = l2z_qcap(QLDS);
  constexpr int BAND = l2z_band(NWQ), BPL = BAND / 64;            // ranks of a band, ranks per lane
  constexpr int TBITS = l2_tbits(NWQ), tshift = 32 - TBITS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t* const BL = lds;                                      // membership bits of the sketch, keyed by the low hash bits (MurmurHash3's finaliser: uniform)
  const uint32_t bm = (1u << (bbl - 5)) - 1u;
  uint16_t* const T = (uint16_t*)((uint8_t*)lds + ((size_t)1 << (bbl - 3)));
  int* const tmaxp = (int*)(T + ((l2_tsize(NWQ) + 1) & ~1));
  uint32_t* const QL = (uint32_t*)((uint8_t*)T + l2_tpart_bytes(NWQ));   // (QLDS only)
  uint8_t* const wbase = (uint8_t*)lds + l2z_shared_bytes(smax, NWQ, QLDS, bbl) + (size_t)wave * l2z_wave_bytes(NWQ, QLDS);
  const int64_t c0 = (int64_t)grp_cand0[blockIdx.x];
  const int r = cand_read[c0];                                   // every wave of the workgroup serves this read
  const int s = sk_n[r];
  const uint64_t qo = mz_off[r];
  const int len = read_len[r];
  const uint32_t* __restrict__ Qg = sk_hash + qo;
  auto qat = [&](int i) -> uint32_t { if constexpr (QLDS) return QL[i]; else return Qg[i]; };

  // ---- per workgroup: bit table, bucket table (l2_bucket, mm_l2_core.hpp), the sketch itself (QLDS) -------------------------------
  if (((int)counters[11] & 0xff) == 10) { if (threadIdx.x < 64 * WAVES && (int)(threadIdx.x >> 6) < grp_n[blockIdx.x] && lane == 0) { L2Result z{}; out[c0 + (threadIdx.x >> 6)] = z; } return; }   // MM_L2_STOP=10: nothing at all (what the launches and the grouping cost)
  for (int i = threadIdx.x; i <= (int)bm; i += 64 * WAVES) BL[i] = 0;
  if (threadIdx.x == 0) *tmaxp = 0;
  // (the sketch into LDS with coalesced, independent loads: the table loop below walked a thread's run of ranks with one dependent global load per rank,
  //  nine round trips per workgroup before its first wave could start)
  if constexpr (QLDS) {
    for (int i = threadIdx.x; i < s; i += 64 * WAVES) QL[i] = Qg[i];
    if (threadIdx.x < L2_QPAD) QL[s + threadIdx.x] = 0xffffffffu;
  }
  __syncthreads();
  // T[b] = first rank whose bucket is >= b (every entry written exactly once, as in l2_kernel).  A thread takes a run of consecutive ranks, so that
  // the bucket of a hash is computed once (the float arithmetic of l2_bucket is most of this set-up) and its predecessor's is at hand.
  {
    const int per = (s + 64 * WAVES) / (64 * WAVES);             // ranks 0 .. s: s + 1 table steps
    const int i0 = (int)threadIdx.x * per, i1 = min(i0 + per, s + 1);
    int bprev = (i0 > 0 && i0 <= s) ? l2_bucket(qat(i0 - 1), tshift) : -1;
    for (int ib = i0; ib < i1; ib += 8) {                        // eight ranks per step: their hashes in flight together
      uint32_t hh[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) hh[u] = qat(max(min(ib + u, s - 1), 0));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = ib + u;
        if (i >= i1) break;
        int hi = 1 << TBITS;
        if (i < s) {
          const uint32_t h = hh[u];
          atomicOr(&BL[(h >> 5) & bm], (1u << (h & 31)) | (1u << ((h >> 20) & 31)));   // two bits of one word: one LDS read per test, ~2 % false positives at 15 bits per hash
          hi = l2_bucket(h, tshift);
        }
        for (int bb = bprev + 1; bb <= hi; ++bb) T[bb] = (uint16_t)i;
        bprev = hi;
      }
    }
  }
  __syncthreads();
  {
    int tm = 0;
    for (int bkt = threadIdx.x; bkt < (1 << TBITS); bkt += 64 * WAVES) tm = max(tm, (int)T[bkt + 1] - (int)T[bkt]);
    tm = wave_max(tm);
    if ((threadIdx.x & 63) == 0) atomicMax(tmaxp, tm);
  }
  __syncthreads();
  const int tsteps = *tmaxp ? 32 - __clz(*tmaxp) : 0;
  const int dbg_flags = (int)counters[11];                       // bit 8 MM_L2_PHASES, bit 9 MM_FORCE_AMB_REDO; 0 in normal operation
  if (wave >= grp_n[blockIdx.x]) return;
  const int64_t c = c0 + wave;
  const int dbg_stop = dbg_flags & 0xff;                        // MM_L2_STOP: leave after a phase WITHOUT RESULTS (timing aid: tools/l2z_stops.py)
  if (dbg_stop) { if (lane == 0) { L2Result z{}; out[c] = z; } if (dbg_stop == 1) return; }

  // rank code of a hash (>= 0: matched rank; < 0: not in Q)
  auto classify = [&](uint32_t h) -> int {
    if constexpr (QLDS) return l2_classify1(QL, T, tshift, tsteps, s, h);
    else {
      const int bkt = l2_bucket(h, tshift);
      int lo = T[bkt], hi = T[bkt + 1];
      for (int it = 0; it < tsteps; ++it) {
        const int m = min((lo + hi) >> 1, s - 1);
        const uint32_t v = Qg[m];
        if (lo < hi) { if (v < h) lo = m + 1; else hi = m; }
      }
      const uint32_t e = Qg[min(lo, s - 1)];
      return (lo < s && e == h) ? lo : -(lo + 1);
    }
  };

  const int contig = cand[3 * c], rs = cand[3 * c + 1], re = cand[3 * c + 2];
  const int cnt = len - (w - 1) - (k - 1);                       // computeMap.hpp:470
  if (cnt < 2) {                                                 // reads shorter than w+k: left to the literal full slide (host launches it on this list)
    if (lane == 0) { L2Result z{}; out[c] = z; ovf_list[atomicAdd(ovf_n, 1u)] = (int32_t)c; }
    return;
  }
  int64_t first0, last0;
  if (cand_rng) { first0 = cand_rng[2 * c]; last0 = cand_rng[2 * c + 1]; }   // (made for all candidates at once: mm_map.hip, l2_ranges_kernel)
  else {
    first0 = contig_lower_bound_wpos(I, contig, rs, lane);                // searchIndex, :466
    last0 = max(first0, contig_lower_bound_wpos(I, contig, re + len, lane));   // :477
  }
  if (last0 - first0 > (int64_t)CAP) {                           // (merged candidates over long repeats)
    if (lane == 0) { L2Result z{}; out[c] = z; big_list[atomicAdd(big_n, 1u)] = (int32_t)c; }
    return;
  }
  const Rec* __restrict__ pos = I.pos + first0;
  const int M = (int)(last0 - first0), last_end = M;
  const int nmax = (int)min((int64_t)0x7fffffff, I.N - 1 - first0);
  int amin = accept_min[r]; if (amin < 1) amin = 1;
  const int nwords = (M + 63) >> 6;
  if (M == 0) {                                                  // nothing to stream (a candidate behind the last entry of its contig)
    if (lane == 0) {
      L2Result o{};
      o.contig = contig; o.strand = -1; o.opt_beg = first0; o.opt_end = first0;
      out[c] = o;
    }
    return;
  }

  // phase clocks (MM_L2_PHASES, builds with -DL2Z_CLOCKS): setup, passA, bounds, rebuild, slide, passB, vote — in LDS.  Compiled out otherwise: as a run-time
  // switch they cost the kernel 0.26 of 9.9 ms (a branch per lap inside the slide, the clock's registers live across it)
#ifdef L2Z_CLOCKS
  const bool prof = (dbg_flags & 0x100) != 0;
#else
  constexpr bool prof = false;                                   // (the clocks cost the slide two scalar registers, a branch per lap and the spills that go with them: a build with -DL2Z_CLOCKS has them)
#endif
  long long* const tphL = (long long*)(wbase + l2z_wave_bytes(NWQ, QLDS) - 64);
  long long tmark = 0;
  if (prof) { if (lane < 8) tphL[lane] = 0; tmark = clock64(); }
  auto lap = [&](int ph) { if (prof) { const long long now = clock64(); if (lane == 0) tphL[ph] += now - tmark; tmark = now; } };

  // ---- scratch slot of this wave (mm_l2.hpp: taken from the share of the XCD the wave runs on, given back at the end) ------------
  int slot;
  {
    unsigned int sidx = blockIdx.x * WAVES + wave;
    if (slot_flags) {
      const unsigned int per = (unsigned int)n_slots >> 3;
      const unsigned int lo = ((unsigned int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) * per;   // hwreg(HW_REG_XCC_ID, 0, 4)
      unsigned int t = (sidx * 2654435761u >> 7) % per;
      if (lane == 0) while (atomicCAS(&slot_flags[lo + t], 0u, 1u) != 0u) t = t + 1u == per ? 0u : t + 1u;
      sidx = lo + (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    }
    slot = (int)sidx;
  }
  auto release_slot = [&]() {
    if (slot_flags) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) atomicExch(&slot_flags[slot], 0u);
    }
  };
  uint32_t* const ML = ml_buf + (size_t)slot * CAP;              // matched list: entry (15 bits) | rank << 15, in stream order
  uint8_t* const mb = mask_buf + (size_t)slot * l2z_mask_bytes(NWQ);
  uint64_t* const mAll = (uint64_t*)mb;                          // matched
  uint64_t* const mLo = mAll + NW1;                              // matched, hash <= Q[r_ref]
  uint64_t* const mA = mLo + NW1;                                // window-only, hash <= Q[r_ref], first occurrence in its contig
  uint64_t* const mZ = mA + NW1;                                 // entries of the band: Q[zb - 1] < hash <= Q[band top]
  uint64_t* const mX = mZ + NW1;                                 // hash <= Q[band top], an earlier occurrence exists in the contig (DP)
  uint16_t* const pAll = (uint16_t*)(mX + NW1);
  uint16_t* const pLo = pAll + NW1;
  uint16_t* const pA = pLo + NW1;
  int* const W0 = (int*)wbase;                                   // wpos of the first entry of every word
  uint8_t* const xb_ = wbase + (l2z_wave_bytes(NWQ, QLDS) - l2z_x_bytes(NWQ, QLDS));
  uint32_t* const RQh = (uint32_t*)xb_;                          // pass A: the ring's hashes ...
  uint16_t* const RQj = (uint16_t*)(xb_ + QCAP * 4);         // ... and entry numbers
  uint64_t* const mL = (uint64_t*)(xb_ + QCAP * 6);          // pass A: matched bits of the current group of 64 words
  uint32_t* const zc = (uint32_t*)xb_;                           // afterwards: gap counters of the band, then their inclusive prefixes
  uint32_t* const BQ = zc + BAND;                            // Q[zb .. zb + 128), 0xffffffff from rank s on
  uint32_t* const pmw = BQ + BAND;                           // matched ranks of the band present in the window (128 bits)
  int* const tst = (int*)(pmw + BAND / 32);
  uint8_t* const fdel = (uint8_t*)(tst + 64);
  uint8_t* const fadd = fdel + 64;

  auto pfx = [&](const uint64_t* m, const uint16_t* p, int j) -> int {   // set bits among entries [0, j)
    const int wd = j >> 6, bit = j & 63;
    return (int)p[wd] + __popcll(m[wd] & ((1ull << bit) - 1ull));
  };
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
  auto valid_mask = [&](int chunk_base) -> uint64_t {
    const int nv = last_end - chunk_base;
    return nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : (1ull << nv) - 1ull);
  };
  auto close_prefix = [&](uint64_t* m, uint16_t* p, int total) {   // the word behind the last group, where pfx(last_end) lands when M is a multiple of 4096
    if ((nwords & 63) == 0 && lane == 0) { m[nwords] = 0; p[nwords] = (uint16_t)total; }
  };

  // ---- the band, predicted.  Pivot rank of a window = number of query hashes among the s smallest of query + window-only hashes: hypergeometric
  // with mean s s / (s + wo), wo = window entries - matched entries.  The window's entries follow from the streamed range (entries per base x cnt),
  // the matched count of the BEST window is about what L1 counted as the candidate's seed hits — so the band is known BEFORE the stream is read
  // and its threshold masks come out of pass A (no second pass over the stream).  Checked against the matched counts afterwards.
  int zb = 0, r_ref = 0;                                         // band start; reference rank of the prefix masks (inside the band)
  uint32_t tau_ref = 0, tau_bl = 0, tau_bh = 0;                  // Q[r_ref]; Q[zb - 1]; Q[min(zb + 128, s) - 1]
  bool has_bl = false;
  const int zb_top = max(0, s - BAND + 1);                    // highest band start: its last rank is the "R = s" sentinel
  auto set_band = [&](float wo_, float centre_sig) -> int {      // first band around where the pivot is expected for windows with wo_ window-only hashes; returns that estimate
    const float pq = (float)s / ((float)s + wo_);
    const float sigma = sqrtf((float)s * pq * (1.0f - pq) * (1.0f - pq));
    const int est = (int)((float)s * pq);
    const int centre = est + (int)(centre_sig * sigma);
    zb = max(0, min(centre - BAND / 2, zb_top));
    r_ref = max(zb, min(centre + max(8, (int)(L2Z_REF_SIG * sigma)), min(zb + BAND, s) - 1));
    return est;
  };
  auto band_thresholds = [&]() {
    tau_bh = (uint32_t)__builtin_amdgcn_readfirstlane((int)qat(min(zb + BAND, s) - 1));   // (wave-uniform: compared from scalar registers)
    has_bl = zb > 0;
    tau_bl = has_bl ? (uint32_t)__builtin_amdgcn_readfirstlane((int)qat(zb - 1)) : 0u;
    tau_ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)qat(r_ref));
  };
  bool fused = false;
  int r_pred = 0;
  {
    const int hc = cand_hint ? cand_hint[c] : 0;
    if (hc > 0 && M >= 2) {
      const int span = max(pw_wpos(pos[M - 1].pw) - pw_wpos(pos[0].pw), 1);
      const float we = (float)M * (float)cnt / (float)span;
      const float wo_p = fmaxf(we - (float)hc, 0.0f);
      r_pred = set_band(wo_p, L2Z_CENTRE_SIG_PRED);
      band_thresholds();
      fused = true;
    }
  }

  // ---- pass A: membership.  Bit table per entry; what passes it goes through the ring and is searched, 64 at a time --------------
  if (dbg_stop == 11) { release_slot(); return; }                 // (the wave's preamble alone: range searches, scratch slot, band prediction)
  int n_ml = 0;
  {
    int head = 0, tail = 0, carry = 0, carryLo = 0, carryA = 0;
    uint64_t rRef = 0, rBH = 0, rBL = 0, rNF = 0;                // (fused: the four ballots of pass B, parked per word — see pass_low below)
    mL[lane] = 0;
    wave_sync();
    auto dense = [&](int n) {
      wave_sync();
      const int slot_ = (head + lane) & (QCAP - 1);
      const uint32_t eh = RQh[slot_];
      const uint32_t ej = RQj[slot_];
      const int code = classify(eh);
      const bool hit = lane < n && code >= 0;
      const uint64_t hm = __ballot(hit);
      if (hit) {
        ML[mbcnt64(hm, n_ml)] = ej | ((uint32_t)code << 15);
        const uint32_t jg = ej & 4095u;
        atomicOr(&((uint32_t*)mL)[jg >> 5], 1u << (jg & 31));
      }
      n_ml += __popcll(hm);
      head += n;
    };
    // Long-read classes (the sketch is searched in global memory, five or six dependent loads of ~1 us): two ring entries per lane, their searches
    // interleaved, so that a step of 128 entries waits as long as one of 64
    auto dense2 = [&](int n) {
      wave_sync();
      const int s0 = (head + lane) & (QCAP - 1), s1 = (head + 64 + lane) & (QCAP - 1);
      const uint32_t h0 = RQh[s0], h1 = RQh[s1];
      const uint32_t j0_ = RQj[s0], j1_ = RQj[s1];
      const int k0 = l2_bucket(h0, tshift), k1 = l2_bucket(h1, tshift);
      int lo0 = T[k0], hi0 = T[k0 + 1], lo1 = T[k1], hi1 = T[k1 + 1];
      for (int it = 0; it < tsteps; ++it) {
        const int m0 = min((lo0 + hi0) >> 1, s - 1), m1 = min((lo1 + hi1) >> 1, s - 1);
        const uint32_t v0 = Qg[m0], v1 = Qg[m1];
        if (lo0 < hi0) { if (v0 < h0) lo0 = m0 + 1; else hi0 = m0; }
        if (lo1 < hi1) { if (v1 < h1) lo1 = m1 + 1; else hi1 = m1; }
      }
      const uint32_t e0 = Qg[min(lo0, s - 1)], e1 = Qg[min(lo1, s - 1)];
      const bool hit0 = lane < n && lo0 < s && e0 == h0, hit1 = 64 + lane < n && lo1 < s && e1 == h1;
      const uint64_t hm0 = __ballot(hit0), hm1 = __ballot(hit1);
      if (hit0) { ML[mbcnt64(hm0, n_ml)] = j0_ | ((uint32_t)lo0 << 15); const uint32_t jg = j0_ & 4095u; atomicOr(&((uint32_t*)mL)[jg >> 5], 1u << (jg & 31)); }
      n_ml += __popcll(hm0);
      if (hit1) { ML[mbcnt64(hm1, n_ml)] = j1_ | ((uint32_t)lo1 << 15); const uint32_t jg = j1_ & 4095u; atomicOr(&((uint32_t*)mL)[jg >> 5], 1u << (jg & 31)); }
      n_ml += __popcll(hm1);
      head += n;
    };
    auto drain = [&](bool all) {                                 // the ring down to less than a step's worth (all: empty)
      if constexpr (QLDS) { while (tail - head >= 64) dense(64); if (all && tail > head) dense(tail - head); }
      else { while (tail - head >= 128) dense2(128); if (all && tail > head) dense2(tail - head); }
    };
    // The eight words of a step without control flow between them: eight table reads in flight, eight ballots, the parks of the fused masks; the ring
    // takes what passed afterwards.  LAST: the step that holds the end of the stream (entries behind it are masked out there and nowhere else).
    auto step = [&](const Rec (&x)[8], int wd0, auto last_tag) __attribute__((always_inline)) {
      constexpr bool LAST = decltype(last_tag)::value;
      uint32_t bw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) bw[i] = BL[(x[i].hash >> 5) & bm];
      bool ps[8]; uint64_t pmk[8];
      int total = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t h = x[i].hash;
        ps[i] = ((bw[i] >> (h & 31)) & (bw[i] >> ((h >> 20) & 31)) & 1u) != 0u;
        if (LAST) ps[i] = ps[i] && (wd0 + i) * 64 + lane < M;
        pmk[i] = __ballot(ps[i]);
        total += __popcll(pmk[i]);
      }
      if (fused) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int wd = wd0 + i;
          if (LAST && wd >= nwords) continue;                    // (wave-uniform)
          const uint32_t h = x[i].hash;
          uint64_t le_bh = __ballot(h <= tau_bh);
          if (LAST) le_bh &= valid_mask(wd * 64);
          park64x3(rBH, le_bh, rRef, __ballot(h <= tau_ref), rBL, has_bl ? __ballot(h <= tau_bl) : 0ull, wd & 63);
          const uint64_t nf = __ballot((x[i].pw & PW_DP) != 0u);
          if (nf) rNF = park64(rNF, nf, wd & 63);
        }
      }
      const bool room = tail - head + total <= QCAP;         // (otherwise — more than a third of the entries passed the table — the ring is emptied after every word)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (ps[i]) { const int sl = (mbcnt64(pmk[i]) + tail) & (QCAP - 1); RQh[sl] = x[i].hash; RQj[sl] = (uint16_t)((wd0 + i) * 64 + lane); }
        tail += __popcll(pmk[i]);
        if (!room) drain(false);
      }
      drain(false);
    };
    Rec nx[8];
    load8(nx, 0);
    for (int wd0 = 0; wd0 < nwords; wd0 += 8) {
      Rec x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = nx[i];
      if (wd0 + 8 < nwords) { load8(nx, (wd0 + 8) * 64); step(x, wd0, std::false_type{}); }
      else step(x, wd0, std::true_type{});
      if (((wd0 + 8) & 63) == 0 || wd0 + 8 >= nwords) {          // end of a group of 64 words: its matched bits are complete once the ring is empty
        drain(true);
        wave_sync();
        const uint64_t reg = mL[lane];
        const int g0 = wd0 & ~63;
        const int cbits = __popcll(reg);
        const int ex = carry + wave_excl_scan(cbits, lane);
        // position of the first entry of every word of the group (just streamed: the lines are in the caches)
        const int w0reg = g0 + lane < nwords ? pw_wpos(pos[(g0 + lane) * 64].pw) : 0x7fffffff;
        mAll[g0 + lane] = reg; pAll[g0 + lane] = (uint16_t)ex; W0[g0 + lane] = w0reg;
        carry = __builtin_amdgcn_readlane(ex, 63) + __builtin_amdgcn_readlane(cbits, 63);
        if (fused) {
          rRef &= rBH;
          const uint64_t rLo = rRef & reg, rA = rRef & ~reg & ~rNF;
          const int cl = __popcll(rLo), ca = __popcll(rA);
          const int exl = carryLo + wave_excl_scan(cl, lane), exa = carryA + wave_excl_scan(ca, lane);
          mLo[g0 + lane] = rLo; pLo[g0 + lane] = (uint16_t)exl;
          mA[g0 + lane] = rA; pA[g0 + lane] = (uint16_t)exa;
          mZ[g0 + lane] = rBH & ~rBL; mX[g0 + lane] = rBH & rNF;
          carryLo = __builtin_amdgcn_readlane(exl, 63) + __builtin_amdgcn_readlane(cl, 63);
          carryA = __builtin_amdgcn_readlane(exa, 63) + __builtin_amdgcn_readlane(ca, 63);
          rRef = rBH = rBL = rNF = 0;
        }
        mL[lane] = 0;
        wave_sync();
      }
    }
    if (fused) { close_prefix(mLo, pLo, carryLo); close_prefix(mA, pA, carryA); }
    close_prefix(mAll, pAll, carry);
  }
  lap(1);
  if (dbg_stop == 2) { release_slot(); return; }

  // ---- blocks of `bspan` window starts; e_min of every block start (first entry with wpos >= wpos[block start] + cnt) ----------------
  int bwl = 0;
  while (((nwords + (1 << bwl) - 1) >> bwl) > L2_NBLK_MAX) ++bwl;
  const int nblk = (nwords + (1 << bwl) - 1) >> bwl;
  const int bspan = 64 << bwl;
  // Per block (128 at most): e_min of its first window start and its bound live in LDS — where W0 lay, which is done with once the searches below are —
  // and the three flags per block in the slide's spare scratch words.  (As lane-distributed registers, two blocks per lane, these seven arrays were live
  // through the whole sweep, and the allocator spilled ~36 registers around it for every candidate.)
  uint16_t* const eLoL = (uint16_t*)W0;                          // [128]
  uint16_t* const ub2L = eLoL + 128;                             // [128] bound + 1 (0: no bound / not eligible)
  uint32_t* const fUpW = (uint32_t*)tst;                         // [4] blocks with a window whose pivot lies above the current band
  uint32_t* const fDnW = fUpW + 4;                               // [4] ... below it
  uint32_t* const eligW = fDnW + 4;                              // [4] blocks this pass may visit
  auto e_of = [&](int kk) -> int { return __builtin_amdgcn_readfirstlane((int)eLoL[kk]); };
  auto ub_of = [&](int kk) -> int { return __builtin_amdgcn_readfirstlane((int)ub2L[kk]) - 1; };
  int ubmax = -1, wo_est = 0;
  {
    int tg[2], lo[2], hi[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int bq = lane + 64 * qq;
      const int v = bq < nblk ? W0[bq << bwl] : 0x7fffffff;
      tg[qq] = v == 0x7fffffff ? v : v + cnt;
      lo[qq] = 0; hi[qq] = nwords;                               // number of words whose first entry lies below the target
    }
    constexpr int WSTEPS = NWQ == 2 ? 8 : 10;                    // log2(64 NWQ) + 1
    for (int it = 0; it < WSTEPS; ++it) {
      const int m0 = min((lo[0] + hi[0]) >> 1, nwords - 1), m1 = min((lo[1] + hi[1]) >> 1, nwords - 1);
      const int p0 = W0[m0], p1 = W0[m1];
      if (lo[0] < hi[0]) { if (p0 < tg[0]) lo[0] = m0 + 1; else hi[0] = m0; }
      if (lo[1] < hi[1]) { if (p1 < tg[1]) lo[1] = m1 + 1; else hi[1] = m1; }
    }
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) { lo[qq] = 64 * max(lo[qq] - 1, 0); hi[qq] = min(lo[qq] + 64, last_end); }
    // the entry inside that word (64 entries): two rounds of eight probes each, all sixteen loads of a round in flight, instead of seven dependent ones
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      const int step = rnd == 0 ? 8 : 1;
      int pv[2][8];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int t = 0; t < 8; ++t) pv[qq][t] = pw_wpos(pos[min(lo[qq] + (t + 1) * step - 1, nmax)].pw);
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        int adv = 0;                                             // probes t with a position below the target (they come first: positions ascend)
#pragma unroll
        for (int t = 0; t < 8; ++t) adv += (lo[qq] + (t + 1) * step - 1 < hi[qq] && pv[qq][t] < tg[qq]) ? 1 : 0;
        lo[qq] = min(lo[qq] + adv * step, hi[qq]);
      }
    }
    int eLo[2], eHi[2], ub_all[2];
    eLo[0] = lane < nblk ? lo[0] : last_end;
    eLo[1] = lane + 64 < nblk ? lo[1] : last_end;
    // per block: largest window [bF, eHi), smallest window [bL, eLo)   (lane l owns blocks l and l+64)
    const int up0 = __shfl_down(eLo[0], 1, 64), up1 = __shfl_down(eLo[1], 1, 64);
    const int e64 = __builtin_amdgcn_readlane(eLo[1], 0);
    eHi[0] = lane == 63 ? e64 : up0;
    eHi[1] = lane == 63 ? last_end : up1;
    for (int q = 0; q < 2; ++q) {
      const int bq = lane + 64 * q;
      ub_all[q] = -1;
      if (bq < nblk) {
        const int bF = bq * bspan, bL = min(bF + bspan - 1, last_end - 1);
        if (!(bL + 1 < last_end)) eHi[q] = last_end;
        if (eLo[q] < last_end) ub_all[q] = pfx(mAll, pAll, eHi[q]) - pfx(mAll, pAll, bF);
      } else eLo[q] = eHi[q] = last_end;
    }
    ubmax = wave_max(max(ub_all[0], ub_all[1]));
    // the block with the most matched entries has the fewest window-only ones: what its smallest window holds besides them
    const int key = max(ub_all[0], ub_all[1]) == ubmax ? ((ub_all[0] == ubmax) ? lane : lane + 64) : 1 << 20;
    const int bkb = wave_min(key);
    const int bLb = min(bkb * bspan + bspan - 1, last_end - 1);
    const int eb = __builtin_amdgcn_readlane(bkb < 64 ? eLo[0] : eLo[1], bkb & 63);
    wo_est = max(eb - bLb - ubmax, 0);
    wave_sync();                                                 // (W0 has been read for the last time)
    eLoL[lane] = (uint16_t)eLo[0]; eLoL[lane + 64] = (uint16_t)eLo[1];
    if (lane < 4) { fUpW[lane] = 0; fDnW[lane] = 0; eligW[lane] = 0xffffffffu; }
    wave_sync();
  }
  lap(2);
  if (dbg_stop == 3) { release_slot(); return; }

  // ---- window state and trackers ------------------------------------------------------------------------------------------------
  int b = 0, e = 0;
  int z0 = 0, cbase = 0, sb = 0, fz = 0;                         // zone [z0, z0 + 64) inside the band, and the state of the current window in it
  uint64_t pm = 0;
  uint32_t Qz = 0xffffffffu, tau_lo = 0, tau_hi = 0;            // Q[z0 + lane]; Q[z0 - 1]; Q[min(z0 + 64, s) - 1]
  bool has_lo = false;
  int best = 0, bestR = 0, beg_pos = 0, last_pos = 0, opt_b = 0, opt_e = 0, last_b = 0;
  uint32_t evals = 0, rebuilds = 0, rounds = 0;                   // (diagnostics of the result record: 32 bits there too)
  int zdir = 0, n_pass = 0, n_low = 0;                                      // 0: first pass, +1: bands above it, -1: bands below it
  auto flag_block = [&](uint32_t* f, int kk) { if (lane == 0) atomicOr(&f[kk >> 5], 1u << (kk & 31)); };
  // pass B: the masks of the band and its reference rank from a second reading of the stream (l2z_pass_b above)
  auto pass_low = [&]() {
    band_thresholds();
    const L2ZPassB pb{pos, M, nmax, nwords, tau_bh, tau_ref, tau_bl, has_bl ? 1 : 0, mAll, mLo, mA, mZ, mX, pLo, pA};
    l2z_pass_b(pb);
  };
  auto fill_band_hashes = [&]() {                                // the band's hashes for the rank searches of the rebuilds
#pragma unroll
    for (int t = 0; t < BPL; ++t) BQ[lane + 64 * t] = zb + lane + 64 * t < s ? qat(zb + lane + 64 * t) : 0xffffffffu;
    wave_sync();
  };

  // State of window [nb, ne) from the masks.  The prefix sums give the distinct entries at or below Q[r_ref]; the band's entries of the window
  // (and the entries with an earlier occurrence in their contig, which count once per window: slidingMap.hpp:148-157) are visited one by one,
  // a word per lane: gap counts and presence bits over the band -> pivot -> the zone is centred on it.
  // A pivot below / above the band leaves the zone at that edge of the band: the slide then sees the window as out of reach and flags its block.
  auto rebuild_state = [&](int nb, int ne) __attribute__((always_inline)) {
    ++rebuilds;
    const int cb_ref = pfx(mA, pA, ne) - pfx(mA, pA, nb);
    const int sb_ref = pfx(mLo, pLo, ne) - pfx(mLo, pLo, nb);
#pragma unroll
    for (int t = 0; t < BPL; ++t) zc[lane + 64 * t] = 0;
    if (lane < BAND / 32) pmw[lane] = 0;
    wave_sync();
    int acc2 = 0;                                                // window-only at or below Q[r_ref] whose earlier occurrence lies outside the window | matched ones whose earlier occurrence lies inside << 16
    const int w_lo = nb >> 6, w_hi = (ne - 1) >> 6;
    auto first_in_window = [&](uint64_t need, int j, uint32_t h, int& dres) {   // saturated distances (windows of 65535+ entries; tests lower dup_sat): scan
      while (need) {
        const int l = __builtin_ctzll(need); need &= need - 1;
        const int jj = __builtin_amdgcn_readlane(j, l);
        const bool dup = wave_has_hash(pos, nb, jj, (uint32_t)__builtin_amdgcn_readlane((int)h, l), lane);
        if (lane == l) dres = dup ? 1 : 0;
      }
    };
    for (int wb = w_lo; wb <= w_hi; wb += 64) {
      const int wd = wb + lane;
      const bool inr = wd <= w_hi;
      uint64_t range = ~0ull;
      if (wd == w_lo) range &= ~0ull << (nb & 63);
      if (wd == w_hi) { const int rr = ne - (wd << 6); if (rr < 64) range &= (1ull << rr) - 1ull; }
      uint64_t z = inr ? mZ[wd] & range : 0ull;
      uint64_t xx = inr ? mX[wd] & range : 0ull;
      const uint64_t ma = inr ? mAll[wd] : 0ull;
      const uint64_t xflag = xx;
      while (__ballot(z != 0ull) != 0ull) {                      // four band entries per lane and step: their hashes are in flight together, the searches interleaved
        bool a[4]; int j[4]; uint32_t h[4]; bool mtc[4], flg[4]; int bi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] = z != 0ull;
          const int bit = a[u] ? __builtin_ctzll(z) : 0;
          if (a[u]) z &= z - 1;
          j[u] = (wd << 6) + bit;
          mtc[u] = (ma >> bit) & 1ull; flg[u] = a[u] && ((xflag >> bit) & 1ull);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) h[u] = a[u] ? pos[j[u]].hash : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) bi[u] = 0;                   // rank inside the band: number of the band's hashes below h (a matched hash finds itself)
#pragma unroll
        for (int st = BAND / 2; st >= 1; st >>= 1) {
          uint32_t v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = BQ[bi[u] + st - 1];
#pragma unroll
          for (int u = 0; u < 4; ++u) bi[u] += v[u] < h[u] ? st : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bi[u] += BQ[bi[u]] < h[u] ? 1 : 0;
          int dres = 0;
          if (__ballot(flg[u]) != 0ull) {                        // (an earlier occurrence in the contig: rare outside repeats)
            dres = flg[u] ? dup_before(I, first0 + j[u], (int64_t)j[u] - nb) : 0;
            first_in_window(__ballot(dres < 0), j[u], h[u], dres);
          }
          if (a[u] && dres == 0) {
            if (mtc[u]) atomicOr(&pmw[bi[u] >> 5], 1u << (bi[u] & 31));
            else atomicAdd(&zc[bi[u]], 1u);
          }
        }
      }
      while (__ballot(xx != 0ull) != 0ull) {
        const bool a = xx != 0ull;
        const int bit = a ? __builtin_ctzll(xx) : 0;
        if (a) xx &= xx - 1;
        const int j = (wd << 6) + bit;
        const bool mtc = (ma >> bit) & 1ull;
        const uint32_t h = a ? pos[j].hash : 0xffffffffu;
        int dres = a ? dup_before(I, first0 + j, (int64_t)j - nb) : 0;
        first_in_window(__ballot(dres < 0), j, h, dres);
        if (a && h <= tau_ref) { if (mtc) { if (dres > 0) acc2 += 1 << 16; } else { if (dres == 0) acc2 += 1; } }
      }
    }
    wave_sync();
    // inclusive prefixes of the band's gap counters (BPL consecutive ones per lane), written back in place
    int inc[BPL];
    {
      int run = 0;
#pragma unroll
      for (int t = 0; t < BPL; ++t) { run += (int)zc[BPL * lane + t]; inc[t] = run; }
      const int before = wave_incl_scan(run) - run;
#pragma unroll
      for (int t = 0; t < BPL; ++t) inc[t] += before;
    }
    const int acc2T = __builtin_amdgcn_readlane(wave_incl_scan(acc2), 63);
    wave_sync();
#pragma unroll
    for (int t = 0; t < BPL; ++t) zc[BPL * lane + t] = (uint32_t)inc[t];
    wave_sync();
    auto band_bits = [&](int wq) -> uint64_t {                   // 64 presence bits of the band from bit 64 wq on (wave-uniform)
      return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)pmw[2 * wq]) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)pmw[2 * wq + 1]) << 32);
    };
    auto popc_below = [&](int n) -> int {                        // matched band ranks present with band index < n
      int c_ = 0;
#pragma unroll
      for (int wq = 0; wq < BPL; ++wq) {
        const int k_ = n - 64 * wq;
        if (k_ > 0) { const uint64_t m_ = band_bits(wq); c_ += __popcll(k_ >= 64 ? m_ : (m_ & ((1ull << k_) - 1ull))); }
      }
      return c_;
    };
    const int kref = r_ref - zb;
    const int cb_lo = cb_ref + (acc2T & 0xffff) - __builtin_amdgcn_readfirstlane((int)zc[kref]);          // C(zb - 1): distinct window-only hashes below the band
    const int sb_lo = sb_ref - (acc2T >> 16) - popc_below(kref + 1);                                       // distinct matched hashes below the band
    // pivot: first band index i with (zb + i) + C(zb + i) >= s; ranks from s on are the "R = s" sentinel
    int mine_first = BAND;
#pragma unroll
    for (int t = BPL - 1; t >= 0; --t) { const int i_ = BPL * lane + t; if (zb + i_ >= s || zb + i_ + cb_lo + inc[t] >= s) mine_first = i_; }
    const int Rb = wave_min(mine_first);
    const bool below = has_bl && zb - 1 + cb_lo >= s;
    const int zrel = below ? 0 : (Rb >= BAND ? BAND - 64 : max(0, min(Rb - 32, BAND - 64)));
    z0 = zb + zrel;
    const int czl = zrel > 0 ? __builtin_amdgcn_readfirstlane((int)zc[zrel - 1]) : 0;
    cbase = cb_lo + czl;
    const int rz = z0 + lane;
    fz = rz < s ? rz + (int)zc[zrel + lane] - czl : (1 << 29);
    sb = sb_lo + popc_below(zrel);
    {
      const int wq = zrel >> 6, sh = zrel & 63;
      const uint64_t m0_ = band_bits(wq), m1_ = (sh && wq + 1 < BPL) ? band_bits(wq + 1) : 0ull;
      pm = sh ? ((m0_ >> sh) | (m1_ << (64 - sh))) : m0_;
    }
    Qz = BQ[zrel + lane];
    has_lo = z0 > 0;
    tau_lo = zrel > 0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)BQ[zrel - 1]) : tau_bl;
    tau_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)BQ[min(zrel + 64, s - zb) - 1]);
  };

  // ---- 64 consecutive windows per round, one lane per window (mm_l2.hpp, block_slide: the merge of the leave / enter times) ----------
  auto rank_search = [&](int arr, int v) -> int {
    int lo = 0;
    for (int st = 32; st >= 1; st >>= 1) { const int x = __shfl(arr, lo + st - 1, 64); if (x < v) lo += st; }
    const int x = __shfl(arr, lo, 64);
    return lo + (x < v ? 1 : 0);
  };
  bool pending_rebuild = false, run_stop = false;
  int bk = 0, j0 = 0, stage = 0, blk_end = 0x7fffffff;
  auto block_slide = [&]() __attribute__((always_inline)) {
    constexpr int INF = 0x7fffffff;
    while (e < last_end && b < last_end) {
      if (pending_rebuild) { lap(4); rebuild_state(b, e); pending_rebuild = false; lap(3); if (dbg_stop == 8) break; }
      ++rounds;
      const Rec xb = pos[min(b + lane, nmax)];
      const Rec xe = pos[min(e + lane, nmax)];
      const int w64 = pw_wpos(pos[min(b + 64, nmax)].pw);
      const int ob = b + lane, oe = e + lane;
      const bool mB = (mAll[min(ob >> 6, NW)] >> (ob & 63)) & 1ull, mE = (mAll[min(oe >> 6, NW)] >> (oe & 63)) & 1ull;
      const int wpb = pw_wpos(xb.pw);
      int nextw = __shfl_down(wpb, 1, 64);
      if (lane == 63) nextw = w64;
      const int tA = (b + lane + 1 < last_end) ? nextw : INF;
      const int tB = (e + lane < last_end) ? pw_wpos(xe.pw) - (cnt - 1) : INF;
      // cross ranks and ties
      const int nB = rank_search(tB, tA), nA = rank_search(tA, tB);
      const int tB_at = __shfl(tB, min(nB, 63), 64);
      const bool tie = tA != INF && tB_at == tA && nB < 64;
      const int tie_ex = wave_excl_scan(tie ? 1 : 0, lane);
      const int ties_all = __builtin_amdgcn_readlane(tie_ex, 63) + (__builtin_amdgcn_readlane((int)tie, 63) ? 1 : 0);
      const int tie_at = __shfl(tie_ex, min(nA, 63), 64);
      const int t_lim = min(__builtin_amdgcn_readlane(tA, 63), __builtin_amdgcn_readlane(tB, 63));
      const bool okA = tA != INF && tA <= t_lim, okB = tB != INF && tB <= t_lim;
      const int kA = okA ? lane + nB - tie_ex : (1 << 20);
      const int kB = okB ? lane + nA - (nA < 64 ? tie_at : ties_all) : (1 << 20);
      const int ksteps = wave_max(max(okA ? kA + 1 : 0, okB ? kB + 1 : 0));
      fdel[lane] = 0; fadd[lane] = 0;
      wave_sync();
      if (kA < 64) fdel[kA] = 1;
      if (kB < 64) fadd[kB] = 1;
      wave_sync();
      const int hasDel = fdel[lane], hasAdd = fadd[lane];
      const int dj = wave_excl_scan(hasDel, lane), aj = wave_excl_scan(hasAdd, lane);   // lane j: window j = [b+dj, e+aj)
      const bool cond = lane < ksteps && (e + aj < last_end) && (b + dj < last_end);
      const uint64_t cm = __ballot(cond);
      const int n_eval = (~cm == 0ull) ? 64 : __builtin_ctzll(~cm);   // windows 0 .. n_eval-1 can be visited (n_eval >= 1)
      // events that count: hash at or below the zone's top, and distinct inside their window (flagged entries are rare)
      bool vE = xe.hash <= tau_hi && kB < n_eval, vB = xb.hash <= tau_hi && kA < n_eval;
      {
        const bool fE = vE && (xe.pw & PW_DP), fB = vB && (xb.pw & PW_DN);
        if (__ballot(fE || fB) != 0ull) {
          const int kEc = min(kB, 63), kBc = min(kA, 63);
          const int hbL = b + __shfl(dj, kEc, 64) + __shfl(hasDel, kEc, 64), weL = e + __shfl(aj, kBc, 64);
          const int rE = fE ? dup_before(I, first0 + e + lane, (int64_t)(e + lane) - hbL) : 0;
          if (rE > 0) vE = false;
          uint64_t fm = __ballot(rE < 0);
          while (fm) {
            const int l = __builtin_ctzll(fm); fm &= fm - 1;
            const int kk = __builtin_amdgcn_readlane(kB, l);
            const int hb = b + __builtin_amdgcn_readlane(dj, kk) + __builtin_amdgcn_readlane(hasDel, kk);
            const bool dup = wave_has_hash(pos, hb, e + l, (uint32_t)__builtin_amdgcn_readlane((int)xe.hash, l), lane);
            if (dup && lane == l) vE = false;
          }
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
      const bool loE = has_lo && xe.hash <= tau_lo, loB = has_lo && xb.hash <= tau_lo;
      const int indE = (vE && loE) ? (mE ? 1 << 16 : 1) : 0;
      const int indB = (vB && loB) ? (mB ? 1 << 16 : 1) : 0;
      const int pE = wave_incl_scan(indE), pB = wave_incl_scan(indB);
      const int gE_ = __shfl(pE, max(aj - 1, 0), 64), gB_ = __shfl(pB, max(dj - 1, 0), 64);
      const int accE = aj > 0 ? gE_ : 0, accB = dj > 0 ? gB_ : 0;
      const int cbase_j = cbase + (accE & 0xffff) - (accB & 0xffff);
      const int sb_j = sb + (accE >> 16) - (accB >> 16);
      const int thr = s - cbase_j;
      auto pivot_of = [&]() -> int { return rank_search(fz, thr); };
      int pj = pivot_of();
      int pc = pj;                                                // the lane's pivot under the zone events applied so far
      uint64_t pm_j = pm;
      uint64_t zE = __ballot(vE && !loE), zB = __ballot(vB && !loB);   // zone events in step order (each one changes the windows after its step)
      const uint64_t mEm = __ballot(mE), mBm = __ballot(mB);
      while (zE | zB) {
        const int lE = zE ? __builtin_ctzll(zE) : 0, lB = zB ? __builtin_ctzll(zB) : 0;
        const int kE_ = zE ? __builtin_amdgcn_readlane(kB, lE) : INF, kB_ = zB ? __builtin_amdgcn_readlane(kA, lB) : INF;
        const bool takeB = kB_ <= kE_;                           // the deletion of a step comes first
        const uint32_t hh = (uint32_t)(takeB ? __builtin_amdgcn_readlane((int)xb.hash, lB) : __builtin_amdgcn_readlane((int)xe.hash, lE));
        const bool mtc = takeB ? ((mBm >> lB) & 1ull) : ((mEm >> lE) & 1ull);
        const int kk = takeB ? kB_ : kE_;
        const int sign = takeB ? -1 : 1;
        if (takeB) zB &= zB - 1; else zE &= zE - 1;
        const int zi = __popcll(__ballot(Qz < hh));              // rank inside the zone
        if (mtc) {
          const uint64_t bit = 1ull << zi;
          pm ^= bit;
          if (lane > kk) pm_j ^= bit;
        } else {
          // fz[r] = r + C(r) is strictly increasing and the event moves it by one from rank zi on: the first rank that reaches the lane's threshold
          // moves by one rank at most (mm_l2_core.hpp, l2_wonly_event) — one look at the neighbour instead of a search of seven steps
          fz += (lane >= zi) ? sign : 0;
          const int at = sign > 0 ? pc - 1 : pc;
          const int fv = __shfl(fz, min(max(at, 0), 63), 64);
          if (sign > 0) { if (pc > 0 && fv >= thr) --pc; }
          else { if (pc < 64 && fv < thr) ++pc; }
          if (lane > kk) pj = pc;
        }
      }
      // A window whose pivot left the zone: the zone is re-centred there if the band still holds the pivot (the windows before it are
      // scored first); at an edge of the band the window is not scored in this pass — its block is visited again with the next band.
      const bool inw = lane < n_eval;
      const bool outU = inw && pj >= 64;
      const bool outD = inw && has_lo && thr <= z0 - 1;           // (z0 - 1) + C(z0 - 1) >= s: the pivot lies below z0
      const uint64_t um = __ballot(outU), dm = __ballot(outD);
      const bool canU = z0 + 64 < zb + BAND, canD = z0 > zb;
      const uint64_t cut = (canU ? um : 0ull) | (canD ? dm : 0ull);
      const int n_vis = cut ? __builtin_ctzll(cut) : n_eval;     // windows 0 .. n_vis-1 are visited in this round
      const uint64_t vis = n_vis >= 64 ? ~0ull : ((1ull << n_vis) - 1ull);
      const uint64_t fu = canU ? 0ull : (um & vis), fd = canD ? 0ull : (dm & vis);
      if ((fu && zdir >= 0) || (fd && zdir <= 0)) {
        const int bk2 = (b + 63 >= blk_end && bk + 1 < nblk) ? bk + 1 : bk;
        if (fu && zdir >= 0) { flag_block(fUpW, bk); flag_block(fUpW, bk2); }
        if (fd && zdir <= 0) { flag_block(fDnW, bk); flag_block(fDnW, bk2); }
      }
      const bool scored = lane < n_vis && !outU && !outD;
      const int sh_j = scored ? sb_j + __popcll(pm_j & ((1ull << (pj & 63)) - 1ull)) : -1;
      const int m = wave_max(sh_j);
      if (m >= 0) {
        const uint64_t at = __ballot(sh_j == m);
        const int j1 = __builtin_ctzll(at), jl = 63 - __builtin_clzll(at);
        // trackers by position, not by evaluation order: the first window reaching the maximum (:510-518) and the last one equal to it (:520-524)
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
        evals += (uint32_t)__popcll(__ballot(scored));
      }
      // state of window n_vis
      int dn, an;
      if (n_vis < 64) { dn = __builtin_amdgcn_readlane(dj, n_vis); an = __builtin_amdgcn_readlane(aj, n_vis); }
      else { dn = __builtin_amdgcn_readlane(dj, 63) + __builtin_amdgcn_readlane(hasDel, 63); an = __builtin_amdgcn_readlane(aj, 63) + __builtin_amdgcn_readlane(hasAdd, 63); }
      b += dn; e += an;
      while (b >= blk_end) {                                     // entered the next block: does its bound still pass?
        ++bk;
        if (bk >= nblk || (stage == 1 && bk == j0) || ub_of(bk) < max(best, amin)) { run_stop = true; break; }
        blk_end += bspan;
      }
      if (run_stop) break;
      if (cut) pending_rebuild = true;
      else {
        const int fE = an > 0 ? __builtin_amdgcn_readlane(pE, an - 1) : 0, fB = dn > 0 ? __builtin_amdgcn_readlane(pB, dn - 1) : 0;
        cbase += (fE & 0xffff) - (fB & 0xffff);
        sb += (fE >> 16) - (fB >> 16);
      }
    }
  };

  // ---- bands: the first one around the expected pivot of the most promising block, then those above and below it that windows asked for ----
  int zb_first = 0, r_est = 0;
  if (dbg_stop == 9) { release_slot(); return; }
  bool any_pass = ubmax >= amin;                                 // otherwise no window can reach the acceptance threshold
  bool masks_ready = false;                                      // the band's masks came out of pass A
  if (any_pass) {
    const int wo = wo_est;
    const int zb_p = zb, r_ref_p = r_ref;
    r_est = set_band((float)wo, L2Z_CENTRE_SIG);                  // (what the matched counts ask for)
    if (fused) {
      // the predicted band stands if the centre this estimate asks for lies within its middle half (or both are clamped to the same edge)
      const int want = zb + BAND / 2;
      masks_ready = zb == zb_p || (want >= zb_p + BAND / 4 && want < zb_p + 3 * BAND / 4);
      if (masks_ready) { zb = zb_p; r_ref = r_ref_p; }
    }
    zb_first = zb;
  }
  while (any_pass) {
    if (++n_pass > (s >> 6) + 4) break;                            // (cannot happen: every pass moves the band by its width in one direction)
    if (!masks_ready) { pass_low(); ++n_low; }
    masks_ready = false;
    if (dbg_stop == 6) { release_slot(); return; }
    fill_band_hashes();
    lap(5);
    if (dbg_stop == 7) { release_slot(); return; }
    wave_sync();
    int ub2[2];
    int lane_b = lane;
    asm volatile("" : "+v"(lane_b));                            // (a lane number the compiler cannot see through: everything below that depends on the lane alone was hoisted out of the band
                                                                 //  loop — which runs once for nine candidates in ten — and kept in scratch: 29 stores per candidate, 5.7 GB per batch)
    for (int q = 0; q < 2; ++q) {
      const int bq = lane_b + 64 * q;
      int u = -1;
      const int eLo_q = bq < nblk ? (int)eLoL[bq] : last_end;
      if (bq < nblk && eLo_q < last_end && ((eligW[bq >> 5] >> (bq & 31)) & 1u)) {
        const int bF = bq * bspan, bL = min(bF + bspan - 1, last_end - 1);
        const int eHi_q = (bL + 1 < last_end && bq + 1 < nblk) ? (int)eLoL[bq + 1] : last_end;   // largest window of the block: up to e_min of the next block's start
        const int a = eLo_q > bL ? pfx(mA, pA, eLo_q) - pfx(mA, pA, bL) : 0;
        // r_ref + a >= s: every window of the block has its pivot at or below r_ref, so it shares at most the matched entries at or below Q[r_ref]
        u = (r_ref + a >= s) ? pfx(mLo, pLo, eHi_q) - pfx(mLo, pLo, bF) : pfx(mAll, pAll, eHi_q) - pfx(mAll, pAll, bF);
      }
      ub2[q] = u;
    }
    ub2L[lane_b] = (uint16_t)(ub2[0] + 1); ub2L[lane_b + 64] = (uint16_t)(ub2[1] + 1);
    wave_sync();
    int bkmax = 0, done_hi = nblk;
    if (zdir == 0) {
      // the sweep starts a little before the block with the largest bound, so that the maximum is known early and
      // the rest (left flank afterwards, right flank on the way) is pruned against it
      const int u2max = wave_max(max(ub2[0], ub2[1]));
      const int key2 = max(ub2[0], ub2[1]) == u2max ? ((ub2[0] == u2max) ? lane_b : lane_b + 64) : 1 << 20;
      bkmax = wave_min(key2);
      j0 = bkmax;
      while (j0 > 0 && bkmax - j0 < 3 && 100 * ub_of(j0 - 1) >= 95 * u2max) --j0;
      stage = 0; bk = j0;
    } else { stage = 1; j0 = -1; bk = 0; }
    lap(2);
    if (dbg_stop == 4) { release_slot(); return; }
    for (;;) {
      // the next block whose bound reaches max(best so far, amin); everything else is provably below the maximum.
      // stage 0: from j0 to the first failing block behind bkmax; stage 1: all other blocks in index order.
      bool found = false;
      for (;;) {
        if (stage == 1 && bk == j0) bk = done_hi;
        if (bk >= nblk) { if (stage == 0) { done_hi = nblk; stage = 1; bk = 0; continue; } break; }
        if (ub_of(bk) >= max(best, amin)) { found = true; break; }
        if (stage == 0 && bk >= bkmax) { done_hi = bk + 1; stage = 1; bk = 0; continue; }
        ++bk;
      }
      if (!found) break;
      const int nb = bk * bspan;
      blk_end = nb + bspan; run_stop = false;
      b = nb; e = e_of(bk); pending_rebuild = true;
      lap(2);
      block_slide();
      lap(4);
      if (dbg_stop == 8) { release_slot(); return; }
      if (e >= last_end || bk >= nblk) {
        if (stage == 1) break;
        done_hi = nblk; stage = 1; bk = 0;
      }
    }
    // the next band: upwards while windows asked for it, then downwards from the first one
    wave_sync();
    const bool wantU = __ballot(lane < 4 && fUpW[lane & 3] != 0u) != 0ull, wantD = __ballot(lane < 4 && fDnW[lane & 3] != 0u) != 0ull;
    if (zdir >= 0 && wantU && zb < zb_top) {
      zdir = 1; zb = min(zb + BAND, zb_top);
      r_ref = min(zb + BAND, s) - 1;
      if (lane < 4) { eligW[lane] = fUpW[lane]; fUpW[lane] = 0; }
      continue;
    }
    if (zdir >= 0) { zdir = -1; zb = zb_first; }
    if (wantD && zb > 0) {
      zb = max(0, zb - BAND);
      r_ref = min(zb + BAND, s) - 1;
      if (lane < 4) { eligW[lane] = fDnW[lane]; fDnW[lane] = 0; }
      continue;
    }
    break;
  }

  // ---- K6 strand vote over the first optimal window (computeMap.hpp:424-433, slidingMap.hpp:232-254): the matched list holds all it needs ----
  lap(4);
  if (dbg_stop == 5) { release_slot(); return; }
  int strand = -1, accepted = 0;
  if (best >= amin) {
    accepted = 1;
    int votes = 0, amb_votes = 0;
    const int i0 = pfx(mAll, pAll, opt_b), i1 = pfx(mAll, pAll, opt_e);
    constexpr int VB = 8;                                        // (the best window of a 10 kb read holds 600-900 matched entries: two steps instead of four)
    for (int ib = i0; ib < i1; ib += 64 * VB) {                  // VB batches of 64 matched entries per step: their three dependent loads each in flight together
      uint32_t ew[VB], sq[VB], pwj[VB]; bool cnt_it[VB]; int jv[VB];
#pragma unroll
      for (int u = 0; u < VB; ++u) { const int i = ib + 64 * u + lane; ew[u] = i < i1 ? ML[i] : 0xffffffffu; }
#pragma unroll
      for (int u = 0; u < VB; ++u) {
        const int rk = (int)((ew[u] >> 15) & 0x7fffu);
        jv[u] = (int)(ew[u] & 0x7fffu);
        cnt_it[u] = ew[u] != 0xffffffffu && rk < bestR;
        sq[u] = cnt_it[u] ? (uint32_t)sk_strand[qo + rk] : 0u;   // bit 0 strand, bit 1 unresolved duplicate (mm_map.hip, K2)
        pwj[u] = cnt_it[u] ? pos[jv[u]].pw : 0u;
      }
#pragma unroll
      for (int u = 0; u < VB; ++u) {
        if (ib + 64 * u >= i1) continue;                         // (wave-uniform)
        const bool unres = (sq[u] & 2u) && amb_used != nullptr;
        const int contrib = cnt_it[u] ? (((sq[u] & 1u) ? 1 : -1) * pw_strand(pwj[u])) : 0;
        const bool flagged = cnt_it[u] && (pwj[u] & PW_DN);      // a later occurrence exists in the contig: inside the window?  (strandR is the LAST occurrence's, :155-156)
        int dres = 0;
        if (__ballot(flagged) != 0ull) dres = flagged ? dup_after(I, first0 + jv[u], (int64_t)opt_e - 1 - jv[u]) : 0;
        if (cnt_it[u] && dres == 0) { if (unres) ++amb_votes; else votes += contrib; }
        uint64_t fm = __ballot(dres < 0);
        while (fm) {
          const int l = __builtin_ctzll(fm); fm &= fm - 1;
          const int jj = __builtin_amdgcn_readlane(jv[u], l);
          const uint32_t hj = pos[jj].hash;
          const bool later = wave_has_hash(pos, jj + 1, opt_e, hj, lane);
          if (!later && lane == l) { if (unres) ++amb_votes; else votes += contrib; }
        }
      }
    }
    votes = wave_sum(votes);
    amb_votes = wave_sum(amb_votes);
    if (amb_votes > 0 && ((votes - amb_votes <= 0 && votes + amb_votes > 0) || (dbg_flags & 0x200)) && lane == 0) amb_used[r] = 1;
    strand = votes > 0 ? 1 : -1;
  }
  release_slot();
  if (lane == 0) {
    L2Result o;
    o.contig = contig; o.mean_pos = (beg_pos + last_pos) / 2;    // :537
    o.shared = best; o.strand = strand; o.accepted = accepted; o.pad = n_low;   // (pad: passes over the stream beyond the first, a diagnostic summed by l2_stats_kernel)
    o.opt_beg = first0 + opt_b; o.opt_end = first0 + opt_e;
    o.n_stream = (uint32_t)M; o.n_evals = evals; o.n_rebuilds = rebuilds; o.pad2 = rounds;
    if (dbg_flags & 0x400) { o.mean_pos = bestR - ((dbg_flags & 0x800) ? r_pred : r_est); o.shared = n_low; }   // MM_L2Z_DBG: where the pivot of the best window lay against the estimate (results are then meaningless)
    out[c] = o;
    lap(6);
    if (prof) for (int i = 0; i < 8; ++i) atomicAdd(&counters[3 + i], (unsigned long long)tphL[i]);   // MM_L2_PHASES only
  }
}

}  // namespace mm
