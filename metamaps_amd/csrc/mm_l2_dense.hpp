// K5 / K6 for long reads — the sliding MinHash window with its state in global memory (replaces Map::computeL2MappedRegions +
// the statistics part of doL2Mapping for reads whose sketch does not fit the LDS-resident classes of mm_l2.hpp;
// computeMap.hpp:396-538, slidingMap.hpp:26-318, MIIteratorL2.hpp:74-96).
//
// The LDS classes of l2_kernel keep the whole window state on chip, which costs them their occupancy as the sketch grows (one wave
// per workgroup from 16 384 hashes on) and ends at 32 768 hashes.  Here the work is split by what it needs:
//
//   l2_range_kernel   one wave per candidate: the streamed index range [first, lastEnd)            (searchIndex, :466, :477)
//   l2_codes_kernel   one workgroup per READ: the sorted sketch Q (LDS up to 32 768 hashes, else global) and a 4 096-bucket
//                     table; every streamed entry of every candidate of the read is classified ONCE: rank of its hash in Q
//                     (matched) or the gap it falls into (window only), parked as a 4-byte code word with the entry's flags
//   l2_dense_kernel   one wave per candidate, no sketch at all: the window sequence 64 windows per round exactly as block_slide
//                     of l2_kernel evaluates it (cross-ranked leave/enter times, pivot zone of 64 ranks in registers), but the
//                     gap counters D[s] and the matched bitmap live in a per-wave slot of global memory and are kept current
//                     with fire-and-forget atomics, so that when the pivot leaves the zone the zone is re-centred from 64
//                     counters next to it instead of being rebuilt from the window.  Every window is evaluated (no bounds),
//                     in the reference's order.
// The dense kernel needs 400 bytes of LDS per wave, so a CU holds as many waves as registers allow, whatever the read length.
#pragma once
#include "mm_l2.hpp"

namespace mm {

constexpr int LD_TBITS = 12;                                    // bucket table over the sketch (l2_bucket)
constexpr int LD_TSIZE = (1 << LD_TBITS) + 1;
constexpr int LD_Q_LDS_MAX = 32768;                             // sketches up to this size are searched in LDS

struct L2Range { int64_t first; int32_t m, pad; };               // streamed entries of a candidate: pos[first .. first + m)

__device__ inline uint32_t ld_make(int code, uint32_t flags) { return ((uint32_t)code & 0x1fffffffu) | (flags << 29); }
__device__ inline int ld_code(uint32_t wd) { return ((int)(wd << 3)) >> 3; }
__device__ inline uint32_t ld_flags(uint32_t wd) { return wd >> 29; }

__global__ void __launch_bounds__(256) l2_range_kernel(IndexView I, const int32_t* __restrict__ cand, const int32_t* __restrict__ cand_read,
                                                       const int32_t* __restrict__ read_len, const int32_t* __restrict__ cand_list, int n_list,
                                                       L2Range* __restrict__ ranges) {
  const int li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (li >= n_list) return;
  const int64_t c = cand_list[li];
  const int contig = cand[3 * c], rs = cand[3 * c + 1], re = cand[3 * c + 2], len = read_len[cand_read[c]];
  const int64_t first0 = contig_lower_bound_wpos(I, contig, rs, lane);                // searchIndex, computeMap.hpp:466
  const int64_t last0 = max(first0, contig_lower_bound_wpos(I, contig, re + len, lane));   // :477
  if (lane == 0) ranges[li] = L2Range{first0, (int32_t)min<int64_t>(last0 - first0, 0x7fffffff), 0};
}

// rank / gap code of hash h: >= 0 matched rank, < 0 window only in gap -code-1 (gap s: above every query hash)
__device__ inline int ld_classify(const uint32_t* __restrict__ Q, const uint32_t* __restrict__ T, int s, uint32_t h) {
  const int bkt = l2_bucket(h, 32 - LD_TBITS);                 // (buckets follow the distribution of sketch hashes: mm_l2.hpp)
  int lo = (int)T[bkt], hi = (int)T[bkt + 1];
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (Q[mid] < h) lo = mid + 1; else hi = mid; }
  return (lo < s && Q[lo] == h) ? lo : -(lo + 1);
}

// grp_first[g] .. grp_first[g+1]: the list positions (consecutive, one read) workgroup g classifies
__global__ void __launch_bounds__(256) l2_codes_kernel(IndexView I, const int32_t* __restrict__ cand_read, const uint32_t* __restrict__ sk_hash,
                                                       const uint64_t* __restrict__ mz_off, const int32_t* __restrict__ sk_n,
                                                       const int32_t* __restrict__ cand_list, const int32_t* __restrict__ grp_first,
                                                       const L2Range* __restrict__ ranges, const uint64_t* __restrict__ code_off,
                                                       uint32_t* __restrict__ codes, int q_in_lds) {
  extern __shared__ __align__(16) uint32_t ld_lds[];
  uint32_t* T = ld_lds;                                          // [LD_TSIZE]
  uint32_t* Ql = ld_lds + ((LD_TSIZE + 3) & ~3);                 // [s] when q_in_lds
  const int l0 = grp_first[blockIdx.x], l1 = grp_first[blockIdx.x + 1];
  const int r = cand_read[cand_list[l0]];
  const int s = sk_n[r];
  const uint32_t* __restrict__ Qg = sk_hash + mz_off[r];
  if (q_in_lds) for (int i = threadIdx.x; i < s; i += 256) Ql[i] = Qg[i];
  // T[b] = first rank whose bucket (l2_bucket) is >= b: element i closes the buckets after Q[i-1]'s up to its own
  for (int i = threadIdx.x; i <= s; i += 256) {
    const int lo = i ? l2_bucket(Qg[i - 1], 32 - LD_TBITS) + 1 : 0;
    const int hi = i < s ? l2_bucket(Qg[i], 32 - LD_TBITS) : (1 << LD_TBITS);
    for (int bb = lo; bb <= hi; ++bb) T[bb] = (uint32_t)i;
  }
  __syncthreads();
  const uint32_t* __restrict__ Q = q_in_lds ? Ql : Qg;
  for (int li = l0; li < l1; ++li) {
    const L2Range R = ranges[li];
    const Rec* __restrict__ pos = I.pos + R.first;
    uint32_t* __restrict__ out = codes + code_off[li];
    for (int base = 0; base < R.m; base += 1024) {               // four entries per thread in flight
      Rec x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = base + threadIdx.x + 256 * u; x[u] = pos[i < R.m ? i : 0]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = base + threadIdx.x + 256 * u; if (i < R.m) out[i] = ld_make(ld_classify(Q, T, s, x[u].hash), x[u].pw & 7u); }
    }
  }
}

// per-wave slot: D[smax] (32-bit gap counters) | mc[(smax + 3) / 4] (one byte per rank: occurrences of the query hash counted into the window)
__host__ __device__ inline size_t l2_dense_slot_words(int smax) { return (size_t)smax + (size_t)((smax + 3) / 4) + 16; }

__global__ void __launch_bounds__(64) l2_dense_kernel(IndexView I, const int32_t* __restrict__ cand, const int32_t* __restrict__ cand_read,
                                                      const uint8_t* __restrict__ sk_strand, const uint64_t* __restrict__ mz_off,
                                                      const int32_t* __restrict__ sk_n, const int32_t* __restrict__ read_len,
                                                      const int32_t* __restrict__ accept_min, int k, int w, int smax, L2Result* __restrict__ out,
                                                      const int32_t* __restrict__ cand_list, int n_list, const L2Range* __restrict__ ranges,
                                                      const uint64_t* __restrict__ code_off, const uint32_t* __restrict__ codes,
                                                      uint32_t* __restrict__ scratch, unsigned int* __restrict__ next_item,
                                                      uint8_t* __restrict__ amb_used /* optional: set per read when a vote needed an unresolved strand (bit 1 of the strand byte) */,
                                                      int force_amb /* tests: flag every read whose vote saw such an entry */,
                                                      int early_stop /* leave a candidate once no later window can reach its best so far (0: evaluate every window, cross-check) */) {
  constexpr int INF = 0x7fffffff;
  __shared__ int tst[64];
  __shared__ uint8_t fdel[64], fadd[64];
  const int lane = threadIdx.x & 63;
  uint32_t* const D = scratch + (size_t)blockIdx.x * l2_dense_slot_words(smax);
  uint32_t* const mc = D + smax;
  auto present = [&](int rk) -> bool { return ((mc[rk >> 2] >> (8 * (rk & 3))) & 0xffu) != 0u; };
  auto rank_search = [&](int arr, int v) -> int {                // number of leading lanes whose (ascending) arr < v
    int lo = 0;
    for (int st = 32; st >= 1; st >>= 1) { const int x = __shfl(arr, lo + st - 1, 64); if (x < v) lo += st; }
    const int x = __shfl(arr, lo, 64);
    return lo + (x < v ? 1 : 0);
  };
  for (;;) {
    int li = 0;
    if (lane == 0) li = (int)atomicAdd(next_item, 1u);
    li = __builtin_amdgcn_readfirstlane(li);
    if (li >= n_list) break;
    const int64_t c = cand_list[li];
    const int r = cand_read[c];
    const int s = sk_n[r];
    const uint64_t qo = mz_off[r];
    const int len = read_len[r];
    const int cnt = len - (w - 1) - (k - 1);                     // computeMap.hpp:470
    const L2Range RG = ranges[li];
    const Rec* __restrict__ pos = I.pos + RG.first;
    const uint32_t* __restrict__ cw = codes + code_off[li];
    const int last_end = RG.m;
    const int nmax = (int)min((int64_t)0x7fffffff, I.N - 1 - RG.first);
    const int cmax = max(last_end - 1, 0);
    int amin = accept_min[r]; if (amin < 1) amin = 1;
    // ---- the state arrays of this candidate
    for (int i = lane; i < s; i += 64) D[i] = 0;
    for (int i = lane; i < (s + 3) / 4; i += 64) mc[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    // A counted event: matched rank -> its byte of mc, window-only gap -> D; fire-and-forget additions, so the order in which the
    // memory system applies the events of a round does not matter (a rank whose last occurrence leaves and whose next occurrence
    // enters within one round passes through 2 or 0 and ends at 1 either way; bytes stay within 0..2, no carry into a neighbour).
    auto apply = [&](bool on, int code, int sign) {
      if (!on) return;
      if (code >= 0) atomicAdd(&mc[code >> 2], sign > 0 ? (1u << (8 * (code & 3))) : (0u - (1u << (8 * (code & 3)))));
      else atomicAdd(&D[-code - 1], (uint32_t)sign);
    };
    // ---- first super-window [0, e0): every entry once, duplicates of a hash inside the window resolved as everywhere
    int b = 0, e = (int)wave_lower_bound_wpos(pos, 0, last_end, pw_wpos(pos[0].pw) + cnt, lane);   // :473, MIIteratorL2.hpp:62
    for (int base = 0; base < e; base += 64) {
      const int j = base + lane;
      const uint32_t wd = cw[min(j, cmax)];
      const int code = ld_code(wd);
      bool on = j < e && code != -(s + 1);
      // an earlier occurrence exists in the contig: inside the window?  (mm_index.hpp: neighbour distances; a scan only beyond their reach)
      const int dres = (on && (ld_flags(wd) & PW_DP)) ? dup_before(I, RG.first + j, (int64_t)j) : 0;
      if (dres > 0) on = false;
      uint64_t fm = __ballot(dres < 0);
      while (fm) {
        const int l = __builtin_ctzll(fm); fm &= fm - 1;
        const bool dup = wave_has_hash(pos, 0, base + l, pos[base + l].hash, lane);
        if (dup && lane == l) on = false;
      }
      apply(on, code, +1);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    // ---- pivot zone: lane l owns rank z0 + l, fz = r + D[z0..r]; cbase = D[0..z0), sb = matched ranks below z0 in the window
    int z0 = 0, cbase = 0, sb = 0, fz = 0;
    uint64_t pm = 0;
    {
      // pivot R = min r with r + C(r) >= s, by a scan over the counters (once per candidate)
      int acc = 0, R = s;
      for (int r0 = 0; r0 < s; r0 += 64) {
        const int rr = r0 + lane;
        const int d = rr < s ? (int)D[rr] : 0;
        const int ex = wave_excl_scan(d, lane);
        const uint64_t m = __ballot(rr < s && rr + acc + ex + d >= s);
        if (m) { R = r0 + __builtin_ctzll(m); break; }
        acc += __builtin_amdgcn_readlane(ex, 63) + __builtin_amdgcn_readlane(d, 63);
      }
      z0 = max(0, min(R - 32, s - 63));
      int cb = 0, sbl = 0;                                       // D[0..z0), matched bits below z0
      for (int r0 = 0; r0 < z0; r0 += 64) { const int rr = r0 + lane; cb += rr < z0 ? (int)D[rr] : 0; }
      cbase = wave_sum(cb);
      for (int r0 = 0; r0 < z0; r0 += 64) { const int rr = r0 + lane; sbl += (rr < z0 && present(rr)) ? 1 : 0; }
      sb = wave_sum(sbl);
    }
    // (re)loads the zone at z0 from the arrays and moves it until the pivot of the current window lies strictly inside
    auto recentre = [&]() {
      for (;;) {
        const int rz = z0 + lane;
        const int dz = rz < s ? (int)D[rz] : 0;
        const int exz = wave_excl_scan(dz, lane);
        fz = rz < s ? rz + exz + dz : (1 << 29);
        pm = __ballot(rz < s && present(rz));
        const uint64_t ge = __ballot(fz >= s - cbase);            // (the sentinel lanes r >= s always qualify)
        const int zmax = max(0, s - 63);
        if (ge == 0ull) {                                         // pivot above the zone
          if (z0 >= zmax) break;
          const int nz = min(z0 + 32, zmax), dlt = nz - z0;
          cbase += __builtin_amdgcn_readlane(exz, dlt);            // D[z0 .. nz)
          sb += __popcll(pm & ((1ull << dlt) - 1ull));
          z0 = nz;
          continue;
        }
        if ((ge & 1ull) && z0 > 0) {                              // pivot at or below the zone's first rank
          const int nz = max(0, z0 - 32), dlt = z0 - nz;
          const int rr = nz + lane;
          const int dd = lane < dlt ? (int)D[rr] : 0;
          const uint64_t mb = __ballot(lane < dlt && present(rr));
          cbase -= wave_sum(dd);
          sb -= __popcll(mb);
          z0 = nz;
          continue;
        }
        break;
      }
    };
    recentre();
    // ---- the reference's loop (computeMap.hpp:496-533): evaluate [b,e), then MIIteratorL2::next — 64 windows per round
    int best = 0, bestR = 0, beg_pos = 0, last_pos = 0, opt_b = 0, opt_e = 0;
    unsigned long long evals = 0, shifts = 0;
    // Early end: a window shares at most as many hashes as it holds entries that carry a sketch hash, and every later window lies in
    // [b, last_end).  Once fewer such entries are left there than the best so far (or than the acceptance threshold while nothing has
    // reached it), no later window can reach OR equal it — neither the maximum nor the last position equal to it can change — and the
    // candidate is done.  On a true hit that is shortly after the optimum: the tail of the range, about half of it, is not slid over.
    int rem = 0;                                                  // entries with a sketch hash in [b, last_end)
    if (early_stop) {
      int a = 0;
      for (int base = 0; base < last_end; base += 256) {
        uint32_t wq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wq[u] = cw[min(base + 64 * u + lane, cmax)];
#pragma unroll
        for (int u = 0; u < 4; ++u) a += (base + 64 * u + lane < last_end && ld_code(wq[u]) >= 0) ? 1 : 0;
      }
      rem = wave_sum(a);
    }
    while (e < last_end) {
      if (early_stop && rem < max(best, amin)) break;
      const Rec xb = pos[min(b + lane, nmax)];
      const Rec xe = pos[min(e + lane, nmax)];
      const int w64 = pw_wpos(pos[min(b + 64, nmax)].pw);
      const int cB = ld_code(cw[min(b + lane, cmax)]), cE = ld_code(cw[min(e + lane, cmax)]);
      const int wpb = pw_wpos(xb.pw);
      int nextw = __shfl_down(wpb, 1, 64);
      if (lane == 63) nextw = w64;
      const int tA = (b + lane + 1 < last_end) ? nextw : INF;    // entry b+lane leaves when sw_pos reaches the next entry's position
      const int tB = (e + lane < last_end) ? pw_wpos(xe.pw) - (cnt - 1) : INF;   // entry e+lane enters
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
      if (kA < 64) { fdel[kA] = 1; tst[kA] = tA; }
      if (kB < 64) { fadd[kB] = 1; tst[kB] = tB; }
      wave_sync();
      const int hasDel = fdel[lane], hasAdd = fadd[lane];
      const int dj = wave_excl_scan(hasDel, lane), aj = wave_excl_scan(hasAdd, lane);   // lane j: window j = [b+dj, e+aj)
      const bool cond = lane < ksteps && (e + aj < last_end);
      const uint64_t cm = __ballot(cond);
      int n_eval = (~cm == 0ull) ? 64 : __builtin_ctzll(~cm);    // windows 0 .. n_eval-1 are evaluated (n_eval >= 1)
      bool vE = cE != -(s + 1) && kB < n_eval, vB = cB != -(s + 1) && kA < n_eval;
      {
        const bool fE = vE && (xe.pw & PW_DP), fB = vB && (xb.pw & PW_DN);
        if (__ballot(fE || fB) != 0ull) {
          // REV: the hash of the entering entry e + lane is already inside [b', x), b' = window start at its step (after the step's deletion)
          const int kEc = min(kB, 63), kBc = min(kA, 63);
          const int hbL = b + __shfl(dj, kEc, 64) + __shfl(hasDel, kEc, 64), weL = e + __shfl(aj, kBc, 64);
          const int rE = fE ? dup_before(I, RG.first + e + lane, (int64_t)(e + lane) - hbL) : 0;
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
          const int rB = fB ? dup_after(I, RG.first + b + lane, (int64_t)weL - 1 - (b + lane)) : 0;
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
      const int gE = -cE - 1, gB = -cB - 1;
      const int indE = vE ? ((cE < 0 && gE < z0 ? 1 : 0) | (cE >= 0 && cE < z0 ? 1 << 16 : 0)) : 0;
      const int indB = vB ? ((cB < 0 && gB < z0 ? 1 : 0) | (cB >= 0 && cB < z0 ? 1 << 16 : 0)) : 0;
      const int pE = wave_excl_scan(indE, lane) + indE, pB = wave_excl_scan(indB, lane) + indB;
      const int gE_ = __shfl(pE, max(aj - 1, 0), 64), gB_ = __shfl(pB, max(dj - 1, 0), 64);
      const int accE = aj > 0 ? gE_ : 0, accB = dj > 0 ? gB_ : 0;
      const int cbase_j = cbase + (accE & 0xffff) - (accB & 0xffff);
      const int sb_j = sb + (accE >> 16) - (accB >> 16);
      const int thr = s - cbase_j;
      auto pivot_of = [&]() -> int { return rank_search(fz, thr); };
      int pj = pivot_of();
      uint64_t pm_j = pm;
      uint64_t zE = __ballot(vE && ((cE >= 0) ? (cE >= z0 && cE < z0 + 64) : (gE >= z0 && gE < z0 + 64)));
      uint64_t zB = __ballot(vB && ((cB >= 0) ? (cB >= z0 && cB < z0 + 64) : (gB >= z0 && gB < z0 + 64)));
      while (zE | zB) {                                          // zone events in step order (each one changes the windows after its step)
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
        if (m > best) {                                          // strict: the first window reaching the maximum (:510-518)
          best = m;
          bestR = z0 + __builtin_amdgcn_readlane(pj, j1);
          const int d1 = __builtin_amdgcn_readlane(dj, j1);
          opt_b = b + d1; opt_e = e + __builtin_amdgcn_readlane(aj, j1);
          beg_pos = __builtin_amdgcn_readlane(wpb, d1);
          last_pos = __builtin_amdgcn_readlane(wpb, __builtin_amdgcn_readlane(dj, jl));
        } else if (m == best) last_pos = __builtin_amdgcn_readlane(wpb, __builtin_amdgcn_readlane(dj, jl));   // :520-524
        evals += (unsigned long long)n_eval;
      }
      // ---- the arrays follow the steps that are consumed: events of steps 0 .. n_eval-1
      apply(vB && kA < n_eval, cB, -1);
      apply(vE && kB < n_eval, cE, +1);
      int dn, an;
      if (n_eval < 64) { dn = __builtin_amdgcn_readlane(dj, n_eval); an = __builtin_amdgcn_readlane(aj, n_eval); }
      else { dn = __builtin_amdgcn_readlane(dj, 63) + __builtin_amdgcn_readlane(hasDel, 63); an = __builtin_amdgcn_readlane(aj, 63) + __builtin_amdgcn_readlane(hasAdd, 63); }
      if (early_stop) rem -= __popcll(__ballot(lane < dn && cB >= 0));   // entries b .. b+dn-1 leave [b, last_end)
      b += dn; e += an;
      {                                                          // scalars of the zone after the consumed steps
        const int fE = an > 0 ? __builtin_amdgcn_readlane(pE, an - 1) : 0, fB = dn > 0 ? __builtin_amdgcn_readlane(pB, dn - 1) : 0;
        cbase += (fE & 0xffff) - (fB & 0xffff);
        sb += (fE >> 16) - (fB >> 16);
      }
      bool shift = zone_exit;
      if (!shift) {
        const uint64_t ge = __ballot(fz >= s - cbase);           // the pivot of the next round's first window must be inside too
        shift = ge == 0ull || ((ge & 1ull) && z0 > 0);
      }
      if (shift) {                                               // (after a zone exit fz / pm hold events of steps that were not consumed: reload)
        ++shifts;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        recentre();
      }
    }
    // ---- K6 strand vote over the first optimal window (computeMap.hpp:424-433, slidingMap.hpp:232-254)
    int strand = -1, accepted = 0;
    if (best >= amin) {
      accepted = 1;
      int votes = 0, amb_votes = 0;                               // votes of resolved strands / number of votes whose query strand is unresolved (l2_kernel's scheme)
      for (int base = opt_b; base < opt_e; base += 64) {
        const int j = base + lane;
        const uint32_t wd = cw[min(j, cmax)];
        const int code = ld_code(wd);
        const bool cnt_it = j < opt_e && code >= 0 && code < bestR;
        const uint32_t sq = cnt_it ? (uint32_t)sk_strand[qo + code] : 0u;
        const bool unres = (sq & 2) && amb_used != nullptr;       // (after the host resolved the read, amb_used is null and bit 1 is gone)
        const int contrib = cnt_it ? ((sq & 1) ? 1 : -1) * pw_strand(ld_flags(wd)) : 0;
        const bool flagged = cnt_it && (ld_flags(wd) & PW_DN);   // a later occurrence exists in the contig: inside the window?
        const int dres = flagged ? dup_after(I, RG.first + j, (int64_t)opt_e - 1 - j) : 0;
        if (cnt_it && dres == 0) { if (unres) ++amb_votes; else votes += contrib; }
        uint64_t fm = __ballot(dres < 0);
        while (fm) {
          const int l = __ffsll((unsigned long long)fm) - 1;
          fm &= fm - 1;
          const bool later = wave_has_hash(pos, base + l + 1, opt_e, pos[base + l].hash, lane);
          if (!later && lane == l) { if (unres) ++amb_votes; else votes += contrib; }
        }
      }
      votes = wave_sum(votes);
      amb_votes = wave_sum(amb_votes);
      // each unresolved vote is +1 or -1: the sign of the total is already decided unless the resolved votes are that close
      if (amb_votes > 0 && ((votes - amb_votes <= 0 && votes + amb_votes > 0) || force_amb) && lane == 0) amb_used[cand_read[c]] = 1;
      strand = votes > 0 ? 1 : -1;
    }
    if (lane == 0) {
      L2Result o;
      o.contig = cand[3 * c]; o.mean_pos = (beg_pos + last_pos) / 2;   // :537
      o.shared = best; o.strand = strand; o.accepted = accepted; o.pad = 0;
      o.opt_beg = RG.first + opt_b; o.opt_end = RG.first + opt_e;
      o.n_stream = (uint32_t)last_end; o.n_evals = (uint32_t)evals; o.n_rebuilds = (uint32_t)shifts; o.pad2 = 0;
      out[c] = o;
    }
    wave_sync();
  }
}

}  // namespace mm
