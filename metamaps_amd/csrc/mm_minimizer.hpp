// K1 — k-mer hashing + winnowing sweep (replaces CommonFunc::addMinimizers, commonFunc.hpp:92-175).
//
// Parallel restatement of the reference's monotone-deque loop.  Let NS be the positions i in
// [0, len-k] whose forward and reverse-complement hashes differ (commonFunc.hpp:130: symmetric k-mers
// take no part at all, not even in window expiry).  For every i in NS with i >= w-1:
//     c(i) = argmin over q in NS ∩ (i-w, i] of the canonical hash, ties -> largest q   (:139-149)
// The loop appends (hash[c(i)], wpos = i-w+1, strand[c(i)]) when c(i) differs from c at the previous
// evaluated position (:157: the saved queue entry carries its wpos, an unsaved one carries 0), except
// that right after the window-0 emission E0 (only if position w-1 is in NS) change points whose
// (hash, strand) equal E0's are swallowed until the first one that differs (the 4-tuple compare
// matches an unsaved entry with wpos 0).  `jstar` = position of that first differing change point,
// found per sequence by a short serial walk (jstar_kernel); emissions in (w-1, jstar) are dropped.
//
// HBM traffic per position: 2 bits in, 8 bytes out per emitted minimizer (~2/(w+1) per position),
// twice (count pass + write pass).  Everything else lives in LDS.
#pragma once
#include <type_traits>
#include "mm_common.hpp"
#include "mm_scan.hpp"

namespace mm {

constexpr int MZ_THREADS = 256;
constexpr int MZ_TILE = 2048;          // positions per workgroup
constexpr int MZ_MAX_W = 4096;
constexpr int MZ_STAGE = 1024;         // staged records per tile in the single-pass scheme
constexpr int MZ_MAX_K = 64;

struct SeqView {                       // device view of an mm_seqset
  const uint32_t* packed;
  const uint64_t* base;                // [n+1]
  const int32_t* len;                  // [n]
  const uint64_t* exc_start;
  const uint32_t* exc_len;
  const uint8_t* exc_byte;
  int64_t n_exc;
  int64_t n;
};

__device__ inline uint32_t code_at(const SeqView& S, uint64_t g) { return (S.packed[g >> 4] >> (2 * (g & 15))) & 3u; }

// first exception run whose end is beyond g
__device__ inline int64_t exc_lower(const SeqView& S, uint64_t g) {
  int64_t lo = 0, hi = S.n_exc;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (S.exc_start[mid] + S.exc_len[mid] <= g) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ inline uint8_t ascii_at(const SeqView& S, uint64_t g) {
  if (S.n_exc) {
    int64_t r = exc_lower(S, g);
    if (r < S.n_exc && S.exc_start[r] <= g) return S.exc_byte[r];
  }
  return ascii_of_code(code_at(S, g));
}

struct KmerInfo { uint32_t hash; bool ns; bool fwd; };
// serial (one thread) evaluation of one position; used only by the jstar walk
// eight 2-bit codes (low 16 bits of x) -> eight ASCII bytes; comp: the complementary bases
__device__ inline uint64_t ascii8_of_codes(uint32_t x, bool comp) {
  const uint32_t flip = comp ? 0x03030303u : 0u;
  const uint32_t a = x & 0xffu, b = (x >> 8) & 0xffu;
  const uint32_t sa = ((a | (a << 6) | (a << 12) | (a << 18)) & 0x03030303u) ^ flip, sb = ((b | (b << 6) | (b << 12) | (b << 18)) & 0x03030303u) ^ flip;
  return (uint64_t)__builtin_amdgcn_perm(0u, 0x54474341u, sa) | ((uint64_t)__builtin_amdgcn_perm(0u, 0x54474341u, sb) << 32);
}
// clean_pos: positions below it have no exception run (non-ACGT bytes) among their k bases — the packed words are then the sequence
__device__ inline KmerInfo kmer_info_serial(const SeqView& S, uint64_t gbase, int p, int k, int clean_pos) {
  if (k == 16 && (S.n_exc == 0 || p < clean_pos)) {             // sixteen codes straddle at most two packed words
    const uint64_t g = gbase + (uint64_t)p;
    const uint32_t off = (uint32_t)(g & 15), w0 = S.packed[g >> 4], w1 = off ? S.packed[(g >> 4) + 1] : 0u;
    const uint32_t codes = off ? (w0 >> (2 * off)) | (w1 << (32 - 2 * off)) : w0;
    const uint64_t lo = ascii8_of_codes(codes, false), hi = ascii8_of_codes(codes >> 16, false);
    const uint64_t clo = ascii8_of_codes(codes, true), chi = ascii8_of_codes(codes >> 16, true);
    const uint32_t hf = murmur16(lo, hi), hb = murmur16(__builtin_bswap64(chi), __builtin_bswap64(clo));
    return KmerInfo{hf < hb ? hf : hb, hf != hb, hf < hb};
  }
  uint8_t f[MZ_MAX_K], c[MZ_MAX_K];
  for (int j = 0; j < k; ++j) { f[j] = ascii_at(S, gbase + p + j); c[j] = complement_ascii(f[j]); }
  uint32_t hf = murmur_bytes<false>(f, k), hb = murmur_bytes<true>(c + k - 1, k);
  return KmerInfo{hf < hb ? hf : hb, hf != hb, hf < hb};
}

// One thread per sequence: position of the first change point after window 0 whose (hash,strand)
// differs from the window-0 emission; w-1 when nothing can be swallowed.
static __global__ void jstar_kernel(SeqView S, const uint8_t* __restrict__ active, int k, int w, int32_t* __restrict__ jstar) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S.n) return;
  int32_t res = w - 1;
  int npos = S.len[s] - k + 1;
  if ((active == nullptr || active[s]) && npos >= w) {
    uint64_t gb = S.base[s];
    // the walk looks at the first few dozen positions: when the sequence's first exception run starts behind them (a reference contig with N runs
    // somewhere inside), they take the packed-word path instead of sixteen binary searches over the exception table per position
    int clean_pos = 0;
    if (S.n_exc) {
      const int64_t r = exc_lower(S, gb);
      const uint64_t first_exc = r < S.n_exc ? S.exc_start[r] : ~0ull;
      clean_pos = first_exc <= gb ? 0 : (int)min<uint64_t>(first_exc - gb, (uint64_t)0x7fffffff) - (k - 1);
    }
    KmerInfo e = kmer_info_serial(S, gb, w - 1, k, clean_pos);
    if (e.ns) {
      // c(w-1): argmin over NS positions 0..w-1, ties -> rightmost
      int cpos = w - 1; uint32_t ch = e.hash; bool cf = e.fwd;
      for (int q = w - 2; q >= 0; --q) {
        KmerInfo t = kmer_info_serial(S, gb, q, k, clean_pos);
        if (t.ns && t.hash < ch) { ch = t.hash; cpos = q; cf = t.fwd; }
      }
      const uint32_t h0 = ch; const bool f0 = cf;
      res = npos;                                  // swallowed to the end unless a differing change point shows up
      for (int p = w; p < npos; ++p) {
        KmerInfo t = kmer_info_serial(S, gb, p, k, clean_pos);
        if (!t.ns) continue;
        int ncpos; uint32_t nch; bool ncf;
        if (t.hash <= ch || cpos <= p - w) {
          if (t.hash <= ch && cpos > p - w) { ncpos = p; nch = t.hash; ncf = t.fwd; }
          else {                                   // previous minimum left the window: rescan
            ncpos = p; nch = t.hash; ncf = t.fwd;
            for (int q = p - 1; q > p - w; --q) {
              KmerInfo u = kmer_info_serial(S, gb, q, k, clean_pos);
              if (u.ns && u.hash < nch) { nch = u.hash; ncpos = q; ncf = u.fwd; }
            }
          }
        } else { ncpos = cpos; nch = ch; ncf = cf; }
        if (ncpos != cpos) {
          if (nch != h0 || ncf != f0) { res = p; break; }
        }
        cpos = ncpos; ch = nch; cf = ncf;
      }
    }
  }
  jstar[s] = res;
}

// tile_first[s] = index of the first tile of sequence s (tile_first[n] = number of tiles)
__device__ inline int64_t seq_of_tile(const uint64_t* __restrict__ tile_first, int64_t n, uint64_t tile) {
  int64_t lo = 0, hi = n;                          // last s with tile_first[s] <= tile
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (tile_first[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

static __device__ int mz_dbg_stop = 0;   // timing aid (MM_MZ_DBG=n: the read kernel leaves after step n and reports no minimizers; tools/stage_ms.py)
// MODE 0: tile_count[tile] = number of emitted minimizers (count pass of the two-pass scheme, index scale).
// MODE 1: records written at tile_out[tile]... (write pass).
// MODE 2: single pass — records staged at tile*MZ_STAGE and counted; compact_tiles_kernel then packs them.  A tile
//         normally emits ~2/(w+1) of its 2048 positions; if one exceeds MZ_STAGE (low-complexity sequence) the
//         overflow flag is raised and the caller falls back to the two-pass scheme.  Used for read batches.
// (Round 5 measured the alternative to MODE 2 + compact_tiles_kernel: ONE launch whose tiles take the place of their records from a chained scan
//  with decoupled look-back inside the launch — ticketed tile numbers, 64-bit {flag, value} descriptors, agent-scope atomics, one wavefront walking
//  back 64 or 256 predecessors per step.  Bit-identical, and slower: the stage 9.7 -> 9.9 ms (64 per step) / 11.4 ms (256), index builds no faster
//  although they hash once instead of twice.  The ~1 800 resident tiles finish as a generation, the nearest known prefix is then hundreds of tiles
//  back and every step of the walk is a round trip to the memory side (the XCDs' L2s are not coherent) with the tile's other three wavefronts
//  parked at the barrier.  Commit b69326c keeps the code.)
// (amdgpu_waves_per_eu(7): the kernel needs 68-70 registers; left to itself the compiler took 107 after an unrelated edit — one more kernel argument, round 5 — and ran
//  four waves per SIMD instead of seven: the stage 9.6 -> 10.6 ms.  Pinned.  The same round measured index builds hashing ONCE — MODE 2 in groups of 2^19 tiles through a
//  4 GB staging area, every group packed behind the previous one into a buffer sized from the winnowing density, the records then moved into an array of their exact size —
//  against the count + write passes below: 1.355 against 1.366 s for the 26.8 Gbp build; the second hashing pass costs what staging, packing and moving 47 GB cost.  Not kept.)
template <int MODE>
__global__ void __launch_bounds__(MZ_THREADS) __attribute__((amdgpu_waves_per_eu(7))) minimizer_kernel(SeqView S, const uint64_t* __restrict__ tile_first, int k, int w,
                                                               const int32_t* __restrict__ jstar, uint32_t* __restrict__ tile_count,
                                                               const uint64_t* __restrict__ tile_out, Rec* __restrict__ out,
                                                               uint32_t* __restrict__ out_seq, int* __restrict__ stage_overflow) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int halo = (2 * (w - 1) + 7) & ~7;         // rounded so that the tile's first byte is 8-byte aligned in LDS
  const int NP = MZ_TILE + halo;                   // positions held
  const int NB = ((NP + k - 1 + 8) + 15) & ~15;    // bytes held (+8: the k=16 path reads whole 8-byte words)
  uint8_t* fwd = smem;
  uint8_t* cmp = fwd + NB;
  uint32_t* hsh = (uint32_t*)(cmp + NB);
  uint16_t* cps = (uint16_t*)(hsh + NP);
  uint8_t* flg = (uint8_t*)(cps + NP);

  const uint64_t tile = blockIdx.x;
  const int64_t s = seq_of_tile(tile_first, S.n, tile);
  const int len = S.len[s];
  const int npos = len - k + 1;
  const int P0 = (int)(tile - tile_first[s]) * MZ_TILE;
  const int Pend = min(P0 + MZ_TILE, npos);
  const int H0 = max(0, P0 - halo);
  const uint64_t gb = S.base[s];
  const int nbytes = (Pend - H0) + k - 1;
  const int tid = threadIdx.x;

  // 1. unpack 2-bit codes -> ASCII (+ complement) in LDS
  {
    const uint64_t g0 = gb + (uint64_t)H0;
    const uint64_t w0 = g0 >> 4, w1 = (g0 + nbytes - 1) >> 4;
    // sixteen bases per packed word -> two 8-byte ASCII words per strand: the four 2-bit codes of a byte are spread into the
    // selector bytes of a byte permute over the table "ACGT" (its complement: selector ^ 3).  The tile starts on a multiple of 8
    // bases, so both halves of a packed word land on 8-byte boundaries of the LDS strings; the last word may run past the
    // sequence into the 8 spare bytes, which no evaluated position reads.
    for (uint64_t wi = w0 + tid; wi <= w1; wi += MZ_THREADS) {
      const uint32_t word = S.packed[wi];
      const int64_t rel = (int64_t)(wi << 4) - (int64_t)g0;    // 0 mod 8
      uint32_t fq[4], cq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t x = (word >> (8 * q)) & 0xffu;
        const uint32_t sel = (x | (x << 6) | (x << 12) | (x << 18)) & 0x03030303u;
        fq[q] = __builtin_amdgcn_perm(0u, 0x54474341u, sel);
        cq[q] = __builtin_amdgcn_perm(0u, 0x54474341u, sel ^ 0x03030303u);
      }
      if (rel >= 0 && rel < nbytes) {
        *reinterpret_cast<uint64_t*>(fwd + rel) = (uint64_t)fq[0] | ((uint64_t)fq[1] << 32);
        *reinterpret_cast<uint64_t*>(cmp + rel) = (uint64_t)cq[0] | ((uint64_t)cq[1] << 32);
      }
      if (rel + 8 >= 0 && rel + 8 < nbytes) {
        *reinterpret_cast<uint64_t*>(fwd + rel + 8) = (uint64_t)fq[2] | ((uint64_t)fq[3] << 32);
        *reinterpret_cast<uint64_t*>(cmp + rel + 8) = (uint64_t)cq[2] | ((uint64_t)cq[3] << 32);
      }
    }
    __syncthreads();
    if (S.n_exc) {                                 // patch non-ACGT bytes (self-complementary, commonFunc.hpp:50)
      const uint64_t g1 = g0 + nbytes;
      for (int64_t r = exc_lower(S, g0); r < S.n_exc && S.exc_start[r] < g1; ++r) {
        uint64_t a = max(S.exc_start[r], g0), b = min(S.exc_start[r] + S.exc_len[r], g1);
        uint8_t ch = S.exc_byte[r];
        for (uint64_t g = a + tid; g < b; g += MZ_THREADS) { fwd[g - g0] = ch; cmp[g - g0] = ch; }
      }
      __syncthreads();
    }
  }
  const int dbgs = mz_dbg_stop;
  if (dbgs == 1) { if (MODE != 1 && tid == 0) tile_count[tile] = 0; return; }
  // 2. canonical hash / strand / non-symmetric flag per position
  const int np = Pend - H0;
  if (k == 16) {
    // eight consecutive positions per thread from three aligned 8-byte words per strand: the k-mer at offset i
    // is a funnel shift of (w0,w1,w2); its reverse complement is the byte-swapped complement words
    const uint64_t* f64 = (const uint64_t*)fwd;
    const uint64_t* c64 = (const uint64_t*)cmp;
    // The tile's own 2 048 positions are exactly one group of eight per thread.  The halo in front of them (16 positions at w = 8) would be two more groups
    // — a second trip through this loop for the whole first wavefront with two lanes at work (a sixth of the kernel's instructions); it is hashed
    // position by position instead, a lane each, by the byte-wise form of the same hash.
    const int hb8 = P0 > H0 ? (P0 - H0) : 0;                      // positions of the halo (a multiple of 8)
    for (int j = tid; j < hb8; j += MZ_THREADS) {
      const uint32_t hf = murmur_bytes<false>(fwd + j, 16);
      const uint32_t hb = murmur_bytes<true>(cmp + j + 15, 16);
      hsh[j] = hf < hb ? hf : hb;
      flg[j] = (uint8_t)((hf != hb ? 1 : 0) | (hf < hb ? 2 : 0));
    }
    for (int g = (hb8 >> 3) + tid; g * 8 < np; g += MZ_THREADS) {
      const uint64_t w0 = f64[g], w1 = f64[g + 1], w2 = f64[g + 2];
      const uint64_t c0 = c64[g], c1 = c64[g + 1], c2 = c64[g + 2];
      uint32_t hq[8]; uint64_t fq = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint64_t lo = i ? (w0 >> (8 * i)) | (w1 << (64 - 8 * i)) : w0;
        const uint64_t hi = i ? (w1 >> (8 * i)) | (w2 << (64 - 8 * i)) : w1;
        const uint64_t clo = i ? (c0 >> (8 * i)) | (c1 << (64 - 8 * i)) : c0;
        const uint64_t chi = i ? (c1 >> (8 * i)) | (c2 << (64 - 8 * i)) : c1;
        const uint32_t hf = murmur16(lo, hi);
        const uint32_t hb = murmur16(__builtin_bswap64(chi), __builtin_bswap64(clo));
        hq[i] = hf < hb ? hf : hb;
        fq |= (uint64_t)((hf != hb ? 1u : 0u) | (hf < hb ? 2u : 0u)) << (8 * i);
      }
      // (positions at or beyond np inside the last group hold bytes past the sequence: written, never read)
      *reinterpret_cast<uint4*>(hsh + 8 * g) = make_uint4(hq[0], hq[1], hq[2], hq[3]);
      *reinterpret_cast<uint4*>(hsh + 8 * g + 4) = make_uint4(hq[4], hq[5], hq[6], hq[7]);
      *reinterpret_cast<uint64_t*>(flg + 8 * g) = fq;
    }
  } else {
    for (int j = tid; j < np; j += MZ_THREADS) {
      const uint32_t hf = murmur_bytes<false>(fwd + j, k);
      const uint32_t hb = murmur_bytes<true>(cmp + j + k - 1, k);
      hsh[j] = hf < hb ? hf : hb;
      flg[j] = (uint8_t)((hf != hb ? 1 : 0) | (hf < hb ? 2 : 0));
    }
  }
  __syncthreads();
  if (dbgs == 2) { if (MODE != 1 && tid == 0) tile_count[tile] = 0; return; }
  // 3. window argmin c(p) for every evaluated position from P0-(w-1) on
  const int jeval0 = max(max(P0 - (w - 1), w - 1), H0) - H0;
  if (w <= 9) {
    // eight consecutive positions per thread: the 16 (hash, flag) pairs they can look at are read once with vector loads and
    // the eight arg-mins are found in registers (same scan order: from the position itself down, strictly smaller wins)
    // (as above: the few evaluated positions of the halo — w - 1 of them — are taken one per lane by the plain scan, the tile's own 2 048 are one group per thread)
    const int hb8 = P0 > H0 ? (P0 - H0) : 0;
    for (int j = jeval0 + tid; j < hb8; j += MZ_THREADS) {
      uint16_t c = 0xFFFF;
      if (flg[j] & 1) {
        uint32_t best = hsh[j]; int bj = j;
        const int qlo = max(j - w + 1, 0);
        for (int q = j - 1; q >= qlo; --q)
          if ((flg[q] & 1) && hsh[q] < best) { best = hsh[q]; bj = q; }
        c = (uint16_t)bj;
      }
      cps[j] = c;
    }
    for (int g = (hb8 >> 3) + tid; g * 8 < np; g += MZ_THREADS) {
      uint32_t h[16]; uint8_t f[16];
      {
        const uint4 a = g ? *reinterpret_cast<const uint4*>(hsh + 8 * g - 8) : make_uint4(0, 0, 0, 0);
        const uint4 b = g ? *reinterpret_cast<const uint4*>(hsh + 8 * g - 4) : make_uint4(0, 0, 0, 0);
        const uint4 c = *reinterpret_cast<const uint4*>(hsh + 8 * g), d = *reinterpret_cast<const uint4*>(hsh + 8 * g + 4);
        h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w; h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w;
        h[8] = c.x; h[9] = c.y; h[10] = c.z; h[11] = c.w; h[12] = d.x; h[13] = d.y; h[14] = d.z; h[15] = d.w;
        const uint64_t fa = g ? *reinterpret_cast<const uint64_t*>(flg + 8 * g - 8) : 0ull, fb = *reinterpret_cast<const uint64_t*>(flg + 8 * g);
#pragma unroll
        for (int t = 0; t < 8; ++t) { f[t] = (uint8_t)(fa >> (8 * t)); f[8 + t] = (uint8_t)(fb >> (8 * t)); }
      }
      // key = hash << 8 | (255 - slot) for non-symmetric positions, all ones otherwise: the minimum over a window is its
      // smallest hash, the rightmost one among equals — found for all eight windows at once by doubling (windows of 2, 4, 8)
      uint64_t key[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) key[t] = (f[t] & 1) ? ((uint64_t)h[t] << 8) | (uint64_t)(255 - t) : ~0ull;
      uint64_t res[8];
      auto mn = [](uint64_t a, uint64_t b) { return a < b ? a : b; };
      auto windows = [&](auto wtag) {
        constexpr int W = decltype(wtag)::value;
        uint64_t m1[16], m2[16], m4[16];
#pragma unroll
        for (int t = 1; t < 16; ++t) m1[t] = mn(key[t], key[t - 1]);
#pragma unroll
        for (int t = 3; t < 16; ++t) m2[t] = mn(m1[t], m1[t - 2]);
#pragma unroll
        for (int t = 7; t < 16; ++t) m4[t] = mn(m2[t], m2[t - 4]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = 8 + i;
          if constexpr (W == 1) res[i] = key[t];
          else if constexpr (W == 2) res[i] = m1[t];
          else if constexpr (W == 3) res[i] = mn(m1[t], m1[t - 1]);
          else if constexpr (W == 4) res[i] = m2[t];
          else if constexpr (W < 8) res[i] = mn(m2[t], m2[t - (W - 4)]);
          else if constexpr (W == 8) res[i] = m4[t];
          else res[i] = mn(m4[t], m4[t - 1]);
        }
      };
      switch (w) {
        case 1: windows(std::integral_constant<int, 1>{}); break;
        case 2: windows(std::integral_constant<int, 2>{}); break;
        case 3: windows(std::integral_constant<int, 3>{}); break;
        case 4: windows(std::integral_constant<int, 4>{}); break;
        case 5: windows(std::integral_constant<int, 5>{}); break;
        case 6: windows(std::integral_constant<int, 6>{}); break;
        case 7: windows(std::integral_constant<int, 7>{}); break;
        case 8: windows(std::integral_constant<int, 8>{}); break;
        default: windows(std::integral_constant<int, 9>{}); break;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = 8 * g + i;
        if (j < jeval0 || j >= np) continue;
        cps[j] = (f[8 + i] & 1) ? (uint16_t)(8 * g - 8 + 255 - (int)(res[i] & 0xffu)) : (uint16_t)0xFFFF;
      }
    }
  } else
  for (int j = jeval0 + tid; j < np; j += MZ_THREADS) {
    uint16_t c = 0xFFFF;
    if (flg[j] & 1) {
      uint32_t best = hsh[j]; int bj = j;
      const int qlo = max(j - w + 1, 0);           // j-w+1 >= 0 whenever p >= H0 + ... ; clamp for p < w-1+H0 cases
      for (int q = j - 1; q >= qlo; --q)
        if ((flg[q] & 1) && hsh[q] < best) { best = hsh[q]; bj = q; }
      c = (uint16_t)bj;
    }
    cps[j] = c;
  }
  __syncthreads();
  if (dbgs == 3) { if (MODE != 1 && tid == 0) tile_count[tile] = 0; return; }
  // 4. emission flags for the tile's own positions, 8 consecutive positions per thread
  const int js = jstar[s];
  const int j0 = (P0 - H0) + tid * (MZ_TILE / MZ_THREADS);
  uint32_t mask = 0;
  static_assert(MZ_TILE / MZ_THREADS == 8, "eight positions per thread");
  // Nearly always all eight positions and the one before them are non-symmetric: the previous evaluated position is then simply the one before, and the
  // flags and window minima of the nine come with three loads.  Anything else (symmetric k-mers, the first window, the end of the sequence) takes the
  // position-by-position form below.
  bool fast = false;
  if (w >= 2 && j0 + 8 <= np && H0 + j0 - 1 >= w - 1 && j0 >= 1) {   // (w = 1: a window holds no earlier position)
    const uint2 fw = *reinterpret_cast<const uint2*>(flg + j0);
    const uint32_t fprev = flg[j0 - 1];
    fast = (fw.x & fw.y & 0x01010101u) == 0x01010101u && (fprev & 1u);
    if (fast) {
      const uint4 cw = *reinterpret_cast<const uint4*>(cps + j0);   // eight 16-bit positions of the minima
      const uint32_t cprev = cps[j0 - 1];
      const uint32_t c[9] = {cprev, cw.x & 0xffffu, cw.x >> 16, cw.y & 0xffffu, cw.y >> 16, cw.z & 0xffffu, cw.z >> 16, cw.w & 0xffffu, cw.w >> 16};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = H0 + j0 + i;
        const bool emit = c[i] != c[i + 1] && !(p > w - 1 && p < js);
        mask |= (emit ? 1u : 0u) << i;
      }
    }
  }
  if (!fast) {
#pragma unroll
  for (int i = 0; i < MZ_TILE / MZ_THREADS; ++i) {
    int j = j0 + i, p = H0 + j;
    if (j < np && p >= w - 1 && (flg[j] & 1)) {
      int pj = -1;
      const int qlo = max(max(j - w + 1, 0), (w - 1) - H0);
      for (int q = j - 1; q >= qlo; --q) if (flg[q] & 1) { pj = q; break; }
      bool emit = (pj < 0) || (cps[pj] != cps[j]);
      if (p > w - 1 && p < js) emit = false;
      if (emit) mask |= 1u << i;
    }
  }
  }
  const uint32_t cnt = __popc(mask);
  uint64_t tot;
  uint64_t ex = block_excl_scan_u64(cnt, &tot);
  if (dbgs == 4) { if (MODE != 1 && tid == 0) tile_count[tile] = 0; return; }
  if (MODE != 1 && tid == 0) tile_count[tile] = (uint32_t)tot;
  if (MODE == 2 && tot > MZ_STAGE) { if (tid == 0) atomicExch(stage_overflow, 1); return; }
  if (MODE != 0) {
    uint64_t o = (MODE == 1 ? tile_out[tile] : tile * (uint64_t)MZ_STAGE) + ex;
#pragma unroll
    for (int i = 0; i < MZ_TILE / MZ_THREADS; ++i) {
      if (mask & (1u << i)) {
        int j = j0 + i, p = H0 + j, c = cps[j];
        uint32_t pw = ((uint32_t)(p - w + 1) << PW_SHIFT) | ((flg[c] & 2) ? PW_STRAND : 0u);
        out[o] = Rec{hsh[c], pw};
        if (out_seq) out_seq[o] = (uint32_t)s;
        ++o;
      }
    }
  }
}

// packs the staged tiles of the single-pass scheme
static __global__ void __launch_bounds__(256) compact_tiles_kernel(const Rec* __restrict__ stage, const uint32_t* __restrict__ tile_count,
                                                                   const uint64_t* __restrict__ tile_out, Rec* __restrict__ out) {
  const uint64_t tile = blockIdx.x;
  const uint32_t n = tile_count[tile];
  const uint64_t o = tile_out[tile];
  for (uint32_t i = threadIdx.x; i < n; i += 256) out[o + i] = stage[tile * (uint64_t)MZ_STAGE + i];
}

inline size_t minimizer_lds_bytes(int k, int w) {
  int halo = (2 * (w - 1) + 7) & ~7, NP = MZ_TILE + halo, NB = ((NP + k - 1 + 8) + 15) & ~15;
  return (size_t)2 * NB + (size_t)NP * 4 + (size_t)NP * 2 + (size_t)NP;
}

// Result of a sweep over a sequence set
struct MinimizerSet {
  DBuf<Rec> rec;                 // all records, sequence-major, position order
  DBuf<uint64_t> off;            // [n+1] first record of each sequence
  DBuf<uint32_t> rec_seq;        // optional owner sequence per record
  std::vector<uint64_t> h_off;   // host copy of off
  int64_t total = 0;
};

// `active` (host, may be empty = all): sequences with active[i]==0 produce nothing.
void run_minimizers(mm_ctx* ctx, const mm_seqset* S, int k, int w, const std::vector<uint8_t>& active, bool want_rec_seq,
                    MinimizerSet& out);

SeqView make_view(const mm_seqset* S);

}  // namespace mm
