// K5 core — the sliding MinHash window as O(1) integer updates on two small arrays
// (replaces SlideMapper's ordered std::map: slidingMap.hpp:26-318).
//
// Let Q[0..s) be the read's sorted unique sketch.  For a window W of reference minimizers:
//   matched rank r   : Q[r] occurs in W                                    -> bit r of mt[]
//   W-only hash h    : h not in Q, gap g = #{Q < h}; D[g] = number of DISTINCT such hashes in W (g < s)
// Q[r] belongs to the s smallest hashes of Q ∪ W  <=>  r + sum_{g<=r} D[g] < s   (it and everything below
// it number at most s).  The left side is strictly increasing in r, so the counted ranks are exactly
// r < R where R = min{ r : r + C(r) >= s } (R = s if none), C(r) = D[0]+..+D[r].  The reference's
//   sharedSketchElements == popcount(mt[0..R))            (slidingMap.hpp:263-316 keeps the same count
// incrementally around its `pivot` iterator).  Inserting/deleting one reference minimizer moves R by at
// most one, so the state is {R, Cb = C(R-1), shared}; every event is a constant number of array accesses.
//
// Distinctness (slidingMap.hpp:148-157 REV / :186-209 NOOP): the caller tells whether another occurrence
// of the same hash is inside the window; the index precomputes per-entry DP/DN flags so that this
// question only needs work for hashes repeated within one contig.
//
// Compiles for host (unit tests: tests/test_l2_core.cpp via g++) and device.
#pragma once
#include <stdint.h>

#ifndef MM_HD
#if defined(__HIPCC__)
#define MM_HD __host__ __device__ inline
#else
#define MM_HD inline
#endif
#endif

namespace mm {

// DT = uint16_t (always sufficient: a window holds < 65536 entries for reads up to 64 kb; longer reads are rejected
// by the caller) or uint8_t (compact variant; `overflow` is raised if a gap ever holds more than 255 distinct
// hashes and the caller redoes the candidate with the wide type).
template <typename DT>
struct L2StateT {
  const uint32_t* Q;   // sorted unique sketch hashes
  DT* D;               // [s] distinct W-only hashes per gap (gap s is never needed)
  uint32_t* mt;        // [(s+31)/32] matched-rank bitmap
  int s;
  int R;               // pivot rank: ranks < R are counted
  int Cb;              // D[0] + ... + D[R-1]
  int shared;
  int overflow;
};
using L2State = L2StateT<uint16_t>;

// code >= 0: matched rank; code < 0: W-only in gap g = -code-1 (g == s means "above every query hash")
MM_HD int l2_classify(const uint32_t* Q, int s, uint32_t h) {
  int lo = 0, hi = s;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (Q[mid] < h) lo = mid + 1; else hi = mid; }
  return (lo < s && Q[lo] == h) ? lo : -(lo + 1);
}

template <typename DT> MM_HD void l2_reset(L2StateT<DT>& S) { S.R = S.s; S.Cb = 0; S.shared = 0; S.overflow = 0; }   // arrays must be zeroed by the caller

template <typename DT> MM_HD bool l2_mt_test(const L2StateT<DT>& S, int r) { return (S.mt[r >> 5] >> (r & 31)) & 1u; }

// Every event gathers all the array cells it can possibly need with INDEPENDENT loads first (on the device:
// one LDS round trip instead of a chain of dependent ones), then decides in registers.

// a matched hash enters the window (first occurrence, sign=+1) / leaves it (last occurrence, sign=-1)
template <typename DT> MM_HD void l2_matched_event(L2StateT<DT>& S, int r, int sign) {
  const uint32_t bit = 1u << (r & 31);
  uint32_t wd = S.mt[r >> 5];
  wd = sign > 0 ? (wd | bit) : (wd & ~bit);
  S.mt[r >> 5] = wd;
  if (r < S.R) S.shared += sign;
}
template <typename DT> MM_HD void l2_add_matched(L2StateT<DT>& S, int r) { l2_matched_event(S, r, +1); }
template <typename DT> MM_HD void l2_del_matched(L2StateT<DT>& S, int r) { l2_matched_event(S, r, -1); }

// a distinct W-only hash of gap g enters (sign=+1) / leaves (sign=-1)
template <typename DT> MM_HD void l2_wonly_event(L2StateT<DT>& S, int g, int sign) {
  if (g >= S.s) return;
  const int R = S.R;
  const int rm1 = R > 0 ? R - 1 : 0;            // rank just below the pivot (dummy 0 when R == 0)
  const int rr = R < S.s ? R : S.s - 1;         // pivot rank (dummy s-1 when R == s)
  int dg = S.D[g];
  int dR = S.D[rr];
  int dRm1 = S.D[rm1];
  const uint32_t wR = S.mt[rr >> 5], wRm1 = S.mt[rm1 >> 5];
  dg += sign;
  if ((long long)dg > (long long)(DT)~(DT)0) S.overflow = 1;
  S.D[g] = (DT)dg;
  if (g == rr) dR = dg;                         // the cells were read before the update
  if (g == rm1) dRm1 = dg;
  if (sign > 0) {
    if (g < R) {
      S.Cb += 1;
      if (R - 1 + S.Cb >= S.s) {                // rank R-1 pushed out of the s smallest
        S.R = R - 1;
        S.Cb -= dRm1;
        if ((wRm1 >> (rm1 & 31)) & 1u) S.shared -= 1;
      }
    }
  } else {
    if (g < R) S.Cb -= 1;
    if (g <= R && R < S.s && R + S.Cb + dR < S.s) {   // rank R now fits among the s smallest
      if ((wR >> (R & 31)) & 1u) S.shared += 1;
      S.Cb += dR;
      S.R = R + 1;
    }
  }
}
template <typename DT> MM_HD void l2_add_wonly(L2StateT<DT>& S, int g) { l2_wonly_event(S, g, +1); }
template <typename DT> MM_HD void l2_del_wonly(L2StateT<DT>& S, int g) { l2_wonly_event(S, g, -1); }

// Bucket of a hash in the rank table T.  Sketch hashes are window minima of minima (two strands, w windows): half of a read's
// sketch lies below 2^28, four fifths below 2^29, 97 % below 2^30 (w = 8), so buckets of equal width (h >> tshift) put 25-35
// hashes of a 10 kb read (40-55 of a 50 kb read) into the lowest ones and every search below ran 5-7 steps on the generic path.
// The buckets follow the distribution instead: b = nb * (1 - (1 - h / 2^32)^10) — the tenth power sits between the shapes of
// w = 6 and w = 16 — which leaves the longest bucket of a read at 8-15 entries in every class: four doubling steps.  Every
// operation below is correctly rounded and monotone (no contraction possible: products feed products, the two fused
// operations are written as such), so the bucket never falls as the hash rises — all that T ("first rank whose bucket
// is >= b") and the searches need — and the table's builder and its readers evaluate the same instructions.
MM_HD int l2_bucket(uint32_t h, int tshift) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float c = (float)(1u << (32 - tshift)) - 0.0625f;       // (just below nb: the product truncates to nb - 1 at most, no clamp)
  const float y = __builtin_fmaf((float)h, -0x1p-32f, 1.0f);    // 1 - h / 2^32, in [0, 1] (v_cvt_f32_u32: to nearest, monotone; the
                                                                 //  round-towards-zero conversion is ten instructions of software here)
  const float y2 = y * y, y4 = y2 * y2, y8 = y4 * y4, y10 = y8 * y2;
  return (int)__builtin_fmaf(-y10, c, c);
}

}  // namespace mm
