// ORACLE — test infrastructure only.  CPU restatement of the MetaMaps hot path.
//
// Nothing under oracle/ is product code: only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may build, link, import or execute it.  The shipped
// path is metamaps_amd/csrc (HIP); it never includes or calls anything from here.
//
// Every function cites the reference file:line (relative to /root/reference/src) whose
// behaviour it restates.  The text is written from the behaviour, not copied.
//
// Pinning status (see DESIGN.md "Oracle"):
//   * murmur3 (A1): checked against the real reference header compiled in place
//     (oracle/_ref/libref_murmur.so, built by oracle/Makefile from
//     /root/reference/src/common/murmur3.h) and against the known answers of SURVEY.md §8a.
//   * Boost.Math binomial pdf / quantile (third-party, not vendored in the reference,
//     version unpinned by configure.ac:49): restated from the documented semantics and
//     pinned by tables generated with scipy 1.15.3 (which embeds Boost.Math) —
//     tests/golden/binom_*.json, generator tests/golden/make_binom_golden.py.
//   * everything else: restated; pinned only by the reference's example outputs
//     (MetaMaps_example_output.zip → tests/golden/example_*) and the known answers the
//     survey recorded.  The reference binary itself cannot be built in this image (hard
//     Boost dependency, base_types.hpp:12, Makefile.in:20) — "parity partially pinned".
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------------------
// A1  MurmurHash3_x64_128, low 32 bits, seed 42
//     common/murmur3.h:226-303 ; map/include/commonFunc.hpp:33,71-81
// ---------------------------------------------------------------------------------------
static inline uint64_t rotl64(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
static inline uint64_t avalanche64(uint64_t v) {
  v ^= v >> 33; v *= 0xff51afd7ed558ccdULL;
  v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL;
  v ^= v >> 33; return v;
}
static inline void murmur3_x64_128(const uint8_t* p, int n, uint32_t seed, uint64_t out[2]) {
  const uint64_t C1 = 0x87c37b91114253d5ULL, C2 = 0x4cf5ad432745937fULL;
  uint64_t a = seed, b = seed;
  int nb = n / 16;
  for (int blk = 0; blk < nb; ++blk) {
    uint64_t x, y;
    memcpy(&x, p + 16 * blk, 8);
    memcpy(&y, p + 16 * blk + 8, 8);
    x *= C1; x = rotl64(x, 31); x *= C2; a ^= x;
    a = rotl64(a, 27); a += b; a = a * 5 + 0x52dce729;
    y *= C2; y = rotl64(y, 33); y *= C1; b ^= y;
    b = rotl64(b, 31); b += a; b = b * 5 + 0x38495ab5;
  }
  const uint8_t* t = p + 16 * nb;
  int rem = n & 15;
  uint64_t x = 0, y = 0;
  for (int i = rem - 1; i >= 8; --i) y |= (uint64_t)t[i] << (8 * (i - 8));
  if (rem > 8) { y *= C2; y = rotl64(y, 33); y *= C1; b ^= y; }
  for (int i = std::min(rem, 8) - 1; i >= 0; --i) x |= (uint64_t)t[i] << (8 * i);
  if (rem > 0) { x *= C1; x = rotl64(x, 31); x *= C2; a ^= x; }
  a ^= (uint64_t)n; b ^= (uint64_t)n;
  a += b; b += a;
  a = avalanche64(a); b = avalanche64(b);
  a += b; b += a;
  out[0] = a; out[1] = b;
}
static inline uint32_t kmer_hash(const char* s, int k) {
  uint64_t o[2];
  murmur3_x64_128((const uint8_t*)s, k, 42u, o);
  return (uint32_t)o[0];
}

// ---------------------------------------------------------------------------------------
// Core records (map/include/base_types.hpp:22-103)
// ---------------------------------------------------------------------------------------
struct Mz {            // MinimizerInfo
  uint32_t hash; int32_t seq; int32_t wpos; int32_t strand;
  bool same(const Mz& o) const {
    return hash == o.hash && seq == o.seq && wpos == o.wpos && strand == o.strand;
  }
};
struct Hit {           // MinimizerMetaData, ordered by (seq,wpos,strand) base_types.hpp:99
  int32_t seq, wpos, strand;
  bool operator<(const Hit& o) const {
    if (seq != o.seq) return seq < o.seq;
    if (wpos != o.wpos) return wpos < o.wpos;
    return strand < o.strand;
  }
};
struct Contig { std::string name; int32_t len; };

// ---------------------------------------------------------------------------------------
// A2  winnowed minimizers — commonFunc.hpp:38-66 (complement / upper-case), :92-175
//     `seq` is upper-cased in place like the reference does.
// ---------------------------------------------------------------------------------------
static inline char complement_base(char c) {
  switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; }
  return c;
}
static inline void add_minimizers(std::vector<Mz>& out, char* seq, int len, int k, int w, int seqId) {
  for (int i = 0; i < len; ++i)
    if (seq[i] > 96 && seq[i] < 123) seq[i] -= 32;            // commonFunc.hpp:57-66
  std::string rc((size_t)std::max(len, 0), 'N');
  for (int i = 0; i < len; ++i) rc[len - 1 - i] = complement_base(seq[i]);   // :38-55

  struct Slot { Mz m; int pos; };
  std::deque<Slot> q;                                           // monotone queue, :99
  for (int i = 0; i + k <= len; ++i) {                          // :114  (len-k+1 k-mers)
    int win = i - w + 1;                                        // :118
    uint32_t hf = kmer_hash(seq + i, k);
    uint32_t hb = kmer_hash(rc.data() + (len - i - k), k);      // :125
    if (hf == hb) continue;                                     // symmetric k-mer: nothing at all, :130
    uint32_t h = std::min(hf, hb);
    int st = hf < hb ? +1 : -1;                                 // :136
    while (!q.empty() && q.front().pos <= i - w) q.pop_front(); // :139
    while (!q.empty() && q.back().m.hash >= h) q.pop_back();    // :144 (ties: newest wins)
    q.push_back(Slot{Mz{h, seqId, 0, st}, i});                  // wpos 0 until saved, :149
    if (win >= 0) {                                             // :154
      if (out.empty() || !out.back().same(q.front().m)) {       // :157
        q.front().m.wpos = win;                                 // :161
        out.push_back(q.front().m);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Binomial helpers standing in for Boost.Math (third-party; see header note).
//   pdf(binomial(n,p),k)                        mapWrap.h:340
//   quantile(complement(binomial(n,p),q))       map_stats.hpp:88   (policy integer_round_outwards,
//        q<0.5 ⇒ upper quantile ⇒ smallest x with P(X>x) <= q ; 0 when 1-q <= pdf(0) ; n when p==1)
//   cdf(complement(binomial(n,p),x))            map_stats.hpp:204  = P(X>x)
// ---------------------------------------------------------------------------------------
static inline long double log_binom_pmf(int n, long double p, int k) {
  // log C(n,k) + k log p + (n-k) log(1-p); p in (0,1)
  return lgammal((long double)n + 1) - lgammal((long double)k + 1) - lgammal((long double)(n - k) + 1)
       + (long double)k * logl(p) + (long double)(n - k) * log1pl(-p);
}
static inline double binom_pmf(int n, double p, int k) {
  if (k < 0 || k > n) return 0.0;
  if (p == 0) return k == 0 ? 1.0 : 0.0;
  if (p == 1) return k == n ? 1.0 : 0.0;
  if (n == 0) return 1.0;
  if (k == 0) return std::pow(1 - p, n);
  if (k == n) return std::pow(p, (double)k);
  return (double)expl(log_binom_pmf(n, (long double)p, k));
}
// terms pmf(lo..n) in long double, built by recurrence away from an anchor computed in log space
static inline void binom_terms_from(int n, double p, int lo, std::vector<long double>& t) {
  t.assign((size_t)(n - lo + 1), 0.0L);
  long double P = p, odds = P / (1.0L - P);
  int mode = (int)std::floor((n + 1) * p);
  int anchor = std::min(std::max(mode, lo), n);
  t[anchor - lo] = expl(log_binom_pmf(n, P, anchor));
  for (int i = anchor; i < n; ++i) t[i + 1 - lo] = t[i - lo] * odds * (long double)(n - i) / (long double)(i + 1);
  for (int i = anchor; i > lo; --i) t[i - 1 - lo] = t[i - lo] / odds * (long double)i / (long double)(n - i + 1);
}
static inline double binom_sf(int n, double p, int x) {      // P(X > x)
  if (x < 0) return 1.0;
  if (x >= n) return 0.0;
  if (p <= 0) return 0.0;
  if (p >= 1) return 1.0;
  std::vector<long double> t;
  binom_terms_from(n, p, x + 1, t);
  long double acc = 0;
  for (size_t i = t.size(); i-- > 0;) acc += t[i];
  return (double)acc;
}
static inline int binom_quantile_upper(int n, double p, double q) {   // smallest x with P(X>x) <= q
  if (p >= 1) return n;
  if (p <= 0) return 0;
  std::vector<long double> t;
  binom_terms_from(n, p, 0, t);
  long double tail = 0;   // P(X > x), x = n
  int x = n;
  while (x > 0 && tail + t[x] <= (long double)q) { tail += t[x]; --x; }
  return x;
}

// ---------------------------------------------------------------------------------------
// A9  statistics — map_stats.hpp:44-256.  float where the reference uses float.
// ---------------------------------------------------------------------------------------
static inline float j2md(float j, int k) {                      // :44-54
  if (j == 0) return 1.0f;
  if (j == 1) return 0.0f;
  float d = (-1.0 / k) * std::log(2.0 * j / (1 + j));
  return d;
}
static inline float md2j(float d, int k) {                      // :62-66
  float j = 1.0 / (2.0 * std::exp(k * d) - 1.0);
  return j;
}
static inline float md_lower_bound(float d, int s, int k, float ci) {   // :79-111 (USE_BOOST branch)
  float q2 = (1.0 - ci) / 2;
  int x = binom_quantile_upper(s, (double)md2j(d, k), (double)q2);
  float jac = float(x) / s;
  return j2md(jac, k);
}
static inline int estimate_min_hits(int s, int k, float pi) {   // :120-132
  float d = 1.0 - pi / 100.0;
  float jac = md2j(d, k);
  return (int)std::ceil(1.0 * s * jac);
}
static inline int estimate_min_hits_relaxed(int s, int k, float pi) {   // :142-167
  int start = estimate_min_hits(s, k, pi), best = start;
  for (int i = start; i >= 0; --i) {
    float jac = 1.0 * i / s;
    float d = j2md(jac, k);
    float lo = md_lower_bound(d, s, k, 0.9);
    float idu = 100.0 * (1.0 - lo);
    if (idu >= pi) best = i; else break;
  }
  return best;
}
static inline double estimate_pvalue(int s, int k, int alphabet, float pi, int qlen, uint64_t rlen) {  // :179-214
  double space = std::pow((double)alphabet, k);
  double px = 1. / (1. + space / qlen), py = px;
  double r = px * py / (px + py - px * py);
  int x = estimate_min_hits_relaxed(s, k, pi);
  double tail = (x == 0) ? 1.0 : binom_sf(s, r, x - 1);
  return rlen * tail;
}
static inline int recommended_window(double pcut, int k, int alphabet, float pi, int qlen, uint64_t rlen) {  // :226-256
  std::vector<int> cand{1, 2, 5};
  for (int i = 10; i < qlen; i += 10) cand.push_back(i);
  int pick = 0; bool found = false;
  for (int s : cand)
    if (estimate_pvalue(s, k, alphabet, pi, qlen, rlen) <= pcut) { pick = s; found = true; break; }
  if (!found) pick = cand.back();   // reference leaves this uninitialised; unreachable for sane inputs
  int w = 2.0 * qlen / pick;
  return std::min(std::max(w, 1), qlen);
}

// text form of a C++ float/double through operator<< at default precision (6 significant, %g)
static inline std::string fmt_g(double v) { char b[64]; snprintf(b, sizeof b, "%g", v); return b; }

}  // namespace orc
