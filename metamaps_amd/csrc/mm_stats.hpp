// Host-side float statistics of the mapper (replaces skch::Stat, map_stats.hpp:44-256, and the
// Boost.Math binomial calls at map_stats.hpp:88,204 and mapWrap.h:340).  These are pure functions of a
// handful of small integers, evaluated once per distinct sketch size and cached — not GPU work.
//
// Float/double mix mirrors the reference expression by expression, because the results are compared
// against thresholds (identity >= --pi) and printed with 6 significant digits.
#pragma once
#include "cpu_budget.hpp"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace mm { namespace stats {

// ---- Binomial(n, p) ---------------------------------------------------------------------------------
// Upper tail built from the top down: tail(x) = P(X > x).  Terms come from the log-pmf at a numerically
// safe anchor (the mode) and the exact ratio pmf(i+1)/pmf(i) = (n-i)/(i+1) * p/(1-p).
class BinomTail {
  int n_; double p_;
  std::vector<long double> pmf_;      // pmf_[i], i in [0,n]
 public:
  BinomTail(int n, double p) : n_(n), p_(p), pmf_((size_t)n + 1, 0.0L) {
    if (p <= 0) { pmf_[0] = 1; return; }
    if (p >= 1) { pmf_[(size_t)n] = 1; return; }
    const long double P = p, Qc = 1.0L - P, ratio = P / Qc;
    int m = (int)std::floor(((long double)n + 1) * P);
    m = std::min(std::max(m, 0), n);
    long double lg = lgammal((long double)n + 1) - lgammal((long double)m + 1) - lgammal((long double)(n - m) + 1) +
                     (long double)m * logl(P) + (long double)(n - m) * log1pl(-P);
    pmf_[(size_t)m] = expl(lg);
    for (int i = m; i < n; ++i) pmf_[(size_t)i + 1] = pmf_[(size_t)i] * ratio * (long double)(n - i) / (long double)(i + 1);
    for (int i = m; i > 0; --i) pmf_[(size_t)i - 1] = pmf_[(size_t)i] * (long double)i / ((long double)(n - i + 1) * ratio);
  }
  double pmf(int k) const { return (k < 0 || k > n_) ? 0.0 : (double)pmf_[(size_t)k]; }
  double upper_tail(int x) const {                  // P(X > x)
    if (x < 0) return 1.0;
    long double t = 0;
    for (int i = n_; i > x; --i) t += pmf_[(size_t)i];
    return (double)t;
  }
  // quantile(complement(binomial(n,p), q)) under Boost's default integer_round_outwards policy for q < 1/2:
  // the smallest x with P(X > x) <= q.
  int upper_quantile(double q) const {
    if (p_ >= 1) return n_;
    long double t = 0;
    int x = n_;
    while (x > 0 && t + pmf_[(size_t)x] <= (long double)q) { t += pmf_[(size_t)x]; --x; }
    return x;
  }
};

inline double binom_pmf(int n, double p, int k) {   // boost::math::pdf(binomial(n,p), k), mapWrap.h:340
  if (k < 0 || k > n) return 0;
  if (p == 0) return k == 0 ? 1 : 0;
  if (p == 1) return k == n ? 1 : 0;
  if (n == 0) return 1;
  if (k == 0) return std::pow(1 - p, n);
  if (k == n) return std::pow(p, (double)k);
  long double P = p;
  return (double)expl(lgammal((long double)n + 1) - lgammal((long double)k + 1) - lgammal((long double)(n - k) + 1) +
                      (long double)k * logl(P) + (long double)(n - k) * log1pl(-P));
}

// ---- map_stats.hpp --------------------------------------------------------------------------------
inline float j2md(float j, int k) {                 // :44
  if (j == 0) return 1.0f;
  if (j == 1) return 0.0f;
  float d = (-1.0 / k) * std::log(2.0 * j / (1 + j));
  return d;
}
inline float md2j(float d, int k) {                 // :62
  float j = 1.0 / (2.0 * std::exp(k * d) - 1.0);
  return j;
}
inline float md_lower_bound(float d, int s, int k, float ci) {   // :79
  float q2 = (1.0 - ci) / 2;
  BinomTail B(s, (double)md2j(d, k));
  int x = B.upper_quantile((double)q2);
  float jac = float(x) / s;
  return j2md(jac, k);
}
inline void identity(int shared, int s, int k, float* ident, float* ident_upper) {   // computeMap.hpp:406-412
  float md = j2md(1.0 * shared / s, k);
  float lo = md_lower_bound(md, s, k, 0.9);
  *ident = 100 * (1 - md);
  *ident_upper = 100 * (1 - lo);
}
inline int min_hits(int s, int k, float pi) {       // :120
  float d = 1.0 - pi / 100.0;
  float jac = md2j(d, k);
  return (int)std::ceil(1.0 * s * jac);
}
inline int min_hits_relaxed(int s, int k, float pi) {   // :142
  int start = min_hits(s, k, pi), best = start;
  for (int i = start; i >= 0; --i) {
    float jac = 1.0 * i / s;
    float d = j2md(jac, k);
    float lo = md_lower_bound(d, s, k, 0.9);
    float ub = 100.0 * (1.0 - lo);                    // double arithmetic here (:158), float in computeMap.hpp:412
    if (ub >= pi) best = i; else break;
  }
  return best;
}
inline double estimate_pvalue(int s, int k, int alphabet, float pi, int qlen, uint64_t rlen) {   // :179
  double space = std::pow((double)alphabet, k);
  double px = 1. / (1. + space / qlen), py = px;
  double r = px * py / (px + py - px * py);
  int x = min_hits_relaxed(s, k, pi);
  double tail = 1.0;
  if (x != 0) { BinomTail B(s, r); tail = B.upper_tail(x - 1); }
  return rlen * tail;
}
inline int recommended_window(double pcut, int k, int alphabet, float pi, int qlen, uint64_t rlen) {   // :226
  std::vector<int> cand{1, 2, 5};
  for (int i = 10; i < qlen; i += 10) cand.push_back(i);
  int pick = cand.back();
  for (int s : cand) if (estimate_pvalue(s, k, alphabet, pi, qlen, rlen) <= pcut) { pick = s; break; }
  int w = 2.0 * qlen / pick;
  return std::min(std::max(w, 1), qlen);
}

// Per-sketch-size thresholds used by the kernels:
//   min_hits   = estimateMinimumHitsRelaxed(s)                   L1 (computeMap.hpp:325)
//   accept_min = smallest shared count whose upper-bound identity reaches --pi        L2 filter (:415)
// identity_upper is non-decreasing in `shared` (the quantile is monotone in p, float ops are monotone),
// so the filter `nucIdentityUpperBound >= pi` is a threshold on `shared`; found by bisection, verified at
// both sides.
struct SketchLut { int min_hits; int accept_min; };
class LutCache {
  int k_; float pi_;
  std::mutex m_;
  std::unordered_map<int, SketchLut> memo_;
  SketchLut compute(int s) const {
    SketchLut L{0, 0};
    if (s > 0) {
      L.min_hits = min_hits_relaxed(s, k_, pi_);
      auto ok = [&](int sh) { float id, ub; identity(sh, s, k_, &id, &ub); return ub >= pi_; };
      if (!ok(s)) L.accept_min = s + 1;             // nothing passes (pi > 100)
      else {
        int lo = 0, hi = s;                         // smallest sh in [0,s] with ok(sh)
        while (lo < hi) { int mid = (lo + hi) / 2; if (ok(mid)) hi = mid; else lo = mid + 1; }
        L.accept_min = lo;
      }
    }
    return L;
  }
 public:
  LutCache(int k, float pi) : k_(k), pi_(pi) {}
  // thresholds of the given (distinct) sketch sizes; sizes not seen before are computed on a few host threads.  One cache per
  // (k, pi) serves every context of the process (for()), so a second context never recomputes what the first one has.
  std::vector<SketchLut> get_many(const std::vector<int>& sizes) {
    std::lock_guard<std::mutex> lk(m_);
    std::vector<int> missing;
    for (int s : sizes) if (!memo_.count(s)) missing.push_back(s);
    if (!missing.empty()) {
      std::vector<SketchLut> out(missing.size());
      const size_t nt = std::min<size_t>({missing.size() / 8 + 1, 16, std::max(1u, mm::cpu_budget())});
      std::vector<std::thread> th;
      for (size_t t = 1; t < nt; ++t) th.emplace_back([&, t]() { for (size_t i = t; i < missing.size(); i += nt) out[i] = compute(missing[i]); });
      for (size_t i = 0; i < missing.size(); i += nt) out[i] = compute(missing[i]);
      for (auto& x : th) x.join();
      for (size_t i = 0; i < missing.size(); ++i) memo_[missing[i]] = out[i];
    }
    std::vector<SketchLut> r; r.reserve(sizes.size());
    for (int s : sizes) r.push_back(memo_.at(s));
    return r;
  }
  SketchLut get(int s) { return get_many({s})[0]; }
  static std::shared_ptr<LutCache> for_params(int k, float pi) {
    static std::mutex gm; static std::map<std::pair<int, float>, std::shared_ptr<LutCache>> all;
    std::lock_guard<std::mutex> lk(gm);
    auto& p = all[{k, pi}];
    if (!p) p = std::make_shared<LutCache>(k, pi);
    return p;
  }
};

}}  // namespace mm::stats
