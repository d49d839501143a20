// ORACLE — test infrastructure only (see orc_core.hpp header).
// Reference sketch, L1 seed-hit candidates, L2 sliding MinHash, read reporting.
#pragma once
#include "orc_core.hpp"
#include "orc_io.hpp"
#include <functional>
#include <chrono>
#include <cstdio>
#include <atomic>
#include <thread>
#include <algorithm>
#include <climits>

namespace orc {

struct Params {
  int k = 16, w = 16, minReadLen = 1000, alphabet = 4, threads = 1;
  float pi = 80;
  double pval = 1e-3;
  uint64_t refSize = 0;
  uint64_t maxMem = 0;          // bytes, 0 = unlimited
  bool reportAll = false;
};

// ---------------------------------------------------------------------------------------
// A3/A4  reference sketch — map/include/winSketch.hpp:68-556
// ---------------------------------------------------------------------------------------
// minimizerPosLookupIndex (winSketch.hpp:119-120): hash -> occurrences in position order.  One unordered_map as in the reference;
// for the timed CPU baseline of bench.py (no --maxmemory, -t N) the same map is split by hash % parts so that N threads can fill
// it — the reference's single-threaded build of a >= 1 Gbp slice would take minutes of untimed set-up.  Lookups see no difference.
struct HashLookup {
  typedef std::unordered_map<uint32_t, std::vector<Hit>> Map;
  std::vector<Map> part = std::vector<Map>(1);
  Map& of(uint32_t h) { return part[part.size() == 1 ? 0 : h % part.size()]; }
  const Map& of(uint32_t h) const { return part[part.size() == 1 ? 0 : h % part.size()]; }
  std::vector<Hit>& operator[](uint32_t h) { return of(h)[h]; }
  const std::vector<Hit>* find(uint32_t h) const { const Map& m = of(h); auto f = m.find(h); return f == m.end() ? nullptr : &f->second; }
  size_t count(uint32_t h) const { return of(h).count(h); }
  size_t size() const { size_t n = 0; for (auto& m : part) n += m.size(); return n; }
  bool empty() const { return size() == 0; }
  void clear() { part.assign(1, Map()); }
};

struct RefSketch {
  std::vector<Contig> meta;                                      // :102
  HashLookup lookup;                                             // :119-120
  std::vector<Mz> byPos;                                         // :129 (seq,wpos) ordered
  std::map<int, int> freqHist;                                   // :133 — never cleared between chunks
  int freqThreshold = INT_MAX;                                   // :94

  // winSketch.hpp:165-178; struct sizes on LP64: vector<Hit> 24, Hit 12, Mz 16, vector<Mz> 24
  static size_t memory_of(size_t hashes, size_t mins) {
    size_t buckets = hashes / 10;
    size_t table = buckets * (8 + 8) + hashes * 8 + hashes * 24 + mins * 12;
    table *= 1.2;                                                // size_t *= double, as in the reference
    size_t vec = 24 + mins * 16;
    return table + vec;
  }

  // winSketch.hpp:452-494
  void compute_freq_hist() {
    if (lookup.empty()) return;
    if (lookup.part.size() == 1) { for (auto& e : lookup.part[0]) freqHist[(int)e.second.size()] += 1; }
    else {                                                       // partitions counted side by side, then added (the histogram is a sum)
      std::vector<std::map<int, int>> h(lookup.part.size());
      std::vector<std::thread> pool;
      for (size_t p = 0; p < h.size(); ++p) pool.emplace_back([&, p] { for (auto& e : lookup.part[p]) h[p][(int)e.second.size()] += 1; });
      for (auto& th : pool) th.join();
      for (auto& hp : h) for (auto& kv : hp) freqHist[kv.first] += kv.second;
    }
    int64_t uniq = (int64_t)lookup.size();
    float pct = 0.001f;                                          // :91
    int64_t ignore = uniq * pct / 100;                           // int64 * float -> float, / int, truncation
    int64_t sum = 0;
    for (auto it = freqHist.rbegin(); it != freqHist.rend(); ++it) {
      sum += it->second;
      if (sum < ignore) freqThreshold = it->first;
      else if (sum == ignore) { freqThreshold = it->first; break; }
      else break;
    }
  }

  // winSketch.hpp:506-518 : first entry with (seq,wpos) >= (seqId,pos)
  size_t search(int seqId, int pos) const {
    size_t lo = 0, hi = byPos.size();
    while (lo < hi) {
      size_t mid = (lo + hi) / 2;
      const Mz& m = byPos[mid];
      bool less = m.seq < seqId || (m.seq == seqId && m.wpos < pos);
      if (less) lo = mid + 1; else hi = mid;
    }
    return lo;
  }

  void clear_chunk() { byPos.clear(); lookup.clear(); meta.clear(); }

  // winSketch.hpp:180-365.  onChunk(this, N) is called once per index chunk (N is 1-based).
  void build(const std::vector<std::string>& fastas, const Params& P,
             const std::function<void(RefSketch&, int)>& onChunk) {
    size_t runHashes = 0, runMins = 0, seen = 0;
    int chunkNo = 1;
    // distinct hashes of `cur` that the lookup does not hold yet (the reference walks a std::set, :274-282; same number)
    auto count_novel = [&](const std::vector<Mz>& cur, bool against_lookup) {
      std::vector<uint32_t> hs(cur.size());
      for (size_t i = 0; i < cur.size(); ++i) hs[i] = cur[i].hash;
      std::sort(hs.begin(), hs.end());
      hs.erase(std::unique(hs.begin(), hs.end()), hs.end());
      if (!against_lookup) return hs.size();
      size_t n = 0;
      for (uint32_t h : hs) if (!lookup.count(h)) ++n;
      return n;
    };
    for (const auto& fn : fastas) {
      // The winnowing of a contig does not depend on anything else, so with -t N the contigs are read first and winnowed by N
      // threads (seqId patched in when the contig is consumed); everything stateful below stays the reference's serial loop.
      struct Pre { std::string name, seq; long len = 0; std::vector<Mz> mz; };
      std::vector<Pre> pre;
      const bool timing = getenv("ORC_TIMING") != nullptr;
      const auto tb0 = std::chrono::steady_clock::now();
      auto lap = [&](const char* what) { if (timing) fprintf(stderr, "ORC_TIMING %s at +%.2f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count()); };
      {
        SeqReader rd(fn);
        long len;
        while ((len = rd.next()) >= 0) { pre.emplace_back(); pre.back().name = rd.name; pre.back().len = len; if (!(len < P.w || len < P.k)) pre.back().seq = rd.seq; }
      }
      {
        std::atomic<size_t> next{0};
        auto work = [&]() {
          for (size_t i = next.fetch_add(1); i < pre.size(); i = next.fetch_add(1)) {
            Pre& x = pre[i];
            if (x.len < P.w || x.len < P.k) continue;
            add_minimizers(x.mz, &x.seq[0], (int)x.len, P.k, P.w, 0);   // :269 (seqId follows below)
            std::string().swap(x.seq);
          }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < std::max(1, P.threads); ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
      }
      lap("contigs read and winnowed");
      // no --maxmemory: the chunk rule decides nothing, so the per-contig count of novel hashes (a lookup per distinct hash) is
      // skipped and, with -t N, the map is filled by N threads, thread t owning the hashes with hash % N == t; every thread walks
      // the contigs in order, so every occurrence list is in position order exactly as the serial loop leaves it
      const bool bulk = P.maxMem == 0 && P.threads > 1 && lookup.empty() && byPos.empty();
      if (bulk) {
        { size_t tot = 0; for (auto& x : pre) tot += x.mz.size(); byPos.reserve(tot); }
        for (auto& x : pre) {
          if (x.len < P.w || x.len < P.k) { meta.push_back(Contig{x.name, (int32_t)x.len}); ++seen; continue; }
          for (auto& e : x.mz) e.seq = (int)seen;
          byPos.insert(byPos.end(), x.mz.begin(), x.mz.end());
          meta.push_back(Contig{x.name, (int32_t)x.len});
          ++seen;
          std::vector<Mz>().swap(x.mz);
        }
        // two passes over the entries: every thread buckets a contiguous range of byPos by partition (indices only), then thread p
        // fills partition p from the buckets in range order — position order inside every occurrence list, as the serial loop leaves it
        lap("byPos assembled");
        const size_t NP = (size_t)std::min(P.threads, 64), NT = NP, N = byPos.size();
        lookup.part.assign(NP, HashLookup::Map());
        std::vector<std::vector<std::vector<uint32_t>>> bucket(NT, std::vector<std::vector<uint32_t>>(NP));
        if (N >= ((size_t)1 << 32)) throw std::runtime_error("oracle: more than 2^32 index entries");
        auto scatter = [&](size_t t) { const size_t a = N * t / NT, b = N * (t + 1) / NT; for (auto& v : bucket[t]) v.reserve((b - a) / NP + 64); for (size_t i = a; i < b; ++i) bucket[t][byPos[i].hash % NP].push_back((uint32_t)i); };
        auto fill = [&](size_t p) { auto& m = lookup.part[p]; m.reserve(N / NP / 8 + 16);
                                    for (size_t t = 0; t < NT; ++t) for (uint32_t i : bucket[t][p]) { const Mz& e = byPos[i]; m[e.hash].push_back(Hit{e.seq, e.wpos, e.strand}); } };
        for (const std::function<void(size_t)>& phase : {std::function<void(size_t)>(scatter), std::function<void(size_t)>(fill)}) {
          std::vector<std::thread> pool;
          for (size_t t = 1; t < NP; ++t) pool.emplace_back([&phase, t] { phase(t); });
          phase(0);
          for (auto& th : pool) th.join();
          lap("lookup phase");
        }
        continue;
      }
      for (auto& x : pre) {
        const long len = x.len;
        if (len < P.w || len < P.k) {                            // :258-264 metadata only
          meta.push_back(Contig{x.name, (int32_t)len});
          ++seen;
          continue;
        }
        std::vector<Mz>& cur = x.mz;
        for (auto& e : cur) e.seq = (int)seen;
        size_t addHashes = count_novel(cur, true), addMins = cur.size();   // :274-282
        size_t totH = runHashes + addHashes, totM = runMins + addMins;
        size_t mem = memory_of(totH, totM);
        if (P.maxMem > 0 && mem > P.maxMem) {                    // :298-329 flush before adding
          compute_freq_hist();
          onChunk(*this, chunkNo);
          clear_chunk();
          runHashes = runMins = seen = 0;
          ++chunkNo;
          for (auto& e : cur) e.seq = 0;
          addHashes = count_novel(cur, false);
          totH = addHashes; totM = addMins;
          mem = memory_of(totH, totM);
          if (mem > P.maxMem)
            throw std::runtime_error("Can't index file " + fn + " within current memory limits - contig " +
                                     x.name + " is too large");
        }
        for (auto& e : cur) lookup[e.hash].push_back(Hit{e.seq, e.wpos, e.strand});   // :331-336
        byPos.insert(byPos.end(), cur.begin(), cur.end());       // :338
        meta.push_back(Contig{x.name, (int32_t)len});
        runHashes = totH; runMins = totM;
        ++seen;
        std::vector<Mz>().swap(cur);
      }
    }
    compute_freq_hist();                                         // :359 → processCurrentState
    onChunk(*this, chunkNo);
  }
};

// ---------------------------------------------------------------------------------------
// A5/A6  L1 — map/include/computeMap.hpp:277-386
// ---------------------------------------------------------------------------------------
struct L1Cand { int seq, start, end; };                           // computeMap.hpp:39-48

struct Query {
  std::string name;
  char* seq; int len;
  int sketch = 0;
  std::vector<Mz> mins;       // after L1: first `sketch` entries = sorted unique-by-hash
  std::vector<Mz> minsRaw;    // winnowing order (debug tap)
};

static inline void l1_candidates(const Query& Q, std::vector<Hit>& hits, int minHits, std::vector<L1Cand>& out) {
  if (minHits < 1) minHits = 1;                                  // :349
  std::sort(hits.begin(), hits.end());                           // :353
  size_t n = hits.size(), m = (size_t)minHits;
  for (size_t i = 0; i + m <= n; ++i) {                          // :355-357
    const Hit& a = hits[i]; const Hit& b = hits[i + m - 1];
    if (b.seq != a.seq || b.wpos - a.wpos >= Q.len) continue;    // :365
    L1Cand c{a.seq, std::max(0, b.wpos - Q.len + 1), a.wpos};    // :368
    if (!out.empty() && out.back().seq == c.seq && out.back().end >= c.start)   // :374-376
      out.back().end = std::max(c.end, out.back().end);
    else
      out.push_back(c);
  }
}

static inline bool by_hash_less(const Mz& a, const Mz& b) { return a.hash < b.hash; }   // base_types.hpp:70
static inline bool by_hash_eq(const Mz& a, const Mz& b) { return a.hash == b.hash; }    // base_types.hpp:66

struct L1Debug { std::vector<Hit> hits; int minHits = 0; };

static inline void do_l1(const RefSketch& R, const Params& P, Query& Q, std::vector<L1Cand>& out,
                         L1Debug* dbg = nullptr) {
  add_minimizers(Q.mins, Q.seq, Q.len, P.k, P.w, 0);             // :285
  Q.minsRaw = Q.mins;
  std::sort(Q.mins.begin(), Q.mins.end(), by_hash_less);         // :292 (libstdc++ introsort, not stable)
  auto ue = std::unique(Q.mins.begin(), Q.mins.end(), by_hash_eq);   // :295
  Q.sketch = (int)(ue - Q.mins.begin());                         // :298
  if (Q.sketch == 0) return;                                     // :302
  std::vector<Hit> hits;
  for (auto it = Q.mins.begin(); it != ue; ++it) {               // :307-323
    const std::vector<Hit>* f = R.lookup.find(it->hash);
    if (!f) continue;
    if (f->size() < (size_t)R.freqThreshold)                     // size_t < int → converted as in the reference
      hits.insert(hits.end(), f->begin(), f->end());
  }
  int minHits = estimate_min_hits_relaxed(Q.sketch, P.k, P.pi);  // :325
  l1_candidates(Q, hits, minHits, out);
  if (dbg) { dbg->hits = hits; dbg->minHits = minHits; }
}

// ---------------------------------------------------------------------------------------
// A7  sliding MinHash window — map/include/slidingMap.hpp:26-318
//     Ordered map hash -> {query side, reference side}; `pivot` = s-th smallest key;
//     `shared` = number of keys at or below the pivot present on both sides.
// ---------------------------------------------------------------------------------------
class SlideWindow {
  static constexpr int NA = INT_MAX;                             // slidingMap.hpp:46
  struct Cell { int wq, sq, wr, sr; };                           // :31-37
  std::map<uint32_t, Cell> m_;
  std::map<uint32_t, Cell>::iterator pivot_;
  int s_;
  static bool both(const Cell& c) { return c.wq != NA && c.wr != NA; }
 public:
  int shared = 0;
  explicit SlideWindow(const Query& Q) : s_(Q.sketch) {          // :100-131
    for (int i = 0; i < Q.sketch; ++i)
      m_.emplace_hint(m_.end(), Q.mins[i].hash, Cell{Q.mins[i].wpos, Q.mins[i].strand, NA, 0});
    pivot_ = std::next(m_.begin(), Q.sketch - 1);
  }
  void insert(const Mz& r) {                                     // :139-160, counters :263-285
    auto it = m_.find(r.hash);
    enum { UNIQ, CPLD, REV } st;
    if (it == m_.end()) { m_[r.hash] = Cell{NA, 0, r.wpos, r.strand}; st = UNIQ; }
    else { st = (it->second.wr == NA) ? CPLD : REV; it->second.wr = r.wpos; it->second.sr = r.strand; }
    if (r.hash <= pivot_->first) {
      if (st == CPLD) shared += 1;
      else if (st == UNIQ) { if (both(pivot_->second)) shared -= 1; --pivot_; }
    }
  }
  void erase(const Mz& r) {                                      // :170-214, counters :293-316
    auto it = m_.find(r.hash);
    enum { DEL, UPD, NOOP } st;
    bool pivotCase = false;
    if (it->second.wr == r.wpos) {
      if (it->second.wq == NA) {
        if (it == pivot_) { ++pivot_; if (both(pivot_->second)) shared += 1; pivotCase = true; }
        m_.erase(it); st = DEL;
      } else { it->second.wr = NA; st = UPD; }
    } else st = NOOP;
    if (pivotCase) return;
    if (r.hash <= pivot_->first) {
      if (st == UPD) shared -= 1;
      else if (st == DEL) { ++pivot_; if (both(pivot_->second)) shared += 1; }
    }
  }
  void stats(int& strandVotes, int& uniqRef) const {             // :232-254
    int seen = 0; strandVotes = uniqRef = 0;
    for (auto& kv : m_) {
      ++seen;
      if (seen <= s_ && both(kv.second)) strandVotes += kv.second.sq * kv.second.sr;
      if (kv.second.wr != NA) ++uniqRef;
    }
  }
};

struct L2Locus { int seq = 0, meanPos = 0; size_t optBeg = 0, optEnd = 0; int shared = 0; bool any = false; };

// computeMap.hpp:460-538 with MIIteratorL2.hpp:74-96 folded in.
static inline void l2_locus(const RefSketch& R, const Params& P, const Query& Q, const L1Cand& c, L2Locus& o,
                            uint64_t* nEvals = nullptr, uint64_t* nStream = nullptr) {
  const std::vector<Mz>& X = R.byPos;
  size_t first = R.search(c.seq, c.start);
  int cnt = Q.len - (P.w - 1) - (P.k - 1);                       // :470
  size_t firstEnd = R.search(c.seq, X[first].wpos + cnt);
  size_t lastEnd = R.search(c.seq, c.end + Q.len);
  if (nStream) *nStream += lastEnd - first;
  SlideWindow sw(Q);
  size_t b = first, e = firstEnd;
  int pos = X[b].wpos;                                           // MIIteratorL2.hpp:62
  for (size_t i = b; i < e; ++i) sw.insert(X[i]);
  size_t pb = b, pe = e;
  int bestBeg = 0, bestLast = 0;
  while (e < lastEnd) {                                          // :496
    if (pb != b) sw.erase(X[pb]);
    if (pe != e) sw.insert(X[pe]);
    if (nEvals) ++*nEvals;
    if (sw.shared > o.shared) {                                  // strict, :510
      o.shared = sw.shared; o.optBeg = b; o.optEnd = e; o.any = true;
      bestBeg = bestLast = X[b].wpos;
    } else if (sw.shared == o.shared) bestLast = X[b].wpos;      // :520
    pb = b; pe = e;
    int lastPos = pos + cnt - 1;                                 // MIIteratorL2.hpp:76-95
    int dBeg = X[b + 1].wpos - pos, dEnd = X[e].wpos - lastPos;
    int adv = std::min(dBeg, dEnd);
    pos += adv;
    if (adv == dBeg) ++b;
    if (adv == dEnd) ++e;
  }
  o.seq = c.seq;
  o.meanPos = (bestBeg + bestLast) / 2;                          // :537
}

struct Mapping {                                                  // base_types.hpp:133-147
  int qlen, rstart, rend, rseq; float ident, identUB; int sketch, shared, strand;
};

// computeMap.hpp:396-451
static inline void do_l2(const RefSketch& R, const Params& P, const Query& Q, const std::vector<L1Cand>& cands,
                         std::vector<Mapping>& out, uint64_t* nEvals = nullptr, uint64_t* nStream = nullptr) {
  for (auto& c : cands) {
    L2Locus l2;
    l2_locus(R, P, Q, c, l2, nEvals, nStream);
    float md = j2md(1.0 * l2.shared / Q.sketch, P.k);
    float lo = md_lower_bound(md, Q.sketch, P.k, 0.9);
    float ident = 100 * (1 - md), identUB = 100 * (1 - lo);
    if (!(identUB >= P.pi)) continue;
    if (!l2.any) continue;   // shared==0: reference would read uninitialised iterators; only reachable for pi<=0
    Mapping m{Q.len, l2.meanPos, l2.meanPos + Q.len - 1, l2.seq, ident, identUB, Q.sketch, l2.shared, 0};
    SlideWindow sw(Q);
    for (size_t i = l2.optBeg; i < l2.optEnd; ++i) sw.insert(R.byPos[i]);
    int votes, uniq;
    sw.stats(votes, uniq);
    m.strand = votes > 0 ? +1 : -1;                              // :433
    out.push_back(m);
  }
}

// computeMap.hpp:546-588 — one line per reported mapping (12 fields)
static inline void report_lines(const RefSketch& R, const Params& P, const std::string& qname,
                                const std::vector<Mapping>& ms, std::string& sink) {
  float best = 0;
  for (auto& e : ms) if (e.ident > best) best = e.ident;
  for (auto& e : ms) {
    if (!(P.reportAll || e.ident >= best - 1.0)) continue;
    std::ostringstream o;
    o << qname << " " << e.qlen << " " << "0" << " " << e.qlen - 1 << " " << (e.strand == 1 ? "+" : "-") << " "
      << R.meta[e.rseq].name << " " << R.meta[e.rseq].len << " " << e.rstart << " " << e.rend << " " << e.ident
      << " " << e.shared << " " << e.sketch << "\n";
    sink += o.str();
  }
}

}  // namespace orc
