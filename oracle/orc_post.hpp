// ORACLE — test infrastructure only (see orc_core.hpp header).
// mapping qualities, mapDirectly file plumbing, EM classification and its output files.
#pragma once
#include "orc_map.hpp"
#include <fstream>
#include <iostream>
#include <regex>
#include <chrono>
#include <thread>

namespace orc {

// meta/util.h:80-118
static inline std::vector<std::string> split(const std::string& in, const std::string& d) {
  std::vector<std::string> out;
  if (in.empty()) return out;
  size_t s = 0, p;
  while ((p = in.find(d, s)) != std::string::npos) { out.push_back(in.substr(s, p - s)); s = p + d.size(); }
  out.push_back(in.substr(s));
  return out;
}
static inline std::string join(const std::vector<std::string>& v, const std::string& d) {   // util.h:57-71
  std::string r;
  for (size_t i = 0; i < v.size(); ++i) { if (i) r += d; r += v[i]; }
  return r;
}

// ---------------------------------------------------------------------------------------
// A12 mapping qualities — map/mapWrap.h:215-323, :332-356
// ---------------------------------------------------------------------------------------
static inline double likelihood_set_sizes(int k, int nKmers, double ident, int sketch, int inter) {   // :332-356
  double surv = std::pow(ident, k);
  double eSurv = std::round(surv * nKmers);
  double eUnion = nKmers + (nKmers - eSurv);
  return binom_pmf(sketch, eSurv / eUnion, inter);
}
static inline void add_mapping_qualities(const Params& P, std::vector<std::string>& lines) {
  if (lines.empty()) return;
  std::vector<double> ids; std::vector<std::pair<int, int>> sizes;
  double maxId = -1; int readLen = 0;
  for (auto& ln : lines) {
    auto f = split(ln, " ");
    readLen = std::stoi(f.at(1));
    double id = std::stod(f.at(9)) / 100.0;                      // :237 — re-parsed 6-digit text
    int inter = std::stoi(f.at(10)), sk = std::stoi(f.at(11));
    if (id > maxId) maxId = id;
    ids.push_back(id); sizes.push_back({sk, inter});
  }
  maxId = std::exp(-(1 - maxId));                                // :261
  int nK = readLen - P.k + 1;                                    // :266
  std::vector<double> L; double sum = 0;
  for (auto& p : sizes) { double l = likelihood_set_sizes(P.k, nK, maxId, p.first, p.second); L.push_back(l); sum += l; }
  if (!(sum > 0)) throw std::runtime_error("likelihood_sum == 0 (reference aborts here, mapWrap.h:298)");
  for (size_t i = 0; i < lines.size(); ++i) {
    double mq = L[i] / sum;
    float corrected = std::exp(-(1 - ids[i]));                   // :311 double exp → float
    std::ostringstream add;
    add << " " << corrected * 100 << " " << mq;                  // :318-320
    lines[i] += add.str();
  }
}

// prettyprint.hpp rendering of std::vector<std::string> as used by mapWrap.h:204-205
static inline std::string pretty(const std::vector<std::string>& v) {
  std::string r = "[";
  for (size_t i = 0; i < v.size(); ++i) { if (i) r += ", "; r += v[i]; }
  return r + "]";
}

// mapWrap.h:34-213 — merges the per-chunk files PREFIX.N in read order and writes the side files.
static inline void unify_files(const std::string& prefix, const Params& P, const std::vector<std::string>& chunkFiles,
                               const std::string& queryFile, const std::string& refFile) {
  std::ofstream out(prefix);
  std::vector<std::ifstream*> in;
  for (auto& f : chunkFiles) in.push_back(new std::ifstream(f));
  std::ofstream unm(prefix + ".meta.unmappedReadsLengths");
  std::set<std::string> done;
  size_t total = 0, mapped = 0, tooShort = 0, notMapped = 0;
  auto pull = [&](size_t fi, const std::string& id) {
    std::vector<std::string> got;
    std::ifstream& s = *in[fi];
    if (!s.good()) return got;
    std::streampos keep = s.tellg();
    std::string ln;
    while (s.good()) {
      std::getline(s, ln);
      size_t sp = ln.find(' ');
      if (sp == std::string::npos) break;
      std::string rid = ln.substr(0, sp);
      if (done.count(rid)) { std::cerr << "Seems that read ID " << rid << " has already been processed\n"; exit(1); }
      if (rid == id) { got.push_back(ln); keep = s.tellg(); } else break;
    }
    s.seekg(keep);
    return got;
  };
  SeqReader rd(queryFile);
  long len;
  while ((len = rd.next()) >= 0) {
    ++total;
    if (len < P.w || len < P.k || len < P.minReadLen) { ++tooShort; continue; }
    std::vector<std::string> lines;
    for (size_t fi = 0; fi < in.size(); ++fi) { auto g = pull(fi, rd.name); lines.insert(lines.end(), g.begin(), g.end()); }
    if (lines.empty()) { ++notMapped; unm << (int)len << "\t" << rd.name << "\n"; } else ++mapped;
    add_mapping_qualities(P, lines);
    for (auto& l : lines) out << l << "\n";
    done.insert(rd.name);
  }
  for (auto* s : in) { s->close(); delete s; }
  std::ofstream meta(prefix + ".meta");
  meta << "TotalReads " << total << "\nReadsTooShort " << tooShort << "\nReadsMapped " << mapped
       << "\nReadsNotMapped " << notMapped << "\n";
  for (auto& f : chunkFiles) std::remove(f.c_str());
  std::ofstream ps(prefix + ".parameters");
  ps << "kmerSize " << P.k << "\nwindowSize " << P.w << "\nminReadLength " << P.minReadLen << "\nalphabetSize "
     << P.alphabet << "\nreferenceSize " << P.refSize << "\npercentageIdentity " << P.pi << "\np_value " << P.pval
     << "\nrefSequences " << pretty({refFile}) << "\nquerySequences " << pretty({queryFile}) << "\noutFileName "
     << prefix << "\nreportAll " << P.reportAll << "\nindex " << "" << "\nmaximumMemory " << P.maxMem << "\n";
}

struct MapCounters { uint64_t reads = 0, bases = 0, sketch = 0, hits = 0, cands = 0, stream = 0, evals = 0, maps = 0, chunks = 0; double map_seconds = 0; };

// computeMap.hpp:104-172 writing PREFIX.N.  The reference's thread pool only guarantees that output order equals
// input order (ThreadPool.hpp:13-17); here `threads` workers take contiguous read ranges and the ranges are
// written in order, which gives the same file.
static inline void map_query_file(const RefSketch& R, const Params& P, const std::string& queryFile,
                                  const std::string& outFile, MapCounters* C = nullptr) {
  std::ofstream out(outFile);
  struct Item { std::string name, seq; };
  std::vector<Item> items;
  {
    SeqReader rd(queryFile);
    long len;
    while ((len = rd.next()) >= 0) {
      if (len < P.w || len < P.k || len < P.minReadLen) continue;
      items.push_back(Item{rd.name, rd.seq});
    }
  }
  auto t0 = std::chrono::steady_clock::now();                    // "Time spent mapping the query", computeMap.hpp:91-96
  const int nthr = std::max(1, std::min<int>(P.threads, (int)items.size()));
  std::vector<std::string> chunks((size_t)nthr);
  std::vector<MapCounters> cs((size_t)nthr);
  auto work = [&](int t) {
    const size_t lo = items.size() * (size_t)t / (size_t)nthr, hi = items.size() * (size_t)(t + 1) / (size_t)nthr;
    for (size_t i = lo; i < hi; ++i) {
      Query Q; Q.name = items[i].name; Q.seq = &items[i].seq[0]; Q.len = (int)items[i].seq.size();
      std::vector<L1Cand> cands; std::vector<Mapping> ms; L1Debug dbg;
      do_l1(R, P, Q, cands, &dbg);
      uint64_t ev = 0, st = 0;
      do_l2(R, P, Q, cands, ms, &ev, &st);
      report_lines(R, P, Q.name, ms, chunks[(size_t)t]);
      MapCounters& c = cs[(size_t)t];
      c.reads++; c.bases += Q.len; c.sketch += Q.sketch; c.hits += dbg.hits.size(); c.cands += cands.size();
      c.stream += st; c.evals += ev; c.maps += ms.size();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nthr; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  for (auto& ch : chunks) out << ch;
  if (C) {
    C->chunks++;
    for (auto& c : cs) { C->reads += c.reads; C->bases += c.bases; C->sketch += c.sketch; C->hits += c.hits; C->cands += c.cands;
                         C->stream += c.stream; C->evals += c.evals; C->maps += c.maps; }
    C->map_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
}

// mapWrap.h:407-441 for one query/prefix pair
static inline void map_directly(Params P, const std::string& refFile, const std::string& queryFile,
                                const std::string& prefix, MapCounters* C = nullptr) {
  std::vector<std::string> chunkFiles;
  RefSketch R;
  R.build({refFile}, P, [&](RefSketch& r, int n) {
    std::string f = prefix + "." + std::to_string(n);
    map_query_file(r, P, queryFile, f, C);
    chunkFiles.push_back(f);
  });
  unify_files(prefix, P, chunkFiles, queryFile, refFile);
}

// ---------------------------------------------------------------------------------------
// Taxonomy — meta/taxonomy.h:137-246, :51-135
// ---------------------------------------------------------------------------------------
struct TaxNode { std::string id, parent, rank, sci; };
struct Taxonomy {
  std::map<std::string, TaxNode> T;
  static std::vector<std::string> fields(std::string ln) {
    static const std::regex re("\\s*\\|\\s*");
    ln = std::regex_replace(ln, re, "|");
    return split(ln, "|");
  }
  explicit Taxonomy(const std::string& dir) {
    std::map<std::string, std::string> sci;
    std::ifstream nm(dir + "/names.dmp");
    if (!nm.is_open()) { std::cerr << "Cannot open file " << dir << "/names.dmp\n"; exit(1); }
    std::string ln;
    while (std::getline(nm, ln)) {
      if (ln.empty()) continue;
      auto f = fields(ln);
      if (f.size() > 3 && f[3] == "scientific name") sci[f[0]] = f[1];
      else if (f.size() > 3 && f[3] == "genbank common name") sci[f[0]];   // entry exists, scientific name may stay empty
    }
    std::ifstream nd(dir + "/nodes.dmp");
    if (!nd.is_open()) { std::cerr << "Cannot open file " << dir << "/nodes.dmp\n"; exit(1); }
    while (std::getline(nd, ln)) {
      if (ln.empty()) continue;
      auto f = fields(ln);
      if (!sci.count(f[0])) { std::cerr << "No name for taxon ID " << f[0] << "\n"; exit(1); }
      T[f[0]] = TaxNode{f[0], f[1], f[2], sci[f[0]]};
    }
  }
  std::vector<std::string> upward(std::string id) const {        // :113-129
    std::vector<std::string> u{id};
    while (id != "1") { id = T.at(id).parent; u.push_back(id); }
    return u;
  }
  std::map<std::string, std::string> upward_by_ranks(const std::string& id, const std::set<std::string>& want) const {  // :76-111
    std::map<std::string, std::string> r;
    for (auto& n : upward(id)) {
      const std::string& rank = T.at(n).rank;
      if (!want.empty() && !want.count(rank)) continue;
      if (rank != "no rank") {
        if (r.count(rank)) { std::cerr << "Node " << id << " has multiple entries for rank " << rank << "\n"; exit(1); }
        r[rank] = n;
      }
    }
    for (auto& w : want) if (!r.count(w)) r[w] = "Undefined";
    return r;
  }
  std::string first_non_x(const std::string& id) const {         // :51-74
    std::string r = id;
    while (r.find('x') != std::string::npos) r = T.at(r).parent;
    return r;
  }
};

// meta/fEM.h:1396-1415
static inline std::string extract_taxon(const std::string& contig) {
  static const std::regex re("kraken:taxid\\|(x?\\d+)");
  std::smatch m;
  if (!std::regex_search(contig, m, re)) throw std::runtime_error("Could not extract taxon ID from contig identifier '" + contig + "'");
  return m[1];
}

struct Loc { std::string taxon, contig; double identity; size_t readLen, start, stop; double p, l; };   // fEM.h:39-50
using TaxonInfo = std::map<std::string, std::map<std::string, size_t>>;

// meta/fEM.h:234-373
static inline std::vector<Loc> mapping_locations(const TaxonInfo& TI, const std::map<std::string, double>& f,
                                                 const std::vector<std::string>& lines) {
  std::set<std::string> sawContig, sawTaxon;
  std::vector<Loc> locs;
  long long readLen = -1;
  for (auto& ln : lines) {
    auto fld = split(ln, " ");
    Loc l;
    l.contig = fld.at(5);
    l.start = std::stoull(fld.at(7)); l.stop = std::stoull(fld.at(8));
    l.taxon = extract_taxon(l.contig);
    if (!TI.count(l.taxon)) { std::cerr << "Unknown taxonID '" << l.taxon << "'\n"; exit(1); }
    double mq;
    try { mq = std::stod(fld.at(13)); }
    catch (const std::out_of_range&) { if (fld.at(13).find("e-") != std::string::npos) mq = 0; else throw; }   // :269-281
    if (readLen == -1) readLen = std::stoi(fld.at(1));
    l.identity = std::stod(fld.at(9)) / 100.0;
    l.p = mq; l.readLen = (size_t)readLen; l.l = 0;
    sawTaxon.insert(l.taxon); sawContig.insert(l.contig);
    locs.push_back(l);
  }
  std::map<std::string, size_t> nLoc;                            // :325-348
  for (auto& t : sawTaxon) {
    size_t n = 0;
    for (auto& c : TI.at(t)) {
      if ((long long)c.second >= readLen) n += c.second - readLen + 1;
      else if (sawContig.count(c.first)) ++n;
    }
    nLoc[t] = n;
  }
  double tot = 0;
  for (auto& l : locs) { l.l = f.at(l.taxon) * (1 / (double)nLoc.at(l.taxon)) * l.p; tot += l.l; }   // :353
  for (auto& l : locs) l.p = l.l / tot;
  return locs;
}

static inline std::vector<std::vector<std::string>> group_reads(const std::string& file) {   // fEM.h:1171-1214
  std::vector<std::vector<std::string>> groups;
  std::ifstream s(file);
  std::string ln, curId; std::vector<std::string> cur;
  while (std::getline(s, ln)) {
    if (ln.empty()) continue;
    std::string id = ln.substr(0, ln.find(' '));
    if (id != curId) { if (!cur.empty()) groups.push_back(cur); curId = id; cur.clear(); }
    cur.push_back(ln);
  }
  if (!cur.empty()) groups.push_back(cur);
  return groups;
}

// meta/util.h:118-172
static inline size_t overlap_first_larger(size_t aL, size_t aR, size_t bL, size_t bR) {
  size_t bLen = bR - bL + 1;
  if (aL <= bL && aR >= bR) return bLen;
  else if (bL >= aL && bL <= aR) return aR - bL + 1;
  else if (bR >= aL && bR <= aR) return bR - aL + 1;
  return 0;
}
static inline size_t overlap(size_t aL, size_t aR, size_t bL, size_t bR) {
  return (aR - aL + 1 > bR - bL + 1) ? overlap_first_larger(aL, aR, bL, bR) : overlap_first_larger(bL, bR, aL, aR);
}
// read coverage in 1000-bp windows of the contigs carrying best mappings — meta/fEM.h:684 (window size), :730-776
// (accumulation, including the unsigned wrap of the last window's length when the contig is not a multiple of the
// window: :744 subtracts after incrementing n_windows), :805-845 (output)
struct Coverage {
  size_t W = 1000;
  std::map<std::string, std::map<std::string, std::vector<size_t>>> cov, cnt;   // bases / reads per window (:685-686)
  std::map<std::string, std::map<std::string, size_t>> lastWin;
  void add(const TaxonInfo& TI, const Loc& b) {
    const size_t L = TI.at(b.taxon).at(b.contig);
    if (!cov[b.taxon].count(b.contig)) {
      size_t n = L / W;
      if (n == 0) { n++; lastWin[b.taxon][b.contig] = L; }
      else if (n * W != L) { n++; lastWin[b.taxon][b.contig] = L - n * W; }
      else lastWin[b.taxon][b.contig] = W;
      cov[b.taxon][b.contig].resize(n, 0);
      cnt[b.taxon][b.contig].resize(n, 0);
    }
    const size_t stop = b.stop >= L ? L - 1 : b.stop;
    for (size_t pos = b.start; pos <= stop; pos += W) {
      const size_t wi = pos / W, ws = wi * W;
      size_t we = (wi + 1) * W - 1;
      if (we > L) we = L - 1;
      cov.at(b.taxon).at(b.contig).at(wi) += overlap(ws, we, b.start, stop);
      cnt.at(b.taxon).at(b.contig).at(wi)++;
    }
  }
  template <typename TaxT> void write(const std::string& file, const TaxT& T) const {
    std::ofstream o(file);
    o << "taxonID\tequalCoverageUnitLabel\tcontigID\tstart\tstop\tnBases\treadCoverage\n";
    for (auto& t : cov) for (auto& c : t.second)
      for (size_t wi = 0; wi < c.second.size(); ++wi) {
        size_t wl = W;
        if (wi == c.second.size() - 1) wl = lastWin.at(t.first).at(c.first);
        o << t.first << "\t" << T.T.at(t.first).sci << "\t" << c.first << "\t" << wi * W << "\t" << (wi + 1) * W - 1 << "\t" << c.second[wi] << "\t"
          << (double)c.second[wi] / (double)wl << "\n";
      }
  }
};

// Distribution functions the unknown-species table takes from Boost.Math (fEM.h:1014, :1064, :1101, :1107); the version
// is unpinned, values only reach the file through std::to_string's six decimals.
// chi-squared, one degree of freedom: P(X <= x) = P(|Z| <= sqrt(x)) = erf(sqrt(x/2))
static inline double chi2_1df_cdf(double x) { return x <= 0 ? 0.0 : std::erf(std::sqrt(x / 2)); }
// binomial P(X <= k), summed term by term in extended precision from the mode outwards being unnecessary here: k+1 terms
static inline double binom_cdf_sum(size_t n, double p, size_t k) {
  if (k >= n) return 1.0;
  if (p <= 0) return 1.0;
  if (p >= 1) return 0.0;
  const long double lp = std::log((long double)p), lq = log1pl(-(long double)p);
  long double acc = 0;
  for (size_t i = 0; i <= k; ++i)
    acc += expl(lgammal((long double)n + 1) - lgammal((long double)i + 1) - lgammal((long double)(n - i) + 1) + (long double)i * lp + (long double)(n - i) * lq);
  return (double)(acc > 1 ? 1.0L : acc);
}

// PREFIX.EM.evidenceUnknownSpecies — meta/fEM.h:846-1132, Ns per window from DBDIR/contigNstats_windowSize_1000.txt (:1421-1470).
// All size_t arithmetic is kept as written there, including running sums fed by the wrapped length of a last partial
// window (Coverage above).  Where the reference would stop on an assert (:1049-1050, an expected count of zero), the
// identity columns of that row are "NA" and a warning goes to stderr.  Returns false (nothing written) when the DB has no
// contigNstats file — the reference asserts there (:1427).
template <typename TaxT>
static inline bool write_unknown_species(const std::string& file, const std::string& dbDir, const TaxT& T, const Coverage& C,
                                         const std::map<std::string, std::vector<double>>& identPerTaxon, long long maxReadLen, size_t minReads) {
  std::map<std::string, std::vector<size_t>> Ns;                 // by contig ID (:1421-1470)
  {
    std::ifstream s(dbDir + "/contigNstats_windowSize_" + std::to_string(C.W) + ".txt");
    if (!s.is_open()) return false;
    std::string ln;
    while (std::getline(s, ln)) {
      while (!ln.empty() && (ln.back() == '\n' || ln.back() == '\r')) ln.pop_back();
      if (ln.empty()) continue;
      auto fl = split(ln, "\t");
      if (fl.size() != 3) throw std::runtime_error("Format error contigNstats; wrong number of fields: " + ln);
      if (!C.cov.count(fl[0]) || !C.cov.at(fl[0]).count(fl[1])) continue;
      auto nf = split(fl[2], ";");
      if (nf.size() != C.cov.at(fl[0]).at(fl[1]).size()) throw std::runtime_error("contigNstats: window count mismatch for " + fl[1]);
      std::vector<size_t> v; for (auto& e : nf) v.push_back(std::stoull(e));
      Ns[fl[1]] = v;
    }
    for (auto& t : C.cov) for (auto& c : t.second) if (!Ns.count(c.first)) throw std::runtime_error("Missing entry " + c.first + " in contigNstats");
  }
  std::map<std::string, std::string> contigTaxon;
  for (auto& t : C.cov) for (auto& c : t.second) contigTaxon[c.first] = t.first;

  // taxon with the highest median identity and the bottom-third quantile of its identities (:846-890)
  std::string bestTaxon; double bestMedian = 0, oneThird = 0, oneThirdP = 0;
  for (auto& e : identPerTaxon) {
    std::vector<double> id = e.second;
    if (id.size() >= 3 && id.size() >= minReads) {
      std::sort(id.begin(), id.end());
      const double med = id.at(id.size() / 2);
      if (bestTaxon.empty() || med > bestMedian) {
        bestMedian = med; bestTaxon = e.first;
        oneThird = id.at((size_t)(id.size() * (1.0 / 3.0)));
        size_t n13 = 0; for (double v : id) if (v <= oneThird) ++n13;
        oneThirdP = (double)n13 / (double)id.size();
      }
    }
  }

  // usable windows: at least maxReadLen bases of N-poor (<= 2 % N) windows on either side (:893-996)
  const size_t need = (size_t)maxReadLen;
  std::map<std::string, size_t> nWin, nUsable, nUsableReads, nUsableZero;
  for (auto& cd : Ns) {
    const std::string& tx = contigTaxon.at(cd.first);
    const std::vector<size_t>& n = cd.second;
    const size_t lastLen = C.lastWin.at(tx).at(cd.first);
    std::vector<size_t> fwd(n.size(), 0), bwd(n.size(), 0);
    size_t run = 0;
    for (size_t i = 0; i < n.size(); ++i) {
      fwd[i] = run;
      const size_t wl = i == n.size() - 1 ? lastLen : C.W;
      if ((double)n[i] / (double)wl <= 0.02) run += wl; else run = 0;
    }
    run = 0;
    for (long long i = (long long)n.size() - 1; i >= 0; --i) {
      bwd[(size_t)i] = run;
      const size_t wl = i == (long long)n.size() - 1 ? lastLen : C.W;
      if ((double)n[(size_t)i] / (double)wl <= 0.02) run += wl; else run = 0;
    }
    size_t use = 0, useReads = 0, useZero = 0;
    const std::vector<size_t>& reads = C.cnt.at(tx).at(cd.first);
    for (size_t i = 0; i < n.size(); ++i)
      if (fwd[i] >= need && bwd[i] >= need) { ++use; useReads += reads.at(i); if (reads.at(i) == 0) ++useZero; }
    nWin[tx] += n.size(); nUsable[tx] += use; nUsableReads[tx] += useReads; nUsableZero[tx] += useZero;
  }

  std::ofstream o(file);
  o << "taxonID\tspecies\tgenus\tnReads\tpropBottomThirdReadIdentities\texpectedPropBottomThirdReadIdentities\tpValue_BottomThirdReadIdentities"
       "\tcoverageWindows_totalGenome\tcoverageWindows_usable\tcoverageWindows_usable_averageCoverage\tcoverageWindows_usable_coverageIsZero"
       "\tcoverageWindows_usable_coverageIsZero_expected\tcoverageWindows_usable_coverageIsZero_P\n";
  for (auto& e : identPerTaxon) {
    const std::string& tx = e.first;
    const std::vector<double>& id = e.second;
    std::string prop = "NA", pId = "NA", exp13 = "NA";
    if (!bestTaxon.empty()) {                                    // :1026-1068
      size_t obs = 0; for (double v : id) if (v <= oneThird) ++obs;
      const size_t obsRest = id.size() - obs;
      const double ex = oneThirdP * id.size(), exRest = id.size() - ex;
      if (ex > 0 && exRest > 0) {
        exp13 = std::to_string(oneThirdP);
        const double stat = std::pow(obs - ex, 2) / ex + std::pow(obsRest - exRest, 2) / exRest;
        prop = std::to_string((double)obs / (double)id.size());
        pId = std::to_string(1 - chi2_1df_cdf(stat));
      } else std::cerr << "evidenceUnknownSpecies: expected count of zero for taxon " << tx << " (the reference asserts here); identity columns NA\n";
    }
    std::string avg = "NA", zeroExp = "NA", zeroP = "NA";        // :1070-1114
    if (nUsable.at(tx) > 0) {
      const double a = (double)nUsableReads.at(tx) / (double)nUsable.at(tx);
      avg = std::to_string(a);
      if (a == 0) { zeroExp = std::to_string(nUsable.at(tx)); zeroP = std::to_string(1); }
      else {
        const double p0 = std::exp(-a);                          // Poisson(a) at 0
        zeroExp = std::to_string(nUsable.at(tx) * p0);
        double pv = 1;
        if (nUsableZero.at(tx) > 0) pv = 1 - binom_cdf_sum(nUsable.at(tx), p0, nUsableZero.at(tx) - 1);
        zeroP = std::to_string(pv);
      }
    }
    auto up = T.upward_by_ranks(tx, {"species", "genus"});
    o << tx << "\t" << up.at("species") << "\t" << up.at("genus") << "\t" << id.size() << "\t" << prop << "\t" << exp13 << "\t" << pId << "\t"
      << nWin.at(tx) << "\t" << nUsable.at(tx) << "\t" << avg << "\t" << nUsableZero.at(tx) << "\t" << zeroExp << "\t" << zeroP << "\n";
  }
  return true;
}

struct EMTrace { std::vector<double> ll; std::map<std::string, double> f; };

// meta/fEM.h:52-215 (WIMP)
static inline void write_wimp(const std::string& fn, const Taxonomy& T, std::map<std::string, double> freq,
                              const std::map<std::string, size_t>& reads, size_t nTotal, size_t nUnmapped, size_t nTooShort) {
  std::set<std::string> levels{"species", "genus", "family", "order", "phylum", "superkingdom"};   // :1417-1420
  std::map<std::string, std::set<std::string>> keys;
  std::map<std::string, std::map<std::string, double>> fL;
  std::map<std::string, std::map<std::string, size_t>> rL;
  for (auto& kv : freq) {
    auto up = T.upward_by_ranks(kv.first, levels); up["definedGenomes"] = kv.first;
    for (auto& u : up) {
      fL[u.first][u.second] += kv.second; keys[u.first].insert(u.second);
      if (fL[u.first][u.second] > 1) fL[u.first][u.second] = 1;  // :98-101
    }
  }
  for (auto& kv : reads) {
    auto up = T.upward_by_ranks(kv.first, levels); up["definedGenomes"] = kv.first;
    for (auto& u : up) {
      if (fL[u.first].count(u.second) == 0) rL[u.first][u.second] = 0;   // :108 (tests f_per_level; harmless)
      rL[u.first][u.second] += kv.second; keys[u.first].insert(u.second);
    }
  }
  long long nMappable = (long long)nTotal - (long long)nTooShort, nMapped = nMappable - (long long)nUnmapped;
  std::ofstream o(fn);
  o << "AnalysisLevel\ttaxonID\tName\tAbsolute\tEMFrequency\tPotFrequency\n";
  for (auto& lv : keys) {
    const std::string& L = lv.first;
    std::map<std::string, double> emF;
    double sumF = 0;
    for (auto& t : lv.second) {
      double f = fL[L].count(t) ? fL[L][t] : 0; size_t r = rL[L].count(t) ? rL[L][t] : 0;
      sumF += f; fL[L][t] = f; rL[L][t] = r;
    }
    for (auto& t : lv.second) { fL[L][t] /= sumF; emF[t] = fL[L][t]; }
    double propMapped = (double)nMapped / nMappable, propNot = (double)nUnmapped / nMappable;
    for (auto& t : lv.second) fL[L][t] *= propMapped;
    double emUnm = 0; size_t nUnmUndef = nUnmapped;
    for (auto& t : lv.second) {
      if (t != "Undefined")
        o << L << "\t" << t << "\t" << T.T.at(t).sci << "\t" << rL[L][t] << "\t" << emF[t] << "\t" << fL[L][t] << "\n";
      else { nUnmUndef += rL[L][t]; emUnm += emF[t]; propNot += fL[L][t]; }
    }
    o << L << "\t" << 0 << "\t" << "Unclassified" << "\t" << nUnmUndef << "\t" << emUnm << "\t" << propNot << "\n";
    o << L << "\t" << -3 << "\t" << "totalReads" << "\t" << nTotal << "\t" << 0 << "\t" << 0 << "\n";
    o << L << "\t" << -3 << "\t" << "readsLongEnough" << "\t" << nMappable << "\t" << 0 << "\t" << 0 << "\n";
    o << L << "\t" << -3 << "\t" << "readsLongEnough_unmapped" << "\t" << nUnmapped << "\t" << 0 << "\t" << 0 << "\n";
  }
}

// everything doEM writes behind its loop (fEM.h:663-803): .EM, .EM.reads2Taxon, .krona, .EM.lengthAndIdentitiesPerMappingUnit, cleanF (:1135-1163),
// .EM.WIMP, .EM.contigCoverage, .EM.evidenceUnknownSpecies.  `locs_of(group)` = the read's mapping locations with their final posteriors.
template <typename LocsOf>
static inline void write_classify_outputs(const std::string& mapped, const std::string& dbDir, const Taxonomy& T, const TaxonInfo& TI, std::map<std::string, double> f,
                                          const std::map<std::string, size_t>& st, const std::vector<std::vector<std::string>>& groups, LocsOf locs_of, size_t minReadsU) {
  const size_t nUnmapped = st.at("ReadsNotMapped"), nTooShort = st.at("ReadsTooShort"), nTotal = st.at("TotalReads");
  std::ofstream em(mapped + ".EM"), r2t(mapped + ".EM.reads2Taxon"), kr(mapped + ".EM.reads2Taxon.krona"), li(mapped + ".EM.lengthAndIdentitiesPerMappingUnit");
  li << "AnalysisLevel\tID\treadI\tIdentity\tLength\n";     // :686
  std::map<std::string, size_t> readsPer;
  Coverage coverage;
  std::map<std::string, std::vector<double>> identPerTaxon;      // :691
  long long maxReadLen = -1;                                     // :692
  size_t readI = 0;
  for (auto& g : groups) {                                       // :684-779
    auto locs = locs_of(g);
    std::string rid;
    for (size_t i = 0; i < g.size(); ++i) {
      auto fld = split(g[i], " "); rid = fld.at(0);
      fld.at(13) = std::to_string(locs[i].p);                    // :705
      em << join(fld, " ") << "\n";
    }
    size_t best = 0;                                             // :217-232 first strict maximum
    for (size_t i = 1; i < locs.size(); ++i) if (locs[i].p > locs[best].p) best = i;
    r2t << rid << "\t" << locs[best].taxon << "\n";
    kr << rid << "\t" << T.first_non_x(locs[best].taxon) << "\t" << locs[best].p << "\n";
    readsPer[locs[best].taxon]++;
    li << "EqualCoverageUnit\t" << locs[best].contig << "\t" << readI++ << "\t" << locs[best].identity << "\t" << locs[best].readLen << "\n";   // :711
    identPerTaxon[locs[best].taxon].push_back(locs[best].identity);   // :718-722
    if ((long long)locs[best].readLen > maxReadLen) maxReadLen = (long long)locs[best].readLen;
    coverage.add(TI, locs[best]);
  }
  {                                                              // :785-790
    std::ifstream s(mapped + ".meta.unmappedReadsLengths"); std::string ln;
    while (std::getline(s, ln)) {
      if (ln.empty()) continue;
      auto fl = split(ln, "\t");
      r2t << fl.at(1) << "\t" << 0 << "\n"; kr << fl.at(1) << "\t" << 0 << "\t" << 0 << "\n";
    }
  }
  {                                                              // cleanF :1135-1163
    double minF = 0.9 * (1.0 / (double)st.at("ReadsMapped"));
    std::set<std::string> drop;
    for (auto& e : f) if (e.second < minF && !readsPer.count(e.first)) drop.insert(e.first);
    for (auto& d : drop) f.erase(d);
    double s = 0; for (auto& e : f) s += e.second;
    for (auto& e : f) e.second /= s;
  }
  write_wimp(mapped + ".EM.WIMP", T, f, readsPer, nTotal, nUnmapped, nTooShort);
  coverage.write(mapped + ".EM.contigCoverage", T);
  if (!write_unknown_species(mapped + ".EM.evidenceUnknownSpecies", dbDir, T, coverage, identPerTaxon, maxReadLen, minReadsU))
    std::cerr << "no contigNstats_windowSize_1000.txt in " << dbDir << ": .EM.evidenceUnknownSpecies not written\n";
}

// meta/fEM.h:466-803 (EM loop + .EM / .EM.reads2Taxon / .krona / .EM.WIMP).  Single summation order
// (the reference sums per OpenMP thread chunk, then across threads; with -t 1 it is exactly this order).
// f0: start frequencies instead of the uniform ones of :491-495 (tests/test_example_pins.py starts the loop AT a fixed point to see it hold it)
static inline EMTrace do_em(const std::string& mapped, const std::string& dbDir, bool writeFiles = true, size_t minReadsU = 10000,
                            const std::map<std::string, double>* f0 = nullptr) {
  EMTrace tr;
  std::set<std::string> taxa;                                    // :1366-1394
  {
    std::ifstream s(mapped); std::string ln;
    while (std::getline(s, ln)) if (!ln.empty()) taxa.insert(extract_taxon(split(ln, " ").at(5)));
  }
  if (taxa.empty()) throw std::runtime_error("No relevant taxon IDs found in your mappings file");
  std::map<std::string, size_t> st;                              // :398-421
  { std::ifstream s(mapped + ".meta"); std::string a; size_t b; while (s >> a >> b) st[a] = b; }
  (void)st.at("ReadsNotMapped"); (void)st.at("ReadsTooShort"); (void)st.at("TotalReads");
  TaxonInfo TI;                                                  // :1320-1364
  {
    std::ifstream s(dbDir + "/taxonInfo.txt"); std::string ln;
    if (!s.is_open()) { std::cerr << "Could not open file " << dbDir << "/taxonInfo.txt\n"; exit(1); }
    while (std::getline(s, ln)) {
      if (ln.empty()) continue;
      auto f = split(ln, " ");
      for (auto& c : split(f.at(1), ";")) { auto kv = split(c, "="); TI[f.at(0)][kv.at(0)] = std::stoull(kv.at(1)); }
    }
  }
  Taxonomy T(dbDir + "/taxonomy");
  std::map<std::string, double> f;
  for (auto& t : taxa) f[t] = f0 ? (f0->count(t) ? f0->at(t) : 0.0) : 1 / (double)taxa.size();   // :491-495
  auto groups = group_reads(mapped);
  double llPrev = 0; size_t iter = 0; bool go = true;
  while (go) {                                                   // :501-661
    std::map<std::string, double> fn = f; for (auto& e : fn) e.second = 0;
    double ll = 0;
    for (auto& g : groups) {
      auto locs = mapping_locations(TI, f, g);
      double lr = 0;
      for (auto& l : locs) { lr += l.l; fn.at(l.taxon) += l.p; }
      ll += std::log(lr);
    }
    double sum = 0; for (auto& e : fn) sum += e.second;
    for (auto& e : fn) e.second /= sum;
    tr.ll.push_back(ll);
    if (iter > 0) {
      double diff = ll - llPrev, rel = 1 - ll / llPrev;
      if (diff <= 1 && rel < 0.0001) go = false;                 // :636
    }
    f = fn; ++iter; llPrev = ll;
  }
  tr.f = f;
  if (!writeFiles) return tr;
  write_classify_outputs(mapped, dbDir, T, TI, f, st, groups, [&](const std::vector<std::string>& g) { return mapping_locations(TI, f, g); }, minReadsU);
  return tr;
}

// The part of doEM behind the loop (fEM.h:663-803) on its own, fed with mappings that ALREADY carry final posteriors in field 14 (the
// format of PREFIX.EM): f is one M step from them (fEM.h:575, :606-615), everything else as in do_em.  Exists so that the reference's
// own example run (tests/golden/example/example.EM, whose field 14 the reference computed) can be pushed through this writer and held
// against the reference's example.EM.WIMP / .reads2Taxon / .krona / .lengthAndIdentitiesPerMappingUnit (tests/test_example_pins.py).
static inline void finish_from_posteriors(const std::string& emFile, const std::string& metaPrefix, const std::string& dbDir, const std::string& outPrefix) {
  std::map<std::string, size_t> st;
  { std::ifstream s(metaPrefix + ".meta"); std::string a; size_t b; while (s >> a >> b) st[a] = b; }
  TaxonInfo TI;
  {
    std::ifstream s(dbDir + "/taxonInfo.txt"); std::string ln;
    while (std::getline(s, ln)) {
      if (ln.empty()) continue;
      auto f = split(ln, " ");
      for (auto& c : split(f.at(1), ";")) { auto kv = split(c, "="); TI[f.at(0)][kv.at(0)] = std::stoull(kv.at(1)); }
    }
  }
  Taxonomy T(dbDir + "/taxonomy");
  auto groups = group_reads(emFile);
  auto locs_of = [&](const std::vector<std::string>& g) {
    std::vector<Loc> locs;
    for (auto& ln : g) {
      auto fld = split(ln, " ");
      Loc l; l.contig = fld.at(5); l.taxon = extract_taxon(l.contig);
      l.start = std::stoull(fld.at(7)); l.stop = std::stoull(fld.at(8));
      l.identity = std::stod(fld.at(9)) / 100.0; l.readLen = (size_t)std::stoi(fld.at(1));
      l.p = std::stod(fld.at(13)); l.l = 0;
      locs.push_back(l);
    }
    return locs;
  };
  std::map<std::string, double> f;
  for (auto& g : groups) for (auto& l : locs_of(g)) f[l.taxon] += l.p;
  double sum = 0; for (auto& e : f) sum += e.second;
  for (auto& e : f) e.second /= sum;
  // outputs next to outPrefix; the unmapped read list is taken from metaPrefix
  {
    std::ifstream src(metaPrefix + ".meta.unmappedReadsLengths"); std::ofstream dst(outPrefix + ".meta.unmappedReadsLengths"); dst << src.rdbuf();
  }
  write_classify_outputs(outPrefix, dbDir, T, TI, f, st, groups, locs_of, 10000);
}

}  // namespace orc
