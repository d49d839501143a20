// ORACLE — test infrastructure only (see orc_core.hpp header).
// Flat C entry points so tests/ can drive the restatement through ctypes.
#include "orc_post.hpp"

using namespace orc;

extern "C" {

uint32_t orc_kmer_hash(const char* s, int k) { return kmer_hash(s, k); }

// winnowed minimizers of one sequence; returns the count (may exceed cap: then only cap are written)
long orc_minimizers(const char* seq, int len, int k, int w, uint32_t* hash, int32_t* wpos, int32_t* strand, long cap) {
  std::string s(seq, (size_t)len);
  std::vector<Mz> v;
  add_minimizers(v, &s[0], len, k, w, 0);
  for (long i = 0; i < (long)v.size() && i < cap; ++i) { hash[i] = v[i].hash; wpos[i] = v[i].wpos; strand[i] = v[i].strand; }
  return (long)v.size();
}

double orc_binom_pmf(int n, double p, int k) { return binom_pmf(n, p, k); }
double orc_binom_sf(int n, double p, int x) { return binom_sf(n, p, x); }
int orc_binom_quantile_upper(int n, double p, double q) { return binom_quantile_upper(n, p, q); }
double orc_binom_cdf_sum(uint64_t n, double p, uint64_t k) { return binom_cdf_sum((size_t)n, p, (size_t)k); }   // evidenceUnknownSpecies columns
double orc_chi2_1df_cdf(double x) { return chi2_1df_cdf(x); }
int orc_min_hits_relaxed(int s, int k, float pi) { return estimate_min_hits_relaxed(s, k, pi); }
int orc_recommended_window(double pval, int k, float pi, int qlen, uint64_t rlen) { return recommended_window(pval, k, 4, pi, qlen, rlen); }
void orc_identity(int shared, int s, int k, float* ident, float* identUB) {
  float md = j2md(1.0 * shared / s, k);
  float lo = md_lower_bound(md, s, k, 0.9);
  *ident = 100 * (1 - md); *identUB = 100 * (1 - lo);
}

// mapping qualities for one read: in = 12-field lines joined by '\n'; out = 14-field lines joined by '\n'
long orc_add_mapq(int k, const char* in, char* out, long cap) {
  Params P; P.k = k;
  auto lines = split(in, "\n");
  while (!lines.empty() && lines.back().empty()) lines.pop_back();
  add_mapping_qualities(P, lines);
  std::string j = join(lines, "\n");
  if ((long)j.size() + 1 > cap) return -(long)j.size() - 1;
  memcpy(out, j.c_str(), j.size() + 1);
  return (long)j.size();
}

struct OrcIndex { RefSketch R; Params P; };

void* orc_index_build(const char* fasta, int k, int w) {
  auto* I = new OrcIndex;
  I->P.k = k; I->P.w = w;
  try { I->R.build({fasta}, I->P, [](RefSketch&, int) {}); } catch (...) { delete I; return nullptr; }
  return I;
}
void orc_index_free(void* h) { delete (OrcIndex*)h; }
long orc_index_entries(void* h) { return (long)((OrcIndex*)h)->R.byPos.size(); }
long orc_index_contigs(void* h) { return (long)((OrcIndex*)h)->R.meta.size(); }
int orc_index_freq_threshold(void* h) { return ((OrcIndex*)h)->R.freqThreshold; }
void orc_index_set_freq_threshold(void* h, int thr) { ((OrcIndex*)h)->R.freqThreshold = thr; }   // tests: hit lists under a forced threshold
long orc_index_unique_hashes(void* h) { return (long)((OrcIndex*)h)->R.lookup.size(); }
void orc_index_dump(void* h, uint32_t* hash, int32_t* seq, int32_t* wpos, int32_t* strand) {
  auto& v = ((OrcIndex*)h)->R.byPos;
  for (size_t i = 0; i < v.size(); ++i) { hash[i] = v[i].hash; seq[i] = v[i].seq; wpos[i] = v[i].wpos; strand[i] = v[i].strand; }
}
int orc_index_contig_len(void* h, long i) { return ((OrcIndex*)h)->R.meta[(size_t)i].len; }

// Map one read and expose every intermediate.  Arrays are caller-owned with the given capacities;
// counts are returned through n[0..4] = {sketch, hits, minHits, candidates, mappings}.
// sk_*: the sorted unique sketch (hash, strand of the surviving occurrence)
// hit_*: post-threshold seed hits sorted by (seq,wpos,strand)
// cand: triples (seq,start,end);  l2: per candidate (seq, meanPos, shared, optBeg, optEnd) before the identity filter
// map: per reported mapping (rseq, rstart, rend, shared, sketch, strand)
int orc_map_read(void* h, const char* seq, int len, float pi, int32_t* n,
                 uint32_t* sk_hash, int32_t* sk_strand, long sk_cap,
                 int32_t* hit_seq, int32_t* hit_wpos, long hit_cap,
                 int32_t* cand, long cand_cap, int64_t* l2, int32_t* map, long map_cap) {
  auto* I = (OrcIndex*)h;
  Params P = I->P; P.pi = pi; P.reportAll = true;
  std::string s(seq, (size_t)len);
  Query Q; Q.seq = &s[0]; Q.len = len;
  std::vector<L1Cand> cands; L1Debug dbg;
  do_l1(I->R, P, Q, cands, &dbg);
  n[0] = Q.sketch; n[1] = (int)dbg.hits.size(); n[2] = dbg.minHits; n[3] = (int)cands.size();
  for (long i = 0; i < Q.sketch && i < sk_cap; ++i) { sk_hash[i] = Q.mins[i].hash; sk_strand[i] = Q.mins[i].strand; }
  for (long i = 0; i < (long)dbg.hits.size() && i < hit_cap; ++i) { hit_seq[i] = dbg.hits[i].seq; hit_wpos[i] = dbg.hits[i].wpos; }
  for (long i = 0; i < (long)cands.size() && i < cand_cap; ++i) {
    cand[3 * i] = cands[i].seq; cand[3 * i + 1] = cands[i].start; cand[3 * i + 2] = cands[i].end;
    L2Locus o; l2_locus(I->R, P, Q, cands[i], o);
    l2[5 * i] = o.seq; l2[5 * i + 1] = o.meanPos; l2[5 * i + 2] = o.shared; l2[5 * i + 3] = (int64_t)o.optBeg; l2[5 * i + 4] = (int64_t)o.optEnd;
  }
  std::vector<Mapping> ms;
  do_l2(I->R, P, Q, cands, ms);
  n[4] = (int)ms.size();
  for (long i = 0; i < (long)ms.size() && i < map_cap; ++i) {
    map[6 * i] = ms[i].rseq; map[6 * i + 1] = ms[i].rstart; map[6 * i + 2] = ms[i].rend;
    map[6 * i + 3] = ms[i].shared; map[6 * i + 4] = ms[i].sketch; map[6 * i + 5] = ms[i].strand;
  }
  return 0;
}

// end-to-end file drivers (same behaviour as the CLI)
int orc_map_directly(const char* ref, const char* query, const char* prefix, int k, int w, int minReadLen, float pi,
                     int reportAll, uint64_t maxMemBytes, uint64_t* counters /*8*/) {
  try {
    Params P; P.k = k; P.w = w; P.minReadLen = minReadLen; P.pi = pi; P.reportAll = reportAll != 0; P.maxMem = maxMemBytes;
    MapCounters C;
    map_directly(P, ref, query, prefix, &C);
    if (counters) { uint64_t v[8] = {C.reads, C.bases, C.sketch, C.hits, C.cands, C.stream, C.evals, C.maps}; memcpy(counters, v, sizeof v); }
    return 0;
  } catch (std::exception& e) { std::cerr << "oracle: " << e.what() << "\n"; return -1; }
}
// returns number of EM iterations; ll[] receives the per-iteration log-likelihoods
int orc_classify(const char* mapped, const char* db, double* ll, int ll_cap) {
  try {
    EMTrace tr = do_em(mapped, db);
    for (int i = 0; i < (int)tr.ll.size() && i < ll_cap; ++i) ll[i] = tr.ll[i];
    return (int)tr.ll.size();
  } catch (std::exception& e) { std::cerr << "oracle: " << e.what() << "\n"; return -1; }
}

// classify with given start frequencies (file of "taxon value" lines) instead of uniform ones
int orc_classify_from(const char* mapped, const char* db, const char* f0File, double* ll, int ll_cap) {
  try {
    std::map<std::string, double> f0;
    { std::ifstream s(f0File); std::string t; double v; while (s >> t >> v) f0[t] = v; }
    EMTrace tr = do_em(mapped, db, true, 10000, &f0);
    for (int i = 0; i < (int)tr.ll.size() && i < ll_cap; ++i) ll[i] = tr.ll[i];
    return (int)tr.ll.size();
  } catch (std::exception& e) { std::cerr << "oracle: " << e.what() << "\n"; return -1; }
}
// the reference's example run pushed through the output writer: see finish_from_posteriors (orc_post.hpp)
int orc_finish_from_posteriors(const char* emFile, const char* metaPrefix, const char* db, const char* outPrefix) {
  try { finish_from_posteriors(emFile, metaPrefix, db, outPrefix); return 0; }
  catch (std::exception& e) { std::cerr << "oracle: " << e.what() << "\n"; return -1; }
}

// every record SeqReader returns, as tests/test_ref_host.py prints them for the real kseq: "name len fnv1a(seq)" lines + "END code"
long orc_read_dump(const char* path, char* out, long cap) {
  SeqReader rd(path);
  long used = 0, len;
  auto fnv = [](const std::string& s) { uint64_t h = 1469598103934665603ull; for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; } return h; };
  while ((len = rd.next()) >= 0) {
    const int n = snprintf(out + used, (size_t)(cap - used), "%s %ld %016llx\n", rd.name.c_str(), len, (unsigned long long)fnv(rd.seq));
    if (n < 0 || used + n >= cap) return -2;
    used += n;
  }
  const int n = snprintf(out + used, (size_t)(cap - used), "END %ld\n", len);
  if (n < 0 || used + n >= cap) return -2;
  return used + n;
}

}  // extern "C"
