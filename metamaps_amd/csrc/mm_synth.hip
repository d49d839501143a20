// Counter-based generators: every base is a pure function of (seed, stream, position), so the kernels are
// embarrassingly parallel and the data are reproducible on any number of GPUs.
//
// Reference model (simplified to substitutions so that it stays position-parallel):
//   genus root  (4 species per genus)        uniform random bases
//   species root = genus root with `genus_divergence` substitutions
//   strain 0 = species root; strain j>0 = species root with `strain_divergence` substitutions
// One contig per genome, genomes in (species, strain) order.
#include "mm_synth.hpp"
#include <cmath>
#include <random>
#include <algorithm>

namespace mm {

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t rnd(uint64_t seed, uint64_t stream, uint64_t ctr) {
  return mix64(mix64(seed ^ (stream * 0xd1342543de82ef95ULL)) + ctr * 0x2545f4914f6cdd1dULL);
}
__host__ __device__ inline float u01(uint64_t r) { return (float)(r >> 40) * (1.0f / 16777216.0f); }

__device__ inline uint32_t ref_base(uint64_t seed, int sp, int st, uint64_t p, float genus_div, float strain_div) {
  uint32_t b = (uint32_t)(rnd(seed, 0x100000000ull + (uint64_t)(sp >> 2), p) & 3);
  uint64_t r1 = rnd(seed, 0x200000000ull + (uint64_t)sp, p);
  if (u01(r1) < genus_div) b = (b + 1 + (uint32_t)((r1 & 0xffff) % 3)) & 3;
  if (st > 0) {
    uint64_t r2 = rnd(seed, 0x300000000ull + ((uint64_t)sp << 12) + (uint64_t)st, p);
    if (u01(r2) < strain_div) b = (b + 1 + (uint32_t)((r2 & 0xffff) % 3)) & 3;
  }
  return b;
}

__global__ void synth_ref_kernel(uint32_t* __restrict__ packed, int64_t words_per_genome, int64_t n_genomes, int genome_len, int strains,
                                 uint64_t seed, float genus_div, float strain_div) {
  int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (wi >= words_per_genome * n_genomes) return;
  int64_t g = wi / words_per_genome, wl = wi - g * words_per_genome;
  int sp = (int)(g / strains), st = (int)(g % strains);
  uint32_t word = 0;
  for (int b = 0; b < 16; ++b) {
    int64_t p = wl * 16 + b;
    if (p < genome_len) word |= ref_base(seed, sp, st, (uint64_t)p, genus_div, strain_div) << (2 * b);
  }
  packed[wi] = word;
}

void synth_reference(mm_ctx* ctx, const mm_synth_ref_params& p, mm_seqset* S) {
  MM_REQUIRE(p.n_species > 0 && p.strains_per_species > 0 && p.strains_per_species < 4096 && p.genome_len >= 64, MM_ERR_ARG, "bad synthetic reference parameters");
  hipStream_t st = ctx->stream;
  const int64_t G = (int64_t)p.n_species * p.strains_per_species;
  const int64_t wpg = ((int64_t)p.genome_len + 15) / 16;
  S->ctx = ctx;
  S->len.assign((size_t)G, p.genome_len);
  S->base.resize((size_t)G + 1);
  for (int64_t g = 0; g <= G; ++g) S->base[(size_t)g] = (uint64_t)(g * wpg * 16);
  S->total_bases = G * p.genome_len;
  S->packed.alloc((size_t)(G * wpg) + 1);
  S->d_base.alloc((size_t)G + 1); S->d_base.upload(S->base.data(), (size_t)G + 1, st);
  S->d_len.alloc((size_t)G); S->d_len.upload(S->len.data(), (size_t)G, st);
  S->n_exc = 0;
  const int64_t nw = G * wpg;
  MM_REQUIRE(ceil_div(nw, 256) < (1LL << 31), MM_ERR_LIMIT, "synthetic reference too large for one launch");
  synth_ref_kernel<<<dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, st>>>(S->packed.p, wpg, G, p.genome_len, p.strains_per_species, p.seed,
                                                                            p.genus_divergence, p.strain_divergence);
  MM_KERNEL_CHECK();
  MM_HIP(mm::stream_sync(st));
  S->frozen = true;
}

// ---------------------------------------------------------------------------------------------------
// SURVEY.md §8 D1 community: what shapes the occurrence-list length distribution (repeats), the exception path of K1 (N runs),
// the contig table (lognormal lengths, shuffled order) and the number of near-identical candidates per read (1..12 strains).
// Still counter based: a base is a pure function of (seed, contig, position); the host only draws the small tables.
// ---------------------------------------------------------------------------------------------------
struct SynthContig {
  uint64_t word0;              // first packed word
  int32_t len, kind;           // kind 0 microbial, 1 human-like
  int32_t species, genus, strain_id;
  float genus_div, strain_div;
  int32_t seg0, nseg;          // microbial: piecewise map strain position -> root position (SynthSeg)
  int32_t human_id;
};
struct SynthSeg { int32_t at, root; };   // strain positions >= at map to root + (p - at); root < 0: inserted (random) bases
constexpr int SYN_GRAN = 256;

__device__ inline uint32_t community_base(const SynthContig& c, const SynthSeg* __restrict__ segs, const float* __restrict__ fam_cum,
                                          const int32_t* __restrict__ fam_gran, int n_fam, float repeat_fraction, uint64_t seed, int64_t p) {
  if (c.kind == 0) {
    int k = 0;
    for (int i = 1; i < c.nseg; ++i) if (segs[c.seg0 + i].at <= p) k = i;
    const SynthSeg sg = segs[c.seg0 + k];
    if (sg.root < 0) return (uint32_t)(rnd(seed, 0x500000000ull + (uint64_t)c.strain_id, (uint64_t)p) & 3);
    const uint64_t q = (uint64_t)sg.root + (uint64_t)(p - sg.at);
    uint32_t b = (uint32_t)(rnd(seed, 0x100000000ull + (uint64_t)c.genus, q) & 3);
    const uint64_t r1 = rnd(seed, 0x200000000ull + (uint64_t)c.species, q);
    if (u01(r1) < c.genus_div) b = (b + 1 + (uint32_t)((r1 & 0xffff) % 3)) & 3;
    const uint64_t r2 = rnd(seed, 0x300000000ull + (uint64_t)c.strain_id, q);
    if (u01(r2) < c.strain_div) b = (b + 1 + (uint32_t)((r2 & 0xffff) % 3)) & 3;
    return b;
  }
  const uint64_t gran = (uint64_t)p / SYN_GRAN;
  const uint64_t h = rnd(seed, 0x600000000ull + (uint64_t)c.human_id, gran);
  if (u01(h) >= repeat_fraction) return (uint32_t)(rnd(seed, 0x700000000ull + (uint64_t)c.human_id, (uint64_t)p) & 3);
  const uint64_t h2 = mix64(h);
  const float u = u01(h2);
  int lo = 0, hi = n_fam - 1;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (fam_cum[mid] < u) lo = mid + 1; else hi = mid; }
  const uint64_t h3 = mix64(h2);
  const int gi = (int)(h3 % (uint64_t)fam_gran[lo]);
  const bool rev = (h3 >> 40) & 1;
  const float div = 0.02f + 0.18f * u01(mix64(h3));
  const int o = (int)(p % SYN_GRAN);
  uint32_t b = (uint32_t)(rnd(seed, 0x800000000ull + (uint64_t)lo, (uint64_t)(gi * SYN_GRAN + (rev ? SYN_GRAN - 1 - o : o))) & 3);
  if (rev) b = 3u - b;
  const uint64_t r = rnd(seed, 0x900000000ull + (uint64_t)c.human_id, (uint64_t)p);
  if (u01(r) < div) b = (b + 1 + (uint32_t)((r & 0xffff) % 3)) & 3;
  return b;
}

__global__ void synth_community_kernel(uint32_t* __restrict__ packed, int64_t n_words, const SynthContig* __restrict__ contigs, int n_contigs,
                                       const SynthSeg* __restrict__ segs, const float* __restrict__ fam_cum, const int32_t* __restrict__ fam_gran,
                                       int n_fam, float repeat_fraction, uint64_t seed) {
  for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_contigs - 1;                              // last contig whose first word is <= wi
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int64_t)contigs[mid].word0 <= wi) lo = mid; else hi = mid - 1; }
    const SynthContig c = contigs[lo];
    const int64_t p0 = (wi - (int64_t)c.word0) * 16;
    uint32_t word = 0;
    for (int b = 0; b < 16; ++b) if (p0 + b < c.len) word |= community_base(c, segs, fam_cum, fam_gran, n_fam, repeat_fraction, seed, p0 + b) << (2 * b);
    packed[wi] = word;
  }
}

// strains per species: 1..12, heavier towards few, summing to n_genomes (the first draws of the generator's random stream)
static std::vector<int> community_strains(const mm_synth_community_params& p, std::mt19937_64& gen) {
  const int SP = p.n_species, NG = p.n_genomes;
  std::vector<int> strains((size_t)SP, 1);
  int left = NG - SP;
  while (left > 0) { const int sp = (int)(gen() % (uint64_t)SP); if (strains[(size_t)sp] < 12) { ++strains[(size_t)sp]; --left; } }   // (terminates: NG <= 12 * SP)
  return strains;
}
// genome (= strain, numbered species by species) -> species of the community mm_synth_community generates from the same parameters
void synth_community_species(const mm_synth_community_params& p, int32_t* genome_species) {
  MM_REQUIRE(p.n_genomes > 0 && p.n_species > 0 && p.n_species <= p.n_genomes && p.n_genomes <= 12 * (int64_t)p.n_species, MM_ERR_ARG, "bad synthetic community parameters");
  std::mt19937_64 gen(p.seed ^ 0x5eedc0ffeeull);
  const std::vector<int> strains = community_strains(p, gen);
  int g = 0;
  for (int sp = 0; sp < p.n_species; ++sp) for (int k = 0; k < strains[(size_t)sp]; ++k) genome_species[g++] = sp;
}

void synth_community(mm_ctx* ctx, const mm_synth_community_params& p, mm_seqset* S, int32_t* contig_genome) {
  MM_REQUIRE(p.n_genomes <= 12 * (int64_t)p.n_species, MM_ERR_ARG, "synthetic community: at most 12 strains per species (n_genomes <= 12 * n_species)");
  MM_REQUIRE(p.n_genomes > 0 && p.n_species > 0 && p.n_species <= p.n_genomes && p.n_genera > 0 && p.n_genera <= p.n_species && p.min_len >= 64 &&
             p.max_len >= p.min_len && p.median_len > 0 && p.human_contigs >= 0 && (p.human_contigs == 0 || (p.human_bases >= 4096 * (int64_t)p.human_contigs && p.n_repeat_families > 0)),
             MM_ERR_ARG, "bad synthetic community parameters");
  hipStream_t st = ctx->stream;
  std::mt19937_64 gen(p.seed ^ 0x5eedc0ffeeull);
  auto uni = [&]() { return (double)(gen() >> 11) * (1.0 / 9007199254740992.0); };
  auto gauss = [&]() { const double u1 = ((gen() >> 11) + 1) * (1.0 / 9007199254740993.0), u2 = uni(); return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); };
  const int SP = p.n_species, NG = p.n_genomes;
  std::vector<int> strains = community_strains(p, gen);
  std::vector<double> sp_len((size_t)SP);
  for (int i = 0; i < SP; ++i) sp_len[(size_t)i] = std::min<double>(p.max_len, std::max<double>(p.min_len, p.median_len * std::exp(p.sigma_len * gauss())));
  std::vector<int64_t> hlen((size_t)p.human_contigs);
  { double tot = 0; std::vector<double> wv((size_t)p.human_contigs);   // chromosome-like proportions (largest about five times the smallest)
    for (int i = 0; i < p.human_contigs; ++i) { wv[(size_t)i] = 1.0 + 4.0 * (double)(p.human_contigs - i) / std::max(p.human_contigs, 1); tot += wv[(size_t)i]; }
    for (int i = 0; i < p.human_contigs; ++i) hlen[(size_t)i] = std::min<int64_t>((int64_t)MAX_SEQ_LEN, std::max<int64_t>(4096, (int64_t)((double)p.human_bases * wv[(size_t)i] / tot))); }
  if (p.total_bases_target > 0) {                                // scale the microbial part so that the whole reference has the asked size
    int64_t human = 0; for (auto L : hlen) human += L;
    double micro = 0; for (int i = 0; i < SP; ++i) micro += sp_len[(size_t)i] * strains[(size_t)i];
    const double f = std::max(0.05, ((double)p.total_bases_target - (double)human) / micro);
    for (auto& L : sp_len) L = std::min<double>(p.max_len, std::max<double>(p.min_len, L * f));
  }
  std::vector<SynthContig> cs; std::vector<SynthSeg> segs; std::vector<int32_t> genome_of;
  int strain_id = 0;
  for (int sp = 0; sp < SP; ++sp) {
    const int genus = (int)((int64_t)sp * p.n_genera / SP);
    const float gdiv = p.genus_div_min + (float)uni() * (p.genus_div_max - p.genus_div_min);
    for (int s = 0; s < strains[(size_t)sp]; ++s, ++strain_id) {
      SynthContig c{};
      c.kind = 0; c.species = sp; c.genus = genus; c.strain_id = strain_id; c.genus_div = gdiv;
      c.strain_div = p.strain_div_min * std::pow(std::max(p.strain_div_max, p.strain_div_min) / std::max(p.strain_div_min, 1e-6f), (float)uni());   // log-uniform
      const int64_t root_len = (int64_t)sp_len[(size_t)sp];
      // block events at increasing root positions -> piecewise position map
      const int ne = p.strain_indel_events > 0 ? (int)(gen() % (uint64_t)(p.strain_indel_events + 1)) : 0;
      std::vector<int64_t> at((size_t)ne);
      for (auto& x : at) x = (int64_t)(uni() * (double)root_len);
      std::sort(at.begin(), at.end());
      c.seg0 = (int32_t)segs.size();
      int64_t spos = 0, rpos = 0;
      segs.push_back(SynthSeg{0, 0});
      for (int e = 0; e < ne; ++e) {
        const int64_t x = std::max(at[(size_t)e], rpos + 16);
        if (x + 6000 >= root_len) break;
        const int64_t d = 50 + (int64_t)(uni() * uni() * 4950);
        spos += x - rpos; rpos = x;
        if (gen() & 1) { rpos += d; segs.push_back(SynthSeg{(int32_t)spos, (int32_t)rpos}); }              // deletion: skip d root bases
        else { segs.push_back(SynthSeg{(int32_t)spos, -1}); spos += d; segs.push_back(SynthSeg{(int32_t)spos, (int32_t)rpos}); }   // insertion of d random bases
      }
      spos += root_len - rpos;
      c.nseg = (int32_t)segs.size() - c.seg0;
      c.len = (int32_t)std::min<int64_t>(spos, (int64_t)MAX_SEQ_LEN);
      c.human_id = -1;
      cs.push_back(c); genome_of.push_back(strain_id);
    }
  }
  for (int i = 0; i < p.human_contigs; ++i) {
    SynthContig c{}; c.kind = 1; c.len = (int32_t)hlen[(size_t)i]; c.human_id = i; c.species = c.genus = -1; c.strain_id = -1;
    cs.push_back(c); genome_of.push_back(NG);
  }
  // shuffled contig order (buildDB.pl:386,547)
  const size_t C = cs.size();
  std::vector<size_t> order(C); for (size_t i = 0; i < C; ++i) order[i] = i;
  for (size_t i = C; i > 1; --i) std::swap(order[i - 1], order[(size_t)(gen() % (uint64_t)i)]);
  std::vector<SynthContig> sc(C);
  S->ctx = ctx; S->len.resize(C); S->base.assign(C + 1, 0); S->total_bases = 0;
  for (size_t i = 0; i < C; ++i) {
    sc[i] = cs[order[i]];
    if (contig_genome) contig_genome[i] = genome_of[order[i]];
    S->len[i] = sc[i].len;
    sc[i].word0 = S->base[i] >> 4;
    S->base[i + 1] = S->base[i] + (((uint64_t)sc[i].len + 15) & ~15ull);
    S->total_bases += sc[i].len;
  }
  // repeat library: Zipf-weighted families of 1..24 granules
  const int NF = std::max(p.n_repeat_families, 1);
  std::vector<float> fam_cum((size_t)NF); std::vector<int32_t> fam_gran((size_t)NF);
  { double tot = 0; for (int f = 0; f < NF; ++f) tot += 1.0 / (f + 1.0);
    double acc = 0; for (int f = 0; f < NF; ++f) { acc += 1.0 / (f + 1.0) / tot; fam_cum[(size_t)f] = (float)acc; fam_gran[(size_t)f] = 1 + (int32_t)(gen() % 24); }
    fam_cum[(size_t)NF - 1] = 2.0f; }
  // N runs of the human-like contigs: five per contig, each a fifth of the contig's share
  std::vector<uint64_t> es; std::vector<uint32_t> el; std::vector<uint8_t> eb;
  for (size_t i = 0; i < C; ++i) {
    if (sc[i].kind != 1 || p.n_fraction <= 0) continue;
    const int64_t L = sc[i].len, run = std::max<int64_t>(1, (int64_t)((double)L * p.n_fraction / 5.0));
    for (int j = 0; j < 5; ++j) {
      const int64_t at = j == 0 ? 0 : (j == 4 ? L - run : (int64_t)((double)L * (0.2 * j + 0.1 * uni())));
      if (at < 0 || at + run > L) continue;
      es.push_back(S->base[i] + (uint64_t)at); el.push_back((uint32_t)run); eb.push_back((uint8_t)'N');
    }
  }
  { std::vector<size_t> o2(es.size()); for (size_t i = 0; i < o2.size(); ++i) o2[i] = i;   // sorted by start, non-overlapping
    std::sort(o2.begin(), o2.end(), [&](size_t a, size_t b) { return es[a] < es[b]; });
    std::vector<uint64_t> es2; std::vector<uint32_t> el2; std::vector<uint8_t> eb2; uint64_t end = 0;
    for (size_t i : o2) { if (es[i] < end) continue; es2.push_back(es[i]); el2.push_back(el[i]); eb2.push_back(eb[i]); end = es[i] + el[i]; }
    es.swap(es2); el.swap(el2); eb.swap(eb2); }
  const int64_t nw = (int64_t)(S->base[C] >> 4);
  S->packed.alloc((size_t)nw + 1);
  S->d_base.alloc(C + 1); S->d_base.upload(S->base.data(), C + 1, st);
  S->d_len.alloc(C); S->d_len.upload(S->len.data(), C, st);
  S->n_exc = (int64_t)es.size();
  if (S->n_exc) {
    S->exc_start.alloc(es.size()); S->exc_start.upload(es.data(), es.size(), st);
    S->exc_len.alloc(el.size()); S->exc_len.upload(el.data(), el.size(), st);
    S->exc_byte.alloc(eb.size()); S->exc_byte.upload(eb.data(), eb.size(), st);
  }
  DBuf<SynthContig> d_c(C); d_c.upload(sc.data(), C, st);
  DBuf<SynthSeg> d_s(std::max<size_t>(segs.size(), 1)); d_s.upload(segs.data(), segs.size(), st);
  DBuf<float> d_fc((size_t)NF); d_fc.upload(fam_cum.data(), (size_t)NF, st);
  DBuf<int32_t> d_fg((size_t)NF); d_fg.upload(fam_gran.data(), (size_t)NF, st);
  synth_community_kernel<<<dim3(ctx->cus * 64), dim3(256), 0, st>>>(S->packed.p, nw, d_c.p, (int)C, d_s.p, d_fc.p, d_fg.p, NF, p.repeat_fraction, p.seed);
  MM_KERNEL_CHECK();
  MM_HIP(mm::stream_sync(st));
  S->frozen = true;
}

// one thread per read: walk the template, apply deletions / substitutions / insertions, pack as we go
__global__ void synth_reads_kernel(const uint32_t* __restrict__ ref, const uint64_t* __restrict__ ref_base_off, const int32_t* __restrict__ ref_len_of,
                                   const int32_t* __restrict__ pick_genome, const float* __restrict__ pick_cum, int n_pick,
                                   int64_t n_reads, int read_len_max, int read_len_min, int64_t stride_words, uint64_t seed, float sub, float ins, float del,
                                   float frac_random, uint32_t* __restrict__ out, int32_t* __restrict__ out_len, int32_t* __restrict__ truth) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t s0 = 0x400000000ull + (uint64_t)r;
  uint64_t h = rnd(seed, s0, 0);
  const bool random_read = u01(h) < frac_random;
  int read_len = read_len_max;
  if (read_len_min > 0 && read_len_min < read_len_max)          // log-uniform lengths
    read_len = min(read_len_max, (int)((float)read_len_min * expf(u01(rnd(seed, s0, 4)) * logf((float)read_len_max / (float)read_len_min))));
  int g = -1; int64_t start = 0; bool rev = false;
  if (!random_read) {
    float u = u01(rnd(seed, s0, 1));
    int lo = 0, hi = n_pick - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (pick_cum[mid] < u) lo = mid + 1; else hi = mid; }
    g = pick_genome[lo];
    start = (int64_t)(rnd(seed, s0, 2) % (uint64_t)(ref_len_of[g] - read_len + 1));
    rev = rnd(seed, s0, 3) & 1;
  }
  uint32_t* dst = out + r * stride_words;
  const int cap = (int)(stride_words * 16);
  uint32_t word = 0; int n = 0;
  auto emit = [&](uint32_t b) {
    if (n >= cap) return;
    word |= b << (2 * (n & 15));
    if ((n & 15) == 15) { dst[n >> 4] = word; word = 0; }
    ++n;
  };
  const uint64_t gb = g >= 0 ? ref_base_off[g] : 0;
  for (int t = 0; t < read_len; ++t) {
    uint64_t e = rnd(seed, s0, 16 + (uint64_t)t);
    uint32_t b;
    if (random_read) b = (uint32_t)(e >> 60) & 3;
    else {
      uint64_t p = rev ? (uint64_t)(start + read_len - 1 - t) : (uint64_t)(start + t);
      uint64_t gp = gb + p;
      b = (ref[gp >> 4] >> (2 * (gp & 15))) & 3u;
      if (rev) b = 3u - b;
    }
    float ue = u01(e);
    if (ue < del) { /* deleted */ }
    else {
      if (ue < del + sub) b = (b + 1 + (uint32_t)((e & 0xffff) % 3)) & 3;
      emit(b);
    }
    uint64_t e2 = mix64(e);
    if (u01(e2) < ins) emit((uint32_t)(e2 & 3));
  }
  if (n & 15) dst[n >> 4] = word;
  out_len[r] = n;
  if (truth) truth[r] = g;
}

void synth_reads(mm_ctx* ctx, const mm_seqset* ref, const mm_synth_read_params& p, mm_seqset* S, int32_t* truth_genome) {
  MM_REQUIRE(ref->frozen && ref->count() > 0, MM_ERR_ARG, "synthetic reads need an uploaded reference");
  MM_REQUIRE(p.n_reads > 0 && p.read_len >= 32, MM_ERR_ARG, "bad synthetic read parameters");
  hipStream_t st = ctx->stream;
  // abundant contigs (among those long enough for the longest read) and their lognormal(sigma=1.5) weights, host side, deterministic
  std::mt19937_64 gen(p.seed ^ 0xabcdef12345ull);
  std::vector<int32_t> ids;
  for (int64_t i = 0; i < ref->count(); ++i) if (ref->len[(size_t)i] >= p.read_len) ids.push_back((int32_t)i);
  const int64_t G = (int64_t)ids.size();
  MM_REQUIRE(G > 0, MM_ERR_ARG, "no reference contig is as long as the reads");
  const int npick = (int)std::min<int64_t>(std::max(p.n_abundant, 1), G);
  for (int i = 0; i < npick; ++i) { size_t j = (size_t)i + (size_t)(gen() % (uint64_t)(G - i)); std::swap(ids[(size_t)i], ids[j]); }
  ids.resize((size_t)npick);
  std::vector<double> wgt((size_t)npick); double tot = 0;
  for (int i = 0; i < npick; ++i) {
    double u1 = ((gen() >> 11) + 1) * (1.0 / 9007199254740993.0), u2 = (gen() >> 11) * (1.0 / 9007199254740992.0);
    double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    wgt[(size_t)i] = std::exp(1.5 * z); tot += wgt[(size_t)i];
  }
  std::vector<float> cum((size_t)npick); double acc = 0;
  for (int i = 0; i < npick; ++i) { acc += wgt[(size_t)i] / tot; cum[(size_t)i] = (float)acc; }
  cum[(size_t)npick - 1] = 2.0f;
  DBuf<int32_t> d_ids((size_t)npick); d_ids.upload(ids.data(), (size_t)npick, st);
  DBuf<float> d_cum((size_t)npick); d_cum.upload(cum.data(), (size_t)npick, st);

  const int64_t cap = (int64_t)(p.read_len * (1.0 + 1.3 * p.ins_rate) + 64);
  const int64_t sw = (cap + 15) / 16;
  S->ctx = ctx;
  S->packed.alloc((size_t)(p.n_reads * sw) + 1);
  DBuf<int32_t> d_truth((size_t)p.n_reads);
  S->d_len.alloc((size_t)p.n_reads);
  synth_reads_kernel<<<dim3((unsigned)ceil_div(p.n_reads, 128)), dim3(128), 0, st>>>(ref->packed.p, ref->d_base.p, ref->d_len.p, d_ids.p, d_cum.p, npick,
      p.n_reads, p.read_len, p.read_len_min, sw, p.seed, p.sub_rate, p.ins_rate, p.del_rate, p.frac_random, S->packed.p, S->d_len.p, d_truth.p);
  MM_KERNEL_CHECK();
  S->len = S->d_len.to_host(st, (size_t)p.n_reads);
  if (truth_genome) { auto t = d_truth.to_host(st); memcpy(truth_genome, t.data(), sizeof(int32_t) * (size_t)p.n_reads); }
  S->base.resize((size_t)p.n_reads + 1);
  S->total_bases = 0;
  for (int64_t r = 0; r <= p.n_reads; ++r) S->base[(size_t)r] = (uint64_t)(r * sw * 16);
  for (auto L : S->len) S->total_bases += L;
  S->d_base.alloc((size_t)p.n_reads + 1); S->d_base.upload(S->base.data(), (size_t)p.n_reads + 1, st);
  S->n_exc = 0;
  MM_HIP(mm::stream_sync(st));
  S->frozen = true;
}

}  // namespace mm
