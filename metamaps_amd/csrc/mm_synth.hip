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
  MM_HIP(hipStreamSynchronize(st));
  S->frozen = true;
}

// one thread per read: walk the template, apply deletions / substitutions / insertions, pack as we go
__global__ void synth_reads_kernel(const uint32_t* __restrict__ ref, const uint64_t* __restrict__ ref_base_off, int ref_len,
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
    start = (int64_t)(rnd(seed, s0, 2) % (uint64_t)(ref_len - read_len + 1));
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
  MM_REQUIRE(ref->frozen && ref->count() > 0 && ref->n_exc == 0, MM_ERR_ARG, "synthetic reads need a synthetic reference");
  const int ref_len = ref->len[0];
  for (auto L : ref->len) MM_REQUIRE(L == ref_len, MM_ERR_ARG, "synthetic reference genomes must have equal length");
  MM_REQUIRE(p.n_reads > 0 && p.read_len >= 32 && p.read_len <= ref_len, MM_ERR_ARG, "bad synthetic read parameters");
  hipStream_t st = ctx->stream;
  const int64_t G = ref->count();
  const int npick = (int)std::min<int64_t>(std::max(p.n_abundant, 1), G);
  // abundant genomes and their lognormal(sigma=1.5) weights, host side, deterministic
  std::mt19937_64 gen(p.seed ^ 0xabcdef12345ull);
  std::vector<int32_t> ids((size_t)G); for (int64_t i = 0; i < G; ++i) ids[(size_t)i] = (int32_t)i;
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
  synth_reads_kernel<<<dim3((unsigned)ceil_div(p.n_reads, 128)), dim3(128), 0, st>>>(ref->packed.p, ref->d_base.p, ref_len, d_ids.p, d_cum.p, npick,
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
  MM_HIP(hipStreamSynchronize(st));
  S->frozen = true;
}

}  // namespace mm
