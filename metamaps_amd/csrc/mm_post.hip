// K8 mapping qualities (replaces mapWrap::addMappingQualities, mapWrap.h:215-323, :332-356) and
// K9 EM classification step (replaces the per-read callback + reduction of meta::doEM, fEM.h:501-615,
// with getMappingLocations' likelihood, fEM.h:350-361) + the RCCL all-reduce of the sufficient statistics.
#include "mm_map.hpp"
#include "mm_em.hpp"
#include <rccl/rccl.h>
#include <rocprim/rocprim.hpp>
#include <cfloat>
#include <numeric>

namespace mm {

// ---------------------------------------------------------------------------------------------------
// Text round trips are part of the reference's numerics: identities and mapping qualities
// are printed with 6 significant digits (ostream default) and parsed back with stod.  parse6() returns
// the double that stod would return for the "%g" rendering of v.  For a float-valued v the scaling by a
// power of ten is exact in double, so ties (…5 exactly) are detected exactly and rounded half-to-even as
// glibc's printf does.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ inline double pow10_int(int t) {
  const double tab[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  if (t >= 0 && t <= 22) return tab[t];
  return pow(10.0, (double)t);
}
__host__ __device__ inline double parse6(double v) {
  if (v == 0.0 || !(v == v)) return v;
  double a = fabs(v);
  int e = (int)floor(log10(a));
  {                                                              // fix log10 rounding at decade boundaries
    double pe = e >= 0 ? pow10_int(e) : 1.0 / pow10_int(-e);
    if (a < pe) --e; else if (a >= pe * 10.0) ++e;
  }
  int t = 5 - e;
  double x = t >= 0 ? a * pow10_int(t) : a / pow10_int(-t);
  double d = rint(x);
  if (d >= 1e6) { d /= 10.0; t -= 1; }
  double r = t >= 0 ? d / pow10_int(t) : d * pow10_int(-t);
  if (r < DBL_MIN) r = 0.0;                                      // stod throws out_of_range → reference uses 0, fEM.h:269-275
  return v < 0 ? -r : r;
}

// float math of Stat::j2md (map_stats.hpp:44) and the identity of computeMap.hpp:406,411
__host__ __device__ inline float dev_identity(int shared, int s, int k) {
  float j = (float)(1.0 * shared / s);
  float md;
  if (j == 0) md = 1.0f;
  else if (j == 1) md = 0.0f;
  else md = (float)((-1.0 / k) * log(2.0 * j / (double)(1 + j)));
  return 100 * (1 - md);
}

__device__ inline double dev_binom_pmf(int n, double p, int k) {  // boost pdf(binomial), mapWrap.h:340
  if (k < 0 || k > n) return 0.0;
  if (p == 0) return k == 0 ? 1.0 : 0.0;
  if (p == 1) return k == n ? 1.0 : 0.0;
  if (n == 0) return 1.0;
  if (k == 0) return pow(1 - p, (double)n);
  if (k == n) return pow(p, (double)k);
  return exp(lgamma((double)n + 1) - lgamma((double)k + 1) - lgamma((double)(n - k) + 1) + k * log(p) + (n - k) * log1p(-p));
}

// K8 in three launches, one thread per MAPPING where the arithmetic is (round 3 ran one thread per READ through three serial loops
// of f64 lgamma / pow: 3.2 ms per 10^5-read batch, 423 k records).
//   mapq_identity_kernel   per record: the identity the mappings file would carry (6 significant digits, mapWrap.h:237)
//   mapq_likelihood_kernel per record: the read's best identity -> p (mapWrap.h:261-266, :335-338), the record's binomial mass (:340)
//   mapq_normalise_kernel  per read:   sum of the masses IN RECORD ORDER (the reference's loop, :279-296) and the division (:301)
__global__ void __launch_bounds__(256) mapq_identity_kernel(const mm_map_record* __restrict__ rec, int64_t n_rec, int k, double* __restrict__ ident) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rec) return;
  ident[i] = parse6((double)dev_identity(rec[i].shared, rec[i].sketch, k)) / 100.0;
}
__global__ void __launch_bounds__(256) mapq_likelihood_kernel(mm_map_record* __restrict__ rec, const uint64_t* __restrict__ rec_off, const int32_t* __restrict__ read_len,
                                                              const double* __restrict__ ident, int64_t n_rec, int k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rec) return;
  const int64_t r = rec[i].read;
  const uint64_t lo = rec_off[r], hi = rec_off[r + 1];
  double maxid = -1;
  for (uint64_t j = lo; j < hi; ++j) { const double id = ident[j]; if (id > maxid) maxid = id; }
  maxid = exp(-(1 - maxid));                                      // :261
  const int nk = read_len[r] - k + 1;                             // :266
  const double surv = pow(maxid, (double)k);                      // :335
  const double es = round(surv * nk);
  const double eu = nk + (nk - es);
  const double p = es / eu;
  rec[i].mapq = dev_binom_pmf(rec[i].sketch, p, rec[i].shared);
}
__global__ void __launch_bounds__(256) mapq_normalise_kernel(mm_map_record* __restrict__ rec, const uint64_t* __restrict__ rec_off, int64_t n_reads, int* __restrict__ err) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t lo = rec_off[r], hi = rec_off[r + 1];
  if (lo == hi) return;
  double sum = 0;
  for (uint64_t i = lo; i < hi; ++i) sum += rec[i].mapq;
  if (!(sum > 0)) { atomicExch(err, 1); return; }                 // reference asserts here, :298
  for (uint64_t i = lo; i < hi; ++i) rec[i].mapq = rec[i].mapq / sum;
}

void mapping_add_qualities(mm_ctx* ctx, mm_mapping* M, int k) {
  hipStream_t st = ctx->stream;
  if (M->n_reads == 0 || M->n_rec == 0) { M->has_mapq = true; return; }
  DBuf<int> err(1); err.zero(st);
  DBuf<double> ident((size_t)M->n_rec);
  const unsigned gb = (unsigned)ceil_div(M->n_rec, 256);
  mapq_identity_kernel<<<dim3(gb), dim3(256), 0, st>>>(M->rec.p, M->n_rec, k, ident.p);
  MM_KERNEL_CHECK();
  mapq_likelihood_kernel<<<dim3(gb), dim3(256), 0, st>>>(M->rec.p, M->rec_off.p, M->d_read_len.p, ident.p, M->n_rec, k);
  MM_KERNEL_CHECK();
  mapq_normalise_kernel<<<dim3((unsigned)ceil_div(M->n_reads, 256)), dim3(256), 0, st>>>(M->rec.p, M->rec_off.p, M->n_reads, err.p);
  MM_KERNEL_CHECK();
  auto he = err.to_host(st);
  MM_REQUIRE(he[0] == 0, MM_ERR_NUMERIC, "likelihood sum of a read is 0 (the reference aborts here, mapWrap.h:298)");
  M->has_mapq = true;
}

// ---------------------------------------------------------------------------------------------------
// K9 EM
// ---------------------------------------------------------------------------------------------------
__global__ void em_estep_kernel(const int64_t* __restrict__ read_off, const int32_t* __restrict__ taxon, const double* __restrict__ mapq,
                                const double* __restrict__ inv_nloc, const double* __restrict__ f, int64_t n_reads,
                                double* __restrict__ post, double* __restrict__ ll_read) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const int64_t lo = read_off[r], hi = read_off[r + 1];
  double sum = 0;
  for (int64_t i = lo; i < hi; ++i) { double l = f[taxon[i]] * inv_nloc[i] * mapq[i]; post[i] = l; sum += l; }   // fEM.h:353
  for (int64_t i = lo; i < hi; ++i) post[i] = post[i] / sum;                                                      // :361
  ll_read[r] = hi > lo ? log(sum) : 0.0;                                                                          // fEM.h:578
}

// fixed-shape sum of v[idx[lo..hi)]: 64 lanes stride the segment, then a butterfly.  The shape depends only on
// the segment length, so two taxa with identical contribution sequences get bit-identical sums.
__global__ void __launch_bounds__(64) em_taxon_sum_kernel(const double* __restrict__ post, const int64_t* __restrict__ tstart,
                                                          const int64_t* __restrict__ perm, double* __restrict__ f_partial) {
  const int t = blockIdx.x, lane = threadIdx.x;
  double acc = 0;
  for (int64_t j = tstart[t] + lane; j < tstart[t + 1]; j += 64) acc += post[perm[j]];
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (lane == 0) f_partial[t] = acc;
}
__global__ void __launch_bounds__(256) sum_blocks_kernel(const double* __restrict__ v, int64_t n, double* __restrict__ block_sum) {
  __shared__ double sh[256];
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  sh[threadIdx.x] = i < n ? v[i] : 0.0;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) block_sum[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(256) sum_final_kernel(const double* __restrict__ block_sum, int64_t nb, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0;
  for (int64_t i = threadIdx.x; i < nb; i += 256) acc += block_sum[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) *out = sh[0];
}
__global__ void em_best_kernel(const int64_t* __restrict__ read_off, const double* __restrict__ post, int64_t n_reads, int64_t* __restrict__ best) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const int64_t lo = read_off[r], hi = read_off[r + 1];
  int64_t b = lo;                                                 // first strict maximum, fEM.h:217-232
  for (int64_t i = lo + 1; i < hi; ++i) if (post[i] > post[b]) b = i;
  best[r] = hi > lo ? b : -1;
}

void em_create(mm_ctx* ctx, int64_t n_reads, const int64_t* read_off, const int32_t* taxon, const double* mapq, const double* inv_nloc,
               int32_t n_taxa, mm_em* E) {
  hipStream_t st = ctx->stream;
  E->ctx = ctx; E->n_reads = n_reads; E->n_taxa = n_taxa;
  const int64_t ne = n_reads > 0 ? read_off[n_reads] : 0;
  E->n_entries = ne;
  for (int64_t i = 0; i < ne; ++i) MM_REQUIRE(taxon[i] >= 0 && taxon[i] < n_taxa, MM_ERR_ARG, "taxon index out of range");
  // CSR by taxon, entries kept in input (read) order inside each taxon
  std::vector<int64_t> ts((size_t)n_taxa + 1, 0), perm((size_t)std::max<int64_t>(ne, 1));
  for (int64_t i = 0; i < ne; ++i) ts[(size_t)taxon[i] + 1]++;
  for (int32_t t = 0; t < n_taxa; ++t) ts[(size_t)t + 1] += ts[(size_t)t];
  { std::vector<int64_t> cur(ts.begin(), ts.end() - 1); for (int64_t i = 0; i < ne; ++i) perm[(size_t)cur[(size_t)taxon[i]]++] = i; }
  E->read_off.alloc((size_t)n_reads + 1); E->read_off.upload(read_off, (size_t)n_reads + 1, st);
  E->taxon.alloc((size_t)std::max<int64_t>(ne, 1)); E->taxon.upload(taxon, (size_t)ne, st);
  E->mapq.alloc((size_t)std::max<int64_t>(ne, 1)); E->mapq.upload(mapq, (size_t)ne, st);
  E->inv_nloc.alloc((size_t)std::max<int64_t>(ne, 1)); E->inv_nloc.upload(inv_nloc, (size_t)ne, st);
  E->tstart.alloc((size_t)n_taxa + 1); E->tstart.upload(ts.data(), ts.size(), st);
  E->perm.alloc(perm.size()); E->perm.upload(perm.data(), (size_t)ne, st);
  E->post.alloc((size_t)std::max<int64_t>(ne, 1));
  E->ll_read.alloc((size_t)std::max<int64_t>(n_reads, 1));
  E->f.alloc((size_t)n_taxa);
  E->partial.alloc((size_t)n_taxa + 2);
  E->block_sum.alloc((size_t)ceil_div(std::max<int64_t>(n_reads, 1), 256));
  MM_HIP(mm::stream_sync(st));
}

// ---------------------------------------------------------------------------------------------------
// EM problem straight from device-resident mapping records (map -> classify without the text file in between)
// ---------------------------------------------------------------------------------------------------
// per record: taxon of its contig, the mapping quality as the file would carry it (6 significant digits,
// mapWrap.h:318-320 -> fEM.h:262), 1/nLoc with nLoc = sum over the taxon's contigs of (len - L + 1) if len >= L, else 1 if the
// read has a mapping on that contig (getMappingLocations, fEM.h:322-346)
__global__ void __launch_bounds__(256) em_entries_kernel(const mm_map_record* __restrict__ rec, const uint64_t* __restrict__ rec_off, int64_t ne,
                                                         const int32_t* __restrict__ read_len, const int32_t* __restrict__ contig_taxon,
                                                         const int32_t* __restrict__ contig_len, const int64_t* __restrict__ tl_off,
                                                         const int32_t* __restrict__ tl_len /* ascending per taxon */,
                                                         const int64_t* __restrict__ tl_suffix /* sum of tl_len[i..end of taxon) */,
                                                         int32_t* __restrict__ taxon, uint32_t* __restrict__ key, uint32_t* __restrict__ val,
                                                         double* __restrict__ mapq, double* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ne) return;
  const mm_map_record x = rec[i];
  const int t = contig_taxon[x.ref_contig];
  const int64_t L = read_len[x.read];
  int64_t lo = tl_off[t], hi = tl_off[t + 1];
  const int64_t end = hi;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (tl_len[mid] < L) lo = mid + 1; else hi = mid; }
  int64_t n = (lo < end ? tl_suffix[lo] : 0) - (end - lo) * (L - 1);
  int prev = -1;                                                   // shorter contigs of the taxon the read maps to, each once
  for (uint64_t j = rec_off[x.read]; j < rec_off[x.read + 1]; ++j) {
    const int c = rec[j].ref_contig;
    if (c != prev && contig_taxon[c] == t && contig_len[c] < L) ++n;
    prev = c;
  }
  taxon[i] = t; key[i] = (uint32_t)t; val[i] = (uint32_t)i;
  mapq[i] = parse6(x.mapq);
  inv[i] = 1.0 / (double)n;
}
__global__ void em_perm_kernel(const uint32_t* __restrict__ val_sorted, int64_t ne, int64_t* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ne) perm[i] = (int64_t)val_sorted[i];
}
__global__ void em_tstart_kernel(const uint32_t* __restrict__ key_sorted, int64_t ne, int32_t n_taxa, int64_t* __restrict__ tstart) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_taxa) return;
  int64_t lo = 0, hi = ne;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)key_sorted[mid] < t) lo = mid + 1; else hi = mid; }
  tstart[t] = lo;
}

void em_create_from_mapping(mm_ctx* ctx, const mm_mapping* M, const int32_t* contig_taxon, const int32_t* contig_len, int32_t n_contigs,
                            int32_t n_taxa, mm_em* E) {
  hipStream_t st = ctx->stream;
  const int64_t n_reads = M->n_reads, ne = M->n_rec;
  MM_REQUIRE(ne < (1LL << 32), MM_ERR_LIMIT, "more than 2^32 mapping records in one EM problem");
  E->ctx = ctx; E->n_reads = n_reads; E->n_taxa = n_taxa; E->n_entries = ne;
  // contig lengths per taxon, ascending, with suffix sums
  std::vector<int64_t> tl_off((size_t)n_taxa + 1, 0);
  for (int32_t c = 0; c < n_contigs; ++c) { MM_REQUIRE(contig_taxon[c] >= 0 && contig_taxon[c] < n_taxa, MM_ERR_ARG, "taxon index out of range"); tl_off[(size_t)contig_taxon[c] + 1]++; }
  for (int32_t t = 0; t < n_taxa; ++t) tl_off[(size_t)t + 1] += tl_off[(size_t)t];
  std::vector<int32_t> tl_len((size_t)std::max(n_contigs, 1));
  { std::vector<int64_t> cur(tl_off.begin(), tl_off.end() - 1); for (int32_t c = 0; c < n_contigs; ++c) tl_len[(size_t)cur[(size_t)contig_taxon[c]]++] = contig_len[c]; }
  std::vector<int64_t> tl_suf((size_t)std::max(n_contigs, 1), 0);
  for (int32_t t = 0; t < n_taxa; ++t) {
    std::sort(tl_len.begin() + tl_off[(size_t)t], tl_len.begin() + tl_off[(size_t)t + 1]);
    int64_t run = 0;
    for (int64_t i = tl_off[(size_t)t + 1] - 1; i >= tl_off[(size_t)t]; --i) { run += tl_len[(size_t)i]; tl_suf[(size_t)i] = run; }
  }
  DBuf<int32_t> d_ct((size_t)std::max(n_contigs, 1)), d_cl((size_t)std::max(n_contigs, 1)), d_tl(tl_len.size());
  DBuf<int64_t> d_toff(tl_off.size()), d_tsuf(tl_suf.size());
  d_ct.upload(contig_taxon, (size_t)n_contigs, st); d_cl.upload(contig_len, (size_t)n_contigs, st);
  d_tl.upload(tl_len.data(), tl_len.size(), st); d_toff.upload(tl_off.data(), tl_off.size(), st); d_tsuf.upload(tl_suf.data(), tl_suf.size(), st);
  const size_t cap = (size_t)std::max<int64_t>(ne, 1);
  E->read_off.alloc((size_t)n_reads + 1);
  MM_HIP(hipMemcpyAsync(E->read_off.p, M->rec_off.p, sizeof(int64_t) * ((size_t)n_reads + 1), hipMemcpyDeviceToDevice, st));
  E->taxon.alloc(cap); E->mapq.alloc(cap); E->inv_nloc.alloc(cap); E->perm.alloc(cap); E->post.alloc(cap);
  E->tstart.alloc((size_t)n_taxa + 1);
  DBuf<uint32_t> key(cap), val(cap), key2(cap), val2(cap);
  if (ne > 0) {
    em_entries_kernel<<<dim3((unsigned)ceil_div(ne, 256)), dim3(256), 0, st>>>(M->rec.p, M->rec_off.p, ne, M->d_read_len.p, d_ct.p, d_cl.p, d_toff.p, d_tl.p,
                                                                          d_tsuf.p, E->taxon.p, key.p, val.p, E->mapq.p, E->inv_nloc.p);
    MM_KERNEL_CHECK();
    // CSR by taxon, entries in read order inside each taxon (stable sort)
    int bits = 1; while ((1LL << bits) < (int64_t)n_taxa) ++bits;
    size_t tmp_bytes = 0;
    MM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key.p, key2.p, val.p, val2.p, (size_t)ne, 0, bits, st));
    DBuf<uint8_t> tmp(tmp_bytes);
    MM_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, key.p, key2.p, val.p, val2.p, (size_t)ne, 0, bits, st));
    em_perm_kernel<<<dim3((unsigned)ceil_div(ne, 256)), dim3(256), 0, st>>>(val2.p, ne, E->perm.p);
    MM_KERNEL_CHECK();
  }
  em_tstart_kernel<<<dim3((unsigned)ceil_div((int64_t)n_taxa + 1, 256)), dim3(256), 0, st>>>(key2.p, ne, n_taxa, E->tstart.p);
  MM_KERNEL_CHECK();
  E->ll_read.alloc((size_t)std::max<int64_t>(n_reads, 1));
  E->f.alloc((size_t)n_taxa);
  E->partial.alloc((size_t)n_taxa + 2);
  E->block_sum.alloc((size_t)ceil_div(std::max<int64_t>(n_reads, 1), 256));
  MM_HIP(mm::stream_sync(st));
}

// device part of one iteration: partial[0..T) = sum of posteriors per taxon, partial[T] = sum of log-likelihoods
static void em_step_device(mm_em* E, const double* f_host) {
  hipStream_t st = E->ctx->stream;
  E->f.upload(f_host, (size_t)E->n_taxa, st);
  if (E->n_reads > 0) {
    em_estep_kernel<<<dim3((unsigned)ceil_div(E->n_reads, 128)), dim3(128), 0, st>>>(E->read_off.p, E->taxon.p, E->mapq.p, E->inv_nloc.p, E->f.p,
                                                                                 E->n_reads, E->post.p, E->ll_read.p);
    MM_KERNEL_CHECK();
  }
  em_taxon_sum_kernel<<<dim3((unsigned)E->n_taxa), dim3(64), 0, st>>>(E->post.p, E->tstart.p, E->perm.p, E->partial.p);
  MM_KERNEL_CHECK();
  const int64_t nb = ceil_div(std::max<int64_t>(E->n_reads, 1), 256);
  sum_blocks_kernel<<<dim3((unsigned)nb), dim3(256), 0, st>>>(E->ll_read.p, E->n_reads, E->block_sum.p);
  MM_KERNEL_CHECK();
  sum_final_kernel<<<dim3(1), dim3(256), 0, st>>>(E->block_sum.p, nb, E->partial.p + E->n_taxa);
  MM_KERNEL_CHECK();
}

void em_iterate(mm_em* E, const double* f, double* f_partial, double* ll_partial) {
  em_step_device(E, f);
  std::vector<double> h = E->partial.to_host(E->ctx->stream);
  memcpy(f_partial, h.data(), sizeof(double) * (size_t)E->n_taxa);
  *ll_partial = h[(size_t)E->n_taxa];
}

void em_iterate_allreduce(mm_em* E, const double* f, double* f_next, double* ll) {
  mm_ctx* ctx = E->ctx;
  em_step_device(E, f);
  if (ctx->comm) {                                                // fEM.h:583-600, across GPUs instead of OpenMP threads
    ncclResult_t rc = ncclAllReduce(E->partial.p, E->partial.p, (size_t)E->n_taxa + 1, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
    MM_REQUIRE(rc == ncclSuccess, MM_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(rc));
  }
  std::vector<double> h = E->partial.to_host(ctx->stream);
  double sum = 0;
  for (int32_t t = 0; t < E->n_taxa; ++t) sum += h[(size_t)t];    // fEM.h:606-615
  for (int32_t t = 0; t < E->n_taxa; ++t) f_next[t] = h[(size_t)t] / sum;
  *ll = h[(size_t)E->n_taxa];
}

// ---------------------------------------------------------------------------------------------------
// The whole EM loop on the device (meta::doEM's while loop, fEM.h:501-661), round 4: ONE resident kernel per run.
//
// An iteration has three phases with a dependency between each:
//   P1  E step, thread per read (fEM.h:350-361, :578): post[i] = l_i / sum l, log-likelihood partial per workgroup
//   P2  per-taxon sums of the posteriors in a FIXED SHAPE: the entries of a taxon (read order, `perm`) are cut into items of <= 512,
//       an item is summed by one wavefront (lane l takes l, l + 64, ...; butterfly), the items of a taxon are added in order.  The
//       shape depends on the number of entries only, so exactly tied taxa stay exactly tied (getBestMapping's first-maximum rule,
//       fEM.h:217-232, sees the same ties as the reference's sequential sums)
//   P3  one workgroup: item sums -> per-taxon sums, their total, f = sum / total (fEM.h:606-615), log-likelihood, stop rule (:624-639)
// Round 3 ran them as five launches + a copy per iteration (136 us per iteration, 4.3 ms of a 48 ms bench step for a 5 MB problem).
// Now the grid (<= 128 workgroups, all resident) loops over the iterations itself; the phases are separated by grid barriers (agent-scope
// release / acquire around one atomic counter; the workgroup that arrives last at the second barrier runs P3 before it releases the
// others).  With several ranks an iteration is kernel A = P1 | barrier | P2 | last arriver: P3' (local sums), the ncclAllReduce, and
// kernel B = normalise + stop rule on the all-reduced sums (every rank decides on identical values, so the collectives stay matched).
// A barrier that is not released within 2 s (a grid that cannot become resident: many contexts of one device inside their EM loops at
// once) raises the abort flag; the host then finishes the run with the same phases as separate launches (MM_EM_SPLIT=1 forces that
// path: bit-identical results, same shapes).
// ctrl[0] = iterations done, ctrl[1] = stopped (1: the stop rule fired, 2: the caller's iteration limit), ctrl[2] = bits of the
// previous log-likelihood, ctrl[3] = first iteration of the current log-likelihood trace, ctrl[4] = a barrier timed out.
// ---------------------------------------------------------------------------------------------------
struct EmLoop {
  const longlong2* span;                                          // per read (file order; MM_EM_ORDER=count: by mapping count): [first, behind-last) mapping
  const int64_t* read_off; const int32_t* eread;                  // [n_reads + 1]; read of every mapping
  const int32_t* taxon; const double* mapq; const double* inv_nloc;
  int64_t n_reads;
  double* post_sorted; const int64_t* pos;                        // posteriors in taxon-sorted order (P1 writes entry i to pos[i], the inverse of perm[])
  const int64_t* item_lo; const int64_t* item_hi; int n_items;
  const int32_t* present; const int32_t* pt_item; int n_present;   // items of present taxon p: [pt_item[p], pt_item[p + 1])
  double* item_sum; double* wg_ll;
  double* f; double* local_partial; int32_t n_taxa;
  long long* ctrl; double* ll_trace; int ll_cap; long long it_limit;
  unsigned* bar;                                                  // [0] arrivals, [1] released generation
  long long barrier_ticks;                                        // a barrier not released within this many ticks of the 100 MHz wall clock gives up (ctrl[4])
  int dbg;                                                        // MM_EM_DBG (timing aid, results then meaningless): 1 = P1 without its scattered stores, 2 = without the f gather
};
constexpr int EM_ITEM = 512;
constexpr int EM_BAR_GROUP = 16;                                  // workgroups per first-level barrier counter
constexpr long long EM_BARRIER_TICKS = 200000000LL;               // 2 s of the 100 MHz wall clock (MM_EM_BARRIER_TICKS: test hook)

// P1.  A thread walks its reads; the mappings of a read are taken EIGHT at a time with every load of the eight issued before the first is used
// (clamped indices instead of branches), so a read of up to eight mappings — 98 % of them at ~4 per read — costs two dependent memory round
// trips; longer reads loop over such chunks twice (sum, then posteriors).  A wave takes as many rounds as its LONGEST read, so the reads
// are visited in the order of their mapping counts (`span`: the reads' [first, behind-last) mapping pairs sorted by count, longest first,
// made once per problem): lanes of a wave then hold reads of equal length.  (Round 4 measured the first form of this phase — four at a time,
// file order, two passes from five mappings on — at 38 of an iteration's 53 us: per wave the maximum over 64 lanes of ~6 rounds of two dependent
// loads.)  The likelihoods are added in mapping order, as the reference adds them (fEM.h:353-358).  The posterior goes straight to its place
// in the taxon-sorted array P2 reads (pos[i]): P2 then streams instead of gathering through perm[].
__device__ inline double em_p1_reads(const EmLoop& a, int64_t q0, int64_t q1) {   // thread-per-read form over reads [q0, q1) of `span`; returns the thread's log-likelihood share
  const int tid = threadIdx.x;
  double ll = 0;
  for (int64_t q = q0 + tid; q < q1; q += 256) {
    int64_t lo, hi;
    if (a.span) { const longlong2 sp = a.span[q]; lo = sp.x; hi = sp.y; } else { lo = a.read_off[q]; hi = a.read_off[q + 1]; }
    if (hi <= lo) continue;
    const int64_t last = hi - 1;
    double sum = 0, l8[8]; int64_t p8[8];
    for (int64_t c = lo; c < hi; c += 8) {
      int t8[8]; double w8[8], q8[8], f8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int64_t i = c + u < last ? c + u : last; t8[u] = a.taxon[i]; w8[u] = a.inv_nloc[i]; q8[u] = a.mapq[i]; p8[u] = a.pos[i]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) f8[u] = a.dbg == 2 ? 1e-4 : a.f[t8[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u) { l8[u] = f8[u] * w8[u] * q8[u]; if (c + u < hi) sum += l8[u]; }   // fEM.h:353
    }
    if (hi - lo <= 8) {                                            // the usual read: everything is still in registers
#pragma unroll
      for (int u = 0; u < 8; ++u) if (lo + u < hi && a.dbg != 1) a.post_sorted[p8[u]] = l8[u] / sum;              // :361
    } else {
      for (int64_t c = lo; c < hi; c += 8) {
        int t8[8]; double w8[8], q8[8], f8[8]; int64_t q_pos[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t i = c + u < last ? c + u : last; t8[u] = a.taxon[i]; w8[u] = a.inv_nloc[i]; q8[u] = a.mapq[i]; q_pos[u] = a.pos[i]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) f8[u] = a.f[t8[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (c + u < hi && a.dbg != 1) a.post_sorted[q_pos[u]] = (f8[u] * w8[u] * q8[u]) / sum;
      }
    }
    ll += log(sum);                                                // :578
  }
  return ll;
}
// P1 of a workgroup = a contiguous block of reads, hence a contiguous block of mappings.  Round 4's phase clocks put the thread-per-read form at
// 40 of an iteration's 53 us whatever its chunk width, grid or read order: it is bound by the ~0.9 M sector requests its strided accesses make
// (four arrays, a lane's mappings 34 bytes from its neighbour's; without the scattered posterior stores -12 us, without the f gather -7 us).
// Here the mappings are walked thread-per-MAPPING (coalesced), the likelihoods parked in LDS, the reads' sums taken from LDS in mapping order (the
// reference's order, fEM.h:353-358), and the posteriors written in a second coalesced walk.  A block with more mappings or reads than the LDS
// arrays hold takes the thread-per-read form.
// 5.5 KB of LDS + 2 KB for the reduction: a P1 workgroup fits beside the seed filter's resident workgroups (152 of a CU's 160 KB), so that the
// EM of one worker context does not wait for the other context's K3 to leave the CUs (with 54 KB it did: 240-320 us per iteration in the bench)
constexpr int EM_LBUF = 512, EM_RBUF = 192;                       // mappings / reads of a block that fit the LDS arrays
__device__ inline void em_p1(const EmLoop& a, int wg, int n_wg, double* sh, double* lbuf, double* rsum) {
  const int tid = threadIdx.x;
  const int64_t rb = (a.n_reads + n_wg - 1) / n_wg, R0 = min((int64_t)wg * rb, a.n_reads), R1 = min(R0 + rb, a.n_reads);
  double ll = 0;
  const int64_t E0 = R1 > R0 ? a.read_off[R0] : 0, E1 = R1 > R0 ? a.read_off[R1] : 0;
  const int nE = (int)min(E1 - E0, (int64_t)EM_LBUF + 1), nR = (int)(R1 - R0);
  if (a.dbg == 3 || E1 - E0 > EM_LBUF || nR > EM_RBUF) ll = em_p1_reads(a, R0, R1);
  else if (nR > 0) {
    for (int e0 = 0; e0 < nE; e0 += 4 * 256) {                   // four coalesced mappings per thread in flight
      int t4[4]; double w4[4], q4[4], f4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int64_t i = E0 + min(e0 + u * 256 + tid, nE - 1); t4[u] = a.taxon[i]; w4[u] = a.inv_nloc[i]; q4[u] = a.mapq[i]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) f4[u] = a.f[t4[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = e0 + u * 256 + tid; if (e < nE) lbuf[e] = f4[u] * w4[u] * q4[u]; }   // fEM.h:353
    }
    __syncthreads();
    for (int r = tid; r < nR; r += 256) {
      const int lo = (int)(a.read_off[R0 + r] - E0), hi = (int)(a.read_off[R0 + r + 1] - E0);
      double sum = 0;
      for (int e = lo; e < hi; ++e) sum += lbuf[e];
      rsum[r] = sum;
      if (hi > lo) ll += log(sum);                                 // :578
    }
    __syncthreads();
    for (int e0 = 0; e0 < nE; e0 += 4 * 256) {
      int64_t p4[4]; int r4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int64_t i = E0 + min(e0 + u * 256 + tid, nE - 1); p4[u] = a.pos[i]; r4[u] = a.eread[i]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = e0 + u * 256 + tid; if (e < nE) a.post_sorted[p4[u]] = lbuf[e] / rsum[r4[u] - (int)R0]; }   // :361
    }
  }
  __syncthreads();
  sh[tid] = ll;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if (tid < d) sh[tid] += sh[tid + d]; __syncthreads(); }
  if (tid == 0) a.wg_ll[wg] = sh[0];
}
// P2.  An item = up to 512 consecutive posteriors of one taxon (read order): lane l adds l, l + 64, ... in order, then a butterfly.
__device__ inline void em_p2(const EmLoop& a, int wg, int n_wg) {
  const int lane = threadIdx.x & 63;
  const int n_waves = n_wg * 4;
  for (int it = wg * 4 + (threadIdx.x >> 6); it < a.n_items; it += n_waves) {
    const int64_t lo = a.item_lo[it], hi = a.item_hi[it], last = hi - 1;
    double v[EM_ITEM / 64];
#pragma unroll
    for (int u = 0; u < EM_ITEM / 64; ++u) { const int64_t j = lo + lane + 64 * u; v[u] = a.post_sorted[j < last ? j : last]; }
    double acc = 0;
#pragma unroll
    for (int u = 0; u < EM_ITEM / 64; ++u) if (lo + lane + 64 * u < hi) acc += v[u];
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) a.item_sum[it] = acc;
  }
}
// fixed-shape sum of v(0..n) by one workgroup of 256: thread t adds t, t + 256, ... in order, then a tree
template <typename F>
__device__ inline double wg_sum256(int n, double* sh, F v) {
  const int tid = threadIdx.x;
  double acc = 0;
  for (int i = tid; i < n; i += 256) acc += v(i);
  __syncthreads();
  sh[tid] = acc;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if (tid < d) sh[tid] += sh[tid + d]; __syncthreads(); }
  const double r = sh[0];
  __syncthreads();
  return r;
}
__device__ inline void em_stop_rule(long long* ctrl, double ll, double* ll_trace, int ll_cap, long long it_limit) {   // one thread
  const long long it = ctrl[0];
  const double ll_prev = __longlong_as_double(ctrl[2]);
  const long long ti = it - ctrl[3];                               // ctrl[3]: iteration the trace buffer starts at (mm_em_continue)
  if (ti >= 0 && ti < ll_cap) ll_trace[ti] = ll;
  if (it > 0 && (ll - ll_prev) <= 1 && (1 - ll / ll_prev) < 0.0001) ctrl[1] = 1;   // fEM.h:624-639
  if (!ctrl[1] && it + 1 >= it_limit) ctrl[1] = 2;                 // the caller's limit, unless the rule has just fired (a rule stop is final, a limit stop is lifted by mm_em_continue)
  ctrl[2] = __double_as_longlong(ll);
  ctrl[0] = it + 1;
}
// P3 (one workgroup).  LOCAL: the per-taxon sums and the log-likelihood of this rank go to local_partial[0..T] (all-reduced next);
// otherwise: normalise over the present taxa, write f, evaluate the stop rule.
template <bool LOCAL>
__device__ inline void em_p3(const EmLoop& a, int n_wg, double* sh) {
  const int tid = threadIdx.x;
  // the item sums of a taxon are added in order by one thread; their loads are independent: eight in flight per round trip
  for (int p = tid; p < a.n_present; p += 256) {
    double s = 0;
    const int i0 = a.pt_item[p], i1 = a.pt_item[p + 1];
    for (int it = i0; it < i1; it += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = a.item_sum[min(it + u, i1 - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) if (it + u < i1) s += v[u];
    }
    a.local_partial[a.present[p]] = s;
  }
  __syncthreads();
  const double ll = wg_sum256(n_wg, sh, [&](int g) { return a.wg_ll[g]; });
  if (LOCAL) { if (tid == 0) a.local_partial[a.n_taxa] = ll; return; }
  const double total = wg_sum256(a.n_present, sh, [&](int p) { return a.local_partial[a.present[p]]; });
  if (a.ctrl[0] == 0) {                                            // taxa without a mapping: 0 / total from the first iteration on (fEM.h:606-615)
    for (int t = tid; t < a.n_taxa; t += 256) a.f[t] = 0.0;
    __syncthreads();
  }
  for (int p = tid; p < a.n_present; p += 256) { const int t = a.present[p]; a.f[t] = a.local_partial[t] / total; }
  __syncthreads();
  if (tid == 0) em_stop_rule(a.ctrl, ll, a.ll_trace, a.ll_cap, a.it_limit);
}

// grid barrier pieces (thread 0 of every workgroup talks; agent-scope fences publish / fetch the other workgroups' plain stores: the
// XCDs' L2s are not coherent with each other, docs/history.md K5 scratch slots)
// bar[0]: groups that have arrived, bar[1]: released generation, bar[16 * (1 + g)]: arrivals of group g (EM_BAR_GROUP workgroups, a
// 64-byte line each).  Two levels because 128 agent-scope atomics on ONE word are served one after the other: ~13 us per barrier,
// more than the phases between them.
__device__ inline bool grid_arrive_is_last(unsigned* bar, unsigned epoch, unsigned n_wg, int* s_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned g = blockIdx.x / EM_BAR_GROUP, n_groups = (n_wg + EM_BAR_GROUP - 1) / EM_BAR_GROUP;
    const unsigned g_size = min((unsigned)EM_BAR_GROUP, n_wg - g * EM_BAR_GROUP);
    int last = 0;
    if (__hip_atomic_fetch_add(&bar[16 * (1 + g)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == epoch * g_size) {
      __threadfence();
      last = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == epoch * n_groups;
    }
    if (last) __threadfence();
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}
__device__ inline void grid_release(unsigned* bar, unsigned epoch) {
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence(); __hip_atomic_store(&bar[1], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
__device__ inline bool grid_wait(unsigned* bar, unsigned epoch, long long* ctrl, int* s_flag, long long ticks) {   // false: timed out / aborted
  if (threadIdx.x == 0) {
    const long long t0 = (long long)wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if ((long long)wall_clock64() - t0 > (ticks < 0 ? -ticks : ticks) || __hip_atomic_load(&ctrl[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(&ctrl[4], 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break;
      }
    }
    __threadfence();
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// ONE_ITERATION = false: the whole run of one rank.  true: kernel A of a multi-rank iteration (P1 | P2 | local sums), leaves after it.
template <bool ONE_ITERATION>
__global__ void __launch_bounds__(256) em_loop_kernel(EmLoop a) {
  __shared__ double sh[256], lbuf[EM_LBUF], rsum[EM_RBUF];
  __shared__ int s_flag;
  const int wg = blockIdx.x, n_wg = gridDim.x;
  if (__hip_atomic_load(&a.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // (iterations enqueued past the stop are no-ops)
  unsigned epoch = 0;
  const bool prof = a.barrier_ticks < 0 && wg == 0 && threadIdx.x == 0;   // MM_EM_PROF: workgroup 0's wall-clock ticks per phase into ctrl[5..7]
  long long t_prev = prof ? (long long)wall_clock64() : 0;
  auto tick = [&](int slot) { if (prof) { const long long t = (long long)wall_clock64(); a.ctrl[slot] += t - t_prev; t_prev = t; } };
  for (;;) {
    em_p1(a, wg, n_wg, sh, lbuf, rsum);
    tick(5);
    ++epoch;
    if (grid_arrive_is_last(a.bar, epoch, n_wg, &s_flag)) grid_release(a.bar, epoch);
    else if (!grid_wait(a.bar, epoch, a.ctrl, &s_flag, a.barrier_ticks)) {
      if (ONE_ITERATION && threadIdx.x == 0) a.local_partial[a.n_taxa + 1] = 1.0;   // all-reduced: every rank learns that this iteration did not happen
      return;
    }
    tick(6);                                                       // (barrier 1)
    em_p2(a, wg, n_wg);
    tick(7);
    ++epoch;
    if (grid_arrive_is_last(a.bar, epoch, n_wg, &s_flag)) {
      em_p3<ONE_ITERATION>(a, n_wg, sh);
      if (ONE_ITERATION) return;
      grid_release(a.bar, epoch);
    } else {
      if (ONE_ITERATION) return;
      if (!grid_wait(a.bar, epoch, a.ctrl, &s_flag, a.barrier_ticks)) return;
    }
    if (__hip_atomic_load(&a.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  }
}
// the same phases as separate launches (no barrier inside): the path after a barrier time-out, and MM_EM_SPLIT=1
__global__ void __launch_bounds__(256) em_p1_kernel(EmLoop a) { __shared__ double sh[256], lbuf[EM_LBUF], rsum[EM_RBUF]; if (a.ctrl[1]) return; em_p1(a, blockIdx.x, gridDim.x, sh, lbuf, rsum); }
__global__ void __launch_bounds__(256) em_p2_kernel(EmLoop a) { if (a.ctrl[1]) return; em_p2(a, blockIdx.x, gridDim.x); }
template <bool LOCAL>
__global__ void __launch_bounds__(256) em_p3_kernel(EmLoop a, int n_wg) { __shared__ double sh[256]; if (a.ctrl[1]) return; em_p3<LOCAL>(a, n_wg, sh); }
// P2 and P3 in one launch (round 6, MM_EM_SPLIT=2): the workgroup that finishes its items last runs P3 — the hand-over of the resident kernel's second barrier
// (arrive, agent-scope release / acquire) without anybody waiting, so it needs no co-residency.  An iteration is then two launches (with several ranks: kernel A',
// this one, the all-reduce and kernel B) instead of three (five).  bar[2]: workgroups done with P2; the last one puts it back to zero.
// NOT the default: it measured slower than the launch it saves (see em_run).
template <bool LOCAL>
__global__ void __launch_bounds__(256) em_p23_kernel(EmLoop a) {
  __shared__ double sh[256];
  __shared__ int s_last;
  if (a.ctrl[1]) return;
  em_p2(a, blockIdx.x, gridDim.x);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int last = __hip_atomic_fetch_add(&a.bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == gridDim.x;
    if (last) { __threadfence(); __hip_atomic_store(&a.bar[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    s_last = last;
  }
  __syncthreads();
  if (s_last) em_p3<LOCAL>(a, (int)gridDim.x, sh);
}
// kernel B of a multi-rank iteration: normalise the all-reduced sums (fEM.h:606-615; fixed-shape sum over the taxa, the same on every
// rank), log-likelihood trace, stop rule (:624-639)
__global__ void __launch_bounds__(256) em_finalize_kernel(const double* __restrict__ partial, int32_t n_taxa, double* __restrict__ f, long long* __restrict__ ctrl,
                                                          double* __restrict__ ll_trace, int ll_cap, long long it_limit) {
  if (ctrl[1]) return;
  __shared__ double sh[256];
  if (partial[n_taxa + 1] > 0) {                                   // some rank's kernel A gave up at its barrier: nothing is applied, every rank repeats the iteration phase by phase
    if (threadIdx.x == 0) { ctrl[4] = 1; ctrl[1] = 3; }
    return;
  }
  const double sum = wg_sum256(n_taxa, sh, [&](int t) { return partial[t]; });
  for (int t = threadIdx.x; t < n_taxa; t += 256) f[t] = partial[t] / sum;
  if (threadIdx.x == 0) em_stop_rule(ctrl, partial[n_taxa], ll_trace, ll_cap, it_limit);
}

__global__ void em_eread_kernel(const int64_t* __restrict__ read_off, int64_t n_reads, int32_t* __restrict__ eread) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_reads) for (int64_t i = read_off[r]; i < read_off[r + 1]; ++i) eread[i] = (int32_t)r;
}
__global__ void em_pos_kernel(const int64_t* __restrict__ perm, int64_t ne, int64_t* __restrict__ pos) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ne) pos[perm[j]] = j;
}
static int em_grid(int64_t n_reads, int64_t n_entries) {         // (fixed per problem: the log-likelihood partials are summed in the grid's shape)
  if (!getenv("MM_EM_GRID") && !getenv("MM_EM_RESIDENT")) {        // launches: as many blocks as keep a block within P1's LDS buffers (a little under EM_RBUF reads and EM_LBUF mappings on average), 256 at least
    const int64_t want = std::max<int64_t>({(int64_t)256, ceil_div(std::max<int64_t>(n_reads, 1), EM_RBUF * 5 / 6), ceil_div(std::max<int64_t>(n_entries, 1), EM_LBUF * 7 / 8)});
    return (int)std::max<int64_t>(1, std::min<int64_t>({want, (int64_t)1 << 20, std::max<int64_t>(n_reads, 1)}));
  }
  const char* e = getenv("MM_EM_GRID");                           // default: 256 workgroups as launches (33 us per iteration against 38 at 128), 128 for the resident kernel (all must be resident together)
  const int cap = std::min(std::max(e ? atoi(e) : (getenv("MM_EM_RESIDENT") ? 128 : 256), 1), 1024);
  return (int)std::max<int64_t>(1, std::min<int64_t>(cap, ceil_div(std::max<int64_t>(n_reads, 1), 256)));
}

// f0 == nullptr continues the loop where the previous call on E left it (same f, iteration count and previous log-likelihood):
// mm_em_continue.  Returns the iterations done by this call; *stopped = the stop rule has fired.
int em_run(mm_em* E, const double* f0, int max_iter, double* f_out, double* ll_trace, int ll_cap, bool* stopped) {
  mm_ctx* ctx = E->ctx;
  hipStream_t st = ctx->stream;
  const int32_t T = E->n_taxa;
  const int cap = 1024;
  if (E->n_present < 0) {                                        // first run: taxa with mappings on this rank, the items of the per-taxon sums
    MM_REQUIRE(f0 != nullptr, MM_ERR_STATE, "mm_em_continue before mm_em_run");
    std::vector<int64_t> ts = E->tstart.to_host(st, (size_t)T + 1);
    std::vector<int32_t> pr, pti(1, 0);
    std::vector<int64_t> ilo, ihi;
    for (int32_t t = 0; t < T; ++t) {
      if (ts[(size_t)t + 1] == ts[(size_t)t]) continue;
      pr.push_back(t);
      for (int64_t j = ts[(size_t)t]; j < ts[(size_t)t + 1]; j += EM_ITEM) { ilo.push_back(j); ihi.push_back(std::min(j + EM_ITEM, ts[(size_t)t + 1])); }
      MM_REQUIRE(ilo.size() < (size_t)INT32_MAX, MM_ERR_LIMIT, "EM problem beyond 2^31 sum items");
      pti.push_back((int32_t)ilo.size());
    }
    E->n_present = (int32_t)pr.size(); E->n_items = (int32_t)ilo.size();
    E->n_wg = em_grid(E->n_reads, E->n_entries);
    E->present.alloc(std::max<size_t>(pr.size(), 1)); E->present.upload(pr.data(), pr.size(), st);
    E->pt_item.alloc(pti.size()); E->pt_item.upload(pti.data(), pti.size(), st);
    E->item_lo.alloc(std::max<size_t>(ilo.size(), 1)); E->item_lo.upload(ilo.data(), ilo.size(), st);
    E->item_hi.alloc(std::max<size_t>(ihi.size(), 1)); E->item_hi.upload(ihi.data(), ihi.size(), st);
    E->item_sum.alloc(std::max<size_t>(ilo.size(), 1));
    if (getenv("MM_EM_ORDER") && !strcmp(getenv("MM_EM_ORDER"), "count")) {   // measurement aid: the thread-per-read form (MM_EM_DBG=3) with the reads by mapping count, longest first
      std::vector<int64_t> ro = E->read_off.to_host(st, (size_t)E->n_reads + 1);
      std::vector<int64_t> sp(2 * (size_t)std::max<int64_t>(E->n_reads, 1), 0);
      int64_t cmax = 0; for (int64_t r = 0; r < E->n_reads; ++r) cmax = std::max(cmax, ro[(size_t)r + 1] - ro[(size_t)r]);
      const int64_t NB = std::min<int64_t>(cmax, 255) + 1;       // (counts beyond 255 share the first bucket: order among them does not matter for what this is for)
      std::vector<int64_t> start((size_t)NB + 1, 0);
      const bool by_count = getenv("MM_EM_ORDER") && !strcmp(getenv("MM_EM_ORDER"), "count");   // default: file order (one bucket)
      auto bucket = [&](int64_t c) { return by_count ? NB - 1 - std::min<int64_t>(c, NB - 1) : (int64_t)0; };
      for (int64_t r = 0; r < E->n_reads; ++r) start[(size_t)bucket(ro[(size_t)r + 1] - ro[(size_t)r]) + 1]++;
      for (int64_t b2 = 0; b2 < NB; ++b2) start[(size_t)b2 + 1] += start[(size_t)b2];
      for (int64_t r = 0; r < E->n_reads; ++r) { const int64_t k2 = start[(size_t)bucket(ro[(size_t)r + 1] - ro[(size_t)r])]++; sp[2 * (size_t)k2] = ro[(size_t)r]; sp[2 * (size_t)k2 + 1] = ro[(size_t)r + 1]; }
      E->span.alloc(sp.size()); E->span.upload(sp.data(), sp.size(), st);
      MM_HIP(mm::stream_sync(st));                          // (sp is the upload's source)
    }
    E->pos.alloc((size_t)std::max<int64_t>(E->n_entries, 1)); E->post_sorted.alloc((size_t)std::max<int64_t>(E->n_entries, 1));
    MM_REQUIRE(E->n_reads < (1LL << 31), MM_ERR_LIMIT, "EM problem beyond 2^31 reads");
    E->eread.alloc((size_t)std::max<int64_t>(E->n_entries, 1));
    if (E->n_reads > 0) { em_eread_kernel<<<dim3((unsigned)ceil_div(E->n_reads, 256)), dim3(256), 0, st>>>(E->read_off.p, E->n_reads, E->eread.p); MM_KERNEL_CHECK(); }
    if (E->n_entries > 0) { em_pos_kernel<<<dim3((unsigned)ceil_div(E->n_entries, 256)), dim3(256), 0, st>>>(E->perm.p, E->n_entries, E->pos.p); MM_KERNEL_CHECK(); }
    E->wg_ll.alloc((size_t)E->n_wg);
    E->local_partial.alloc((size_t)T + 2);
    E->ll_trace.alloc((size_t)cap);
    E->f_run.alloc((size_t)T);
    E->ctrl.alloc(8);
    E->bar.alloc(16 * (size_t)(1 + ceil_div(E->n_wg, EM_BAR_GROUP)));
    MM_HIP(mm::stream_sync(st));
  }
  long long h_ctrl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (f0) {
    E->f_run.upload(f0, (size_t)T, st);
    E->local_partial.zero(st);
    E->ctrl.zero(st);
  } else {                                                       // go on: a limit stop is lifted, a rule stop stays; the trace starts here
    MM_HIP(hipMemcpyAsync(h_ctrl, E->ctrl.p, sizeof h_ctrl, hipMemcpyDeviceToHost, st));
    MM_HIP(mm::stream_sync(st));
    if (h_ctrl[1] == 2) h_ctrl[1] = 0;
    h_ctrl[3] = h_ctrl[0]; h_ctrl[4] = 0;
    MM_HIP(hipMemcpyAsync(E->ctrl.p, h_ctrl, sizeof h_ctrl, hipMemcpyHostToDevice, st));
    MM_HIP(mm::stream_sync(st));
  }
  const long long it0 = h_ctrl[0], it_limit = it0 + max_iter;
  EmLoop a{(const longlong2*)(E->span.n ? E->span.p : nullptr), E->read_off.p, E->eread.p, E->taxon.p, E->mapq.p, E->inv_nloc.p, E->n_reads, E->post_sorted.p, E->pos.p, E->item_lo.p, E->item_hi.p, E->n_items,
           E->present.p, E->pt_item.p, E->n_present, E->item_sum.p, E->wg_ll.p, E->f_run.p, E->local_partial.p, T, E->ctrl.p, E->ll_trace.p, cap, it_limit, E->bar.p,
           (getenv("MM_EM_PROF") ? -1 : 1) * (getenv("MM_EM_BARRIER_TICKS") ? atoll(getenv("MM_EM_BARRIER_TICKS")) : EM_BARRIER_TICKS),
           getenv("MM_EM_DBG") ? atoi(getenv("MM_EM_DBG")) : 0};
  const dim3 grid((unsigned)E->n_wg), blk(256);
  // One launch per phase is the default: measured against the resident kernel (MM_EM_RESIDENT=1, same phases behind grid barriers, bit-identical)
  // it is the faster form on an idle GPU (35 against 44 us per iteration, tools/em_latency.py) and no slower beside another context's kernels
  // (bench: 1.9 against 2.2 ms of EM per step) — the barriers' L2 write-back / invalidate and the wait for the slowest workgroup cost more
  // than three launches on one stream do.
  const bool force_split = getenv("MM_EM_RESIDENT") == nullptr || getenv("MM_EM_SPLIT") != nullptr;
  bool split = force_split || ctx->em_split;
  // MM_EM_SPLIT=2: P2 and P3 in one launch (em_p23_kernel).  Measured in round 6 (tools/em_latency.py, idle GPU): 48.4 us per iteration against 34.7 with a launch
  // per phase — 256 agent-scope releases (an L2 write-back each) and the arrival counter cost more than the launch they save.  Kept as the record of that.
  const bool three_launches = !(getenv("MM_EM_SPLIT") && atoi(getenv("MM_EM_SPLIT")) == 2);
  if (force_split) E->bar.zero(st);                              // (bar[2]: em_p23_kernel's arrival count)
  // a communicator of ONE rank has nothing to exchange: the run is the resident kernel, as without a communicator (MM_EM_FORCE_COLLECTIVE=1
  // keeps kernel A | ncclAllReduce | kernel B also then: how the tests drive the collective path on a one-GPU box)
  const bool collective = ctx->comm && (ctx->comm_size > 1 || getenv("MM_EM_FORCE_COLLECTIVE") != nullptr);
  auto fetch_ctrl = [&] {
    MM_HIP(hipMemcpyAsync(h_ctrl, E->ctrl.p, sizeof h_ctrl, hipMemcpyDeviceToHost, st));
    MM_HIP(mm::stream_sync(st));
    if (h_ctrl[4]) {                                             // a grid barrier timed out: the state is that of the last completed iteration (P1 / P2 only write scratch)
      if (!ctx->em_split) fprintf(stderr, "libmetamaps_hip: the resident EM kernel could not get its %d workgroups onto device %d together; "
                                          "this context goes on with one launch per phase\n", E->n_wg, ctx->device);
      ctx->em_split = split = true;
      h_ctrl[4] = 0;
      if (h_ctrl[1] == 3) h_ctrl[1] = 0;                         // (several ranks: the all-reduced abort mark stopped the rest of the enqueued group on every rank)
      MM_HIP(hipMemcpyAsync(E->ctrl.p, h_ctrl, sizeof h_ctrl, hipMemcpyHostToDevice, st));
      MM_HIP(hipMemsetAsync(E->local_partial.p + T + 1, 0, sizeof(double), st));
      MM_HIP(mm::stream_sync(st));
    }
  };
  // iterations are enqueued in groups with one read of the control word behind each; iterations behind the stop are no-ops (every kernel leaves at once, the
  // collective still runs: the ranks' sequences must match).  The reference's runs take 20-40 iterations: a first group of 24, then eights — two host round
  // trips for a run of 29 instead of four (round 5; each is a copy, a wait and, under a small CPU budget, a sleep: mm::stream_sync).
  int group_no = 0;
  while (!h_ctrl[1] && h_ctrl[0] < it_limit) {
    const int GROUP = group_no++ == 0 ? 24 : 8;
    if (!collective && !split) {                                 // one rank: the whole run is one launch
      E->bar.zero(st);
      em_loop_kernel<false><<<grid, blk, 0, st>>>(a);
      MM_KERNEL_CHECK();
    } else {
      const int g_n = (int)std::min<long long>(GROUP, it_limit - h_ctrl[0]);   // (the same on every rank: h_ctrl holds all-reduced decisions)
      for (int g = 0; g < g_n; ++g) {
        if (!split) {
          E->bar.zero(st);
          em_loop_kernel<true><<<grid, blk, 0, st>>>(a);
          MM_KERNEL_CHECK();
        } else if (three_launches) {
          em_p1_kernel<<<grid, blk, 0, st>>>(a); MM_KERNEL_CHECK();
          em_p2_kernel<<<grid, blk, 0, st>>>(a); MM_KERNEL_CHECK();
          if (collective) em_p3_kernel<true><<<dim3(1), blk, 0, st>>>(a, E->n_wg); else em_p3_kernel<false><<<dim3(1), blk, 0, st>>>(a, E->n_wg);
          MM_KERNEL_CHECK();
        } else {
          em_p1_kernel<<<grid, blk, 0, st>>>(a); MM_KERNEL_CHECK();
          if (collective) em_p23_kernel<true><<<grid, blk, 0, st>>>(a); else em_p23_kernel<false><<<grid, blk, 0, st>>>(a);
          MM_KERNEL_CHECK();
        }
        if (collective) {                                        // fEM.h:583-600, across GPUs instead of OpenMP threads
          ncclResult_t rc = ncclAllReduce(E->local_partial.p, E->partial.p, (size_t)T + 2, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, st);
          MM_REQUIRE(rc == ncclSuccess, MM_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(rc));
          em_finalize_kernel<<<dim3(1), blk, 0, st>>>(E->partial.p, T, E->f_run.p, E->ctrl.p, E->ll_trace.p, cap, it_limit);
          MM_KERNEL_CHECK();
        }
      }
    }
    fetch_ctrl();
  }
  const int n_iter = (int)(h_ctrl[0] - it0);
  if (getenv("MM_EM_PROF") && n_iter > 0)                        // ticks of the 100 MHz clock, workgroup 0: P1 | barrier 1 | P2 — the rest of an iteration is barrier 2 + P3
    fprintf(stderr, "MM_EM_PROF %d iterations: P1 %.1f us, barrier 1 %.1f us, P2 %.1f us per iteration\n", n_iter, h_ctrl[5] / 100.0 / n_iter, h_ctrl[6] / 100.0 / n_iter, h_ctrl[7] / 100.0 / n_iter);
  if (stopped) *stopped = h_ctrl[1] == 1;
  if (f_out) E->f_run.download(f_out, (size_t)T, st);
  if (ll_trace && ll_cap > 0 && n_iter > 0) E->ll_trace.download(ll_trace, (size_t)std::min(std::min(n_iter, ll_cap), cap), st);
  MM_HIP(mm::stream_sync(st));
  return n_iter;
}

void em_posteriors(mm_em* E, const double* f, double* post, int64_t* best) {
  hipStream_t st = E->ctx->stream;
  E->f.upload(f, (size_t)E->n_taxa, st);                         // the E step alone (fEM.h:696-716): no sums
  if (E->n_reads > 0) {
    em_estep_kernel<<<dim3((unsigned)ceil_div(E->n_reads, 128)), dim3(128), 0, st>>>(E->read_off.p, E->taxon.p, E->mapq.p, E->inv_nloc.p, E->f.p,
                                                                                 E->n_reads, E->post.p, E->ll_read.p);
    MM_KERNEL_CHECK();
  }
  if (post) E->post.download(post, (size_t)E->n_entries, st);
  if (best && E->n_reads > 0) {
    DBuf<int64_t> b((size_t)E->n_reads);
    em_best_kernel<<<dim3((unsigned)ceil_div(E->n_reads, 128)), dim3(128), 0, st>>>(E->read_off.p, E->post.p, E->n_reads, b.p);
    MM_KERNEL_CHECK();
    b.download(best, (size_t)E->n_reads, st);
    MM_HIP(mm::stream_sync(st));
  }
  MM_HIP(mm::stream_sync(st));
}

// ---------------------------------------------------------------------------------------------------
// communicator
// ---------------------------------------------------------------------------------------------------
void comm_unique_id(char* id) {
  static_assert(sizeof(ncclUniqueId) <= MM_COMM_ID_BYTES, "ncclUniqueId larger than MM_COMM_ID_BYTES");
  ncclUniqueId u;
  ncclResult_t rc = ncclGetUniqueId(&u);
  MM_REQUIRE(rc == ncclSuccess, MM_ERR_COMM, std::string("ncclGetUniqueId: ") + ncclGetErrorString(rc));
  memset(id, 0, MM_COMM_ID_BYTES);
  memcpy(id, &u, sizeof u);
}
void comm_init(mm_ctx* ctx, const char* id, int rank, int nranks) {
  MM_REQUIRE(ctx->comm == nullptr, MM_ERR_STATE, "communicator already initialised");
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t c;
  MM_HIP(hipSetDevice(ctx->device));
  ncclResult_t rc = ncclCommInitRank(&c, nranks, u, rank);
  MM_REQUIRE(rc == ncclSuccess, MM_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(rc));
  ctx->comm = c; ctx->comm_rank = rank; ctx->comm_size = nranks;
}
void comm_allreduce_f64(mm_ctx* ctx, double* host, int64_t n) {
  if (!ctx->comm || n <= 0) return;
  DBuf<double> d((size_t)n);
  d.upload(host, (size_t)n, ctx->stream);
  ncclResult_t rc = ncclAllReduce(d.p, d.p, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
  MM_REQUIRE(rc == ncclSuccess, MM_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(rc));
  d.download(host, (size_t)n, ctx->stream);
  MM_HIP(mm::stream_sync(ctx->stream));
}
void comm_destroy(mm_ctx* ctx) {
  if (ctx->comm) { if (!ctx->comm_shared) ncclCommDestroy((ncclComm_t)ctx->comm); ctx->comm = nullptr; ctx->comm_shared = false; ctx->comm_size = 1; ctx->comm_rank = 0; }
}

}  // namespace mm
