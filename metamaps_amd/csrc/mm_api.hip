// extern "C" surface of libmetamaps_hip.so (include/metamaps_hip.h).  Converts internal exceptions into
// status codes + mm_last_error(); owns handle lifetimes.  No CPU fallback anywhere: without a gfx950
// device mm_ctx_create fails and nothing else can be called.
#include "mm_env.hpp"
#include "mm_map.hpp"
#include "mm_em.hpp"
#include "mm_stats.hpp"
#include "mm_synth.hpp"
#include <chrono>
#include <new>
#include <rccl/rccl.h>
#include <rocprim/rocprim.hpp>

namespace mm {
void seqset_upload(mm_seqset* s);
void seqset_save(mm_seqset* s, const char* path);
void seqset_load(mm_seqset* s, const char* path);
void seqset_fetch(mm_seqset* s, int64_t i, char* out, int64_t cap);
void seqset_fetch_range(mm_seqset* s, int64_t first, int64_t count, char* out, int64_t cap);
void seqset_slice(const mm_seqset* s, int64_t first, int64_t count, mm_seqset* o);
void seqset_concat(const mm_seqset* const* parts, int n_parts, mm_seqset* o);
}

namespace {
template <typename F>
int guarded(mm_ctx* ctx, F&& f) {
  // a context is bound to the calling thread for the duration of the call: its device (hipSetDevice is per thread), its
  // stream and its allocator.  Several contexts — one per GPU, or several on one GPU — may be driven from different threads.
  if (ctx) { if (ctx->device >= 0 && ctx->stream) (void)hipSetDevice(ctx->device); mm::current_stream() = ctx->stream; mm::current_alloc() = &ctx->alloc; }
  try { f(); return MM_OK; }
  catch (const mm::Error& e) { if (ctx) ctx->err = e.what(); return e.status; }
  catch (const std::bad_alloc&) { if (ctx) ctx->err = "host allocation failed"; return MM_ERR_NOMEM; }
  catch (const std::exception& e) { if (ctx) ctx->err = e.what(); return MM_ERR_DEVICE; }
}
}  // namespace

extern "C" {

int mm_abi_version(void) { return MM_ABI_VERSION; }

namespace {
// MM_CTX_TRACE=1: what the process' first HIP calls cost (right behind a process that gave back a device-filling index they wait for the driver)
struct CtxTrace {
  const bool on = getenv("MM_CTX_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "MM_CTX_TRACE %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
}  // namespace

int mm_device_count(void) {
  int n = 0;
  CtxTrace tr;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  tr.lap("hipGetDeviceCount");
  return n;
}
int mm_ctx_create(int device_id, mm_ctx** out) {
  if (!out) return MM_ERR_ARG;
  *out = nullptr;
  if (mm::env_strict()) {                                          // MM_STRICT_ENV=1: a MM_* variable the table (mm_env.hpp) does not know is an error, not a silent default
    const std::string bad = mm::env_unknown();
    if (!bad.empty()) { fprintf(stderr, "mm_ctx_create: unknown MM_* environment switch(es): %s (MM_STRICT_ENV is set; the table is metamaps_amd/csrc/mm_env.hpp / INTEGRATION.md)\n", bad.c_str()); return MM_ERR_ARG; }
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MM_ERR_DEVICE;   // no GPU: fail loudly, never fall back
  if (device_id < 0 || device_id >= n) return MM_ERR_ARG;
  mm_ctx* c = new (std::nothrow) mm_ctx;
  if (!c) return MM_ERR_NOMEM;
  c->device = device_id;
  int st = guarded(c, [&] {
    CtxTrace tr;
    MM_HIP(hipSetDevice(device_id));
    tr.lap("hipSetDevice");
    hipDeviceProp_t p;
    MM_HIP(hipGetDeviceProperties(&p, device_id));
    tr.lap("hipGetDeviceProperties");
    MM_REQUIRE(std::string(p.gcnArchName).rfind("gfx950", 0) == 0, MM_ERR_DEVICE,
               std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only");
    c->cus = p.multiProcessorCount;
    MM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    mm::stream_event_register(c->stream);
    tr.lap("hipStreamCreateWithFlags");
    if (tr.on) { void* q = nullptr; if (hipMalloc(&q, 1 << 20) == hipSuccess) { tr.lap("first hipMalloc (1 MiB)"); (void)hipFree(q); } }
    c->alloc.stream = c->stream;
    mm::alloc_register(&c->alloc, c->device);
  });
  if (st != MM_OK) { delete c; return st; }
  *out = c;
  return MM_OK;
}
void mm_ctx_destroy(mm_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)mm::stream_sync(ctx->stream);
  ctx->alloc.trim();
  mm::big_pool_trim(ctx->device);                                  // (recycled index-scale blocks go back to the driver with any context of the device)
  mm::comm_destroy(ctx);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pinned_up) (void)hipHostFree(ctx->pinned_up);
  if (ctx->l2_codes) mm::dev_free(ctx->l2_codes, ctx->l2_codes_bytes);
  if (ctx->l2_masks) mm::dev_free(ctx->l2_masks, ctx->l2_masks_bytes);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->aux_stream) { mm::stream_event_unregister(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
  if (ctx->stream) { mm::stream_event_unregister(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
  delete ctx;
}
const char* mm_last_error(const mm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int mm_ctx_device_info(mm_ctx* ctx, char* name, size_t name_cap, int* cus, uint64_t* hbm_total, uint64_t* hbm_free) {
  if (!ctx) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t p;
    MM_HIP(hipGetDeviceProperties(&p, ctx->device));
    if (name && name_cap) { snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName); }
    if (cus) *cus = p.multiProcessorCount;
    size_t f = 0, t = 0;
    MM_HIP(mm::dev_mem_info(&f, &t));                             // (capped by the test hook MM_DEVICE_BYTES_CAP, mm_common.hpp)
    if (hbm_total) *hbm_total = t;
    if (hbm_free) *hbm_free = f;
  });
}
int mm_ctx_synchronize(mm_ctx* ctx) {
  if (!ctx) return MM_ERR_ARG;
  return guarded(ctx, [&] { MM_HIP(mm::stream_sync(ctx->stream)); });
}
int mm_ctx_release_cached(mm_ctx* ctx) {
  if (!ctx) return MM_ERR_ARG;
  return guarded(ctx, [&] { MM_HIP(hipSetDevice(ctx->device)); ctx->alloc.trim(); mm::big_pool_trim(ctx->device); });
}
void* mm_ctx_stream(mm_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ---- sequences ----------------------------------------------------------------------------------------
int mm_seqset_create(mm_ctx* ctx, mm_seqset** out) {
  if (!ctx || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] { auto* s = new mm_seqset; s->ctx = ctx; *out = s; });
}
void mm_seqset_destroy(mm_seqset* s) { if (s) { (void)hipSetDevice(s->ctx->device); mm::current_stream() = s->ctx->stream; mm::current_alloc() = &s->ctx->alloc; delete s; } }
int mm_seqset_add(mm_seqset* s, const char* ascii, int64_t len) {
  if (!s || (!ascii && len > 0) || len < 0) return MM_ERR_ARG;
  return guarded(s->ctx, [&] {
    MM_REQUIRE(!s->frozen, MM_ERR_STATE, "sequence set already uploaded");
    s->owned.emplace_back(ascii ? ascii : "", (size_t)len);
    s->staged.emplace_back(s->owned.back().data(), (size_t)len);
  });
}
int mm_seqset_add_view(mm_seqset* s, const char* ascii, int64_t len) {
  if (!s || (!ascii && len > 0) || len < 0) return MM_ERR_ARG;
  return guarded(s->ctx, [&] {
    MM_REQUIRE(!s->frozen, MM_ERR_STATE, "sequence set already uploaded");
    s->staged.emplace_back(ascii, (size_t)len);
  });
}
int mm_seqset_upload(mm_seqset* s) {
  if (!s) return MM_ERR_ARG;
  return guarded(s->ctx, [&] { MM_HIP(hipSetDevice(s->ctx->device)); mm::seqset_upload(s); });
}
int mm_seqset_save(mm_seqset* s, const char* path) {
  if (!s || !path) return MM_ERR_ARG;
  return guarded(s->ctx, [&] { MM_HIP(hipSetDevice(s->ctx->device)); mm::seqset_save(s, path); });
}
int mm_seqset_load(mm_ctx* ctx, const char* path, mm_seqset** out) {
  if (!ctx || !path || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    auto* S = new mm_seqset; S->ctx = ctx;
    try { mm::seqset_load(S, path); } catch (...) { delete S; throw; }
    *out = S;
  });
}
int mm_seqset_slice(mm_ctx* ctx, const mm_seqset* set, int64_t first, int64_t count, mm_seqset** out) {
  if (!ctx || !set || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    MM_REQUIRE(set->ctx->device == ctx->device, MM_ERR_ARG, "mm_seqset_slice: the set lives on another device than the context");
    auto* S = new mm_seqset; S->ctx = ctx;
    try { mm::seqset_slice(set, first, count, S); } catch (...) { delete S; throw; }
    *out = S;
  });
}
int mm_seqset_concat(mm_ctx* ctx, const mm_seqset* const* parts, int n_parts, mm_seqset** out) {
  if (!ctx || !parts || n_parts <= 0 || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    auto* S = new mm_seqset; S->ctx = ctx;
    try { mm::seqset_concat(parts, n_parts, S); } catch (...) { delete S; throw; }
    *out = S;
  });
}
int64_t mm_seqset_count(const mm_seqset* s) { return s ? (s->frozen ? s->count() : (int64_t)s->staged.size()) : 0; }
int64_t mm_seqset_total_bases(const mm_seqset* s) { return s ? s->total_bases : 0; }
int mm_seqset_lengths(const mm_seqset* s, int32_t* len_out) {
  if (!s || !len_out || !s->frozen) return MM_ERR_ARG;
  memcpy(len_out, s->len.data(), s->len.size() * sizeof(int32_t));
  return MM_OK;
}
int mm_seqset_fetch(mm_seqset* s, int64_t i, char* ascii_out, int64_t cap) {
  if (!s || !ascii_out) return MM_ERR_ARG;
  return guarded(s->ctx, [&] { mm::seqset_fetch(s, i, ascii_out, cap); });
}

int mm_seqset_fetch_range(mm_seqset* s, int64_t first, int64_t count, char* ascii_out, int64_t cap) {
  if (!s || (!ascii_out && cap > 0)) return MM_ERR_ARG;
  return guarded(s->ctx, [&] { mm::seqset_fetch_range(s, first, count, ascii_out, cap); });
}

int mm_synth_reference(mm_ctx* ctx, const mm_synth_ref_params* p, mm_seqset** out) {
  if (!ctx || !p || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] { MM_HIP(hipSetDevice(ctx->device)); auto* s = new mm_seqset; s->ctx = ctx; try { mm::synth_reference(ctx, *p, s); } catch (...) { delete s; throw; } *out = s; });
}
int mm_synth_community(mm_ctx* ctx, const mm_synth_community_params* p, mm_seqset** out, int32_t* contig_genome) {
  if (!ctx || !p || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] { auto* s = new mm_seqset; s->ctx = ctx; try { mm::synth_community(ctx, *p, s, contig_genome); } catch (...) { delete s; throw; } *out = s; });
}
int mm_synth_community_species(const mm_synth_community_params* p, int32_t* genome_species) {
  if (!p || !genome_species) return MM_ERR_ARG;
  return guarded(nullptr, [&] { mm::synth_community_species(*p, genome_species); });
}
int mm_synth_reads(mm_ctx* ctx, const mm_seqset* reference, const mm_synth_read_params* p, mm_seqset** out, int32_t* truth_genome) {
  if (!ctx || !reference || !p || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] { MM_HIP(hipSetDevice(ctx->device)); auto* s = new mm_seqset; s->ctx = ctx; try { mm::synth_reads(ctx, reference, *p, s, truth_genome); } catch (...) { delete s; throw; } *out = s; });
}

// ---- minimizer tap ------------------------------------------------------------------------------------
int mm_minimizers(mm_ctx* ctx, const mm_seqset* s, int k, int w, int64_t* offsets, uint32_t* hash, int32_t* wpos, int32_t* strand, int64_t cap) {
  if (!ctx || !s) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    mm::MinimizerSet ms;
    mm::run_minimizers(ctx, s, k, w, {}, false, ms);
    if (offsets) for (size_t i = 0; i < ms.h_off.size(); ++i) offsets[i] = (int64_t)ms.h_off[i];
    if (hash || wpos || strand) {
      MM_REQUIRE(cap >= ms.total, MM_ERR_ARG, "output capacity too small");
      auto h = ms.rec.to_host(ctx->stream, (size_t)ms.total);
      for (int64_t i = 0; i < ms.total; ++i) {
        if (hash) hash[i] = h[(size_t)i].hash;
        if (wpos) wpos[i] = mm::pw_wpos(h[(size_t)i].pw);
        if (strand) strand[i] = mm::pw_strand(h[(size_t)i].pw);
      }
    }
  });
}

// ---- index --------------------------------------------------------------------------------------------
int mm_index_build(mm_ctx* ctx, const mm_seqset* contigs, int k, int w, mm_index** out) {
  if (!ctx || !contigs || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    auto* I = new mm_index;
    try { mm::index_build(ctx, contigs, k, w, I); } catch (...) { delete I; throw; }
    *out = I;
  });
}
int mm_index_plan_chunks(mm_ctx* ctx, const mm_index* whole, uint64_t max_memory_bytes, int32_t* first_contig, int32_t cap, int32_t* n_chunks) {
  if (!ctx || !whole || !n_chunks) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    std::vector<int32_t> fc;
    mm::index_plan_chunks(ctx, whole, max_memory_bytes, fc);
    *n_chunks = (int32_t)fc.size();
    if (first_contig) {
      MM_REQUIRE(cap >= (int32_t)fc.size(), MM_ERR_ARG, "output capacity too small");
      for (size_t i = 0; i < fc.size(); ++i) first_contig[i] = fc[i];
    }
  });
}
int mm_index_save(mm_index* idx, const char* path) {
  if (!idx || !path) return MM_ERR_ARG;
  return guarded(idx->ctx, [&] {
    MM_HIP(hipSetDevice(idx->ctx->device));
    mm::index_save(idx, path);
  });
}
int mm_index_load(mm_ctx* ctx, const char* path, mm_index** out) {
  if (!ctx || !path || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    auto* I = new mm_index;
    I->ctx = ctx;
    try { mm::index_load(ctx, path, I); } catch (...) { delete I; throw; }
    *out = I;
  });
}
void mm_index_destroy(mm_index* idx) { if (idx) { (void)hipSetDevice(idx->ctx->device); mm::current_stream() = idx->ctx->stream; mm::current_alloc() = &idx->ctx->alloc; delete idx; } }
int mm_index_get_info(const mm_index* idx, mm_index_info* out) {
  if (!idx || !out) return MM_ERR_ARG;
  out->n_contigs = idx->n_contigs; out->n_entries = idx->N; out->n_unique_hashes = idx->U; out->n_dup_flagged = idx->n_dup;
  out->hbm_bytes = idx->hbm_bytes();
  return MM_OK;
}
int mm_index_freq_hist(mm_index* idx, int64_t* counts, int64_t* n_hashes, int64_t cap, int64_t* n_out) {
  if (!idx || !n_out) return MM_ERR_ARG;
  *n_out = (int64_t)idx->hist.size();
  if (!counts || !n_hashes) return MM_OK;
  if (cap < *n_out) return MM_ERR_ARG;
  int64_t i = 0;
  for (auto& kv : idx->hist) { counts[i] = kv.first; n_hashes[i] = kv.second; ++i; }
  return MM_OK;
}
// winSketch.hpp:452-494 on an accumulated histogram.  `prev_threshold` is the threshold left by the previous
// chunk (INT_MAX for the first): the reference keeps the old value when the loop breaks before assigning.
int mm_freq_threshold_from_hist(const int64_t* counts, const int64_t* n_hashes, int64_t n, int64_t n_unique_hashes, int prev_threshold) {
  int thr = prev_threshold;
  if (n_unique_hashes <= 0) return thr;                        // :456 — empty lookup: nothing happens
  float pct = 0.001f;                                          // :91
  int64_t ignore = n_unique_hashes * pct / 100;                // :466 float arithmetic, truncated
  int64_t sum = 0;
  for (int64_t i = n - 1; i >= 0; --i) {                       // most frequent first
    sum += n_hashes[i];
    if (sum < ignore) thr = (int)counts[i];
    else if (sum == ignore) { thr = (int)counts[i]; break; }
    else break;
  }
  return thr;
}
int mm_index_set_freq_threshold(mm_index* idx, int threshold) {
  if (!idx) return MM_ERR_ARG;
  idx->freq_threshold = threshold;
  return MM_OK;
}
int mm_index_entries(mm_index* idx, uint32_t* hash, int32_t* contig, int32_t* wpos, int32_t* strand, int64_t cap) {
  if (!idx) return MM_ERR_ARG;
  return guarded(idx->ctx, [&] {
    MM_REQUIRE(cap >= idx->N, MM_ERR_ARG, "output capacity too small");
    auto h = idx->pos.to_host(idx->ctx->stream, (size_t)idx->N);
    int64_t c = 0;
    for (int64_t i = 0; i < idx->N; ++i) {
      while (c + 1 < (int64_t)idx->h_cstart.size() && (uint64_t)i >= idx->h_cstart[(size_t)c + 1]) ++c;
      if (hash) hash[i] = h[(size_t)i].hash;
      if (contig) contig[i] = (int32_t)c;
      if (wpos) wpos[i] = mm::pw_wpos(h[(size_t)i].pw);
      if (strand) strand[i] = mm::pw_strand(h[(size_t)i].pw);
    }
  });
}

int mm_index_dup_neighbours(mm_index* idx, int32_t* prev_dist, int32_t* next_dist, int64_t cap) {
  if (!idx || !prev_dist || !next_dist) return MM_ERR_ARG;
  return guarded(idx->ctx, [&] {
    MM_REQUIRE(cap >= idx->N, MM_ERR_ARG, "output capacity too small");
    if (idx->N == 0) return;
    hipStream_t st = idx->ctx->stream;
    auto h = idx->pos.to_host(st, (size_t)idx->N);
    auto bits = idx->dup_bits.to_host(st), rank = idx->dup_rank.to_host(st);
    auto dist = idx->dup_dist.to_host(st);
    for (int64_t i = 0; i < idx->N; ++i) {
      prev_dist[i] = 0; next_dist[i] = 0;
      const uint32_t fl = h[(size_t)i].pw & (mm::PW_DP | mm::PW_DN);
      const bool bit = (bits[(size_t)(i >> 6)] >> (i & 63)) & 1ull;
      MM_REQUIRE(bit == (fl != 0), MM_ERR_DEVICE, "duplicate bitmap and entry flags disagree");
      if (!fl) continue;
      const uint64_t r = rank[(size_t)(i >> 6)] + (uint64_t)__builtin_popcountll(bits[(size_t)(i >> 6)] & ((1ull << (i & 63)) - 1ull));
      MM_REQUIRE(r < dist.size(), MM_ERR_DEVICE, "duplicate rank out of range");
      if (fl & mm::PW_DP) prev_dist[i] = (int32_t)(dist[(size_t)r] & 0xffffu);
      if (fl & mm::PW_DN) next_dist[i] = (int32_t)(dist[(size_t)r] >> 16);
    }
  });
}

// ---- statistics ---------------------------------------------------------------------------------------
int mm_recommended_window(double p_value, int k, float pi, int min_read_len, uint64_t reference_size) {
  return mm::stats::recommended_window(p_value, k, 4, pi, min_read_len, reference_size);
}
double mm_estimate_pvalue(int s, int k, float pi, int min_read_len, uint64_t reference_size) {
  return mm::stats::estimate_pvalue(s, k, 4, pi, min_read_len, reference_size);
}
int mm_min_hits_relaxed(int s, int k, float pi) { return mm::stats::min_hits_relaxed(s, k, pi); }
void mm_identity(int shared, int s, int k, float* ident, float* ident_upper) {
  if (!ident_upper) {                                            // the estimate alone needs no binomial quantile
    if (ident) *ident = 100 * (1 - mm::stats::j2md(1.0 * shared / s, k));
    return;
  }
  float a, b;
  mm::stats::identity(shared, s, k, &a, &b);
  if (ident) *ident = a;
  *ident_upper = b;
}

// ---- mapping ------------------------------------------------------------------------------------------
int mm_map_batch(mm_ctx* ctx, const mm_index* idx, const mm_seqset* reads, const mm_map_params* p, mm_mapping** out) {
  if (!ctx || !idx || !reads || !p || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    MM_REQUIRE(idx->ctx->device == ctx->device, MM_ERR_ARG, "mm_map_batch: the index lives on another device than the context");
    MM_REQUIRE(reads->ctx->device == ctx->device, MM_ERR_ARG, "mm_map_batch: the reads live on another device than the context");
    auto* M = new mm_mapping;
    try { mm::map_batch(ctx, idx, reads, *p, M); } catch (...) { delete M; throw; }
    *out = M;
  });
}
int mm_map_batch_reusing(mm_ctx* ctx, const mm_index* idx, const mm_seqset* reads, const mm_map_params* p, const mm_mapping* sketch_of, mm_mapping** out) {
  if (!ctx || !idx || !reads || !p || !sketch_of || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    MM_REQUIRE(idx->ctx->device == ctx->device && reads->ctx->device == ctx->device && sketch_of->ctx->device == ctx->device, MM_ERR_ARG,
               "mm_map_batch_reusing: index, reads and donor mapping must live on the context's device");
    MM_REQUIRE(!sketch_of->released, MM_ERR_STATE, "mm_map_batch_reusing: the donor mapping has released its intermediates");
    MM_REQUIRE(sketch_of->n_reads == reads->count() && sketch_of->read_len == reads->len && sketch_of->params.k == p->k && sketch_of->params.w == p->w &&
               sketch_of->params.min_read_len == p->min_read_len, MM_ERR_ARG, "mm_map_batch_reusing: the donor mapping is of other reads or other parameters");
    auto* M = new mm_mapping;
    M->sketch_donor = sketch_of;
    try { mm::map_batch(ctx, idx, reads, *p, M); } catch (...) { delete M; throw; }
    M->sketch_donor = nullptr;
    *out = M;
  });
}
int mm_sketch_batch(mm_ctx* ctx, const mm_seqset* reads, const mm_map_params* p, mm_mapping** out) {
  if (!ctx || !reads || !p || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    MM_REQUIRE(reads->ctx->device == ctx->device, MM_ERR_ARG, "mm_sketch_batch: the reads live on another device than the context");
    auto* M = new mm_mapping;
    M->sketch_only = true;
    try { mm::map_batch(ctx, nullptr, reads, *p, M); } catch (...) { delete M; throw; }
    *out = M;
  });
}
int mm_map_batch_phased(mm_ctx* ctx, const mm_index* idx, const mm_seqset* reads, const mm_map_params* p, void (*at_stage)(void*, int), void* user, mm_mapping** out) {
  if (!ctx || !idx || !reads || !p || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    MM_REQUIRE(idx->ctx->device == ctx->device, MM_ERR_ARG, "mm_map_batch_phased: the index lives on another device than the context");
    MM_REQUIRE(reads->ctx->device == ctx->device, MM_ERR_ARG, "mm_map_batch_phased: the reads live on another device than the context");
    auto* M = new mm_mapping;
    M->at_stage = at_stage; M->at_stage_user = user;
    try { mm::map_batch(ctx, idx, reads, *p, M); } catch (...) { delete M; throw; }
    *out = M;
  });
}
void mm_mapping_destroy(mm_mapping* m) { if (m) { (void)hipSetDevice(m->ctx->device); mm::current_stream() = m->ctx->stream; mm::current_alloc() = &m->ctx->alloc; delete m; } }
int mm_mapping_release_intermediates(mm_mapping* m) {
  if (!m) return MM_ERR_ARG;
  return guarded(m->ctx, [&] {
    MM_HIP(hipSetDevice(m->ctx->device));
    m->mz = mm::MinimizerSet{};
    m->sk_hash.release(); m->sk_strand.release(); m->sk_n.release(); m->amb.release();
    m->min_hits.release(); m->accept_min.release();
    m->read_hit_off.release(); m->hits.release(); m->cand_off.release(); m->cand.release(); m->cand_read.release(); m->l2.release();
    std::vector<int32_t>().swap(m->h_sk_n); std::vector<int32_t>().swap(m->h_min_hits);
    std::vector<uint64_t>().swap(m->h_read_hit_off); std::vector<uint64_t>().swap(m->h_cand_off);
    m->n_cand = 0; m->released = true;
  });
}
int mm_mapping_get_stats(const mm_mapping* m, mm_map_stats* out) {
  if (!m || !out) return MM_ERR_ARG;
  *out = m->stats;
  return MM_OK;
}
int mm_mapping_fetch(mm_mapping* m, int64_t* offsets, mm_map_record* records, int64_t cap) {
  if (!m) return MM_ERR_ARG;
  return guarded(m->ctx, [&] {
    if (offsets) for (size_t i = 0; i < m->h_rec_off.size(); ++i) offsets[i] = (int64_t)m->h_rec_off[i];
    if (records) {
      MM_REQUIRE(cap >= m->n_rec, MM_ERR_ARG, "output capacity too small");
      const size_t bytes = (size_t)m->n_rec * sizeof(mm_map_record);
      if (bytes) {                                               // device -> pinned bounce buffer -> caller memory
        void* pin = m->ctx->pinned_at_least(bytes);
        MM_HIP(hipMemcpyAsync(pin, m->rec.p, bytes, hipMemcpyDeviceToHost, m->ctx->stream));
        MM_HIP(mm::stream_sync(m->ctx->stream));
        memcpy(records, pin, bytes);
      }
    }
  });
}
int mm_mapping_add_qualities(mm_ctx* ctx, mm_mapping* m, const mm_seqset* reads, int k) {
  if (!ctx || !m) return MM_ERR_ARG;
  (void)reads;
  return guarded(ctx, [&] { MM_HIP(hipSetDevice(ctx->device)); mm::mapping_add_qualities(ctx, m, k); });
}
// read-wise concatenation of per-chunk record lists in chunk order (unifyFiles, mapWrap.h:128-132); small, done on the host
// ---- read-wise merge of the records of several index chunks, chunk order preserved (unifyFiles, mapWrap.h:128-132) — on the device.
// Rounds 1-3 merged on the host (download, a serial loop, upload), also for chunk indexes resident on ONE device.
struct PartDev { const uint64_t* off; const mm_map_record* rec; int32_t base; int32_t pad; };
__global__ void __launch_bounds__(256) merge_count_kernel(const PartDev* __restrict__ parts, int n_parts, int64_t n, uint64_t* __restrict__ cnt) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > n) return;
  uint64_t c = 0;
  if (r < n) for (int p = 0; p < n_parts; ++p) c += parts[p].off[r + 1] - parts[p].off[r];
  cnt[r] = c;
}
__global__ void __launch_bounds__(256) merge_copy_kernel(const PartDev* __restrict__ parts, int n_parts, int64_t n, const uint64_t* __restrict__ off,
                                                         mm_map_record* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  uint64_t o = off[r];
  for (int p = 0; p < n_parts; ++p) {
    const PartDev P = parts[p];
    for (uint64_t i = P.off[r]; i < P.off[r + 1]; ++i) { mm_map_record x = P.rec[i]; x.ref_contig += P.base; out[o++] = x; }
  }
}
static mm_mapping* merge_parts_device(mm_ctx* ctx, int64_t n, const std::vector<int32_t>& read_len, const mm_map_params& params, const std::vector<PartDev>& parts) {
  hipStream_t st = ctx->stream;
  auto* M = new mm_mapping;
  try {
    M->ctx = ctx; M->n_reads = n; M->params = params; M->read_len = read_len;
    M->active.assign((size_t)n, 0);
    M->stats = mm_map_stats{};
    M->stats.n_reads = n;
    for (int64_t r = 0; r < n; ++r) {
      const int L = read_len[(size_t)r];
      const bool ok = !(L < params.w || L < params.k || L < params.min_read_len);   // computeMap.hpp:137
      M->active[(size_t)r] = ok;
      if (ok) { M->stats.n_reads_long_enough++; M->stats.bases_long_enough += L; }
    }
    mm::DBuf<PartDev> d_parts(parts.size()); d_parts.upload(parts.data(), parts.size(), st);
    mm::DBuf<uint64_t> cnt((size_t)n + 1);
    M->rec_off.alloc((size_t)n + 1);
    const unsigned gb = (unsigned)mm::ceil_div(n + 1, 256);
    merge_count_kernel<<<dim3(gb), dim3(256), 0, st>>>(d_parts.p, (int)parts.size(), n, cnt.p);
    MM_KERNEL_CHECK();
    size_t tmp_bytes = 0;
    MM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt.p, M->rec_off.p, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), st));
    mm::DBuf<uint8_t> tmp(std::max<size_t>(tmp_bytes, 1));
    MM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, cnt.p, M->rec_off.p, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), st));
    M->h_rec_off = M->rec_off.to_host(st);
    M->n_rec = (int64_t)M->h_rec_off[(size_t)n];
    M->rec.alloc(std::max<size_t>((size_t)M->n_rec, 1));
    if (n > 0 && M->n_rec > 0) {
      merge_copy_kernel<<<dim3((unsigned)mm::ceil_div(n, 256)), dim3(256), 0, st>>>(d_parts.p, (int)parts.size(), n, M->rec_off.p, M->rec.p);
      MM_KERNEL_CHECK();
    }
    M->d_read_len.alloc((size_t)std::max<int64_t>(n, 1)); M->d_read_len.upload(M->read_len.data(), (size_t)n, st);
    M->stats.n_mappings = M->n_rec;
    for (int64_t r = 0; r < n; ++r) if (M->h_rec_off[(size_t)r + 1] > M->h_rec_off[(size_t)r]) M->stats.n_reads_mapped++;
    M->released = true;                                          // records only: the debug taps have nothing to show
    MM_HIP(mm::stream_sync(st));
  } catch (...) { delete M; throw; }
  return M;
}
static void add_part_stats(mm_mapping* M, mm_mapping* const* parts, int n_parts) {   // work counters and stage times: summed over the chunks; per-read facts: of chunk 0
  for (int p = 0; p < n_parts; ++p) {
    const mm_map_stats& S = parts[p]->stats;
    M->stats.sum_hits += S.sum_hits; M->stats.n_candidates += S.n_candidates; M->stats.sum_hits_kept += S.sum_hits_kept;
    M->stats.sum_l2_stream_entries += S.sum_l2_stream_entries; M->stats.sum_l2_evals += S.sum_l2_evals;
    M->stats.n_l2_rebuilds += S.n_l2_rebuilds; M->stats.n_l2_wide_redo += S.n_l2_wide_redo;
    M->stats.ms_minimizer += S.ms_minimizer; M->stats.ms_sketch += S.ms_sketch; M->stats.ms_probe_gather += S.ms_probe_gather;
    M->stats.ms_sort_hits += S.ms_sort_hits; M->stats.ms_l1_scan += S.ms_l1_scan; M->stats.ms_l2 += S.ms_l2; M->stats.ms_compact += S.ms_compact;
    M->stats.ms_total += S.ms_total; M->stats.ms_hit_filter += S.ms_hit_filter;
  }
  if (n_parts > 0) {
    M->stats.sum_sketch = parts[0]->stats.sum_sketch; M->stats.n_ambiguous_sketch_reads = parts[0]->stats.n_ambiguous_sketch_reads;
    M->stats.n_reads_giant = parts[0]->stats.n_reads_giant;
  }
}
// Parts may live in any context of this process: those of another DEVICE come over with a peer copy (xGMI), nothing is staged on the host.
int mm_mapping_concat(mm_ctx* ctx, mm_mapping* const* parts, const int32_t* contig_base, int n_parts, mm_mapping** out) {
  if (!ctx || !parts || n_parts <= 0 || !out) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts[0]->n_reads;
    std::vector<PartDev> pv((size_t)n_parts);
    std::vector<mm::DBuf<uint64_t>> t_off((size_t)n_parts); std::vector<mm::DBuf<mm_map_record>> t_rec((size_t)n_parts);
    for (int p = 0; p < n_parts; ++p) {
      mm_mapping* P = parts[p];
      MM_REQUIRE(P->n_reads == n, MM_ERR_ARG, "chunk results cover different read sets");
      MM_REQUIRE(!P->sketch_only && P->rec_off.p && (P->n_rec == 0 || P->rec.p), MM_ERR_STATE, "mm_mapping_concat: a part holds no records (a mm_sketch_batch result is no chunk mapping)");
      pv[(size_t)p] = PartDev{P->rec_off.p, P->rec.p, contig_base ? contig_base[p] : 0, 0};
      if (P->ctx->device != ctx->device) {                       // (the part's own stream has been waited for by the call that made it)
        t_off[(size_t)p].alloc((size_t)n + 1); t_rec[(size_t)p].alloc(std::max<size_t>((size_t)P->n_rec, 1));
        MM_HIP(hipMemcpyPeerAsync(t_off[(size_t)p].p, ctx->device, P->rec_off.p, P->ctx->device, sizeof(uint64_t) * ((size_t)n + 1), ctx->stream));
        if (P->n_rec > 0) MM_HIP(hipMemcpyPeerAsync(t_rec[(size_t)p].p, ctx->device, P->rec.p, P->ctx->device, sizeof(mm_map_record) * (size_t)P->n_rec, ctx->stream));
        pv[(size_t)p].off = t_off[(size_t)p].p; pv[(size_t)p].rec = t_rec[(size_t)p].p;
      }
    }
    mm_mapping* M = merge_parts_device(ctx, n, parts[0]->read_len, parts[0]->params, pv);
    add_part_stats(M, parts, n_parts);
    *out = M;
  });
}
int mm_mapping_from_parts(mm_ctx* ctx, int64_t n_reads, const int32_t* read_len, const mm_map_params* p, int n_parts,
                          const int64_t* const* offsets, const mm_map_record* const* records, const int32_t* contig_base, mm_mapping** out) {
  if (!ctx || n_reads < 0 || (!read_len && n_reads > 0) || !p || n_parts <= 0 || !offsets || !records || !out) return MM_ERR_ARG;
  for (int i = 0; i < n_parts; ++i) if (!offsets[i] || (!records[i] && offsets[i][n_reads] > 0)) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    std::vector<int32_t> len(read_len, read_len + n_reads);
    std::vector<PartDev> pv((size_t)n_parts);
    std::vector<mm::DBuf<uint64_t>> t_off((size_t)n_parts); std::vector<mm::DBuf<mm_map_record>> t_rec((size_t)n_parts);
    for (int i = 0; i < n_parts; ++i) {
      const size_t nr = (size_t)offsets[i][n_reads];
      t_off[(size_t)i].alloc((size_t)n_reads + 1); t_off[(size_t)i].upload((const uint64_t*)offsets[i], (size_t)n_reads + 1, ctx->stream);
      t_rec[(size_t)i].alloc(std::max<size_t>(nr, 1)); t_rec[(size_t)i].upload(records[i], nr, ctx->stream);
      pv[(size_t)i] = PartDev{t_off[(size_t)i].p, t_rec[(size_t)i].p, contig_base ? contig_base[i] : 0, 0};
    }
    *out = merge_parts_device(ctx, n_reads, len, *p, pv);
  });
}
// The same exchange between the ranks of a communicator (one process per GPU, or one thread per GPU of one process): ncclSend / ncclRecv of the
// offsets, then of the records, straight between the devices.  Collective: every rank calls it for the same batch.
int mm_mapping_gather(mm_ctx* ctx, int owner, int64_t n_reads, const int32_t* read_len, const mm_map_params* p, mm_mapping* const* parts, const int32_t* chunk_id,
                      int n_parts, int n_chunks, const int32_t* chunk_rank, const int32_t* contig_base, mm_mapping** out) {
  if (!ctx || !p || n_reads < 0 || (!read_len && n_reads > 0) || n_parts < 0 || (n_parts > 0 && (!parts || !chunk_id)) || n_chunks <= 0 || !chunk_rank || !out) return MM_ERR_ARG;
  *out = nullptr;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int rank = ctx->comm ? ctx->comm_rank : 0, nranks = ctx->comm ? ctx->comm_size : 1;
    MM_REQUIRE(owner >= 0 && owner < nranks, MM_ERR_ARG, "mm_mapping_gather: owner is no rank of the communicator");
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    const bool self_send = getenv("MM_GATHER_SELF_SEND") != nullptr && comm;   // test hook: the owner's own parts go through ncclSend / ncclRecv too
    auto nccl_ok = [&](ncclResult_t rc, const char* what) { MM_REQUIRE(rc == ncclSuccess, MM_ERR_COMM, std::string(what) + ": " + ncclGetErrorString(rc)); };
    std::vector<int> part_of((size_t)n_chunks, -1);              // chunk -> index into parts[] (this rank's chunks)
    for (int i = 0; i < n_parts; ++i) {
      MM_REQUIRE(chunk_id[i] >= 0 && chunk_id[i] < n_chunks && chunk_rank[chunk_id[i]] == rank && parts[i] && parts[i]->n_reads == n_reads, MM_ERR_ARG, "mm_mapping_gather: a part is not a chunk of this rank");
      MM_REQUIRE(parts[i]->ctx->device == ctx->device, MM_ERR_ARG, "mm_mapping_gather: parts live on the rank's own device");
      MM_REQUIRE(!parts[i]->sketch_only && parts[i]->rec_off.p && (parts[i]->n_rec == 0 || parts[i]->rec.p), MM_ERR_STATE, "mm_mapping_gather: a part holds no records (a mm_sketch_batch result is no chunk mapping)");
      part_of[(size_t)chunk_id[i]] = i;
    }
    for (int c = 0; c < n_chunks; ++c) MM_REQUIRE(chunk_rank[c] != rank || part_of[(size_t)c] >= 0, MM_ERR_ARG, "mm_mapping_gather: a chunk of this rank has no part");
    const size_t n1 = (size_t)n_reads + 1;
    if (rank != owner) {                                         // my chunks, ascending: all offset arrays, then all record arrays
      MM_REQUIRE(comm, MM_ERR_STATE, "mm_mapping_gather needs a communicator (mm_comm_init)");
      nccl_ok(ncclGroupStart(), "ncclGroupStart");
      for (int c = 0; c < n_chunks; ++c) if (chunk_rank[c] == rank) nccl_ok(ncclSend(parts[part_of[(size_t)c]]->rec_off.p, n1, ncclUint64, owner, comm, st), "ncclSend");
      nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
      nccl_ok(ncclGroupStart(), "ncclGroupStart");
      for (int c = 0; c < n_chunks; ++c) if (chunk_rank[c] == rank) { mm_mapping* P = parts[part_of[(size_t)c]]; if (P->n_rec > 0) nccl_ok(ncclSend(P->rec.p, sizeof(mm_map_record) * (size_t)P->n_rec, ncclInt8, owner, comm, st), "ncclSend"); }
      nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
      MM_HIP(mm::stream_sync(st));
      return;
    }
    std::vector<mm::DBuf<uint64_t>> t_off((size_t)n_chunks); std::vector<mm::DBuf<mm_map_record>> t_rec((size_t)n_chunks);
    auto remote = [&](int c) { return chunk_rank[c] != rank || self_send; };
    bool any_remote = false;
    for (int c = 0; c < n_chunks; ++c) any_remote = any_remote || remote(c);
    std::vector<uint64_t> n_rec_of((size_t)n_chunks, 0);
    if (any_remote) {
      MM_REQUIRE(comm, MM_ERR_STATE, "mm_mapping_gather needs a communicator (mm_comm_init)");
      nccl_ok(ncclGroupStart(), "ncclGroupStart");
      for (int c = 0; c < n_chunks; ++c) {
        if (!remote(c)) continue;
        t_off[(size_t)c].alloc(n1);
        if (chunk_rank[c] == rank) nccl_ok(ncclSend(parts[part_of[(size_t)c]]->rec_off.p, n1, ncclUint64, rank, comm, st), "ncclSend");
        nccl_ok(ncclRecv(t_off[(size_t)c].p, n1, ncclUint64, chunk_rank[c], comm, st), "ncclRecv");
      }
      nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
      for (int c = 0; c < n_chunks; ++c) if (remote(c)) t_off[(size_t)c].download(&n_rec_of[(size_t)c], 1, st, (size_t)n_reads);
      MM_HIP(mm::stream_sync(st));
      nccl_ok(ncclGroupStart(), "ncclGroupStart");
      for (int c = 0; c < n_chunks; ++c) {
        if (!remote(c) || n_rec_of[(size_t)c] == 0) continue;
        t_rec[(size_t)c].alloc((size_t)n_rec_of[(size_t)c]);
        if (chunk_rank[c] == rank) nccl_ok(ncclSend(parts[part_of[(size_t)c]]->rec.p, sizeof(mm_map_record) * (size_t)n_rec_of[(size_t)c], ncclInt8, rank, comm, st), "ncclSend");
        nccl_ok(ncclRecv(t_rec[(size_t)c].p, sizeof(mm_map_record) * (size_t)n_rec_of[(size_t)c], ncclInt8, chunk_rank[c], comm, st), "ncclRecv");
      }
      nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
    }
    std::vector<PartDev> pv((size_t)n_chunks);
    for (int c = 0; c < n_chunks; ++c) {
      const int32_t base = contig_base ? contig_base[c] : 0;
      if (remote(c)) pv[(size_t)c] = PartDev{t_off[(size_t)c].p, t_rec[(size_t)c].p, base, 0};
      else { mm_mapping* P = parts[part_of[(size_t)c]]; pv[(size_t)c] = PartDev{P->rec_off.p, P->rec.p, base, 0}; }
    }
    std::vector<int32_t> len(read_len, read_len + n_reads);
    *out = merge_parts_device(ctx, n_reads, len, *p, pv);
    if (*out) {
      std::vector<mm_mapping*> mine(parts, parts + n_parts);
      add_part_stats(*out, mine.data(), n_parts);                // (of the owner's own chunks: the counters of the other ranks stay with them)
    }
  });
}

int mm_mapping_keep_best(mm_ctx* ctx, mm_mapping* m, int k) {
  if (!ctx || !m) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    // default (non --all) reporting, computeMap.hpp:546-587: per read keep the mappings whose (float) identity is
    // >= best - 1.0; the comparison happens in double exactly as there.  Record lists are small: host side.
    std::vector<mm_map_record> recs((size_t)m->n_rec), kept;
    m->rec.download(recs.data(), recs.size(), ctx->stream);
    MM_HIP(mm::stream_sync(ctx->stream));
    std::vector<uint64_t> off((size_t)m->n_reads + 1, 0);
    int64_t mapped = 0;
    for (int64_t r = 0; r < m->n_reads; ++r) {
      const uint64_t a = m->h_rec_off[(size_t)r], b = m->h_rec_off[(size_t)r + 1];
      float best = 0;
      std::vector<float> id((size_t)(b - a));
      for (uint64_t i = a; i < b; ++i) {
        float ub; mm::stats::identity(recs[(size_t)i].shared, recs[(size_t)i].sketch, k, &id[(size_t)(i - a)], &ub);
        if (id[(size_t)(i - a)] > best) best = id[(size_t)(i - a)];
      }
      for (uint64_t i = a; i < b; ++i) if (id[(size_t)(i - a)] >= best - 1.0) kept.push_back(recs[(size_t)i]);
      off[(size_t)r + 1] = kept.size();
      if (off[(size_t)r + 1] > off[(size_t)r]) ++mapped;
    }
    m->n_rec = (int64_t)kept.size();
    m->rec.alloc(std::max<size_t>(kept.size(), 1)); m->rec.upload(kept.data(), kept.size(), ctx->stream);
    m->rec_off.upload(off.data(), off.size(), ctx->stream);
    m->h_rec_off = off;
    m->stats.n_mappings = m->n_rec; m->stats.n_reads_mapped = mapped;
    MM_HIP(mm::stream_sync(ctx->stream));
  });
}

int mm_debug_sketch(mm_mapping* m, int64_t* offsets, uint32_t* hash, int32_t* strand, int64_t cap) {
  if (m && m->released) return MM_ERR_STATE;
  if (!m) return MM_ERR_ARG;
  return guarded(m->ctx, [&] {
    hipStream_t st = m->ctx->stream;
    int64_t tot = 0;
    for (int64_t r = 0; r < m->n_reads; ++r) { if (offsets) offsets[r] = tot; tot += m->h_sk_n[(size_t)r]; }
    if (offsets) offsets[m->n_reads] = tot;
    if (!hash && !strand) return;
    MM_REQUIRE(cap >= tot, MM_ERR_ARG, "output capacity too small");
    auto hh = m->sk_hash.to_host(st); auto hs = m->sk_strand.to_host(st);
    int64_t o = 0;
    for (int64_t r = 0; r < m->n_reads; ++r)
      for (int i = 0; i < m->h_sk_n[(size_t)r]; ++i, ++o) {
        size_t src = (size_t)m->mz.h_off[(size_t)r] + (size_t)i;
        if (hash) hash[o] = hh[src];
        if (strand) strand[o] = (hs[src] & 1) ? 1 : -1;       // (bit 1: duplicate of the other strand not resolved, mm_map.hip K2)
      }
  });
}
int mm_debug_hits(mm_mapping* m, int64_t* offsets, int32_t* contig, int32_t* wpos, int64_t cap) {
  if (m && m->released) return MM_ERR_STATE;
  if (!m) return MM_ERR_ARG;
  return guarded(m->ctx, [&] {
    const int64_t tot = (int64_t)m->h_read_hit_off[(size_t)m->n_reads];
    if (offsets) for (size_t i = 0; i < m->h_read_hit_off.size(); ++i) offsets[i] = (int64_t)m->h_read_hit_off[i];
    if (!contig && !wpos) return;
    MM_REQUIRE(cap >= tot, MM_ERR_ARG, "output capacity too small");
    auto h = m->hits.to_host(m->ctx->stream, (size_t)tot);
    for (int64_t i = 0; i < tot; ++i) {
      if (contig) contig[i] = (int32_t)(h[(size_t)i] >> 32);
      if (wpos) wpos[i] = mm::pw_wpos((uint32_t)h[(size_t)i]);
    }
  });
}
int mm_debug_candidates(mm_mapping* m, int64_t* offsets, int32_t* triples, int64_t cap) {
  if (m && m->released) return MM_ERR_STATE;
  if (!m) return MM_ERR_ARG;
  return guarded(m->ctx, [&] {
    if (offsets) for (size_t i = 0; i < m->h_cand_off.size(); ++i) offsets[i] = (int64_t)m->h_cand_off[i];
    if (!triples) return;
    MM_REQUIRE(cap >= m->n_cand, MM_ERR_ARG, "output capacity too small");
    m->cand.download(triples, (size_t)(3 * m->n_cand), m->ctx->stream);
    MM_HIP(mm::stream_sync(m->ctx->stream));
  });
}
int mm_debug_l2(mm_mapping* m, int64_t* per_cand, int64_t cap) {
  if (m && m->released) return MM_ERR_STATE;
  if (!m || !per_cand) return MM_ERR_ARG;
  return guarded(m->ctx, [&] {
    MM_REQUIRE(cap >= m->n_cand, MM_ERR_ARG, "output capacity too small");
    auto h = m->l2.to_host(m->ctx->stream, (size_t)m->n_cand);
    for (int64_t i = 0; i < m->n_cand; ++i) {
      per_cand[6 * i] = h[(size_t)i].contig; per_cand[6 * i + 1] = h[(size_t)i].mean_pos; per_cand[6 * i + 2] = h[(size_t)i].shared;
      per_cand[6 * i + 3] = h[(size_t)i].opt_beg; per_cand[6 * i + 4] = h[(size_t)i].opt_end; per_cand[6 * i + 5] = h[(size_t)i].accepted;
    }
  });
}
int mm_debug_min_hits(mm_mapping* m, int32_t* min_hits) {
  if (m && m->released) return MM_ERR_STATE;
  if (!m || !min_hits) return MM_ERR_ARG;
  memcpy(min_hits, m->h_min_hits.data(), m->h_min_hits.size() * sizeof(int32_t));
  return MM_OK;
}

int mm_debug_probed_lists(mm_mapping* m, const mm_index* idx, int64_t* hist, int32_t n_bins) {
  if (m && m->released) return MM_ERR_STATE;
  if (!m || !idx || !hist || n_bins < 3) return MM_ERR_ARG;
  return guarded(m->ctx, [&] {
    MM_HIP(hipSetDevice(m->ctx->device));
    mm::probed_list_hist(m->ctx, idx, m, n_bins, hist);
  });
}

// ---- EM -----------------------------------------------------------------------------------------------
int mm_em_create(mm_ctx* ctx, int64_t n_reads, const int64_t* read_off, const int32_t* taxon, const double* mapq, const double* inv_nloc,
                 int32_t n_taxa, mm_em** out) {
  if (!ctx || !read_off || !out || n_reads < 0 || n_taxa <= 0) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    auto* E = new mm_em;
    try { mm::em_create(ctx, n_reads, read_off, taxon, mapq, inv_nloc, n_taxa, E); } catch (...) { delete E; throw; }
    *out = E;
  });
}
int mm_em_create_from_mapping(mm_ctx* ctx, const mm_mapping* m, const int32_t* contig_taxon, const int32_t* contig_len, int32_t n_contigs,
                              int32_t n_taxa, mm_em** out) {
  if (!ctx || !m || !contig_taxon || !contig_len || !out || n_contigs < 0 || n_taxa <= 0) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_HIP(hipSetDevice(ctx->device));
    auto* E = new mm_em;
    try { mm::em_create_from_mapping(ctx, m, contig_taxon, contig_len, n_contigs, n_taxa, E); } catch (...) { delete E; throw; }
    *out = E;
  });
}
int mm_em_taxon_counts(mm_em* em, int64_t* counts) {
  if (!em || !counts) return MM_ERR_ARG;
  return guarded(em->ctx, [&] {
    auto ts = em->tstart.to_host(em->ctx->stream, (size_t)em->n_taxa + 1);
    for (int32_t t = 0; t < em->n_taxa; ++t) counts[t] = ts[(size_t)t + 1] - ts[(size_t)t];
  });
}
int mm_em_sizes(const mm_em* em, int64_t* n_reads, int64_t* n_entries, int32_t* n_taxa) {
  if (!em) return MM_ERR_ARG;
  if (n_reads) *n_reads = em->n_reads;
  if (n_entries) *n_entries = em->n_entries;
  if (n_taxa) *n_taxa = em->n_taxa;
  return MM_OK;
}
void mm_em_destroy(mm_em* em) { if (em) { (void)hipSetDevice(em->ctx->device); mm::current_stream() = em->ctx->stream; mm::current_alloc() = &em->ctx->alloc; delete em; } }
int mm_em_iterate(mm_em* em, const double* f, double* f_partial, double* ll_partial) {
  if (!em || !f || !f_partial || !ll_partial) return MM_ERR_ARG;
  return guarded(em->ctx, [&] { MM_HIP(hipSetDevice(em->ctx->device)); mm::em_iterate(em, f, f_partial, ll_partial); });
}
int mm_em_iterate_allreduce(mm_em* em, const double* f, double* f_next, double* ll) {
  if (!em || !f || !f_next || !ll) return MM_ERR_ARG;
  return guarded(em->ctx, [&] { MM_HIP(hipSetDevice(em->ctx->device)); mm::em_iterate_allreduce(em, f, f_next, ll); });
}
int mm_em_run(mm_em* em, const double* f0, int max_iter, double* f_out, double* ll_trace, int ll_cap, int* n_iter) {
  if (!em || !f0 || max_iter <= 0 || !n_iter) return MM_ERR_ARG;
  return guarded(em->ctx, [&] { MM_HIP(hipSetDevice(em->ctx->device)); *n_iter = mm::em_run(em, f0, max_iter, f_out, ll_trace, ll_cap, nullptr); });
}
int mm_em_continue(mm_em* em, int max_iter, double* f_out, double* ll_trace, int ll_cap, int* n_iter, int* stopped) {
  if (!em || max_iter <= 0 || !n_iter) return MM_ERR_ARG;
  return guarded(em->ctx, [&] {
    MM_HIP(hipSetDevice(em->ctx->device));
    bool st = false;
    *n_iter = mm::em_run(em, nullptr, max_iter, f_out, ll_trace, ll_cap, &st);
    if (stopped) *stopped = st ? 1 : 0;
  });
}
int mm_em_posteriors(mm_em* em, const double* f, double* post, int64_t* best) {
  if (!em || !f) return MM_ERR_ARG;
  return guarded(em->ctx, [&] { MM_HIP(hipSetDevice(em->ctx->device)); mm::em_posteriors(em, f, post, best); });
}

// ---- communicator -------------------------------------------------------------------------------------
int mm_comm_unique_id(char id[MM_COMM_ID_BYTES]) {
  if (!id) return MM_ERR_ARG;
  return guarded(nullptr, [&] { mm::comm_unique_id(id); });
}
int mm_comm_init(mm_ctx* ctx, const char id[MM_COMM_ID_BYTES], int rank, int nranks) {
  if (!ctx || !id || rank < 0 || rank >= nranks) return MM_ERR_ARG;
  return guarded(ctx, [&] { mm::comm_init(ctx, id, rank, nranks); });
}
int mm_comm_info(mm_ctx* ctx, int* n_ranks, int* rank) {
  if (!ctx) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    int n = 1, r = 0;
    if (ctx->comm) {
      MM_REQUIRE(ncclCommCount((ncclComm_t)ctx->comm, &n) == ncclSuccess && ncclCommUserRank((ncclComm_t)ctx->comm, &r) == ncclSuccess, MM_ERR_COMM, "ncclCommCount / ncclCommUserRank failed");
    }
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
  });
}
int mm_comm_share(mm_ctx* ctx, mm_ctx* owner) {
  if (!ctx || !owner || ctx == owner) return MM_ERR_ARG;
  return guarded(ctx, [&] {
    MM_REQUIRE(ctx->comm == nullptr, MM_ERR_STATE, "communicator already initialised");
    MM_REQUIRE(owner->comm != nullptr && !owner->comm_shared && owner->device == ctx->device, MM_ERR_ARG, "mm_comm_share: the owner must hold a communicator on the same device");
    ctx->comm = owner->comm; ctx->comm_shared = true; ctx->comm_rank = owner->comm_rank; ctx->comm_size = owner->comm_size;
  });
}
int mm_comm_allreduce_f64(mm_ctx* ctx, double* host_inout, int64_t n) {
  if (!ctx || (!host_inout && n > 0)) return MM_ERR_ARG;
  return guarded(ctx, [&] { MM_HIP(hipSetDevice(ctx->device)); mm::comm_allreduce_f64(ctx, host_inout, n); });
}
void mm_comm_destroy(mm_ctx* ctx) { if (ctx) mm::comm_destroy(ctx); }

}  // extern "C"
