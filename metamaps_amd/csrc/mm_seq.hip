// Sequence sets in HBM (packed 2-bit + exception runs) and the K1 driver.
#include "mm_minimizer.hpp"
#include <algorithm>
#include <thread>
#include <atomic>
#include <cstdio>
#include <cstring>

extern "C" int mm_host_has_avx2();
extern "C" size_t mm_pack_acgt_blocks(const uint8_t* p, size_t n_bases, uint32_t* out);   // host_pack.cpp

namespace mm {

SeqView make_view(const mm_seqset* S) {
  return SeqView{S->packed.p, S->d_base.p, S->d_len.p, S->exc_start.p, S->exc_len.p, S->exc_byte.p, S->n_exc, S->count()};
}

__global__ void gather_offsets_kernel(const uint64_t* __restrict__ tile_first, const uint64_t* __restrict__ tile_out, int64_t n,
                                      uint64_t* __restrict__ off) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s <= n) off[s] = tile_out[tile_first[s]];
}

void run_minimizers(mm_ctx* ctx, const mm_seqset* S, int k, int w, const std::vector<uint8_t>& active, bool want_rec_seq,
                    MinimizerSet& out) {
  MM_REQUIRE(S->frozen, MM_ERR_STATE, "sequence set not uploaded");
  MM_REQUIRE(k >= 1 && k <= MZ_MAX_K, MM_ERR_ARG, "k must be in [1,64]");
  MM_REQUIRE(w >= 1 && w <= MZ_MAX_W, MM_ERR_ARG, "window size must be in [1,4096]");
  hipStream_t st = ctx->stream;
  const int64_t n = S->count();
  std::vector<uint64_t> tf((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    int64_t npos = (int64_t)S->len[i] - k + 1;
    bool on = active.empty() || active[(size_t)i];
    int64_t nt = (on && npos >= w) ? ceil_div(npos, MZ_TILE) : 0;   // no window fits => nothing can be emitted
    tf[i + 1] = tf[i] + (uint64_t)nt;
  }
  const int64_t ntiles = (int64_t)tf[n];
  out.off.alloc((size_t)n + 1);
  out.h_off.assign((size_t)n + 1, 0);
  out.total = 0;
  if (ntiles == 0) { out.off.zero(st); out.rec.alloc(0); MM_HIP(mm::stream_sync(st)); return; }
  MM_REQUIRE(ntiles < (1LL << 31), MM_ERR_LIMIT, "too many tiles for one launch");

  DBuf<uint64_t> d_tf((size_t)n + 1);
  d_tf.upload(tf.data(), tf.size(), st);
  DBuf<uint8_t> d_act;
  if (!active.empty()) { d_act.alloc(active.size()); d_act.upload(active.data(), active.size(), st); }
  DBuf<int32_t> d_js((size_t)n);
  SeqView V = make_view(S);
  jstar_kernel<<<dim3((unsigned)ceil_div(n, 128)), dim3(128), 0, st>>>(V, d_act.p, k, w, d_js.p);
  MM_KERNEL_CHECK();

  const size_t lds = minimizer_lds_bytes(k, w);
  if (lds > 64 * 1024) {
    MM_HIP(hipFuncSetAttribute((const void*)minimizer_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MM_HIP(hipFuncSetAttribute((const void*)minimizer_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MM_HIP(hipFuncSetAttribute((const void*)minimizer_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  DBuf<uint32_t> tcount((size_t)ntiles);
  DBuf<uint64_t> tout((size_t)ntiles + 1), tmp;
  // hashing is the expensive part: do it once when a per-tile staging area (16 KiB per tile) is affordable
  bool single_pass = !want_rec_seq && (size_t)ntiles * MZ_STAGE * sizeof(Rec) <= ((size_t)6 << 30);
  DBuf<Rec> stage;
  DBuf<int> d_ovf(1); d_ovf.zero(st);
  if (single_pass) {
    stage.alloc((size_t)ntiles * MZ_STAGE);
    { const char* e = getenv("MM_MZ_DBG"); int v = e ? atoi(e) : 0; MM_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(mz_dbg_stop), &v, sizeof v, 0, hipMemcpyHostToDevice, st)); }   // timing aid, see mm_minimizer.hpp
    minimizer_kernel<2><<<dim3((unsigned)ntiles), dim3(MZ_THREADS), lds, st>>>(V, d_tf.p, k, w, d_js.p, tcount.p, nullptr, stage.p, nullptr, d_ovf.p);
  } else {
    minimizer_kernel<0><<<dim3((unsigned)ntiles), dim3(MZ_THREADS), lds, st>>>(V, d_tf.p, k, w, d_js.p, tcount.p, nullptr, nullptr, nullptr, nullptr);
  }
  MM_KERNEL_CHECK();
  exclusive_scan_u32_u64(tcount.p, ntiles, tout.p, tmp, st);
  uint64_t total = 0;
  int h_ovf = 0;
  MM_HIP(hipMemcpyAsync(&total, tout.p + ntiles, sizeof total, hipMemcpyDeviceToHost, st));
  MM_HIP(hipMemcpyAsync(&h_ovf, d_ovf.p, sizeof h_ovf, hipMemcpyDeviceToHost, st));
  MM_HIP(mm::stream_sync(st));
  if (h_ovf) single_pass = false;                                // some tile emitted more than MZ_STAGE records: counts are right, redo the write
  out.total = (int64_t)total;
  out.rec.alloc((size_t)total);
  if (want_rec_seq) out.rec_seq.alloc((size_t)total);
  if (total) {
    if (single_pass)
      compact_tiles_kernel<<<dim3((unsigned)ntiles), dim3(256), 0, st>>>(stage.p, tcount.p, tout.p, out.rec.p);
    else
      minimizer_kernel<1><<<dim3((unsigned)ntiles), dim3(MZ_THREADS), lds, st>>>(V, d_tf.p, k, w, d_js.p, nullptr, tout.p, out.rec.p,
                                                                               want_rec_seq ? out.rec_seq.p : nullptr, nullptr);
    MM_KERNEL_CHECK();
  }
  gather_offsets_kernel<<<dim3((unsigned)ceil_div(n + 1, 256)), dim3(256), 0, st>>>(d_tf.p, tout.p, n, out.off.p);
  MM_KERNEL_CHECK();
  out.off.download(out.h_off.data(), (size_t)n + 1, st);
  MM_HIP(mm::stream_sync(st));
}

// ---- host packing -----------------------------------------------------------------------------------
static inline int code_of(uint8_t c) {
  switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; }
  return -1;
}

void seqset_upload(mm_seqset* s) {
  MM_REQUIRE(!s->frozen, MM_ERR_STATE, "sequence set already uploaded");
  hipStream_t st = s->ctx->stream;
  const size_t n = s->staged.size();
  s->len.resize(n);
  s->base.assign(n + 1, 0);
  s->total_bases = 0;
  for (size_t i = 0; i < n; ++i) {
    MM_REQUIRE((int64_t)s->staged[i].second <= MAX_SEQ_LEN, MM_ERR_LIMIT, "sequence longer than 2^29-1 bases");
    s->len[i] = (int32_t)s->staged[i].second;
    s->base[i + 1] = s->base[i] + (((uint64_t)s->len[i] + 15) & ~15ull);
    s->total_bases += s->len[i];
  }
  const size_t nwords = (size_t)(s->base[n] >> 4);
  // the packed words are written straight into the context's pinned upload buffer (kept across calls: a fresh 256 MB vector per Gbase group
  // cost more in page faults and zero-filling than the packing itself, and pageable memory uploads at a fraction of the pinned rate)
  uint32_t* const words = (uint32_t*)s->ctx->pinned_up_at_least((nwords + 1) * sizeof(uint32_t));
  words[nwords] = 0;
  std::vector<uint64_t> es; std::vector<uint32_t> el; std::vector<uint8_t> eb;
  {
    // 16 bases per step through a byte table (code, or 0x80 for anything but ACGT after upper-casing, commonFunc.hpp:57-66);
    // only words with such a base take the per-base path.  Work items are pieces of at most 4 Mbases of one sequence (sequences start on
    // word boundaries and pieces on multiples of 16 bases, so items own disjoint words); the exception runs of the items are concatenated in
    // order, a run that crosses a piece boundary inside a sequence is joined again.
    static const struct Lut { uint8_t t[256]; Lut() { for (int c = 0; c < 256; ++c) { int u = (c > 96 && c < 123) ? c - 32 : c; int k = code_of((uint8_t)u); t[c] = k >= 0 ? (uint8_t)k : 0x80; } } } lut;
    struct Item { size_t seq; size_t j0, j1; };
    struct Runs { std::vector<uint64_t> es; std::vector<uint32_t> el; std::vector<uint8_t> eb; };
    constexpr size_t PIECE = (size_t)4 << 20;
    std::vector<Item> items;
    for (size_t i = 0; i < n; ++i) { const size_t L = s->staged[i].second; for (size_t j0 = 0; j0 < L; j0 += PIECE) items.push_back(Item{i, j0, std::min(L, j0 + PIECE)}); }
    std::vector<Runs> runs(items.size());
    const unsigned hw = mm::cpu_budget();                          // (not hardware_concurrency: cpu_budget.hpp)
    const size_t nthr = (size_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::max(1u, hw / 2), 32, (uint64_t)(s->total_bases >> 22) + 1, (uint64_t)std::max<size_t>(items.size(), 1)}));
    std::atomic<size_t> next_item{0};
    const bool simd = mm_host_has_avx2() != 0 && !getenv("MM_PACK_SCALAR");
    auto work = [&]() {
      for (;;) {
        const size_t it = next_item.fetch_add(1);
        if (it >= items.size()) return;
        const Item I = items[it];
        Runs& R = runs[it];
        const uint8_t* p = (const uint8_t*)s->staged[I.seq].first;
        uint32_t* wp = words + (s->base[I.seq] >> 4);
        const uint64_t b0 = s->base[I.seq];
        bool open = false;
        for (size_t j0 = I.j0; j0 < I.j1; j0 += 16) {
          if (simd && j0 + 32 <= I.j1) {                             // runs of plain ACGT, 32 bases per step (host_pack.cpp); a block with any other byte falls through
            const size_t done = mm_pack_acgt_blocks(p + j0, (I.j1 - j0) & ~(size_t)31, wp + (j0 >> 4));
            if (done) { open = false; j0 += done; if (j0 >= I.j1) break; }
          }
          const size_t m = std::min<size_t>(16, I.j1 - j0);
          uint32_t wv = 0; uint8_t bad = 0;
          for (size_t j = 0; j < m; ++j) { const uint8_t k = lut.t[p[j0 + j]]; bad |= k; wv |= (uint32_t)(k & 3) << (2 * j); }
          if (!(bad & 0x80)) { wp[j0 >> 4] = wv; open = false; continue; }
          wv = 0;
          for (size_t j = 0; j < m; ++j) {
            uint8_t c = p[j0 + j];
            if (c > 96 && c < 123) c -= 32;
            const uint8_t k = lut.t[c];
            if (!(k & 0x80)) { wv |= (uint32_t)k << (2 * j); open = false; }
            else if (open && R.eb.back() == c && R.el.back() < 0xFFFFFFFFu) R.el.back()++;
            else { R.es.push_back(b0 + j0 + j); R.el.push_back(1); R.eb.push_back(c); open = true; }
          }
          wp[j0 >> 4] = wv;
        }
      }
    };
    // (the context's own helpers, there from upload to upload: a context uploads one set at a time)
    if (nthr > 1) { if (!s->ctx->pack_pool) s->ctx->pack_pool = std::make_unique<TaskPool>(31); s->ctx->pack_pool->run(nthr, [&](size_t) { work(); }); }
    else work();
    for (size_t it = 0; it < items.size(); ++it) {
      Runs& R = runs[it];
      size_t from = 0;
      // (the sequential loop continues a run across what is a piece boundary here: same sequence, next base, same byte)
      if (it > 0 && items[it].seq == items[it - 1].seq && !R.es.empty() && !es.empty() && es.back() + el.back() == R.es[0] && eb.back() == R.eb[0] &&
          R.es[0] == s->base[items[it].seq] + items[it].j0 && (uint64_t)el.back() + R.el[0] <= 0xFFFFFFFFull) { el.back() += R.el[0]; from = 1; }
      es.insert(es.end(), R.es.begin() + (long)from, R.es.end()); el.insert(el.end(), R.el.begin() + (long)from, R.el.end()); eb.insert(eb.end(), R.eb.begin() + (long)from, R.eb.end());
    }
  }
  s->packed.alloc(nwords + 1);
  s->packed.upload(words, nwords + 1, st);
  s->d_base.alloc(n + 1); s->d_base.upload(s->base.data(), n + 1, st);
  s->d_len.alloc(std::max<size_t>(n, 1)); s->d_len.upload(s->len.data(), n, st);
  s->n_exc = (int64_t)es.size();
  if (s->n_exc) {
    s->exc_start.alloc(es.size()); s->exc_start.upload(es.data(), es.size(), st);
    s->exc_len.alloc(el.size()); s->exc_len.upload(el.data(), el.size(), st);
    s->exc_byte.alloc(eb.size()); s->exc_byte.upload(eb.data(), eb.size(), st);
  }
  MM_HIP(mm::stream_sync(st));
  s->staged.clear(); s->staged.shrink_to_fit(); s->owned.clear(); s->owned.shrink_to_fit();
  s->frozen = true;
}

// ---------------------------------------------------------------------------------------------------
// Persistent packed form (what `metamaps index` writes instead of the reference's Boost archives of the sketch,
// mapWrap.h:358-405): the 2-bit stream, the exception runs and the lengths.  The device index is rebuilt from it in
// seconds, so nothing derived is stored.  Little endian, versioned:
//   "MMSEQSET" u32 version=1 u32 reserved  i64 n  i64 total_bases  i64 n_words  i64 n_exc
//   i32 len[n]  u32 words[n_words]  u64 exc_start[n_exc]  u32 exc_len[n_exc]  u8 exc_byte[n_exc]
// ---------------------------------------------------------------------------------------------------
namespace {
struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
template <typename T> void write_all(FILE* f, const T* p, size_t n, const char* what) {
  MM_REQUIRE(n == 0 || fwrite(p, sizeof(T), n, f) == n, MM_ERR_ARG, std::string("short write (") + what + ")");
}
template <typename T> void read_all(FILE* f, T* p, size_t n, const char* what) {
  MM_REQUIRE(n == 0 || fread(p, sizeof(T), n, f) == n, MM_ERR_ARG, std::string("truncated sequence-set file (") + what + ")");
}
}  // namespace

void seqset_save(mm_seqset* s, const char* path) {
  MM_REQUIRE(s->frozen, MM_ERR_STATE, "sequence set not uploaded yet");
  hipStream_t st = s->ctx->stream;
  FileCloser fc{fopen(path, "wb")};
  MM_REQUIRE(fc.f != nullptr, MM_ERR_ARG, std::string("cannot open ") + path + " for writing");
  const int64_t n = s->count(), nwords = (int64_t)s->packed.n, nexc = s->n_exc;
  const char magic[8] = {'M', 'M', 'S', 'E', 'Q', 'S', 'E', 'T'};
  const uint32_t ver[2] = {1u, 0u};
  const int64_t hdr[4] = {n, s->total_bases, nwords, nexc};
  write_all(fc.f, magic, 8, "magic"); write_all(fc.f, ver, 2, "version"); write_all(fc.f, hdr, 4, "header");
  write_all(fc.f, s->len.data(), (size_t)n, "lengths");
  {
    std::vector<uint32_t> w = s->packed.to_host(st);
    write_all(fc.f, w.data(), (size_t)nwords, "bases");
  }
  if (nexc) {
    auto es = s->exc_start.to_host(st); auto el = s->exc_len.to_host(st); auto eb = s->exc_byte.to_host(st);
    write_all(fc.f, es.data(), (size_t)nexc, "exception starts"); write_all(fc.f, el.data(), (size_t)nexc, "exception lengths");
    write_all(fc.f, eb.data(), (size_t)nexc, "exception bytes");
  }
}

void seqset_load(mm_seqset* s, const char* path) {
  hipStream_t st = s->ctx->stream;
  FileCloser fc{fopen(path, "rb")};
  MM_REQUIRE(fc.f != nullptr, MM_ERR_ARG, std::string("cannot open ") + path);
  char magic[8]; uint32_t ver[2]; int64_t hdr[4];
  read_all(fc.f, magic, 8, "magic"); read_all(fc.f, ver, 2, "version"); read_all(fc.f, hdr, 4, "header");
  MM_REQUIRE(memcmp(magic, "MMSEQSET", 8) == 0 && ver[0] == 1u, MM_ERR_ARG, std::string(path) + " is not a sequence-set file of this version");
  const int64_t n = hdr[0], nwords = hdr[2], nexc = hdr[3];
  MM_REQUIRE(n >= 0 && nwords >= 1 && nexc >= 0, MM_ERR_ARG, "corrupt sequence-set header");
  s->len.resize((size_t)n);
  read_all(fc.f, s->len.data(), (size_t)n, "lengths");
  s->base.assign((size_t)n + 1, 0);
  s->total_bases = 0;
  for (int64_t i = 0; i < n; ++i) {
    MM_REQUIRE(s->len[(size_t)i] >= 0 && (int64_t)s->len[(size_t)i] <= MAX_SEQ_LEN, MM_ERR_ARG, "corrupt sequence length");
    s->base[(size_t)i + 1] = s->base[(size_t)i] + (((uint64_t)s->len[(size_t)i] + 15) & ~15ull);
    s->total_bases += s->len[(size_t)i];
  }
  MM_REQUIRE(s->total_bases == hdr[1] && (int64_t)(s->base[(size_t)n] >> 4) + 1 == nwords, MM_ERR_ARG, "sequence-set file is inconsistent");
  {
    std::vector<uint32_t> w((size_t)nwords);
    read_all(fc.f, w.data(), (size_t)nwords, "bases");
    s->packed.alloc((size_t)nwords); s->packed.upload(w.data(), (size_t)nwords, st);
    MM_HIP(mm::stream_sync(st));
  }
  s->d_base.alloc((size_t)n + 1); s->d_base.upload(s->base.data(), (size_t)n + 1, st);
  s->d_len.alloc(std::max<size_t>((size_t)n, 1)); s->d_len.upload(s->len.data(), (size_t)n, st);
  s->n_exc = nexc;
  if (nexc) {
    std::vector<uint64_t> es((size_t)nexc); std::vector<uint32_t> el((size_t)nexc); std::vector<uint8_t> eb((size_t)nexc);
    read_all(fc.f, es.data(), (size_t)nexc, "exception starts"); read_all(fc.f, el.data(), (size_t)nexc, "exception lengths");
    read_all(fc.f, eb.data(), (size_t)nexc, "exception bytes");
    s->exc_start.alloc(es.size()); s->exc_start.upload(es.data(), es.size(), st);
    s->exc_len.alloc(el.size()); s->exc_len.upload(el.data(), el.size(), st);
    s->exc_byte.alloc(eb.size()); s->exc_byte.upload(eb.data(), eb.size(), st);
    MM_HIP(mm::stream_sync(st));
  }
  MM_HIP(mm::stream_sync(st));
  s->frozen = true;
}

// ---------------------------------------------------------------------------------------------------
// Slices and concatenations of uploaded sets, on the device: every sequence starts on a word boundary of the packed stream, so a
// run of sequences is a run of words and only the stream coordinates (sequence starts, exception runs) shift.  The CLI packs the
// reference once, in bounded groups as the parser delivers the contigs (winSketch.hpp:242-252 streams contig by contig),
// concatenates the groups and cuts the index chunks of --maxmemory out of the resident whole.
// ---------------------------------------------------------------------------------------------------
static void exc_to_host(const mm_seqset* s, std::vector<uint64_t>& es, std::vector<uint32_t>& el, std::vector<uint8_t>& eb) {
  hipStream_t st = s->ctx->stream;
  if (s->n_exc) { es = s->exc_start.to_host(st, (size_t)s->n_exc); el = s->exc_len.to_host(st, (size_t)s->n_exc); eb = s->exc_byte.to_host(st, (size_t)s->n_exc); }
}
static void finish_derived(mm_seqset* o, const std::vector<uint64_t>& es, const std::vector<uint32_t>& el, const std::vector<uint8_t>& eb) {
  hipStream_t st = o->ctx->stream;
  const size_t n = o->len.size();
  o->d_base.alloc(n + 1); o->d_base.upload(o->base.data(), n + 1, st);
  o->d_len.alloc(std::max<size_t>(n, 1)); o->d_len.upload(o->len.data(), n, st);
  o->n_exc = (int64_t)es.size();
  if (o->n_exc) {
    o->exc_start.alloc(es.size()); o->exc_start.upload(es.data(), es.size(), st);
    o->exc_len.alloc(el.size()); o->exc_len.upload(el.data(), el.size(), st);
    o->exc_byte.alloc(eb.size()); o->exc_byte.upload(eb.data(), eb.size(), st);
  }
  MM_HIP(mm::stream_sync(st));
  o->frozen = true;
}
void seqset_slice(const mm_seqset* s, int64_t first, int64_t count, mm_seqset* o) {
  MM_REQUIRE(s->frozen, MM_ERR_STATE, "sequence set not uploaded");
  MM_REQUIRE(first >= 0 && count >= 0 && first + count <= s->count(), MM_ERR_ARG, "slice out of range");
  hipStream_t st = o->ctx->stream;
  const uint64_t b0 = s->base[(size_t)first], b1 = s->base[(size_t)(first + count)];
  o->len.assign(s->len.begin() + first, s->len.begin() + first + count);
  o->base.resize((size_t)count + 1);
  o->total_bases = 0;
  for (int64_t i = 0; i <= count; ++i) o->base[(size_t)i] = s->base[(size_t)(first + i)] - b0;
  for (int64_t i = 0; i < count; ++i) o->total_bases += o->len[(size_t)i];
  const size_t nw = (size_t)((b1 - b0) >> 4);
  o->packed.alloc(nw + 1);                                       // (+1: the pad word every set ends on)
  if (nw) MM_HIP(hipMemcpyAsync(o->packed.p, s->packed.p + (b0 >> 4), nw * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  MM_HIP(hipMemsetAsync(o->packed.p + nw, 0, sizeof(uint32_t), st));
  std::vector<uint64_t> es, oes; std::vector<uint32_t> el, oel; std::vector<uint8_t> eb, oeb;
  exc_to_host(s, es, el, eb);
  const size_t lo = (size_t)(std::lower_bound(es.begin(), es.end(), b0) - es.begin()), hi = (size_t)(std::lower_bound(es.begin(), es.end(), b1) - es.begin());
  for (size_t r = lo; r < hi; ++r) { oes.push_back(es[r] - b0); oel.push_back(el[r]); oeb.push_back(eb[r]); }   // (a run never crosses a sequence border)
  finish_derived(o, oes, oel, oeb);
}
void seqset_concat(const mm_seqset* const* parts, int n_parts, mm_seqset* o) {
  hipStream_t st = o->ctx->stream;
  uint64_t bases = 0; size_t nseq = 0;
  for (int p = 0; p < n_parts; ++p) {
    MM_REQUIRE(parts[p] && parts[p]->frozen, MM_ERR_STATE, "sequence set not uploaded");
    MM_REQUIRE(parts[p]->ctx->device == o->ctx->device, MM_ERR_ARG, "mm_seqset_concat: the parts live on another device");
    bases += parts[p]->base.back(); nseq += parts[p]->len.size();
  }
  o->len.clear(); o->len.reserve(nseq); o->base.clear(); o->base.reserve(nseq + 1);
  o->total_bases = 0;
  o->packed.alloc((size_t)(bases >> 4) + 1);
  std::vector<uint64_t> oes; std::vector<uint32_t> oel; std::vector<uint8_t> oeb;
  uint64_t b0 = 0;
  for (int p = 0; p < n_parts; ++p) {
    const mm_seqset* s = parts[p];
    const size_t nw = (size_t)(s->base.back() >> 4);
    if (nw) MM_HIP(hipMemcpyAsync(o->packed.p + (b0 >> 4), s->packed.p, nw * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    for (size_t i = 0; i < s->len.size(); ++i) { o->len.push_back(s->len[i]); o->base.push_back(b0 + s->base[i]); }
    o->total_bases += s->total_bases;
    std::vector<uint64_t> es; std::vector<uint32_t> el; std::vector<uint8_t> eb;
    exc_to_host(s, es, el, eb);
    for (size_t r = 0; r < es.size(); ++r) { oes.push_back(es[r] + b0); oel.push_back(el[r]); oeb.push_back(eb[r]); }
    b0 += s->base.back();
  }
  o->base.push_back(b0);
  MM_HIP(hipMemsetAsync(o->packed.p + (b0 >> 4), 0, sizeof(uint32_t), st));
  finish_derived(o, oes, oel, oeb);
}

void seqset_fetch(mm_seqset* s, int64_t i, char* out, int64_t cap) {
  MM_REQUIRE(s->frozen, MM_ERR_STATE, "sequence set not uploaded");
  MM_REQUIRE(i >= 0 && i < s->count(), MM_ERR_ARG, "sequence index out of range");
  const int64_t L = s->len[(size_t)i];
  MM_REQUIRE(cap >= L, MM_ERR_ARG, "output buffer too small");
  hipStream_t st = s->ctx->stream;
  const uint64_t b0 = s->base[(size_t)i];
  const size_t nw = (size_t)((L + 15) >> 4);
  std::vector<uint32_t> w(nw);
  if (nw) MM_HIP(hipMemcpyAsync(w.data(), s->packed.p + (b0 >> 4), nw * 4, hipMemcpyDeviceToHost, st));
  std::vector<uint64_t> es; std::vector<uint32_t> el; std::vector<uint8_t> eb;
  if (s->n_exc) { es = s->exc_start.to_host(st); el = s->exc_len.to_host(st); eb = s->exc_byte.to_host(st); }
  MM_HIP(mm::stream_sync(st));
  {                                                              // four bases per table look-up (bench.py writes 26.8 Gbases of FASTA through this)
    static const struct Lut { uint32_t t[256]; Lut() { for (int b = 0; b < 256; ++b) { uint32_t v = 0; for (int j = 0; j < 4; ++j) v |= (uint32_t)ascii_of_code((uint32_t)(b >> (2 * j)) & 3u) << (8 * j); t[b] = v; } } } lut;
    const int64_t full = L >> 4;
    for (int64_t q = 0; q < full; ++q) {
      const uint32_t x = w[(size_t)q];
      uint32_t v[4] = {lut.t[x & 255], lut.t[(x >> 8) & 255], lut.t[(x >> 16) & 255], lut.t[x >> 24]};
      memcpy(out + (q << 4), v, 16);
    }
    for (int64_t j = full << 4; j < L; ++j) out[j] = (char)ascii_of_code((w[(size_t)(j >> 4)] >> (2 * (j & 15))) & 3u);
  }
  for (size_t r = 0; r < es.size(); ++r) {
    if (es[r] + el[r] <= b0 || es[r] >= b0 + (uint64_t)L) continue;
    for (uint64_t g = std::max(es[r], b0); g < std::min(es[r] + el[r], b0 + (uint64_t)L); ++g) out[g - b0] = (char)eb[r];
  }
}

// sequences [first, first + count) as ASCII, one behind the other without separators (the caller has the lengths): one copy of the packed
// words, unpacked by several host threads (bench.py writes a million reads of FASTQ through this; one call per read cost 30 us)
void seqset_fetch_range(mm_seqset* s, int64_t first, int64_t count, char* out, int64_t cap) {
  MM_REQUIRE(s->frozen, MM_ERR_STATE, "sequence set not uploaded");
  MM_REQUIRE(first >= 0 && count >= 0 && first + count <= s->count(), MM_ERR_ARG, "sequence range out of bounds");
  if (count == 0) return;
  std::vector<int64_t> at((size_t)count + 1, 0);
  for (int64_t i = 0; i < count; ++i) at[(size_t)i + 1] = at[(size_t)i] + s->len[(size_t)(first + i)];
  MM_REQUIRE(cap >= at[(size_t)count], MM_ERR_ARG, "output buffer too small");
  hipStream_t st = s->ctx->stream;
  const uint64_t b0 = s->base[(size_t)first], b1 = s->base[(size_t)(first + count)];
  std::vector<uint32_t> w((size_t)((b1 - b0) >> 4) + 1);
  if (b1 > b0) MM_HIP(hipMemcpyAsync(w.data(), s->packed.p + (b0 >> 4), (size_t)((b1 - b0) >> 4) * 4, hipMemcpyDeviceToHost, st));
  std::vector<uint64_t> es; std::vector<uint32_t> el; std::vector<uint8_t> eb;
  if (s->n_exc) { es = s->exc_start.to_host(st); el = s->exc_len.to_host(st); eb = s->exc_byte.to_host(st); }
  MM_HIP(mm::stream_sync(st));
  static const struct Lut { uint32_t t[256]; Lut() { for (int b = 0; b < 256; ++b) { uint32_t v = 0; for (int j = 0; j < 4; ++j) v |= (uint32_t)ascii_of_code((uint32_t)(b >> (2 * j)) & 3u) << (8 * j); t[b] = v; } } } lut;
  const unsigned nthr = (unsigned)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::max(1u, mm::cpu_budget() / 2), 16, count}));
  std::atomic<int64_t> next{0};
  auto work = [&]() {
    for (;;) {
      const int64_t i0 = next.fetch_add(256);
      if (i0 >= count) return;
      for (int64_t i = i0; i < std::min(count, i0 + 256); ++i) {
        const int64_t L = s->len[(size_t)(first + i)];
        const uint32_t* const ws = w.data() + ((s->base[(size_t)(first + i)] - b0) >> 4);
        char* const o = out + at[(size_t)i];
        const int64_t full = L >> 4;
        for (int64_t q = 0; q < full; ++q) { const uint32_t x = ws[q]; const uint32_t v[4] = {lut.t[x & 255], lut.t[(x >> 8) & 255], lut.t[(x >> 16) & 255], lut.t[x >> 24]}; memcpy(o + (q << 4), v, 16); }
        for (int64_t j = full << 4; j < L; ++j) o[j] = (char)ascii_of_code((ws[j >> 4] >> (2 * (j & 15))) & 3u);
      }
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (!es.empty()) {                                             // exception runs (sorted by start): the sequence a run lies in by binary search over the bases
    for (size_t r = 0; r < es.size(); ++r) {
      if (es[r] + el[r] <= b0 || es[r] >= b1) continue;
      for (uint64_t g = std::max(es[r], b0); g < std::min<uint64_t>(es[r] + el[r], b1); ++g) {
        const size_t i = (size_t)(std::upper_bound(s->base.begin() + first, s->base.begin() + first + count + 1, g) - s->base.begin()) - 1;
        const uint64_t in = g - s->base[i];
        if (i < (size_t)(first + count) && in < (uint64_t)s->len[i]) out[at[i - (size_t)first] + (int64_t)in] = (char)eb[r];
      }
    }
  }
}

}  // namespace mm
