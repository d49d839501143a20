// `metamaps` — drop-in command line front end for the MI355X hot path.
//
// Keeps the reference's sub-commands, flags and on-disk formats (map/include/parseCmdArgs.hpp:33-117,
// map/mash_map.cpp:257-317; files: map/mapWrap.h:39-211, meta/fEM.h:663-803) and drives the device
// exclusively through the C ABI in include/metamaps_hip.h.  Host work here is what stays host work in the
// reference too: argument parsing, FASTA/FASTQ(.gz) reading, text formatting, taxonomy bookkeeping.
//
//   metamaps mapDirectly [--all] -r DB.fa -q reads.fq -o PREFIX [-k 16] [-w W] [-m 1000] [--pi 80] [-p 1e-3] [-t N] [--mm G] [--gpus N]
//   metamaps index -r DB.fa -i IDX [same reference options]          metamaps mapAgainstIndex [--all] -i IDX -q reads.fq -o PREFIX [--gpus N]
//   metamaps classify --DB DBDIR --mappings PREFIX [--minreads N] [-t N] [--gpus N]
//
// --gpus N uses devices 0..N-1 of the node, one context per device on its own host thread (where the reference has -t N worker
// threads, computeMap.hpp:104-176 / fEM.h:1229): mapping shards the read batches (index replicated) or the index chunks
// (--shard-index / --stream-chunks) over the devices and writes the output in input order; classify shards the reads and
// all-reduces the per-taxon EM sums over RCCL every iteration.  (--devices a,b,c names the devices explicitly; a device may
// repeat — several contexts on one GPU — which is how the multi-device paths are tested on a one-GPU box.)
//
// --mm G splits the reference into the same index chunks the reference would build under that limit
// (mm_index_plan_chunks); all chunk indexes stay resident in HBM and every read batch is mapped against each.
// (--maxmemory-bytes N gives the limit in bytes: a test hook, sub-GiB limits make small references chunk.)
// --stream-chunks (automatic when the chunk indexes cannot all be resident): one chunk index on the device at a time,
// every read batch (kept packed on the device) mapped against it, merge at the end — mapWrap.h:417-437 with HBM in
// place of the PREFIX.N files.  (--stream-range-bases N: test hook, size of the contig ranges the chunk rule is evaluated on.)
//
// `index` stores the packed reference per chunk (own versioned format, include/metamaps_hip.h: mm_seqset_save) instead of
// the reference's Boost archives of the sketch; the device index is rebuilt from it in seconds.  `index --full-index` stores the
// device index itself (IDX.N.mmidx, mm_index_save: the arrays as they lie in HBM) and mapAgainstIndex loads it without running a
// kernel — the persistent index of SURVEY N2; which of the two is faster is a question of file bandwidth against build time.
//
// Not provided: classifyU (disabled upstream).

#include "../mm_env.hpp"
#include "../cpu_budget.hpp"
#include "../../../include/metamaps_hip.h"
#include "seq_reader.hpp"
#include "host_util.hpp"
#include "id_set.hpp"
#include "fast_format.hpp"
#include "huge_new.hpp"
#include "../task_pool.hpp"
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <zlib.h>
#include <algorithm>
#include <functional>
#include <cctype>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <condition_variable>
#include <deque>
#include <map>
#include <unordered_map>
#include <cerrno>
#include <thread>
#include <mutex>
#include <memory>
#include <regex>
#include <set>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace {

// MM_CLI_TIMING=1: wall time per phase on stderr at exit
struct PhaseClock {
  std::map<std::string, double> acc; std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now(), t0 = t; std::mutex m;
  void add(const char* name, double seconds) { std::lock_guard<std::mutex> lk(m); acc[name] += seconds; }   // worker threads: summed over the workers
  void lap(const char* name) { auto n = std::chrono::steady_clock::now(); add(name, std::chrono::duration<double>(n - t).count()); t = n;
                               if (getenv("MM_CLI_TIMING")) std::cerr << "INFO, lap " << name << " at +" << std::chrono::duration<double>(n - t0).count() << " s\n"; }
  bool reported = false;
  void report() { if (reported) return; reported = true; if (getenv("MM_CLI_TIMING")) for (auto& kv : acc) std::cerr << "INFO, time " << kv.first << " " << kv.second << " s\n"; }
  ~PhaseClock() { report(); }
};

// Every output file is written and closed: leave without the orderly teardown.  Returning 150 GB of index to the driver allocation by
// allocation (hipFree) took 2.5 s of a 13 s run at miniSeq+H scale; the operating system reclaims the process' device memory as a
// whole.  (MM_CLI_FULL_TEARDOWN=1 keeps the orderly path: the tests of handle lifetimes under a leak checker use it.)
[[noreturn]] void finish_fast() { std::cout.flush(); std::cerr.flush(); fflush(nullptr); _exit(0); }

// An error exit leaves through _exit: helper threads (the HIP runtime coming up beside the parse of `classify`, the worker contexts' prewarm, the
// readers) may be inside the driver at this moment, and exit() would run static destructors and the runtime's atexit handlers under them.
[[noreturn]] void die(const std::string& m) { std::cerr << m << std::endl; std::cout.flush(); fflush(nullptr); _exit(1); }
// a helper thread that is joined on every way out of its scope (an exception that passes a joinable std::thread ends in std::terminate)
struct JoinOnExit { std::thread& t; ~JoinOnExit() { if (t.joinable()) t.join(); } };
void ck(mm_ctx* ctx, int st, const char* what) { if (st != MM_OK) die(std::string(what) + ": " + mm_last_error(ctx)); }

struct Options { std::map<std::string, std::string> v; bool all = false, stream = false, shard = false, em_host = false; };
Options parse(int argc, char** argv) {
  static const std::map<std::string, std::string> alias{{"-r", "reference"}, {"-q", "query"}, {"-o", "output"}, {"-k", "kmer"}, {"-p", "pval"},
      {"-w", "window"}, {"-m", "minReadLen"}, {"-t", "threads"}, {"--mm", "maxmemory"}, {"--pi", "perc_identity"}, {"-i", "index"}};
  Options o;
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--all") { o.all = true; continue; }
    if (a == "--stream-chunks") { o.stream = true; continue; }
    if (a == "--shard-index") { o.shard = true; continue; }
    if (a == "--em-host-reduce") { o.em_host = true; continue; }
    if (a == "--host-gather" || a == "--peer-gather" || a == "--full-index") { o.v[a.substr(2)] = "1"; continue; }
    if (a == "-h" || a == "--help") { std::cout << "see the header of metamaps_main.cpp / the reference's README\n"; exit(0); }
    std::string key = alias.count(a) ? alias.at(a) : (a.rfind("--", 0) == 0 ? a.substr(2) : "");
    if (key.empty() || i + 1 >= argc) die("Unknown or incomplete option " + a);
    o.v[key] = argv[++i];
  }
  return o;
}

uint64_t file_size(const std::string& f) {                       // commonFunc.hpp:211-231
  struct stat st; if (stat(f.c_str(), &st) != 0) die("Cannot open " + f + " for size determination.");
  return (uint64_t)st.st_size;
}

// ------------------------------------------------------------------------------------------------------
// mapDirectly, index and mapAgainstIndex share everything but where the reference comes from:
//   index            FASTA -> chunk plan -> PREFIX.N.seqset per chunk (+ PREFIX.index / .arguments / .contigs)   mapWrap.h:358-405
//   mapAgainstIndex  those files -> device indexes -> map                                                         mapWrap.h:443-554
//   mapDirectly      FASTA -> chunk plan -> device indexes -> map                                                 mapWrap.h:407-441
// The stored form is the packed reference, not the reference's Boost archive of the sketch: rebuilding the device
// index takes seconds and the file stays a third of the FASTA's size.
//
// Several GPUs (--gpus N; the reference's -t N worker pool, computeMap.hpp:104-176, becomes one context per device):
//   replicated   every device holds every chunk index; read batches go to whichever worker is free and the output is written
//                in batch order (= input order, all ThreadPool.hpp:13-17 guarantees).  No exchange between devices.
//   sharded      (--shard-index, or automatic when the chunk indexes fit the devices together but not one of them) chunk c
//                lives on device c mod N, every read batch visits every device, the records stay on the device that made them
//                and go to the batch's owner device — RCCL send / receive between physical devices (mm_mapping_gather), device-to-
//                device copies between logical devices of one GPU (mm_mapping_concat), through the host only with --host-gather —
//                for the merge in chunk order and the mapping qualities: what the reference does with its PREFIX.N files
//                (mapWrap.h:417-437, :128-145).
//   streamed     (--stream-chunks, or automatic when not even that fits) rounds of N chunks, one per device, built, mapped
//                against every (device-resident) read batch and dropped.
// A batch's sequences live back to back in one arena (huge pages when the system grants them) that is handed to the library by
// reference (mm_seqset_add_view) and recycled: no allocation, copy or page fault per read.
struct Batch {
  std::vector<std::string> names; std::vector<int> lens; std::vector<size_t> off;
  std::vector<const char*> view;                                 // per read: where the sequence lies in a mapped query file, or nullptr (then arena + off)
  char* arena = nullptr; size_t cap = 0, used = 0;
  size_t seq = 0, file = 0;
  ~Batch() { free(arena); }
  void reserve(size_t want) {
    if (want <= cap) return;
    const size_t HP = (size_t)2 << 20, ncap = (std::max(want, cap + cap / 2) + HP - 1) / HP * HP;
    char* na = (char*)aligned_alloc(HP, ncap);
    if (!na) die("out of host memory for the read batch");
    madvise(na, ncap, MADV_HUGEPAGE);
    if (used) memcpy(na, arena, used);
    free(arena); arena = na; cap = ncap;
  }
  void put(const std::string& q) { reserve(used + q.size() + 1); memcpy(arena + used, q.data(), q.size()); off.push_back(used); view.push_back(nullptr); used += q.size(); }
  void put_view(const char* p) { off.push_back(0); view.push_back(p); }
  const char* seq_of(size_t r) const { return view[r] ? view[r] : arena + off[r]; }
  void add(SeqFile& f) {                                         // the record `f` just returned
    names.push_back(f.name); lens.push_back((int)f.length());
    if (f.view) put_view(f.view); else put(f.seq);
  }
  void reset() { names.clear(); lens.clear(); off.clear(); view.clear(); used = 0; }
  int64_t bases() const { int64_t b = 0; for (int L : lens) b += L; return b; }
  void absorb(Batch& o) {                                        // o's reads behind this batch's (the block parser's small batches joined up to the batch limits)
    const size_t base = used;
    if (o.used) { reserve(used + o.used); memcpy(arena + used, o.arena, o.used); used += o.used; }
    for (size_t i = 0; i < o.names.size(); ++i) { names.push_back(std::move(o.names[i])); lens.push_back(o.lens[i]); view.push_back(o.view[i]); off.push_back(o.view[i] ? 0 : base + o.off[i]); }
  }
};

// one logical GPU: a context (stream + allocator) on a physical device, and the chunk indexes that live there
struct Dev { int phys = 0; mm_ctx* ctx = nullptr; std::vector<mm_index*> idx; };

template <typename F> void on_each(size_t n, F&& fn) {           // fn(i) for i < n, concurrently
  if (n == 1) { fn(0); return; }
  std::vector<std::thread> th;
  for (size_t i = 0; i < n; ++i) th.emplace_back([&fn, i] { fn(i); });
  for (auto& t : th) t.join();
}

void check_devices(const std::vector<int>& phys) {                // (the first HIP call of the process: the runtime comes up here)
  const int n = mm_device_count();
  if (n <= 0) die("No MI355X (gfx950) device available — this build has no CPU path");
  for (int p : phys) if (p < 0 || p >= n) die("device " + std::to_string(p) + " requested but only " + std::to_string(n) + " visible");
}

std::vector<int> device_list(const Options& o, bool check = true) {   // --gpus N: devices 0..N-1; --devices a,b,..: explicit (a device may repeat: test hook)
  std::vector<int> phys;
  if (o.v.count("devices")) for (auto& s : split(o.v.at("devices"), ",")) phys.push_back(std::stoi(s));
  else { const int g = o.v.count("gpus") ? std::stoi(o.v.at("gpus")) : 1; for (int i = 0; i < g; ++i) phys.push_back(i); }
  if (phys.empty()) die("--gpus must be at least 1");
  if (check) check_devices(phys);
  return phys;
}

// records of one batch -> the text of PREFIX (computeMap.hpp:565-581 + the two fields of mapWrap.h:311-320), reads in order
// fields 10 and 13 of a mapping line are functions of (conserved sketches, sketch size) alone: formatted once per pair and kept.  The table belongs
// to the CALLER (one per formatting slot of a worker thread) and lives as long as that thread: the pool threads of format_records are new with every
// batch, and a table that was theirs (thread_local) was rebuilt — 0.8 MB cleared, every pair formatted again — by every one of them for every batch:
// 19 ms per batch of 85 000 lines, the whole of a worker's "finish" time.
struct FormatCache {
  struct Pair { uint64_t key; char ids[16], corr[16]; uint8_t n_ids, n_corr; double ident; };   // ident: the printed identity read back / 100 (what classify parses, fEM.h:264)
  static constexpr size_t CB = 1 << 14;
  std::vector<Pair> slots; int k = -1;
  void prepare(int k_now) { if (slots.size() != CB || k != k_now) { slots.assign(CB, Pair{~0ull, {0}, {0}, 0, 0, 0.0}); k = k_now; } }
};
// A mapping line as `classify` sees it once it has tokenised the file (fEM.h:234-275): where the line lies in the text, and the values of the fields it reads
// — identity and mapping quality as the PRINTED text parses, not as the floats they were printed from.  `mapDirectly --then-classify` keeps these beside the
// text it writes, so that classify in the same process neither reads the file back nor tokenises it.
struct LineMeta { uint32_t beg, ls /* the blank before field 14, relative to beg */, n /* length without the newline */; int32_t contig /* index into the reference's contigs */, len, start; double ident, mapq; };
static double mapq_as_classify_reads_it(const char* p, size_t n) {
  double v;
  if (parse_g6_text(p, n, &v)) return v;
  const std::string t(p, n);
  errno = 0; v = strtod(t.c_str(), nullptr);
  if (errno == ERANGE) v = t.find("e-") != std::string::npos ? 0.0 : v;   // (std::stod throws on a denormal; the reference then takes 0, fEM.h:269-275 — an overflow cannot be printed by this program)
  return v;
}
static void format_range(const std::vector<std::string>& names, const std::vector<int>& lens, const std::vector<int64_t>& off,
                         const std::vector<mm_map_record>& rec, const std::vector<std::string>& cname, const std::vector<int>& clen, int k, size_t r0, size_t r1, std::string& out,
                         FormatCache& fc, std::vector<LineMeta>* meta) {
  out.clear();
  if (meta) { meta->clear(); meta->reserve((size_t)(off[r1] - off[r0])); }
  out.reserve((size_t)(off[r1] - off[r0]) * 160);
  // no printf anywhere on the line (fast_format.hpp) — 4.2 M lines took 2 s of the mapping phase of a million reads
  using Pair = FormatCache::Pair;
  fc.prepare(k);
  std::vector<Pair>& cache = fc.slots;
  std::string tmp;
  for (size_t r = r0; r < r1; ++r) {
    const int len = lens[r];
    for (int64_t i = off[r]; i < off[r + 1]; ++i) {
      const mm_map_record& x = rec[(size_t)i];
      const uint64_t key = (uint64_t)(uint32_t)x.sketch << 32 | (uint32_t)x.shared;
      Pair& P = cache[(size_t)((key * 0x9E3779B97F4A7C15ull) >> 50)];
      if (P.key != key) {
        float id; mm_identity(x.shared, x.sketch, k, &id, nullptr);
        tmp.clear(); append_g6(tmp, (double)id);                   // operator<<(float): %g with 6 significant digits; printed, then re-parsed (mapWrap.h:237)
        P.n_ids = (uint8_t)tmp.size(); memcpy(P.ids, tmp.data(), tmp.size());
        const double reported = strtod(tmp.c_str(), nullptr) / 100.0;
        P.ident = reported;
        const float corrected = std::exp(-(1 - reported));        // mapWrap.h:311
        tmp.clear(); append_g6(tmp, (double)(corrected * 100));
        P.n_corr = (uint8_t)tmp.size(); memcpy(P.corr, tmp.data(), tmp.size());
        P.key = key;
      }
      const size_t line_beg = out.size();
      out += names[r];
      out += ' '; append_int(out, len); out += " 0 "; append_int(out, len - 1); out += ' '; out += x.strand == 1 ? '+' : '-'; out += ' ';
      out += cname[(size_t)x.ref_contig];
      out += ' '; append_int(out, clen[(size_t)x.ref_contig]);
      out += ' '; append_int(out, x.ref_start); out += ' '; append_int(out, (long long)x.ref_start + len - 1);
      out += ' '; out.append(P.ids, P.n_ids);
      out += ' '; append_int(out, x.shared); out += ' '; append_int(out, x.sketch);
      out += ' '; out.append(P.corr, P.n_corr);
      const size_t ls = out.size();
      out += ' '; append_g6(out, x.mapq);                          // :318-320
      if (meta) meta->push_back(LineMeta{(uint32_t)line_beg, (uint32_t)(ls - line_beg), (uint32_t)(out.size() - line_beg), (int32_t)x.ref_contig, (int32_t)len, (int32_t)x.ref_start,
                                         P.ident, mapq_as_classify_reads_it(out.data() + ls + 1, out.size() - ls - 1)});
      out += '\n';
    }
  }
}
// the mapping lines of a batch (mapWrap.h:300-323): ranges of reads formatted by a few threads, joined in read order
void format_records(const std::vector<std::string>& names, const std::vector<int>& lens, const std::vector<int64_t>& off,
                    const std::vector<mm_map_record>& rec, const std::vector<std::string>& cname, const std::vector<int>& clen, int k, std::string& out, std::vector<LineMeta>* meta) {
  const size_t n = names.size();
  static const size_t per_part = getenv("MM_CLI_FORMAT_PART") ? (size_t)std::max(1, atoi(getenv("MM_CLI_FORMAT_PART"))) : 10000;   // (tests: several parts for small batches too)
  const size_t T = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)std::max(per_part < 10000 ? 8u : 1u, mm::cpu_budget() / 4), rec.size() / per_part + 1}));   // (a quarter of the CPU budget per worker: four workers rarely format at the same moment)
  static thread_local std::vector<FormatCache> caches(8);          // (the calling thread's: a worker of mapDirectly formats batch after batch)
  if (T == 1) { format_range(names, lens, off, rec, cname, clen, k, 0, n, out, caches[0], meta); return; }
  std::vector<size_t> cut(T + 1, n);
  cut[0] = 0;
  { size_t t = 1; for (size_t r = 0; r < n && t < T; ++r) if ((uint64_t)off[r] >= (uint64_t)rec.size() * t / T) cut[t++] = r; }
  static thread_local std::vector<std::string> part_store(8);      // (kept with their capacity: fresh text buffers are page faults, batch after batch)
  std::vector<std::string>& part = part_store;
  FormatCache* const fcs = caches.data();
  static thread_local std::vector<std::vector<LineMeta>> meta_store(8);
  std::vector<std::vector<LineMeta>>& metas = meta_store;        // (the CALLING thread's: the helpers below must not name the thread_local themselves)
  const auto q0 = std::chrono::steady_clock::now();
  std::vector<double> took(T, 0.0);
  auto timed = [&](size_t t) { const auto a = std::chrono::steady_clock::now(); format_range(names, lens, off, rec, cname, clen, k, cut[t], cut[t + 1], part[t], fcs[t], meta ? &metas[t] : nullptr);
                               took[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
  static thread_local TaskPool helpers(7);                         // (task_pool.hpp: the calling worker's own helpers, there from batch to batch)
  const auto q1 = q0;
  helpers.run(T, timed);
  const auto q2 = std::chrono::steady_clock::now();
  size_t total = 0; for (size_t t = 0; t < T; ++t) total += part[t].size();
  out.clear(); out.reserve(total);
  if (meta) { meta->clear(); meta->reserve(rec.size()); }
  for (size_t t = 0; t < T; ++t) {
    if (meta) for (LineMeta lm : metas[t]) { lm.beg += (uint32_t)out.size(); meta->push_back(lm); }
    out += part[t];
  }
  if (getenv("MM_CLI_FORMAT_TRACE")) {
    const auto q3 = std::chrono::steady_clock::now();
    double mx = 0; for (double x : took) mx = std::max(mx, x);
    fprintf(stderr, "FORMAT_TRACE %zu records, %zu threads: all parts %.2f ms (own part %.2f ms, slowest part %.2f ms), join text %.2f ms\n", rec.size(), T,
            std::chrono::duration<double, std::milli>(q2 - q1).count(), took[0] * 1e3, mx * 1e3, std::chrono::duration<double, std::milli>(q3 - q2).count());
  }
}

// (defined behind map_mode; `mapDirectly --then-classify DBDIR` runs it in-process on the files it has just written)
enum class EmReduce { None, Rccl, Host };
struct KeptLines {                                               // the mapping lines of one output prefix as mapDirectly wrote them, batch after batch, with their parsed fields
  struct Part { const char* text; const LineMeta* meta; size_t n_lines; const int64_t* off; size_t n_reads; };
  std::vector<Part> parts; const std::vector<std::string>* cname = nullptr;
};
int classify_one(const std::vector<Dev>& devs, EmReduce reduce, const std::string& mapped, const std::string& db, size_t minReadsU,
                 const std::function<void()>& leave_now, const std::function<void()>& need_devices, const KeptLines* kept = nullptr);

// One run of mapDirectly / index / mapAgainstIndex.  The state every stage shares lives in the object; the stages are its methods, in the order run()
// calls them: parameters -> devices -> reference (parsed, packed, uploaded) or stored index -> chunk plan -> placement of the chunk indexes
// (replicated / sharded / streamed) -> read batches through the worker pipeline (replicated) or chunk-major rounds with the exchange of the
// records (sharded / streamed) -> writers -> optionally classify in-process.  (Until round 5 this was one 800-line function.)
struct MapRun {
  const Options& o; const std::string mode;
  const bool from_index, only_index;
  std::string ref; uint64_t refSize = 0, maxMem = 0; int k = 16, w = 0, minLen = 1000; double pval = 1e-3; float pi = 80;
  std::string ipre;
  std::vector<std::string> queries, prefixes;
  PhaseClock pc;
  std::vector<Dev> devs; size_t G = 0; mm_ctx* ctx0 = nullptr;
  std::vector<std::string> cname; std::vector<int> clen;
  struct Chunk { int first, count; std::string file; };
  std::vector<Chunk> chunks;
  // The packed reference (2 bits per base + exception runs, a quarter of the FASTA's size) lives on every device that builds indexes
  // from it; the host holds contig names and lengths only.  Index chunks are cut out of it on the device (mm_seqset_slice).
  std::vector<mm_seqset*> refset;
  uint64_t hbm_free = 0;
  int64_t BATCH_READS = 100000, BATCH_BASES = 256000000LL;
  size_t WPD = 4;                                               // worker contexts per device (replicated mode)
  std::vector<mm_ctx*> wctx;
  uint64_t ref_bases = 0;
  mm_index* whole = nullptr;                                     // index of the whole reference on device 0, when one was built for the chunk plan
  size_t NC = 0;
  enum class Place { Replicated, Sharded, Streamed } place = Place::Replicated;
  std::vector<int> thr_of;
  std::map<int64_t, int64_t> thr_acc; int thr = INT_MAX;          // occurrence histogram accumulated over the chunks, never cleared (winSketch.hpp:452-494)
  mm_map_params mp{};
  std::vector<int32_t> chunk_base;
  // what a worker hands to the writer: the finished text of one batch
  struct Done { size_t file = 0; std::vector<std::string> names; std::vector<int> lens; std::vector<int64_t> off; std::string text; std::vector<LineMeta> meta; double t_mapq = 0, t_fetch = 0, t_format = 0; };
  // --then-classify: the batches of every query file as they were written, in order (text + the parsed fields of every line): what classify takes instead of the file
  const bool keep_lines = o.v.count("then-classify") && !getenv("MM_CLI_CLASSIFY_FROM_FILE");
  std::vector<std::vector<std::unique_ptr<Done>>> kept;
  // the writer: batches in input order -> PREFIX, .meta.unmappedReadsLengths, .meta, .parameters of every query file (mapWrap.h:34-213)
  struct Writer {
    std::mutex m; std::condition_variable cv; std::map<size_t, std::unique_ptr<Done>> ready;
    void put(size_t seq, std::unique_ptr<Done> d) { std::lock_guard<std::mutex> lk(m); ready[seq] = std::move(d); cv.notify_all(); }
  } writer;
  // a reader thread parses the query files into batches (bounded queue); `take` hands them out in order, nullptr at the end
  struct Reader {
    std::mutex m; std::condition_variable cv; std::deque<std::unique_ptr<Batch>> queue, spare; bool done = false, started = false; size_t max_queued = 2;
    std::vector<size_t> file_end;                                // file_end[f] = number of batches of files 0..f (set when file f has been read to its end)
    std::thread th;
    std::unique_ptr<Batch> take() {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return !queue.empty() || done; });
      if (queue.empty()) return nullptr;
      auto b = std::move(queue.front()); queue.pop_front();
      cv.notify_all();
      return b;
    }
    void recycle(std::unique_ptr<Batch> b) { b->reset(); std::lock_guard<std::mutex> lk(m); spare.push_back(std::move(b)); }
    ~Reader() { if (th.joinable()) th.join(); }
  } reader;
  std::deque<MappedFile> mapped;                                 // query files whose sequences the batches point into: alive until the end
  std::thread prewarm;                                           // (declared last: joined first)

  MapRun(const Options& o_, const std::string& mode_) : o(o_), mode(mode_), from_index(mode_ == "mapAgainstIndex"), only_index(mode_ == "index") {}
  ~MapRun() { if (prewarm.joinable()) prewarm.join(); }

  void read_parameters() {
    if (!from_index && !o.v.count("reference")) die("Provide reference file (s)");
    if ((from_index || only_index) && !o.v.count("index")) die("Please provide index");
    if (!only_index && !o.v.count("query")) die("Provide query file (s)");
    if (!only_index && !o.v.count("output")) die("Provide output file");
    ipre = o.v.count("index") ? o.v.at("index") : "";
    if (!from_index) {
      ref = o.v.at("reference");
      refSize = file_size(ref);
      maxMem = o.v.count("maxmemory") ? (uint64_t)(std::pow(1024, 3) * std::stoull(o.v.at("maxmemory"))) : 0;
      if (o.v.count("maxmemory-bytes")) maxMem = std::stoull(o.v.at("maxmemory-bytes"));
      k = o.v.count("kmer") ? std::stoi(o.v.at("kmer")) : 16;
      pval = o.v.count("pval") ? std::stod(o.v.at("pval")) : 1e-3;
      minLen = o.v.count("minReadLen") ? std::stoi(o.v.at("minReadLen")) : 1000;
      pi = o.v.count("perc_identity") ? std::stof(o.v.at("perc_identity")) : 80;
      if (o.v.count("window")) {                                   // parseCmdArgs.hpp:363-374
        w = std::stoi(o.v.at("window"));
        pval = mm_estimate_pvalue(minLen * 2 / w, k, pi, minLen, refSize);
      } else w = mm_recommended_window(pval, k, pi, minLen, refSize);
    } else {                                                       // the parameters travel with the index (mapWrap.h:447-461)
      std::ifstream a(ipre + ".arguments");
      if (!a.is_open()) die("Cannot open file " + ipre + ".arguments for deserialization.");
      std::string key, val; std::map<std::string, std::string> kv;
      while (a >> key && std::getline(a, val)) { while (!val.empty() && val[0] == ' ') val.erase(0, 1); kv[key] = val; }
      for (const char* need : {"kmerSize", "windowSize", "minReadLength", "percentageIdentity", "p_value", "referenceSize", "maximumMemory", "reference"})
        if (!kv.count(need)) die("Index " + ipre + " is incomplete (" + need + " missing in .arguments)");
      k = std::stoi(kv["kmerSize"]); w = std::stoi(kv["windowSize"]); minLen = std::stoi(kv["minReadLength"]); pi = std::stof(kv["percentageIdentity"]);
      pval = std::stod(kv["p_value"]); refSize = std::stoull(kv["referenceSize"]); maxMem = std::stoull(kv["maximumMemory"]); ref = kv["reference"];
    }
    if (!only_index) {
      queries = split(o.v.at("query"), ","); prefixes = split(o.v.at("output"), ",");
      if (queries.size() != prefixes.size()) die("Please specify an equal number of input and output files (as comma-separated lists)");
    }
  }

  void open_devices() {
    for (int p : device_list(o)) { Dev d; d.phys = p; devs.push_back(d); }
    if (only_index) devs.resize(1);
    G = devs.size();
    for (auto& d : devs) if (mm_ctx_create(d.phys, &d.ctx) != MM_OK) die("No MI355X (gfx950) device available — this build has no CPU path");
    ctx0 = devs[0].ctx;
    pc.lap("0 context");
    refset.assign(G, nullptr);
    query_free();
    // ---- reads (computeMap.hpp:104-172 + unifyFiles mapWrap.h:34-213)
    // ~0.25 Gbp per device batch (16 ms of mapping); the next ones are parsed meanwhile.  (MM_CLI_BATCH_READS: test hook, small batches)
    if (getenv("MM_CLI_BATCH_READS")) BATCH_READS = std::max(1, atoi(getenv("MM_CLI_BATCH_READS")));
    if (getenv("MM_CLI_BATCH_MBASES")) BATCH_BASES = (int64_t)std::max(1, atoi(getenv("MM_CLI_BATCH_MBASES"))) * 1000000LL;
    reader.max_queued = std::max<size_t>(2, 2 * G);
    // worker contexts of the replicated mode (WPD per device, --workers-per-gpu).  The ones beside the device's first context come up while the
    // index is built, each with its upload staging in place (a batch-sized dummy goes through mm_seqset_upload once: pinned buffer, device
    // block): the first batch of a worker used to spend 40-60 ms there, and 38 ms creating its stream, with the device idle.
    WPD = o.v.count("workers-per-gpu") ? (size_t)std::max(1, std::stoi(o.v.at("workers-per-gpu")))
        : getenv("MM_CLI_WORKERS") ? (size_t)std::max(1, atoi(getenv("MM_CLI_WORKERS"))) : 4;
    wctx.assign(G * WPD, nullptr);
  }

  void query_free() {
    char nm[8]; int cus; uint64_t tot; mm_ctx_device_info(ctx0, nm, sizeof nm, &cus, &tot, &hbm_free);
    size_t share = 0; for (auto& d : devs) share += d.phys == devs[0].phys;   // logical devices of one physical device (--devices 0,0,..) share its memory
    hbm_free /= std::max<size_t>(share, 1);
  }

  // Resident bytes of the index of `bases` reference bases: N = 2 bases / (w + 1) entries; U distinct hashes — minimizer
  // hashes are window minima, so they crowd into the low end of the 32-bit space: measured 5.92e8 distinct among 5.94e9 entries at w = 8,
  // i.e. an effective space of H = 1.3 * 2^32 / (w + 1) values that fills as U = H (1 - exp(-N / H)); pos 8 N + occurrence lists padded to
  // 64-byte sectors 8 (N + 7 U) at most + a quarter of that in bin codes + 29 U of table.  Per base this FALLS with the size of the
  // reference: 6 bytes at 26.8 Gbp, 22 at 1 Gbp, where nearly every hash is a list of one padded to eight (a flat 5.5 bytes per base,
  // rounds 1-3, let a 0.5 Gbp planning range ask for 5.5 GiB on a device with 2 GiB left — found with MM_DEVICE_BYTES_CAP).  The build
  // holds another 12 N of sort buffers at its peak.
  double index_bytes(uint64_t bases, bool peak) const {
    const double N = 2.0 * (double)bases / (double)(w + 1), H = 1.3 * 4294967296.0 / (double)(w + 1), U = H * (1 - std::exp(-N / H));
    return 18.0 * N + 99.0 * U + (peak ? 12.0 * N : 0.0);
  }
  // `share` of the index of `bases` bases fits beside what the device already holds (the estimate errs on the large side by ~10 %)
  bool fits(uint64_t bases, double share) const { return index_bytes(bases, share >= 1.0) * share <= 0.8 * (double)hbm_free; }
  mm_seqset* make_part(size_t d, int a, int bnd) {               // contigs [a, bnd) of the reference as a set of their own, on device d
    mm_seqset* part; ck(devs[d].ctx, mm_seqset_slice(devs[d].ctx, refset[d], a, bnd - a, &part), "reference chunk");
    return part;
  }
  void drop_refsets() { for (auto*& r : refset) if (r) { mm_seqset_destroy(r); r = nullptr; } }

  // (started as soon as the reference has been parsed: the first batches are ready when the index is)
  void start_reader() { if (reader.th.joinable() || reader.started) return; reader.started = true; reader.th = std::thread([this]() { reader_main(); }); }
  // the reader thread: every query file in turn -> batches in the bounded queue
  void reader_main() {
    size_t seq = 0;
    const auto r_t0 = std::chrono::steady_clock::now();           // the reader's own rate (MM_CLI_TIMING): its wall time without what it waited for a free queue slot
    double r_waited = 0;
    auto fresh = [&]() {
      std::unique_ptr<Batch> b;
      { std::lock_guard<std::mutex> lk(reader.m); if (!reader.spare.empty()) { b = std::move(reader.spare.back()); reader.spare.pop_back(); } }
      if (!b) b = std::make_unique<Batch>();
      return b;
    };
    auto enqueue = [&](std::unique_ptr<Batch> b, size_t fi) {
      b->seq = seq++; b->file = fi;
      if (getenv("MM_CLI_TIMING")) std::cerr << "INFO, reader: batch of " << b->names.size() << " reads parsed at +" << std::chrono::duration<double>(std::chrono::steady_clock::now() - pc.t0).count() << " s\n";
      std::unique_lock<std::mutex> lk(reader.m);
      const auto w0 = std::chrono::steady_clock::now();
      reader.cv.wait(lk, [&] { return reader.queue.size() < reader.max_queued; });
      r_waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
      reader.queue.push_back(std::move(b));
      reader.cv.notify_all();
    };
    // records of `f` while they start before `stop` (memory mode; (size_t)-1: all) -> batches handed to `emit`; false if the reader
    // gave up on the file (a truncated quality string ends the file for kseq, kseq.h:204)
    auto parse_into = [&](SeqFile& f, size_t stop, const std::function<void(std::unique_ptr<Batch>)>& emit) -> bool {
      bool more = true, gave_up = false;
      while (more) {
        std::unique_ptr<Batch> b = fresh();
        int64_t bases = 0;
        while ((int64_t)b->names.size() < BATCH_READS && bases < BATCH_BASES) {
          if (stop != (size_t)-1) { const size_t ps = f.peek_start(); if (ps == (size_t)-1 || ps >= stop) { more = false; break; } }
          if (!(more = f.next())) { gave_up = stop != (size_t)-1; break; }
          if (b->names.empty() && !f.view) b->reserve((size_t)std::min<int64_t>(BATCH_BASES, (int64_t)f.length() * BATCH_READS) + ((size_t)64 << 20));
          bases += (int64_t)f.length();
          b->add(f);
        }
        if (b->names.empty()) { std::lock_guard<std::mutex> lk(reader.m); reader.spare.push_back(std::move(b)); break; }
        emit(std::move(b));
      }
      return !gave_up;
    };
    for (size_t fi = 0; fi < queries.size(); ++fi) {
      mapped.emplace_back();
      MappedFile& mf = mapped.back();
      if (getenv("MM_CLI_NO_MMAP") || !mf.open(queries[fi])) {   // gzip, pipes, ...: the sequential reader
        mapped.pop_back();
        SeqFile f(queries[fi]);
        parse_into(f, (size_t)-1, [&](std::unique_ptr<Batch> b) { enqueue(std::move(b), fi); });
      } else {
        // blocks of the mapped file, parsed by several threads, handed on in file order; a block's batches only go out once the
        // block before it has been seen to end exactly where this one starts
        const size_t blk = getenv("MM_CLI_BLOCK_BYTES") ? (size_t)std::max(1, atoi(getenv("MM_CLI_BLOCK_BYTES"))) : (size_t)128 << 20;
        const size_t nb = std::max<size_t>(1, (mf.size + blk - 1) / blk);
        std::vector<size_t> start(nb + 1, mf.size);
        start[0] = 0;
        struct Block { std::vector<std::unique_ptr<Batch>> out; size_t next = 0; bool done = false, empty = false, over = false; };   // next: first record start behind the block's records
        std::vector<Block> blocks(nb);
        std::mutex bm; std::condition_variable bcv; size_t next_block = 0, consumed = 0; bool abandon = false;
        const unsigned P = (unsigned)std::max<size_t>(1, std::min<size_t>({nb, (size_t)8, (size_t)std::max(1u, mm::cpu_budget() / 2)}));
        auto worker = [&]() {
          for (;;) {
            size_t j;
            {
              std::unique_lock<std::mutex> lk(bm);
              bcv.wait(lk, [&] { return abandon || next_block >= nb || next_block < consumed + P + 2; });   // not too far ahead of the consumer
              if (abandon || next_block >= nb) return;
              j = next_block++;
            }
            const auto b_t0 = std::chrono::steady_clock::now();
            if (j > 0) start[j] = mf.sync(j * blk, std::min(mf.size, (j + 1) * blk));   // (only this thread writes start[j]; read after `done`)
            Block& B = blocks[j];
            const size_t lim = std::min(mf.size, (j + 1) * blk);
            if (j == 0 || start[j] < lim) {                        // records that start in [start[j], lim)
              SeqFile f(mf.data, j == 0 ? 0 : start[j], mf.size);
              B.over = !parse_into(f, lim, [&](std::unique_ptr<Batch> b) { B.out.push_back(std::move(b)); });
              B.next = f.peek_start();
            } else B.empty = true;                                 // no record start was recognised in this block
            pc.add("R parse threads busy (summed over the block parser's threads)", std::chrono::duration<double>(std::chrono::steady_clock::now() - b_t0).count());
            { std::lock_guard<std::mutex> lk(bm); B.done = true; }
            bcv.notify_all();
          }
        };
        pc.add("R parse threads", (double)P);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < P; ++t) pool.emplace_back(worker);
        // `expect`: where the parse stands = the start of the first record not handed on yet.  A block continues the parse iff it
        // starts exactly there (block 0 starts at the file's first record by construction).
        size_t expect = 0; bool chain_ok = true, file_over = false;
        std::unique_ptr<Batch> pend; int64_t pend_bases = 0;
        for (size_t j = 0; j < nb && chain_ok && !file_over; ++j) {
          { std::unique_lock<std::mutex> lk(bm); bcv.wait(lk, [&] { return blocks[j].done; }); }
          Block& B = blocks[j];
          if (!B.empty) {
            if (j > 0 && start[j] != expect) { chain_ok = false; break; }
            for (auto& b : B.out) {                                 // a block ends where its 128 MB end, not where a batch is full: its last batch goes on in the next block
              if (pend && ((int64_t)(pend->names.size() + b->names.size()) > BATCH_READS || pend_bases + b->bases() > BATCH_BASES)) { enqueue(std::move(pend), fi); pend_bases = 0; }
              if (!pend) { pend_bases = b->bases(); pend = std::move(b); }
              else { pend_bases += b->bases(); pend->absorb(*b); reader.recycle(std::move(b)); }
            }
            B.out.clear();
            if (B.over || B.next == (size_t)-1) { file_over = true; break; }
            expect = B.next;
          } else if (expect < std::min(mf.size, (j + 1) * blk)) { chain_ok = false; break; }   // a record starts in this block, but none was recognised
          { std::lock_guard<std::mutex> lk(bm); consumed = j + 1; } bcv.notify_all();
        }
        { std::lock_guard<std::mutex> lk(bm); abandon = true; } bcv.notify_all();
        for (auto& t : pool) t.join();
        if (pend) enqueue(std::move(pend), fi);
        if (!chain_ok) {                                           // a block did not start where the parse stood: the rest sequentially, from there
          for (auto& B : blocks) for (auto& b : B.out) reader.recycle(std::move(b));
          SeqFile f(mf.data, expect, mf.size);
          parse_into(f, (size_t)-1, [&](std::unique_ptr<Batch> b) { enqueue(std::move(b), fi); });
        }
      }
      std::lock_guard<std::mutex> lk(reader.m); reader.file_end.push_back(seq); reader.cv.notify_all();
    }
    { const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - r_t0).count();
      pc.add("R reader thread wall time without waiting for a queue slot", wall - r_waited); pc.add("R reader waited for a queue slot", r_waited); }
    std::lock_guard<std::mutex> lk(reader.m); reader.done = true; reader.cv.notify_all();
  }

  void start_prewarm() {
    if (prewarm.joinable() || getenv("MM_CLI_NO_PREWARM")) return;
    int64_t query_bytes = 0; for (auto& q : queries) query_bytes += (int64_t)file_size(q);
    const int64_t warm_bases = std::min<int64_t>(BATCH_BASES, query_bytes / 2);   // (a FASTQ is two bytes per base; small inputs get small staging)
    prewarm = std::thread([&, warm_bases]() {
      static const std::string dummy((size_t)1 << 20, 'A');
      std::vector<std::thread> th;
      for (size_t d = 0; d < G; ++d) for (size_t wi = 1; wi < WPD; ++wi) th.emplace_back([&, d, wi]() {
        mm_ctx* c = nullptr;
        if (mm_ctx_create(devs[d].phys, &c) != MM_OK) die("cannot create a worker context");
        mm_seqset* sq = nullptr;
        if (mm_seqset_create(c, &sq) == MM_OK) {                  // (a failure here only means the first batch pays for its staging itself)
          bool ok = true;
          for (int64_t b = 0; ok && b < warm_bases; b += (int64_t)dummy.size()) ok = mm_seqset_add_view(sq, dummy.data(), (int64_t)dummy.size()) == MM_OK;
          if (ok) (void)mm_seqset_upload(sq);
          mm_seqset_destroy(sq);
        }
        wctx[d * WPD + wi] = c;
      });
      for (auto& t : th) t.join();
    });
  }

  // ---- reference (winSketch.hpp:180-365): parsed, packed and uploaded to every device that builds indexes from it
  void load_reference() {
    struct Group { std::deque<std::string> seq; std::vector<std::string> names; uint64_t bases = 0; };
    const uint64_t GROUP_BASES = getenv("MM_CLI_REF_GROUP_BASES") ? std::stoull(getenv("MM_CLI_REF_GROUP_BASES")) : (uint64_t)1 << 30;   // (test hook: small groups)
    std::vector<std::vector<mm_seqset*>> parts(only_index ? 1 : G);
    double t_pack = 0;
    auto consume = [&](Group& g) {                               // names and lengths in file order, then pack + upload to every device
      for (size_t i = 0; i < g.seq.size(); ++i) { cname.push_back(std::move(g.names[i])); clen.push_back((int)g.seq[i].size()); ref_bases += g.seq[i].size(); }
      const auto t0 = std::chrono::steady_clock::now();
      on_each(parts.size(), [&](size_t d) {
        mm_seqset* p; ck(devs[d].ctx, mm_seqset_create(devs[d].ctx, &p), "seqset");
        for (auto& q : g.seq) ck(devs[d].ctx, mm_seqset_add_view(p, q.data(), (int64_t)q.size()), "add contig");
        ck(devs[d].ctx, mm_seqset_upload(p), "upload reference");
        parts[d].push_back(p);
      });
      t_pack += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    // records of `f` (all, or those that start before `stop` in memory mode) in groups of GROUP_BASES handed to `emit`; false when the
    // reader gave up before `stop` (a truncated quality string ends the file for kseq, kseq.h:204)
    auto parse_groups = [&](SeqFile& f, size_t stop, const std::function<void(std::unique_ptr<Group>)>& emit) -> bool {
      auto g = std::make_unique<Group>();
      bool ok = true;
      for (;;) {
        if (stop != (size_t)-1) { const size_t ps = f.peek_start(); if (ps == (size_t)-1 || ps >= stop) break; }
        if (!f.next()) { ok = stop == (size_t)-1; break; }
        g->names.push_back(f.name);
        if (f.view) g->seq.emplace_back(f.view, f.view_len); else { g->seq.push_back(std::move(f.seq)); f.seq.clear(); }
        g->bases += g->seq.back().size();
        if (g->bases >= GROUP_BASES) { emit(std::move(g)); g = std::make_unique<Group>(); }
      }
      if (!g->seq.empty()) emit(std::move(g));
      return ok;
    };
    MappedFile rmf;
    if (!getenv("MM_CLI_NO_MMAP") && !getenv("MM_CLI_REF_SEQUENTIAL") && rmf.open(ref)) {
      // A plain file: blocks of the mapping parsed by several threads (the block parser of the query files below: a block's records
      // count only once the block before it has been seen to end exactly where this one starts), consumed — packed, uploaded — in file
      // order.  The winSketch.hpp:242-252 loop reads contig by contig; here the text of at most P + 2 blocks of 256 MB is resident, and the
      // mapped pages of a block are given back once it is consumed (they would count as resident until the end otherwise: 27 GB).
      const size_t blk = getenv("MM_CLI_REF_BLOCK_BYTES") ? (size_t)std::max(1, atoi(getenv("MM_CLI_REF_BLOCK_BYTES"))) : (size_t)std::min<uint64_t>(GROUP_BASES, (uint64_t)256 << 20);
      const size_t nb = std::max<size_t>(1, (rmf.size + blk - 1) / blk);
      std::vector<size_t> start(nb + 1, rmf.size);
      start[0] = 0;
      struct Block { std::vector<std::unique_ptr<Group>> out; size_t next = 0; bool done = false, empty = false, over = false; };
      std::vector<Block> blocks(nb);
      std::mutex bm; std::condition_variable bcv; size_t next_block = 0, consumed = 0; bool abandon = false;
      const unsigned P = (unsigned)std::max<size_t>(1, std::min<size_t>({nb, (size_t)8, (size_t)std::max(1u, mm::cpu_budget() / 2)}));
      auto worker = [&]() {
        for (;;) {
          size_t j;
          {
            std::unique_lock<std::mutex> lk(bm);
            bcv.wait(lk, [&] { return abandon || next_block >= nb || next_block < consumed + P + 1; });   // not too far ahead of the consumer
            if (abandon || next_block >= nb) return;
            j = next_block++;
          }
          if (j > 0) start[j] = rmf.sync(j * blk, std::min(rmf.size, (j + 1) * blk));
          Block& B = blocks[j];
          const size_t lim = std::min(rmf.size, (j + 1) * blk);
          if (j == 0 || start[j] < lim) {
            SeqFile f(rmf.data, j == 0 ? 0 : start[j], rmf.size);
            B.over = !parse_groups(f, lim, [&](std::unique_ptr<Group> g) { B.out.push_back(std::move(g)); });
            B.next = f.peek_start();
          } else B.empty = true;
          { std::lock_guard<std::mutex> lk(bm); B.done = true; }
          bcv.notify_all();
        }
      };
      std::vector<std::thread> pool;
      for (unsigned t = 0; t < P; ++t) pool.emplace_back(worker);
      size_t expect = 0; bool chain_ok = true, file_over = false;
      for (size_t j = 0; j < nb && chain_ok && !file_over; ++j) {
        { std::unique_lock<std::mutex> lk(bm); bcv.wait(lk, [&] { return blocks[j].done; }); }
        Block& B = blocks[j];
        if (!B.empty) {
          if (j > 0 && start[j] != expect) { chain_ok = false; break; }
          for (auto& g : B.out) consume(*g);
          B.out.clear();
          if (B.over || B.next == (size_t)-1) { file_over = true; break; }
          expect = B.next;
        } else if (expect < std::min(rmf.size, (j + 1) * blk)) { chain_ok = false; break; }
        { std::lock_guard<std::mutex> lk(bm); consumed = j + 1; } bcv.notify_all();
        if (j > 0) rmf.drop(((j - 1) * blk) & ~(size_t)4095, (j * blk) & ~(size_t)4095);   // (block j - 1: its last record may end inside block j, parsed by now)
      }
      { std::lock_guard<std::mutex> lk(bm); abandon = true; } bcv.notify_all();
      for (auto& t : pool) t.join();
      if (!chain_ok) {                                           // a block did not start where the parse stood: the rest sequentially, from there
        for (auto& B : blocks) B.out.clear();
        SeqFile f(rmf.data, expect, rmf.size);
        parse_groups(f, (size_t)-1, [&](std::unique_ptr<Group> g) { consume(*g); });
      }
    } else {
      // gzip, pipes: a parser thread fills groups, the main thread packs and uploads each while the next one is parsed.  Host memory: two groups.
      std::mutex gm; std::condition_variable gcv; std::deque<std::unique_ptr<Group>> ready; bool parsed = false;
      std::thread parser([&]() {
        SeqFile f(ref);
        parse_groups(f, (size_t)-1, [&](std::unique_ptr<Group> g) {
          std::unique_lock<std::mutex> lk(gm); gcv.wait(lk, [&] { return ready.size() < 2; }); ready.push_back(std::move(g)); gcv.notify_all();
        });
        std::lock_guard<std::mutex> lk(gm); parsed = true; gcv.notify_all();
      });
      for (;;) {
        std::unique_ptr<Group> g;
        { std::unique_lock<std::mutex> lk(gm); gcv.wait(lk, [&] { return !ready.empty() || parsed; }); if (ready.empty()) break; g = std::move(ready.front()); ready.pop_front(); gcv.notify_all(); }
        consume(*g);
      }
      parser.join();
    }
    on_each(parts.size(), [&](size_t d) {
      if (parts[d].size() == 1) { refset[d] = parts[d][0]; return; }
      if (parts[d].empty()) { ck(devs[d].ctx, mm_seqset_create(devs[d].ctx, &refset[d]), "seqset"); ck(devs[d].ctx, mm_seqset_upload(refset[d]), "upload reference"); return; }
      ck(devs[d].ctx, mm_seqset_concat(devs[d].ctx, parts[d].data(), (int)parts[d].size(), &refset[d]), "reference");
      for (auto* p : parts[d]) mm_seqset_destroy(p);
    });
    pc.lap("1 reference parse + pack + upload");
    pc.add("2 reference pack+upload (inside 1)", t_pack);
    if (!only_index) { if (!getenv("MM_CLI_LATE_READER")) start_reader(); start_prewarm(); }   // (MM_CLI_LATE_READER: measurement aid — the reader starts when the index is built)
    query_free();                                                // the packed reference now lives on the device (0.25 B per base, for as long as chunks are cut out of it): what is left is what the indexes get
  }

  // ---- the chunk plan of --maxmemory (winSketch.hpp:274-329): on the index of the whole reference when that fits, on contig ranges otherwise
  void plan_chunks() {
  std::vector<int32_t> first(1, 0);
  if (!maxMem || fits(ref_bases, 1.0)) {
    // the index of the whole reference: the only chunk, or what the chunk rule of --maxmemory is evaluated on
    if (!maxMem && o.stream) die("--stream-chunks needs --maxmemory (the chunk rule of the reference, winSketch.hpp:274-329)");
    mm_seqset* contigs = refset[0];
    ck(ctx0, mm_index_build(ctx0, contigs, k, w, &whole), "index");
    pc.lap("3 index build");
    if (maxMem) {
      int32_t n = 0;
      ck(ctx0, mm_index_plan_chunks(ctx0, whole, maxMem, nullptr, 0, &n), "chunk plan");
      first.resize((size_t)n);
      ck(ctx0, mm_index_plan_chunks(ctx0, whole, maxMem, first.data(), n, &n), "chunk plan");
    }
    if (only_index && first.size() == 1 && !o.v.count("full-index")) ck(ctx0, mm_seqset_save(contigs, (ipre + ".1.seqset").c_str()), "store index chunk");
  } else {
    // The chunk rule without an index of the whole reference: it decides to close a chunk from the chunk's own content
    // and the next contig only, so it can be evaluated on the index of a contig range that fits the device.  Every cut
    // inside the range is final; the range's last chunk is not (it may go on), so the next range starts there.
    std::cout << "INFO, the index of " << ref_bases << " reference bases does not fit one device's " << (hbm_free >> 30) << " GiB: the chunk rule is evaluated on contig ranges\n";
    const int C = (int)cname.size();
    uint64_t range_bases = 0;
    if (o.v.count("stream-range-bases")) range_bases = std::stoull(o.v.at("stream-range-bases"));
    else {                                                     // the largest range whose index BUILD stays within 70 % of what is free
      uint64_t lo = 1, hi = ref_bases;
      while (lo < hi) { const uint64_t mid = lo + (hi - lo + 1) / 2; if (index_bytes(mid, true) <= 0.7 * (double)hbm_free) lo = mid; else hi = mid - 1; }
      range_bases = lo;
    }
    int c0 = 0;
    while (c0 < C) {
      int c1 = c0; uint64_t bases = 0;
      while (c1 < C && (bases < range_bases || c1 == c0)) bases += (uint64_t)clen[(size_t)c1++];
      mm_seqset* part = make_part(0, c0, c1);
      mm_index* ri; ck(ctx0, mm_index_build(ctx0, part, k, w, &ri), "index (chunk planning range)");
      mm_seqset_destroy(part);
      int32_t n = 0;
      ck(ctx0, mm_index_plan_chunks(ctx0, ri, maxMem, nullptr, 0, &n), "chunk plan");
      std::vector<int32_t> loc((size_t)n);
      ck(ctx0, mm_index_plan_chunks(ctx0, ri, maxMem, loc.data(), n, &n), "chunk plan");
      mm_index_destroy(ri);
      if (n == 1 && c1 < C) {                                   // the chunk that starts at c0 is longer than the range
        if (index_bytes(bases * 2, true) > 0.9 * (double)hbm_free && !o.v.count("stream-range-bases"))
          die("--maxmemory describes index chunks larger than this device can hold one at a time");
        range_bases = bases * 2; continue;
      }
      for (int32_t j = 1; j < n; ++j) first.push_back(c0 + loc[(size_t)j]);
      if (c1 == C) break;
      c0 += loc[(size_t)n - 1];
    }
    pc.lap("3 index build");
  }
  for (size_t c = 0; c < first.size(); ++c) {
    const int a = first[c], b = c + 1 < first.size() ? first[c + 1] : (int)cname.size();
    chunks.push_back(Chunk{a, b - a, ""});
  }
  }

  // `metamaps index`: PREFIX.N.seqset (or .mmidx with --full-index) per chunk + PREFIX.index / .arguments / .contigs (mapWrap.h:358-405)
  int write_index_files() {
    { std::ofstream flag(ipre + ".index"); if (!flag.is_open()) die("Cannot open " + ipre + ".index"); flag << 0 << "\n"; }   // mapWrap.h:363-366
    std::vector<std::string> chunk_files;
    const bool full = o.v.count("full-index") != 0;            // the device index itself (mm_index_save) instead of the packed reference it is rebuilt from
    for (size_t c = 0; c < chunks.size(); ++c) {
      chunk_files.push_back(ipre + "." + std::to_string(c + 1) + (full ? ".mmidx" : ".seqset"));
      if (full) {
        mm_index* ix = whole;
        if (!(chunks.size() == 1 && whole)) {
          if (whole) { mm_index_destroy(whole); whole = nullptr; }   // (the chunk rule is done with it)
          mm_seqset* part = make_part(0, chunks[c].first, chunks[c].first + chunks[c].count);
          ck(ctx0, mm_index_build(ctx0, part, k, w, &ix), "index chunk");
          mm_seqset_destroy(part);
        }
        ck(ctx0, mm_index_save(ix, chunk_files.back().c_str()), "store index chunk");
        if (ix != whole) mm_index_destroy(ix);
        continue;
      }
      if (chunks.size() == 1 && whole) continue;                // stored above, from the set the index was built on
      mm_seqset* part = make_part(0, chunks[c].first, chunks[c].first + chunks[c].count);
      ck(ctx0, mm_seqset_save(part, chunk_files.back().c_str()), "store index chunk");
      mm_seqset_destroy(part);
    }
    if (whole) mm_index_destroy(whole);
    drop_refsets();
    std::ofstream args(ipre + ".arguments");
    if (!args.is_open()) die("Cannot open file " + ipre + ".arguments for serialization.");
    args.precision(17);
    args << "kmerSize " << k << "\nwindowSize " << w << "\nminReadLength " << minLen << "\npercentageIdentity " << pi << "\np_value " << pval
         << "\nreferenceSize " << refSize << "\nmaximumMemory " << maxMem << "\nreference " << ref << "\n";
    std::ofstream cf(ipre + ".contigs");
    for (size_t c = 0; c < chunks.size(); ++c)
      for (int i = chunks[c].first; i < chunks[c].first + chunks[c].count; ++i) cf << cname[(size_t)i] << "\t" << clen[(size_t)i] << "\t" << c + 1 << "\n";
    std::ofstream flag(ipre + ".index");                       // mapWrap.h:395-402
    flag << 1 << "\n";
    for (auto& fn : chunk_files) { flag << fn << "\n"; std::cout << "Stored state in file " << fn << "\n"; }
    mm_ctx_destroy(ctx0);
    return 0;
  }

  // `metamaps mapAgainstIndex`: the chunk list and the contig table of a stored index (mapWrap.h:443-554)
  void read_index_files() {
    std::ifstream flag(ipre + ".index");
    if (!flag.is_open()) die("Index " + ipre + " not found (" + ipre + ".index)");
    int done = 0; flag >> done;
    if (done != 1) die("Index " + ipre + " is not complete.");    // mapWrap.h:466-470
    std::vector<std::string> chunk_files; std::string fn;
    while (flag >> fn) chunk_files.push_back(fn);
    std::ifstream cf(ipre + ".contigs");
    if (!cf.is_open()) die("Cannot open " + ipre + ".contigs");
    std::vector<int> chunk_of; std::string line;
    while (std::getline(cf, line)) {
      auto fl = split(line, "\t");
      if (fl.size() != 3) die("Weird line in " + ipre + ".contigs");
      cname.push_back(fl[0]); clen.push_back(std::stoi(fl[1])); chunk_of.push_back(std::stoi(fl[2])); ref_bases += (uint64_t)clen.back();
    }
    for (size_t c = 0; c < chunk_files.size(); ++c) {
      int first = -1, count = 0;
      for (size_t i = 0; i < chunk_of.size(); ++i) if (chunk_of[i] == (int)c + 1) { if (first < 0) first = (int)i; ++count; }
      chunks.push_back(Chunk{first < 0 ? 0 : first, count, chunk_files[c]});
    }
  }

  // ---- where the chunk indexes live
  void decide_placement() {
    NC = chunks.size();
    if (o.stream) place = Place::Streamed;
    else if (o.shard) place = Place::Sharded;
    else if (NC > 1) {
      // every chunk index resident on every device / chunk c on device c mod G / one round of G chunks at a time: the first that fits
      // (a chunk index costs more per base than the whole reference's: fewer occurrences per hash, the same padding per list)
      std::vector<double> per_dev(G, 0.0); double all = 0, build_extra = 0;
      for (size_t c = 0; c < NC; ++c) {
        uint64_t cb = 0; for (int i = chunks[c].first; i < chunks[c].first + chunks[c].count; ++i) cb += (uint64_t)clen[(size_t)i];
        const double b = index_bytes(cb, false);
        all += b; per_dev[c % G] += b; build_extra = std::max(build_extra, index_bytes(cb, true) - b);
      }
      const double room = 0.8 * (double)hbm_free;
      if (all + build_extra <= room) place = Place::Replicated;
      else place = (G > 1 && *std::max_element(per_dev.begin(), per_dev.end()) + build_extra <= room) ? Place::Sharded : Place::Streamed;
    }
    if (place != Place::Replicated && !o.stream && !o.shard) {
      std::cout << "INFO, the index of " << ref_bases << " reference bases does not fit one device's " << (hbm_free >> 30) << " GiB: "
                << (place == Place::Sharded ? "the chunk indexes are spread over the devices" : "chunk indexes are built and mapped one after the other") << "\n";
    }
    if (place != Place::Replicated && NC == 1 && !from_index && !maxMem) die("--stream-chunks / --shard-index need --maxmemory (the chunk rule of the reference, winSketch.hpp:274-329)");
    for (auto& d : devs) d.idx.assign(NC, nullptr);
    thr_of.assign(NC, INT_MAX);
  }

  void build_chunk(Dev& d, size_t c) {                            // the index of chunk c on device d
    const Chunk& ch = chunks[c];
    if (d.idx[c]) return;
    if (whole && NC == 1 && &d == &devs[0]) { d.idx[c] = whole; whole = nullptr; return; }
    mm_seqset* part;
    if (ch.file.size() > 6 && ch.file.compare(ch.file.size() - 6, 6, ".mmidx") == 0) {   // `index --full-index`: the stored device index, nothing to build
      ck(d.ctx, mm_index_load(d.ctx, ch.file.c_str(), &d.idx[c]), "load index chunk");
      mm_index_info info; mm_index_get_info(d.idx[c], &info);
      if ((int64_t)ch.count != info.n_contigs) die("Index chunk " + ch.file + " does not match " + ipre + ".contigs");
      return;
    }
    if (!ch.file.empty()) {
      ck(d.ctx, mm_seqset_load(d.ctx, ch.file.c_str(), &part), "load index chunk");
      if ((int64_t)ch.count != mm_seqset_count(part)) die("Index chunk " + ch.file + " does not match " + ipre + ".contigs");
    } else if (NC == 1) part = refset[(size_t)(&d - &devs[0])];    // the whole reference is the chunk: no copy
    else part = make_part((size_t)(&d - &devs[0]), ch.first, ch.first + ch.count);
    ck(d.ctx, mm_index_build(d.ctx, part, k, w, &d.idx[c]), "index chunk");
    if (!(ch.file.empty() && NC == 1)) mm_seqset_destroy(part);
  }
  // freqThreshold of chunk c from the histogram accumulated over chunks 0..c: call once per chunk, in chunk order, after some
  // device has built it; the value is then set on every copy of that chunk
  void settle_threshold(size_t c) {
    mm_index* any = nullptr;
    for (auto& d : devs) if (d.idx[c]) { any = d.idx[c]; break; }
    int64_t n = 0; mm_index_freq_hist(any, nullptr, nullptr, 0, &n);
    std::vector<int64_t> cc((size_t)n), hh((size_t)n); mm_index_freq_hist(any, cc.data(), hh.data(), n, &n);
    for (int64_t i = 0; i < n; ++i) thr_acc[cc[(size_t)i]] += hh[(size_t)i];
    mm_index_info info; mm_index_get_info(any, &info);
    if (info.n_unique_hashes > 0) {
      std::vector<int64_t> ac, ah; for (auto& kv : thr_acc) { ac.push_back(kv.first); ah.push_back(kv.second); }
      thr = mm_freq_threshold_from_hist(ac.data(), ah.data(), (int64_t)ac.size(), info.n_unique_hashes, thr);
    }
    thr_of[c] = thr;
    for (auto& d : devs) if (d.idx[c]) mm_index_set_freq_threshold(d.idx[c], thr);
    std::cout << "INFO, index chunk " << c + 1 << "/" << NC << ": contigs " << chunks[c].first << ".." << chunks[c].first + chunks[c].count - 1
              << ", " << info.n_entries << " minimizers, " << info.n_unique_hashes << " unique hashes\n";
  }

  // replicated: every chunk index on every device before the first batch; the packed reference goes
  void build_resident_indexes() {
    if (whole && !(NC == 1 && place == Place::Replicated)) { mm_index_destroy(whole); whole = nullptr; }
    if (place == Place::Replicated) {
      on_each(G, [&](size_t d) { for (size_t c = 0; c < NC; ++c) build_chunk(devs[d], c); });
      for (size_t c = 0; c < NC; ++c) settle_threshold(c);
      drop_refsets();
      pc.lap("3 index build");
    }
  }

  mm_seqset* upload_batch(mm_ctx* ctx, const Batch& bt) {
    mm_seqset* reads; ck(ctx, mm_seqset_create(ctx, &reads), "seqset");
    for (size_t r = 0; r < bt.names.size(); ++r) ck(ctx, mm_seqset_add_view(reads, bt.seq_of(r), (int64_t)bt.lens[r]), "add read");
    ck(ctx, mm_seqset_upload(reads), "upload reads");
    return reads;
  }
  // one "PREFIX.N" per chunk in the reference (mapWrap.h:419-437); `sketch_of`: an earlier mapping of the same batch on this device,
  // whose minimizers and sketches are reused (they do not depend on the index)
  mm_mapping* map_chunk(mm_ctx* ctx, mm_index* idx, mm_seqset* reads, const mm_mapping* sketch_of = nullptr) {
    mm_mapping* pm;
    if (sketch_of) ck(ctx, mm_map_batch_reusing(ctx, idx, reads, &mp, sketch_of, &pm), "map");
    else ck(ctx, mm_map_batch(ctx, idx, reads, &mp, &pm), "map");
    if (!o.all) ck(ctx, mm_mapping_keep_best(ctx, pm, k), "best mappings");
    return pm;
  }
  std::unique_ptr<Done> finish_mapping(mm_ctx* ctx, mm_mapping* m, std::vector<std::string>&& names, std::vector<int>&& lens, size_t file) {   // mapping qualities + text; consumes m
    auto dn = std::make_unique<Done>();
    dn->file = file; dn->names = std::move(names); dn->lens = std::move(lens);
    const auto f0 = std::chrono::steady_clock::now();
    ck(ctx, mm_mapping_add_qualities(ctx, m, nullptr, k), "mapping qualities");
    dn->off.resize(dn->names.size() + 1);
    ck(ctx, mm_mapping_fetch(m, dn->off.data(), nullptr, 0), "fetch");
    const auto f1 = std::chrono::steady_clock::now();
    std::vector<mm_map_record> rec((size_t)dn->off.back());
    ck(ctx, mm_mapping_fetch(m, dn->off.data(), rec.data(), (int64_t)rec.size()), "fetch");
    mm_mapping_destroy(m);
    const auto f2 = std::chrono::steady_clock::now();
    format_records(dn->names, dn->lens, dn->off, rec, cname, clen, k, dn->text, keep_lines ? &dn->meta : nullptr);
    const auto f3 = std::chrono::steady_clock::now();
    pc.add("7a mapping qualities + offsets", std::chrono::duration<double>(f1 - f0).count());
    pc.add("7b fetch records", std::chrono::duration<double>(f2 - f1).count());
    pc.add("7c format", std::chrono::duration<double>(f3 - f2).count());
    dn->t_mapq = std::chrono::duration<double>(f1 - f0).count(); dn->t_fetch = std::chrono::duration<double>(f2 - f1).count(); dn->t_format = std::chrono::duration<double>(f3 - f2).count();
    return dn;
  }
  void write_all(const std::function<std::unique_ptr<Done>(size_t, size_t)>& next /* (file, seq): batch `seq` if it belongs to that file, nullptr once the file has ended */) {
    size_t seq = 0;
    for (size_t fi = 0; fi < queries.size(); ++fi) {
      const std::string& prefix = prefixes[fi];
      std::ofstream out(prefix), unm(prefix + ".meta.unmappedReadsLengths");
      if (!out.is_open()) die("Cannot open output file " + prefix);
      size_t total = 0, tooShort = 0, mapped = 0, notMapped = 0; IdSet seen;   // (id_set.hpp: a std::set of 10^6 IDs bounded the mapping phase)
      for (;;) {
        std::unique_ptr<Done> d = next(fi, seq);
        if (!d) break;
        if (d->file != fi) die("internal error: batch order");
        ++seq;
        for (size_t r = 0; r < d->names.size(); ++r) {
          ++total;
          const int len = d->lens[r];
          if (len < w || len < k || len < minLen) { ++tooShort; continue; }
          // mapWrap.h:71-75 checks the IDs of mapping LINES against the reads already handled: a repeated ID only stops the run
          // when the repeat carries mappings; every handled read's ID is remembered (:154-157)
          if (d->off[r] == d->off[r + 1]) { ++notMapped; unm << len << "\t" << d->names[r] << "\n"; seen.insert(d->names[r]); continue; }
          if (!seen.insert(d->names[r])) die("Seems that read ID " + d->names[r] + " has already been processed");
          ++mapped;
        }
        out << d->text;
        if (keep_lines) { if (kept.size() <= fi) kept.resize(fi + 1); d->names.clear(); d->names.shrink_to_fit(); kept[fi].push_back(std::move(d)); }
      }
      if (keep_lines && kept.size() <= fi) kept.resize(fi + 1);
      std::ofstream meta(prefix + ".meta");                      // mapWrap.h:178-184
      meta << "TotalReads " << total << "\nReadsTooShort " << tooShort << "\nReadsMapped " << mapped << "\nReadsNotMapped " << notMapped << "\n";
      std::ofstream ps(prefix + ".parameters");                  // mapWrap.h:196-211
      ps << "kmerSize " << k << "\nwindowSize " << w << "\nminReadLength " << minLen << "\nalphabetSize " << 4 << "\nreferenceSize " << refSize
         << "\npercentageIdentity " << pi << "\np_value " << pval << "\nrefSequences [" << ref << "]\nquerySequences [" << queries[fi]
         << "]\noutFileName " << prefix << "\nreportAll " << o.all << "\nindex " << "" << "\nmaximumMemory " << maxMem << "\n";
      std::cout << "INFO, [count of mapped reads, reads qualified for mapping, total input reads] = [" << mapped << ", " << total - tooShort << ", " << total << "]\n";
    }
  }

  void run_replicated() {
    // ---- workers: four contexts per device (--workers-per-gpu; three until round 4: with ten batches of 10^5 reads in one file the GPU idled 60 % of the mapping phase), so that packing, result download and text formatting of one batch overlap the
    // kernels of the other; the device's chunk indexes are shared (read-only) by its contexts
    // The kernels of a batch fill the device; batches mapped side by side only take turns on it, and four workers that start together
    // then also finish together: they packed, fetched and formatted at the same time with the device idle, and mapped at the same time
    // in each other's way (the done-times of the workers came in groups of four, 60 ms apart).  So at most MAP_SLOTS batches per device are
    // inside their mapping section at a time (two: one fills the host-side gaps of the other), which staggers the workers.
    const size_t MAP_SLOTS = getenv("MM_CLI_MAP_SLOTS") ? (size_t)std::max(1, atoi(getenv("MM_CLI_MAP_SLOTS"))) : 2;
    struct Slots { std::mutex m; std::condition_variable cv; size_t free_ = 0;
                   void acquire() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return free_ > 0; }); --free_; }
                   void release() { { std::lock_guard<std::mutex> lk(m); ++free_; } cv.notify_one(); } };
    std::vector<Slots> map_slots(G);
    for (auto& sl : map_slots) sl.free_ = MAP_SLOTS;
    std::vector<std::thread> workers;
    for (size_t d = 0; d < G; ++d) for (size_t wi = 0; wi < WPD; ++wi) workers.emplace_back([&, d, wi]() {
      mm_ctx* ctx = wi == 0 ? devs[d].ctx : wctx[d * WPD + wi];
      if (!ctx && mm_ctx_create(devs[d].phys, &ctx) != MM_OK) die("cannot create a worker context");
      while (std::unique_ptr<Batch> bt = reader.take()) {
        const auto t0 = std::chrono::steady_clock::now();
        mm_seqset* reads = upload_batch(ctx, *bt);
        const auto t1 = std::chrono::steady_clock::now();
        std::vector<mm_mapping*> parts;
        map_slots[d].acquire();
        const auto t1a = std::chrono::steady_clock::now();
        for (size_t c = 0; c < NC; ++c) parts.push_back(map_chunk(ctx, devs[d].idx[c], reads, c ? parts[0] : nullptr));
        map_slots[d].release();
        mm_map_stats gst{}; if (getenv("MM_CLI_TIMING")) mm_mapping_get_stats(parts[0], &gst);   // (device time of the batch's stages by the library's own events)
        mm_mapping* m = parts[0];
        if (parts.size() > 1) {                                   // unifyFiles: read-wise concatenation in chunk order
          ck(ctx, mm_mapping_concat(ctx, parts.data(), chunk_base.data(), (int)parts.size(), &m), "merge chunks");
          for (auto* pm : parts) mm_mapping_destroy(pm);
        }
        mm_seqset_destroy(reads);
        const auto t2 = std::chrono::steady_clock::now();
        const size_t seq = bt->seq;
        auto dn = finish_mapping(ctx, m, std::move(bt->names), std::move(bt->lens), bt->file);
        const auto t3 = std::chrono::steady_clock::now();
        pc.add("5 reads pack+upload", std::chrono::duration<double>(t1 - t0).count());
        pc.add("6 map", std::chrono::duration<double>(t2 - t1a).count());
        pc.add("6a waited for the device", std::chrono::duration<double>(t1a - t1).count());
        pc.add("7 mapq+fetch+format", std::chrono::duration<double>(t3 - t2).count());
        if (getenv("MM_CLI_TIMING")) { std::ostringstream os; os << "INFO, worker " << d << "." << wi << " batch " << seq << ": upload " << std::chrono::duration<double>(t1 - t0).count() << " map "
          << std::chrono::duration<double>(t2 - t1a).count() << " (waited " << std::chrono::duration<double>(t1a - t1).count() << "; device ms: K1 " << gst.ms_minimizer << " K2 " << gst.ms_sketch << " K3 " << gst.ms_probe_gather << " K4 " << gst.ms_sort_hits + gst.ms_l1_scan << " K5 " << gst.ms_l2 << " all " << gst.ms_total << ") finish " << std::chrono::duration<double>(t3 - t2).count() << " (mapq " << dn->t_mapq << " fetch " << dn->t_fetch << " format " << dn->t_format << ") done at +" << std::chrono::duration<double>(t3 - pc.t0).count() << " s\n"; std::cerr << os.str(); }
        reader.recycle(std::move(bt));
        writer.put(seq, std::move(dn));
      }
      if (wi > 0) mm_ctx_destroy(ctx);
    });
    write_all([&](size_t fi, size_t seq) -> std::unique_ptr<Done> {
      std::unique_lock<std::mutex> lk(writer.m);
      for (;;) {
        { std::lock_guard<std::mutex> rl(reader.m); if (reader.file_end.size() > fi && reader.file_end[fi] == seq) return nullptr; }   // file fi ended before batch `seq`
        auto it = writer.ready.find(seq);
        if (it != writer.ready.end()) { auto d = std::move(it->second); writer.ready.erase(it); return d; }
        writer.cv.wait_for(lk, std::chrono::milliseconds(20));
      }
    });
    for (auto& t : workers) t.join();
  }

  void run_chunk_major() {
    // ---- sharded / streamed: every read batch is packed onto every device and stays there (2 bits per base) ...
    struct Held { size_t file = 0; std::vector<std::string> names; std::vector<int> lens; std::vector<mm_seqset*> reads;
                  std::vector<mm_mapping*> sk;                 // per device: the batch's minimizers + sketches (mm_sketch_batch), computed once for all chunks
                  std::vector<mm_mapping*> part;               // per chunk: the batch's records against that chunk, on the device that holds the chunk (c mod G)
                  std::vector<std::vector<int64_t>> poff; std::vector<std::vector<mm_map_record>> prec; };   // --host-gather: the same in host memory (rounds 1-3)
    // How the records of a batch reach the device that merges them (unifyFiles, mapWrap.h:128-145, in place of the PREFIX.N files):
    //   rccl  (several physical devices) mm_mapping_gather: ncclSend / ncclRecv over xGMI, one collective per batch
    //   peer  (logical devices of one GPU, or --peer-gather) mm_mapping_concat pulls the parts of other contexts with device-to-device copies
    //   host  (--host-gather) mm_mapping_fetch + mm_mapping_from_parts: through host memory, the path of rounds 1-3, kept as the cross-check
    bool distinct = true; for (size_t a = 0; a < G; ++a) for (size_t b2 = a + 1; b2 < G; ++b2) distinct = distinct && devs[a].phys != devs[b2].phys;
    enum class Gather { Rccl, Peer, Host } gather = o.v.count("host-gather") ? Gather::Host : (G > 1 && distinct && !o.v.count("peer-gather")) ? Gather::Rccl : Gather::Peer;
    if (getenv("MM_CLI_TIMING")) std::cerr << "INFO, records of the chunks are gathered by " << (gather == Gather::Rccl ? "RCCL send / receive" : gather == Gather::Peer ? "device-to-device copies" : "the host") << "\n";
    std::vector<Held> held;
    while (std::unique_ptr<Batch> bt = reader.take()) {
      held.emplace_back();
      Held& h = held.back();
      h.file = bt->file; h.reads.assign(G, nullptr); h.sk.assign(G, nullptr); h.part.assign(NC, nullptr); h.poff.resize(NC); h.prec.resize(NC);
      on_each(G, [&](size_t d) { h.reads[d] = upload_batch(devs[d].ctx, *bt); });
      h.names = std::move(bt->names); h.lens = std::move(bt->lens);
      reader.recycle(std::move(bt));
    }
    pc.lap("5 reads pack+upload");
    // ... then the chunks in rounds: chunk c on device c mod N — all of them at once when they fit together (sharded), N at a time
    // otherwise (streamed: built, mapped, dropped).  A round's indexes are built concurrently, their thresholds follow in chunk
    // order from the accumulated histogram, then every device maps every batch against its chunks; the records of a pass go to the
    // host, where the reference keeps its PREFIX.N files (mapWrap.h:417-437).
    const size_t per_round = place == Place::Streamed ? G : NC;
    // Minimizers and sketches do not depend on the chunk: a batch keeps them on its device from its first chunk on (about 3 bytes per read
    // base, twelve times the packed reads), as long as all of them stay within an eighth of the device's memory; batches beyond that
    // recompute them per chunk (MM_CLI_NO_SKETCH_REUSE=1: all of them, the cross-check).
    std::vector<uint64_t> sk_used(G, 0), sk_budget(G, 0);
    for (size_t d = 0; d < G; ++d) {
      uint64_t tot = 0, fr = 0; char nm[8]; int cus = 0;
      if (mm_ctx_device_info(devs[d].ctx, nm, sizeof nm, &cus, &tot, &fr) == MM_OK && !getenv("MM_CLI_NO_SKETCH_REUSE")) sk_budget[d] = tot / 8;
    }
    for (size_t c0 = 0; c0 < NC; c0 += per_round) {
      const size_t c1 = std::min(NC, c0 + per_round);
      on_each(G, [&](size_t d) { for (size_t c = c0; c < c1; ++c) if (c % G == d) build_chunk(devs[d], c); });
      for (size_t c = c0; c < c1; ++c) settle_threshold(c);
      pc.lap("3 index build");
      on_each(G, [&](size_t d) {
        for (size_t c = c0; c < c1; ++c) {
          if (c % G != d) continue;
          for (auto& h : held) {
            if (!h.sk[d] && sk_budget[d]) {
              uint64_t bases = 0; for (int L : h.lens) bases += (uint64_t)L;
              if (sk_used[d] + 3 * bases <= sk_budget[d]) { ck(devs[d].ctx, mm_sketch_batch(devs[d].ctx, h.reads[d], &mp, &h.sk[d]), "sketch"); sk_used[d] += 3 * bases; }
            }
            mm_mapping* pm = map_chunk(devs[d].ctx, devs[d].idx[c], h.reads[d], h.sk[d]);
            if (gather == Gather::Host) {
              h.poff[c].resize(h.names.size() + 1);
              ck(devs[d].ctx, mm_mapping_fetch(pm, h.poff[c].data(), nullptr, 0), "fetch");
              h.prec[c].resize((size_t)h.poff[c].back());
              ck(devs[d].ctx, mm_mapping_fetch(pm, h.poff[c].data(), h.prec[c].data(), (int64_t)h.prec[c].size()), "fetch");
              mm_mapping_destroy(pm);
            } else { ck(devs[d].ctx, mm_mapping_release_intermediates(pm), "release"); h.part[c] = pm; }   // the records stay where they were made
          }
          if (place == Place::Streamed) { mm_index_destroy(devs[d].idx[c]); devs[d].idx[c] = nullptr; }
        }
      });
      pc.lap("6 map");
    }
    // merge in chunk order (unifyFiles), mapping qualities over the union and text: batch b on device b mod N
    std::vector<std::unique_ptr<Done>> results(held.size());
    std::vector<int32_t> chunk_rank(NC); for (size_t c = 0; c < NC; ++c) chunk_rank[c] = (int32_t)(c % G);
    char comm_id[MM_COMM_ID_BYTES];
    if (gather == Gather::Rccl && mm_comm_unique_id(comm_id) != MM_OK) die("RCCL: cannot create a communicator id");
    std::vector<mm_mapping*> merged(held.size(), nullptr);
    on_each(G, [&](size_t d) {
      mm_ctx* ctx = devs[d].ctx;
      for (auto& h : held) { if (h.sk[d]) mm_mapping_destroy(h.sk[d]); mm_seqset_destroy(h.reads[d]); }
      if (gather == Gather::Rccl) {                                // every rank takes part in the gather of every batch, batch b ends on rank b mod G
        ck(ctx, mm_comm_init(ctx, comm_id, (int)d, (int)G), "RCCL communicator");
        for (size_t b = 0; b < held.size(); ++b) {
          Held& h = held[b];
          std::vector<mm_mapping*> mine; std::vector<int32_t> ids;
          for (size_t c = d; c < NC; c += G) { mine.push_back(h.part[c]); ids.push_back((int32_t)c); }
          mm_mapping* m = nullptr;
          ck(ctx, mm_mapping_gather(ctx, (int)(b % G), (int64_t)h.names.size(), h.lens.data(), &mp, mine.data(), ids.data(), (int)mine.size(), (int)NC, chunk_rank.data(), chunk_base.data(), &m), "gather chunks");
          if (b % G == d) merged[b] = m;
          for (auto* pm : mine) mm_mapping_destroy(pm);
        }
        mm_comm_destroy(ctx);
      }
    });
    on_each(G, [&](size_t d) {
      for (size_t b = d; b < held.size(); b += G) {
        Held& h = held[b];
        mm_mapping* m = merged[b];
        if (gather == Gather::Peer) {
          ck(devs[d].ctx, mm_mapping_concat(devs[d].ctx, h.part.data(), chunk_base.data(), (int)NC, &m), "merge chunks");
        } else if (gather == Gather::Host) {
          std::vector<const int64_t*> op; std::vector<const mm_map_record*> rp;
          for (size_t c = 0; c < NC; ++c) { op.push_back(h.poff[c].data()); rp.push_back(h.prec[c].data()); }
          ck(devs[d].ctx, mm_mapping_from_parts(devs[d].ctx, (int64_t)h.names.size(), h.lens.data(), &mp, (int)NC, op.data(), rp.data(), chunk_base.data(), &m), "merge chunks");
        }
        results[b] = finish_mapping(devs[d].ctx, m, std::move(h.names), std::move(h.lens), h.file);
        std::vector<std::vector<int64_t>>().swap(h.poff); std::vector<std::vector<mm_map_record>>().swap(h.prec);
      }
    });
    if (gather == Gather::Peer) for (auto& h : held) for (size_t c = 0; c < NC; ++c) if (h.part[c]) {   // (after every owner has pulled what it needed; destroyed through its own context)
      mm_mapping_destroy(h.part[c]); h.part[c] = nullptr; }
    pc.lap("7 mapq+fetch+format");
    write_all([&](size_t fi, size_t seq) -> std::unique_ptr<Done> {
      if (seq >= results.size() || results[seq]->file != fi) return nullptr;
      return std::move(results[seq]);
    });
  }

  // --then-classify DBDIR (not in the reference): `metamaps classify --DB DBDIR --mappings PREFIX` for every output prefix, in THIS process, on the
  // files just written — the same code (classify_one) on the same bytes, so the same .EM* files as the two-process form, which stays the tested
  // default.  What it saves is what lies between the two processes: this one's exit (150 GB of index handed back), the next one's HIP
  // initialisation behind it (1.3-1.9 s waiting for the driver) and its contexts: the live contexts are used.
  void then_classify() {
    if (!o.v.count("then-classify") || only_index) return;
    if (reader.th.joinable()) reader.th.join();
    const EmReduce reduce = o.em_host ? EmReduce::Host : ((devs.size() > 1 || o.v.count("gpus") || o.v.count("devices")) ? EmReduce::Rccl : EmReduce::None);
    const size_t minReadsU = o.v.count("minreads") ? std::stoull(o.v.at("minreads")) : 10000;   // parseCmdArgs.hpp:462-471
    // the last prefix ends the process from inside classify_one, as the last file of `classify` does: everything is written and closed, the
    // gigabyte of line tables and text is not taken apart first (MM_CLI_FULL_TEARDOWN=1: the orderly way)
    const std::function<void()> leave = [&] { pc.lap("9 classify"); pc.report(); };
    for (size_t fi = 0; fi < prefixes.size(); ++fi) {
      const bool last = fi + 1 == prefixes.size();
      KeptLines kl; kl.cname = &cname;
      if (keep_lines && fi < kept.size()) for (const auto& d : kept[fi]) kl.parts.push_back(KeptLines::Part{d->text.data(), d->meta.data(), d->meta.size(), d->off.data(), d->lens.size()});
      classify_one(devs, reduce, prefixes[fi], o.v.at("then-classify"), minReadsU, last ? leave : std::function<void()>(), nullptr, keep_lines ? &kl : nullptr);
      if (keep_lines && fi < kept.size()) kept[fi].clear();
      for (auto& d : devs) mm_comm_destroy(d.ctx);
      pc.lap("9 classify");
    }
  }

  int run() {
    read_parameters();
    open_devices();
    if (!from_index) {
      load_reference();
      plan_chunks();
      if (only_index) return write_index_files();
    } else read_index_files();
    decide_placement();
    build_resident_indexes();
    mp = mm_map_params{k, w, pi, minLen};
    for (auto& ch : chunks) chunk_base.push_back(ch.first);
    start_reader();
    if (prewarm.joinable()) prewarm.join();
    if (place != Place::Replicated) for (auto*& c : wctx) if (c) { mm_ctx_destroy(c); c = nullptr; }   // (the other modes drive one context per device)
    if (place == Place::Replicated) run_replicated(); else run_chunk_major();
    pc.lap("8 write");
    then_classify();
    if (!getenv("MM_CLI_FULL_TEARDOWN")) { if (reader.th.joinable()) reader.th.join(); pc.report(); finish_fast(); }
    drop_refsets();
    for (auto& d : devs) { for (auto* ix : d.idx) if (ix) mm_index_destroy(ix); mm_ctx_destroy(d.ctx); }
    return 0;
  }
};

int map_mode(const Options& o, const std::string& mode) { MapRun run(o, mode); return run.run(); }

// ------------------------------------------------------------------------------------------------------
struct TaxNode { std::string parent, rank, sci; };
struct Taxonomy {                                                // meta/taxonomy.h:137-246
  std::map<std::string, TaxNode> T;
  // split(regex_replace(line, "\\s*\\|\\s*", "|"), "|") of taxonomy.h:150-175 without std::regex: cut at every '|', drop the white space
  // that touches a '|' (not the one at the very start or end of the line)
  static std::vector<std::string> fields(const std::string& ln) {
    std::vector<std::string> out;
    if (ln.empty()) return out;
    size_t a = 0;
    for (bool first = true;; first = false) {
      const size_t bar = ln.find('|', a);
      size_t lo = a, hi = bar == std::string::npos ? ln.size() : bar;
      if (!first) while (lo < hi && isspace((unsigned char)ln[lo])) ++lo;
      if (bar != std::string::npos) while (hi > lo && isspace((unsigned char)ln[hi - 1])) --hi;
      out.push_back(ln.substr(lo, hi - lo));
      if (bar == std::string::npos) break;
      a = bar + 1;
    }
    return out;
  }
  explicit Taxonomy(const std::string& dir) {
    std::map<std::string, std::string> sci; std::string ln;
    std::ifstream nm(dir + "/names.dmp"); if (!nm.is_open()) die("Cannot open file " + dir + "/names.dmp -- is '" + dir + "' a valid NCBI taxonomy?");
    while (std::getline(nm, ln)) { if (ln.empty()) continue; auto f = fields(ln); if (f.size() > 3 && f[3] == "scientific name") sci[f[0]] = f[1]; else if (f.size() > 3 && f[3] == "genbank common name") sci[f[0]]; }
    std::ifstream nd(dir + "/nodes.dmp"); if (!nd.is_open()) die("Cannot open file " + dir + "/nodes.dmp");
    while (std::getline(nd, ln)) { if (ln.empty()) continue; auto f = fields(ln); if (!sci.count(f[0])) die("No name for taxon ID " + f[0] + " in taxonomy directory " + dir); T[f[0]] = TaxNode{f[1], f[2], sci[f[0]]}; }
    std::cout << "Read taxonomy from " << dir << " -- have " << T.size() << " nodes." << std::endl;
  }
  std::map<std::string, std::string> upward_by_ranks(std::string id, const std::set<std::string>& want) const {   // taxonomy.h:76-111
    std::map<std::string, std::string> r;
    std::vector<std::string> up{id};
    while (id != "1") { id = T.at(id).parent; up.push_back(id); }
    for (auto& n : up) { const std::string& rank = T.at(n).rank; if (!want.count(rank)) continue; if (rank != "no rank") { if (r.count(rank)) die("Node " + up[0] + " has multiple entries for rank " + rank); r[rank] = n; } }
    for (auto& w : want) if (!r.count(w)) r[w] = "Undefined";
    return r;
  }
  std::string first_non_x(std::string id) const { while (id.find('x') != std::string::npos) id = T.at(id).parent; return id; }   // :51-74
};

// first match of the reference's regex  kraken:taxid\|(x?\d+)  (fEM.h:1396), without std::regex (called per mapping line)
std::string extract_taxon(const std::string& contig) {
  static const std::string key = "kraken:taxid|";
  for (size_t p = contig.find(key); p != std::string::npos; p = contig.find(key, p + 1)) {
    size_t a = p + key.size(), b = a;
    if (b < contig.size() && contig[b] == 'x') ++b;
    const size_t d0 = b;
    while (b < contig.size() && contig[b] >= '0' && contig[b] <= '9') ++b;
    if (b > d0) return contig.substr(a, b - a);
  }
  die("Could not extract taxon ID from contig identifier '" + contig + "' - did you use the MetMaps build scripts to construct your database?");
}

void write_wimp(const std::string& fn, const Taxonomy& T, const std::map<std::string, double>& freq, const std::map<std::string, size_t>& reads,
                size_t nTotal, size_t nUnmapped, size_t nTooShort) {   // fEM.h:52-215
  const std::set<std::string> levels{"species", "genus", "family", "order", "phylum", "superkingdom"};
  std::map<std::string, std::set<std::string>> keys;
  std::map<std::string, std::map<std::string, double>> fL; std::map<std::string, std::map<std::string, size_t>> rL;
  for (auto& kv : freq) { auto up = T.upward_by_ranks(kv.first, levels); up["definedGenomes"] = kv.first;
    for (auto& u : up) { fL[u.first][u.second] += kv.second; keys[u.first].insert(u.second); if (fL[u.first][u.second] > 1) fL[u.first][u.second] = 1; } }
  for (auto& kv : reads) { auto up = T.upward_by_ranks(kv.first, levels); up["definedGenomes"] = kv.first;
    for (auto& u : up) { rL[u.first][u.second] += kv.second; keys[u.first].insert(u.second); } }
  const long long nMappable = (long long)nTotal - (long long)nTooShort, nMapped = nMappable - (long long)nUnmapped;
  std::ofstream o(fn);
  o << "AnalysisLevel\ttaxonID\tName\tAbsolute\tEMFrequency\tPotFrequency\n";
  for (auto& lv : keys) {
    const std::string& L = lv.first; std::map<std::string, double> emF; double sumF = 0;
    for (auto& t : lv.second) { double f = fL[L].count(t) ? fL[L][t] : 0; size_t r = rL[L].count(t) ? rL[L][t] : 0; sumF += f; fL[L][t] = f; rL[L][t] = r; }
    for (auto& t : lv.second) { fL[L][t] /= sumF; emF[t] = fL[L][t]; }
    const double propMapped = (double)nMapped / nMappable; double propNot = (double)nUnmapped / nMappable;
    for (auto& t : lv.second) fL[L][t] *= propMapped;
    double emUnm = 0; size_t nUnmUndef = nUnmapped;
    for (auto& t : lv.second) {
      if (t != "Undefined") o << L << "\t" << t << "\t" << T.T.at(t).sci << "\t" << rL[L][t] << "\t" << emF[t] << "\t" << fL[L][t] << "\n";
      else { nUnmUndef += rL[L][t]; emUnm += emF[t]; propNot += fL[L][t]; }
    }
    o << L << "\t" << 0 << "\tUnclassified\t" << nUnmUndef << "\t" << emUnm << "\t" << propNot << "\n";
    o << L << "\t" << -3 << "\ttotalReads\t" << nTotal << "\t" << 0 << "\t" << 0 << "\n";
    o << L << "\t" << -3 << "\treadsLongEnough\t" << nMappable << "\t" << 0 << "\t" << 0 << "\n";
    o << L << "\t" << -3 << "\treadsLongEnough_unmapped\t" << nUnmapped << "\t" << 0 << "\t" << 0 << "\n";
  }
}

// .EM.contigCoverage: bases of best mappings per 1000-bp window of every contig that carries one (fEM.h:684, :730-776,
// :805-845).  Kept as the reference computes it, including the length it assigns to the last window of a contig that is
// not a multiple of the window size (:744 subtracts after incrementing the window count, so the unsigned value wraps).
struct ContigCoverage {
  const size_t W = 1000;
  std::map<std::string, std::map<std::string, std::vector<size_t>>> cov, reads;   // bases / best mappings per window
  std::map<std::string, std::map<std::string, size_t>> last;
  struct Slot { std::vector<size_t>* v = nullptr; std::vector<size_t>* nr = nullptr; };   // the two window vectors of a contig (map nodes do not move)
  Slot slot(const std::string& tx, const std::string& cg, size_t L) {
    auto& per = cov[tx];
    if (!per.count(cg)) {
      size_t n = L / W;
      if (n == 0) { n = 1; last[tx][cg] = L; }
      else if (n * W != L) { ++n; last[tx][cg] = L - n * W; }
      else last[tx][cg] = W;
      per[cg].assign(n, 0);
      reads[tx][cg].assign(n, 0);
    }
    return Slot{&per[cg], &reads[tx][cg]};
  }
  void add(const std::string& tx, const std::string& cg, size_t L, size_t start, size_t stop_in) { add(slot(tx, cg, L), L, start, stop_in); }
  void add(const Slot& sl, size_t L, size_t start, size_t stop_in) {
    const size_t stop = stop_in >= L ? L - 1 : stop_in;
    std::vector<size_t>& v = *sl.v;
    std::vector<size_t>& nr = *sl.nr;
    for (size_t p = start; p <= stop; p += W) {
      const size_t wi = p / W, ws = wi * W;
      size_t we = (wi + 1) * W - 1;
      if (we > L) we = L - 1;
      v.at(wi) += iv_overlap(ws, we, start, stop);
      nr.at(wi)++;
    }
  }
  void write(const std::string& fn, const Taxonomy& T) const {   // fEM.h:805-832; one line per 1000-base window of every contig with a best mapping: contigs formatted by several threads
    std::ofstream o(fn);
    o << "taxonID\tequalCoverageUnitLabel\tcontigID\tstart\tstop\tnBases\treadCoverage\n";
    struct Item { const std::string* tx; const std::string* sci; const std::string* cg; const std::vector<size_t>* v; size_t last; };
    std::vector<Item> items;
    for (auto& t : cov) for (auto& c : t.second) items.push_back(Item{&t.first, &T.T.at(t.first).sci, &c.first, &c.second, last.at(t.first).at(c.first)});
    std::vector<std::string> txt(items.size());
    std::atomic<size_t> nx{0};
    auto work = [&] {
      for (;;) {
        const size_t i = nx.fetch_add(1);
        if (i >= items.size()) return;
        const Item& it = items[i]; std::string& s = txt[i];
        s.reserve(it.v->size() * (it.tx->size() + it.sci->size() + it.cg->size() + 40));
        for (size_t wi = 0; wi < it.v->size(); ++wi) {
          const size_t wl = wi + 1 == it.v->size() ? it.last : W;
          s += *it.tx; s += '\t'; s += *it.sci; s += '\t'; s += *it.cg; s += '\t'; append_uint(s, wi * W); s += '\t'; append_uint(s, (wi + 1) * W - 1); s += '\t';
          append_uint(s, (*it.v)[wi]); s += '\t'; append_g6(s, (double)(*it.v)[wi] / (double)wl); s += '\n';
        }
      }
    };
    std::vector<std::thread> pool;
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::max(1u, mm::cpu_budget() / 4), items.size()}));
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    for (auto& s : txt) o.write(s.data(), (std::streamsize)s.size());
  }
};

// Regularised incomplete beta I_x(a, b) by the continued fraction (modified Lentz), used for the binomial tail below.
double reg_inc_beta(double a, double b, double x) {
  if (x <= 0) return 0;
  if (x >= 1) return 1;
  if (x > (a + 1) / (a + b + 2)) return 1 - reg_inc_beta(b, a, 1 - x);
  const double lead = std::exp(std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b) + a * std::log(x) + b * std::log1p(-x)) / a;
  const double tiny = 1e-300;
  double f = 1, c = 1, d = 0;
  for (int i = 0; i <= 100000; ++i) {
    const int m = i / 2;
    double num;
    if (i == 0) num = 1;
    else if (i % 2 == 0) num = (m * (b - m) * x) / ((a + 2.0 * m - 1) * (a + 2.0 * m));
    else num = -((a + m) * (a + b + m) * x) / ((a + 2.0 * m) * (a + 2.0 * m + 1));
    d = 1 + num * d; if (std::fabs(d) < tiny) d = tiny; d = 1 / d;
    c = 1 + num / c; if (std::fabs(c) < tiny) c = tiny;
    const double cd = c * d;
    f *= cd;
    if (std::fabs(1 - cd) < 1e-16) break;
  }
  return lead * (f - 1);
}
// P(X <= k), X ~ Binomial(n, p)  (boost::math::cdf(binomial_distribution, k), fEM.h:1107)
double binomial_cdf(double n, double p, double k) {
  if (k >= n || p <= 0) return 1;
  if (p >= 1) return 0;
  return reg_inc_beta(n - k, k + 1, 1 - p);
}

// .EM.evidenceUnknownSpecies (fEM.h:846-1132): per taxon with best mappings, (1) a one-degree-of-freedom chi-square test of
// the share of its reads whose identity lies in the bottom third of the best-identity taxon's distribution, (2) the
// number of zero-coverage windows among the "usable" ones (at least a maximum read length of N-poor windows on both
// sides; N counts per 1000-bp window come from DBDIR/contigNstats_windowSize_1000.txt, :1421-1470) against a Poisson
// expectation.  Integer arithmetic as in the reference (size_t, including the wrapped last-window length kept by
// ContigCoverage).  The reference asserts when the contigNstats file is missing (:1427) or an expected count is zero
// (:1049-1050): here the file is skipped with a warning, respectively the row's identity columns are "NA".
bool write_unknown_species(const std::string& fn, const std::string& db, const Taxonomy& T, const ContigCoverage& C,
                           const std::map<std::string, std::vector<double>>& idents, long long maxReadLen, size_t minReads) {
  std::ifstream ns(db + "/contigNstats_windowSize_" + std::to_string(C.W) + ".txt");
  if (!ns.is_open()) return false;
  struct PerTaxon { size_t windows = 0, usable = 0, usableReads = 0, usableZero = 0; };
  std::map<std::string, PerTaxon> G;
  std::set<std::string> seenContigs;
  const size_t need = (size_t)maxReadLen;
  std::string ln;
  while (std::getline(ns, ln)) {
    while (!ln.empty() && (ln.back() == '\r' || ln.back() == '\n')) ln.pop_back();
    if (ln.empty()) continue;
    auto fl = split(ln, "\t");
    if (fl.size() != 3) die("Format error " + db + "/contigNstats_windowSize_1000.txt; wrong number of fields:\n" + ln);
    auto ct = C.cov.find(fl[0]);
    if (ct == C.cov.end() || !ct->second.count(fl[1])) continue;
    if (!seenContigs.insert(fl[1]).second) die("contigNstats: duplicate contig " + fl[1]);
    const std::vector<size_t>& nreads = C.reads.at(fl[0]).at(fl[1]);
    auto nf = split(fl[2], ";");
    if (nf.size() != nreads.size()) die("contigNstats: " + fl[1] + " has " + std::to_string(nf.size()) + " windows, expected " + std::to_string(nreads.size()));
    const size_t nw = nf.size(), lastLen = C.last.at(fl[0]).at(fl[1]);
    std::vector<uint8_t> poor(nw);                               // window has <= 2 % N
    for (size_t i = 0; i < nw; ++i) poor[i] = (double)std::stoull(nf[i]) / (double)(i + 1 == nw ? lastLen : C.W) <= 0.02;
    std::vector<size_t> before(nw), after(nw);                   // N-poor bases running up to / following each window
    size_t run = 0;
    for (size_t i = 0; i < nw; ++i) { before[i] = run; if (poor[i]) run += i + 1 == nw ? lastLen : C.W; else run = 0; }
    run = 0;
    for (size_t i = nw; i-- > 0;) { after[i] = run; if (poor[i]) run += i + 1 == nw ? lastLen : C.W; else run = 0; }
    PerTaxon& g = G[fl[0]];
    g.windows += nw;
    for (size_t i = 0; i < nw; ++i) if (before[i] >= need && after[i] >= need) { ++g.usable; g.usableReads += nreads[i]; g.usableZero += nreads[i] == 0; }
  }
  for (auto& t : C.cov) for (auto& c : t.second) if (!seenContigs.count(c.first)) die("Missing entry " + c.first + " in " + db + "/contigNstats_windowSize_1000.txt");

  // reference distribution: the taxon with the highest median identity among those with enough reads (:846-890)
  bool haveRef = false; double refMedian = 0, cut = 0, cutP = 0;
  for (auto& e : idents) {
    if (e.second.size() < 3 || e.second.size() < minReads) continue;
    std::vector<double> v = e.second; std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    if (haveRef && !(med > refMedian)) continue;
    haveRef = true; refMedian = med;
    cut = v.at((size_t)(v.size() * (1.0 / 3.0)));
    cutP = (double)(std::upper_bound(v.begin(), v.end(), cut) - v.begin()) / (double)v.size();
  }

  std::ofstream o(fn);
  o << "taxonID\tspecies\tgenus\tnReads\tpropBottomThirdReadIdentities\texpectedPropBottomThirdReadIdentities\tpValue_BottomThirdReadIdentities\t"
       "coverageWindows_totalGenome\tcoverageWindows_usable\tcoverageWindows_usable_averageCoverage\tcoverageWindows_usable_coverageIsZero\t"
       "coverageWindows_usable_coverageIsZero_expected\tcoverageWindows_usable_coverageIsZero_P\n";
  for (auto& e : idents) {
    const size_t n = e.second.size();
    std::string c5 = "NA", c6 = "NA", c7 = "NA", c10 = "NA", c12 = "NA", c13 = "NA";
    if (haveRef) {
      size_t low = 0; for (double v : e.second) low += v <= cut;
      const double expLow = cutP * n, expRest = n - expLow;
      if (expLow > 0 && expRest > 0) {
        const double dl = (double)low - expLow, dr = (double)(n - low) - expRest;
        const double stat = dl * dl / expLow + dr * dr / expRest;
        c5 = std::to_string((double)low / (double)n);
        c6 = std::to_string(cutP);
        c7 = std::to_string(1 - std::erf(std::sqrt(stat / 2)));   // 1 - cdf(chi_squared(1), stat)
      } else std::cerr << "evidenceUnknownSpecies: expected count of zero for taxon " << e.first << " (the reference asserts here); identity columns NA\n";
    }
    const PerTaxon& g = G.at(e.first);
    if (g.usable > 0) {
      const double avg = (double)g.usableReads / (double)g.usable;
      c10 = std::to_string(avg);
      if (avg == 0) { c12 = std::to_string(g.usable); c13 = std::to_string(1); }
      else {
        const double p0 = std::exp(-avg);                        // Poisson(avg) mass at zero
        c12 = std::to_string(g.usable * p0);
        c13 = std::to_string(g.usableZero > 0 ? 1 - binomial_cdf((double)g.usable, p0, (double)(g.usableZero - 1)) : 1.0);
      }
    }
    auto up = T.upward_by_ranks(e.first, {"species", "genus"});
    o << e.first << "\t" << up.at("species") << "\t" << up.at("genus") << "\t" << n << "\t" << c5 << "\t" << c6 << "\t" << c7 << "\t" << g.windows << "\t"
      << g.usable << "\t" << c10 << "\t" << g.usableZero << "\t" << c12 << "\t" << c13 << "\n";
  }
  return true;
}

// ------------------------------------------------------------------------------------------------------
// The EM loop of classify across devices (meta::doEM, fEM.h:501-661).  The reads are sharded contiguously — rank order = read
// order, as the reference shards them over OpenMP threads (:1229) —, every rank computes the per-taxon posterior sums and the
// log-likelihood of its reads, the sums of the ranks are added (the merge of the per-thread sums, :583-600), and every rank
// normalises and evaluates the stop rule (:624-639) on identical values.
//   Rccl  one ncclAllReduce(f64, T+1) per iteration inside the device-resident loop (mm_em_run / mm_em_continue): the production path
//   Host  each rank's partial sums (mm_em_iterate) added on the host in rank order — what the all-reduce delivers —; several ranks
//         may then share one device, which is how everything AROUND the collective is tested on a one-GPU box (--em-host-reduce)
//   None  one rank, no communicator
struct EmShard { size_t lo = 0, hi = 0, e0 = 0; std::vector<int64_t> soff; };   // reads [lo, hi); e0: first mapping of the shard; soff: shard-local offsets
EmShard em_shard(const std::vector<int64_t>& off, size_t G, size_t d) {
  const size_t NR = off.size() - 1, base = NR / G, rem = NR % G;
  EmShard s;
  s.lo = d * base + std::min(d, rem); s.hi = s.lo + base + (d < rem ? 1 : 0);
  s.e0 = (size_t)off[s.lo];
  s.soff.resize(s.hi - s.lo + 1);
  for (size_t i = 0; i <= s.hi - s.lo; ++i) s.soff[i] = off[s.lo + i] - off[s.lo];
  return s;
}
struct ThreadBarrier {
  std::mutex m; std::condition_variable cv; const size_t n; size_t waiting = 0, gen = 0;
  explicit ThreadBarrier(size_t n_) : n(n_) {}
  void wait() { std::unique_lock<std::mutex> lk(m); const size_t g = gen; if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g; }); }
};
void print_em_round(long long it, double ll, double ll_prev) {  // the per-round lines of the reference's log (fEM.h:503, :602-603, :631-632)
  std::cout << "EM round " << it << std::endl << "\n\tLog likelihood: " << ll << std::endl;
  if (it > 0) std::cout << "\tImprovement: " << ll - ll_prev << "\n\tRelative   : " << ll / ll_prev << std::endl;
}
// f: start frequencies in, final frequencies out; post[mapping], best[read] (index into the whole mapping list) out
void run_em_sharded(const std::vector<Dev>& devs, EmReduce reduce, const std::vector<int64_t>& off, const std::vector<int32_t>& taxon,
                    const std::vector<double>& mapq, const std::vector<double>& inv, size_t NT, std::vector<double>& f,
                    std::vector<double>& post, std::vector<int64_t>& best) {
  const size_t G = devs.size();
  if (reduce == EmReduce::None && G != 1) die("internal error: several EM ranks without a reduction");
  char comm_id[MM_COMM_ID_BYTES];
  if (reduce == EmReduce::Rccl && mm_comm_unique_id(comm_id) != MM_OK) die("RCCL: cannot create a communicator id");
  const std::vector<double> f0 = f;
  std::vector<std::vector<double>> part(G, std::vector<double>(NT + 1, 0.0));   // Host: the ranks' partial sums of one iteration
  std::vector<double> f_cur = f0; bool host_stop = false; double ll_prev = 0;
  ThreadBarrier bar(G);
  const long long MAX_ITER = getenv("MM_EM_MAX_ITER") ? atoll(getenv("MM_EM_MAX_ITER")) : LLONG_MAX;   // (test hook; the reference has no cap)
  const int SLICE = getenv("MM_EM_SLICE") ? std::max(1, atoi(getenv("MM_EM_SLICE"))) : 1024;           // iterations per device-resident call (test hook)
  on_each(G, [&](size_t d) {
    mm_ctx* ctx = devs[d].ctx;
    if (reduce == EmReduce::Rccl) ck(ctx, mm_comm_init(ctx, comm_id, (int)d, (int)G), "RCCL communicator");
    const EmShard sh = em_shard(off, G, d);
    const size_t n = sh.hi - sh.lo;
    mm_em* em; ck(ctx, mm_em_create(ctx, (int64_t)n, sh.soff.data(), taxon.data() + sh.e0, mapq.data() + sh.e0, inv.data() + sh.e0, (int32_t)NT, &em), "em");
    std::vector<double> fl(NT);
    if (reduce != EmReduce::Host) {
      // the loop itself runs on the device (E step, sums, all-reduce, normalisation and the stop rule per iteration, no host round
      // trip), in slices of <= 1024 iterations so that every round's log-likelihood reaches the log as in the reference
      std::vector<double> lls((size_t)std::min(SLICE, 1024));
      long long done = 0; double prev = 0;
      for (bool first = true;; first = false) {
        int n_iter = 0, stopped = 0;
        const int want = (int)std::min<long long>((long long)lls.size(), MAX_ITER - done);
        if (want <= 0) break;
        if (first) { ck(ctx, mm_em_run(em, f0.data(), want, fl.data(), lls.data(), (int)lls.size(), &n_iter), "em"); stopped = n_iter < want; }
        else ck(ctx, mm_em_continue(em, want, fl.data(), lls.data(), (int)lls.size(), &n_iter, &stopped), "em");
        if (d == 0) for (int it = 0; it < n_iter; ++it) { print_em_round(done + it, lls[(size_t)it], prev); prev = lls[(size_t)it]; }
        done += n_iter;
        if (stopped || n_iter == 0) break;
      }
    } else {
      for (long long it = 0; it < MAX_ITER; ++it) {
        ck(ctx, mm_em_iterate(em, f_cur.data(), part[d].data(), &part[d][NT]), "em");
        bar.wait();
        if (d == 0) {                                            // the sum over the ranks, in rank order; normalisation (fEM.h:606-615); stop rule (:624-639)
          std::vector<double> tot(NT + 1, 0.0);
          for (size_t g = 0; g < G; ++g) for (size_t t = 0; t <= NT; ++t) tot[t] += part[g][t];
          double sum = 0; for (size_t t = 0; t < NT; ++t) sum += tot[t];
          for (size_t t = 0; t < NT; ++t) f_cur[t] = tot[t] / sum;
          const double ll = tot[NT];
          print_em_round(it, ll, ll_prev);
          if (it > 0 && (ll - ll_prev) <= 1 && (1 - ll / ll_prev) < 0.0001) host_stop = true;
          ll_prev = ll;
        }
        bar.wait();
        if (host_stop) break;
      }
      fl = f_cur;
    }
    std::vector<int64_t> bl(n);
    ck(ctx, mm_em_posteriors(em, fl.data(), post.data() + sh.e0, bl.data()), "posteriors");
    for (size_t i = 0; i < n; ++i) best[sh.lo + i] = bl[i] < 0 ? -1 : bl[i] + (int64_t)sh.e0;   // rank-local index -> index into the whole mapping list
    mm_em_destroy(em);
    bar.wait();                                                  // (every rank has read f_cur)
    if (d == 0) f = fl;
  });
}

// One `classify` of one mappings file (meta::doEM, fEM.h:466-803): the file read and tokenised, the database's tables, the EM on the devices, every output
// file.  The stages are the methods, in the order run() calls them.  (Until round 5 one 280-line function.)
struct ClassifyRun {
  const std::vector<Dev>& devs; const EmReduce reduce; const std::string& mapped; const std::string& db; const size_t minReadsU;
  const std::function<void()>& leave_now;      // the last file: called once everything is written; the process ends there (may be empty)
  const std::function<void()>& need_devices;   // called before the first device call: the contexts are created beside the parsing of the file (may be empty)
  PhaseClock pc;
  const unsigned HW = mm::cpu_budget();                        // CPUs this process may keep busy (cpu_budget.hpp: a container's quota counts, not the 256 the machine shows)
  const unsigned WIDE = std::max(1u, HW - std::max(1u, HW / 8));   // width of the pools that compute flat out: under a CPU quota (16 CPUs' worth of time per 100 ms) sixteen such threads plus
                                                               // whatever else runs use the period up, and every thread of the process is stopped for the rest of it (bench: c1 0.11 -> 0.21 s)
  struct TextBuf {                                               // the file's bytes + a terminating 0, not zero-filled first (std::string::resize spent 0.1 s on that per 0.5 GB)
    char* p = nullptr; size_t n = 0;
    void resize(size_t k) { p = new (std::nothrow) char[k + 1]; if (!p) die("out of host memory for the mappings file"); n = k; p[k] = 0; }   // (huge_new.hpp: on huge pages)
    size_t size() const { return n; } const char* c_str() const { return p; } char& operator[](size_t i) { return p[i]; }
    ~TextBuf() { delete[] p; }
  };
  TextBuf text;
  struct MapLine { const char* p; uint32_t last_space, n; int contig; long long len; size_t start, stop; double ident, mapq; };   // [p, p + n): the line; p + last_space: the blank before field 14
  const KeptLines* kept = nullptr;                                // mapDirectly --then-classify: the lines in memory (no file is read)
  std::vector<MapLine> lines; std::vector<int64_t> off{0};       // read r owns lines [off[r], off[r+1])
  std::vector<std::string> contig_id; std::unordered_map<std::string, int> contig_index;
  size_t NRD = 0;
  std::vector<std::string> contig_taxon_id; std::set<std::string> taxaSet;
  std::map<std::string, size_t> st; size_t nUnmapped = 0, nTooShort = 0, nTotal = 0;
  std::map<std::string, std::map<std::string, size_t>> TI;       // fEM.h:1320-1364
  std::unique_ptr<Taxonomy> tax;
  std::vector<std::string> taxa;
  std::vector<int> contig_tx; std::vector<long long> contig_len_ti;   // per contig: taxon index; length per taxonInfo (-1: not listed)
  std::vector<int32_t> taxon; std::vector<double> mapq, inv;          // per mapping
  std::vector<double> f, post; std::vector<int64_t> best;

  ClassifyRun(const std::vector<Dev>& devs_, EmReduce reduce_, const std::string& mapped_, const std::string& db_, size_t minReadsU_, const std::function<void()>& leave_now_,
              const std::function<void()>& need_devices_) : devs(devs_), reduce(reduce_), mapped(mapped_), db(db_), minReadsU(minReadsU_), leave_now(leave_now_), need_devices(need_devices_) {}

  // The mappings file once through: every line is tokenised where it lies (the reference splits every line again in every EM round,
  // fEM.h:1171-1214, :234-373), lines of one read are consecutive (mapWrap.h:128-149), contig IDs are interned.
  // Round 4: read and tokenised by several threads — pieces of the file that begin on a read boundary are parsed on their own and joined in
  // file order (read offsets shifted, contig IDs interned in the order a single pass would meet them): 4.2 M lines took 1.3 s on one thread.
  void read_file() {
    {
      const int fd = ::open(mapped.c_str(), O_RDONLY);
      if (fd < 0) die("Cannot open mappings file " + mapped);
      struct stat stt; if (fstat(fd, &stt) != 0) die("Cannot open mappings file " + mapped);
      text.resize((size_t)stt.st_size);
      const size_t PIECE = (size_t)32 << 20, np = (text.size() + PIECE - 1) / PIECE;
      std::atomic<size_t> nx{0}; std::atomic<bool> bad{false};
      auto rd = [&] { for (;;) { const size_t i = nx.fetch_add(1); if (i >= np) return; size_t a0 = i * PIECE; const size_t e0 = std::min(text.size(), a0 + PIECE);
                        while (a0 < e0) { const ssize_t g = pread(fd, &text[a0], e0 - a0, (off_t)a0); if (g <= 0) { bad = true; return; } a0 += (size_t)g; } } };
      std::vector<std::thread> pool; for (unsigned t = 1; t < std::min<unsigned>({8u, HW, (unsigned)std::max<size_t>(np, 1)}); ++t) pool.emplace_back(rd);
      rd(); for (auto& t : pool) t.join();
      ::close(fd);
      if (bad) die("Cannot read mappings file " + mapped);
    }
  }
  void tokenise() {
    {
      const char* const T0 = text.c_str();
      const size_t TS = text.size();
      // the read ID of the line that starts at p (text up to the first blank or the line's end)
      auto id_of = [&](size_t p, size_t& len) { const char* nl = (const char*)memchr(T0 + p, '\n', TS - p); const size_t e = nl ? (size_t)(nl - T0) : TS;
                                                const char* sp = (const char*)memchr(T0 + p, ' ', e - p); len = (sp ? (size_t)(sp - T0) : e) - p; };
      auto next_line = [&](size_t p) { const char* nl = (const char*)memchr(T0 + p, '\n', TS - p); return nl ? (size_t)(nl - T0) + 1 : TS; };
      // first read boundary at or after x: a line start whose ID differs from the ID of the last non-empty line before it
      auto read_boundary = [&](size_t x) {
        if (x == 0) return (size_t)0;
        size_t p = next_line(x - 1);                                // start of the first line that begins at or after x
        while (p < TS) {
          if (T0[p] == '\n') { ++p; continue; }                     // empty line
          size_t q = p;                                            // start of the previous non-empty line
          for (;;) { if (q == 0) return p; size_t e = q - 1; size_t b0 = e; while (b0 > 0 && T0[b0 - 1] != '\n') --b0; if (e > b0) { q = b0; break; } q = b0; }
          size_t la, lb; id_of(p, la); id_of(q, lb);
          if (la != lb || memcmp(T0 + p, T0 + q, la) != 0) return p;
          p = next_line(p);
        }
        return TS;
      };
      // (MM_CLASSIFY_THREADS=n: exactly n pieces, whatever the size of the file — the tests cut small files into many)
      const size_t NTH = getenv("MM_CLASSIFY_THREADS") ? (size_t)std::min(256, std::max(1, atoi(getenv("MM_CLASSIFY_THREADS"))))
                                                       : std::max<size_t>(1, std::min<size_t>({(size_t)32, (size_t)WIDE, TS / ((size_t)4 << 20) + 1}));
      std::vector<size_t> cut(NTH + 1, TS);
      cut[0] = 0;
      for (size_t t = 1; t < NTH; ++t) cut[t] = std::max(cut[t - 1], read_boundary(TS / NTH * t));
      struct Piece { std::vector<MapLine> lines; std::vector<int64_t> starts; std::vector<std::string> cid; std::unordered_map<std::string, int> cix; };
      std::vector<Piece> pieces(NTH);
      auto parse_piece = [&](size_t t) {
        Piece& P = pieces[t];
        size_t cur_beg = 0, cur_len = (size_t)-1;                  // the current read's ID, as a span of `text`
        for (size_t p = cut[t]; p < cut[t + 1];) {
          const char* nl = (const char*)memchr(T0 + p, '\n', cut[t + 1] - p);
          const size_t e = nl ? (size_t)(nl - T0) : cut[t + 1];
          if (e == p) { p = e + 1; continue; }                     // empty line
          size_t fb[16], fe[16]; int nf = 0;                       // fields (single blanks, util.h:80)
          for (size_t q = p; nf < 16;) { const char* sp = (const char*)memchr(T0 + q, ' ', e - q); fb[nf] = q; fe[nf] = sp ? (size_t)(sp - T0) : e; ++nf; if (!sp) break; q = fe[nf - 1] + 1; }
          if (nf < 6) die("File " + mapped + " has weird format - is this a mappings file generated by MetaMap?");
          if (nf < 14) die("File " + mapped + " has lines with fewer than 14 fields - is this a mappings file generated by MetaMap?");
          if (fe[0] - fb[0] != cur_len || memcmp(T0 + fb[0], T0 + cur_beg, cur_len) != 0) { P.starts.push_back((int64_t)P.lines.size()); cur_beg = fb[0]; cur_len = fe[0] - fb[0]; }
          MapLine L{};
          L.p = T0 + p; L.n = (uint32_t)(e - p); L.last_space = (uint32_t)(fb[13] - 1 - p);
          std::string cid(T0 + fb[5], fe[5] - fb[5]);
          auto it = P.cix.find(cid);
          if (it == P.cix.end()) { it = P.cix.emplace(cid, (int)P.cid.size()).first; P.cid.push_back(cid); }
          L.contig = it->second;
          L.len = strtoll(T0 + fb[1], nullptr, 10);
          L.start = strtoull(T0 + fb[7], nullptr, 10); L.stop = strtoull(T0 + fb[8], nullptr, 10);
          L.ident = strtod(T0 + fb[9], nullptr) / 100.0;
          { errno = 0; char* endp = nullptr; L.mapq = strtod(T0 + fb[13], &endp);   // std::stod: out of range (also a denormal) throws; the reference then takes 0 for "…e-…" (fEM.h:269-275)
            if (errno == ERANGE) { if (std::string(T0 + fb[13], fe[13] - fb[13]).find("e-") != std::string::npos) L.mapq = 0; else die("mapping quality out of range in " + mapped); }
            if (endp == T0 + fb[13]) die("File " + mapped + " has a mapping quality that is not a number"); }
          P.lines.push_back(L);
          p = e + 1;
        }
      };
      { std::vector<std::thread> pool; for (size_t t = 1; t < NTH; ++t) pool.emplace_back(parse_piece, t); parse_piece(0); for (auto& th : pool) th.join(); }
      // join: contig IDs in the order of their first line, read offsets shifted by the lines before the piece
      std::vector<std::vector<int>> remap(NTH);
      std::vector<size_t> line0(NTH + 1, 0);
      for (size_t t = 0; t < NTH; ++t) {
        line0[t + 1] = line0[t] + pieces[t].lines.size();
        remap[t].resize(pieces[t].cid.size());
        for (size_t c = 0; c < pieces[t].cid.size(); ++c) {
          auto it = contig_index.find(pieces[t].cid[c]);
          if (it == contig_index.end()) { it = contig_index.emplace(pieces[t].cid[c], (int)contig_id.size()).first; contig_id.push_back(pieces[t].cid[c]); }
          remap[t][c] = it->second;
        }
      }
      lines.resize(line0[NTH]);
      off.clear();
      for (size_t t = 0; t < NTH; ++t) for (int64_t st0 : pieces[t].starts) off.push_back(st0 + (int64_t)line0[t]);
      if (off.empty()) off.push_back(0);
      auto place = [&](size_t t) { MapLine* o = lines.data() + line0[t]; const auto& src = pieces[t].lines; for (size_t i = 0; i < src.size(); ++i) { o[i] = src[i]; o[i].contig = remap[t][(size_t)src[i].contig]; } };
      { std::vector<std::thread> pool; for (size_t t = 1; t < NTH; ++t) pool.emplace_back(place, t); place(0); for (auto& th : pool) th.join(); }
      if (!lines.empty()) off.push_back((int64_t)lines.size());
    }
    NRD = off.size() - 1;
  }
  // the taxa of the mapped contigs, PREFIX.meta, DB/taxonInfo.txt
  void read_tables() {
    contig_taxon_id.assign(contig_id.size(), std::string());
    for (size_t c = 0; c < contig_id.size(); ++c) { contig_taxon_id[c] = extract_taxon(contig_id[c]); taxaSet.insert(contig_taxon_id[c]); }
    if (taxaSet.empty()) die("No relevant taxon IDs found in your mappings file - is it possible that none of your reads are mapped?");
    { std::ifstream s(mapped + ".meta"); if (!s.is_open()) die("The file " + mapped + ".meta is not present or could not be opened - this file is generated automatically as part of the mapping process, so please check whether the mapping process finished successfully.");
      std::string a; size_t b; while (s >> a >> b) st[a] = b; }
    nUnmapped = st.at("ReadsNotMapped"); nTooShort = st.at("ReadsTooShort"); nTotal = st.at("TotalReads");
    { std::ifstream s(db + "/taxonInfo.txt"); if (!s.is_open()) die("Could not open file " + db + "/taxonInfo.txt -- perhaps you have specified an incomplete DB?");
      std::string ln; while (std::getline(s, ln)) { if (ln.empty()) continue; auto f = split(ln, " "); for (auto& c : split(f.at(1), ";")) { auto kv = split(c, "="); TI[f.at(0)][kv.at(0)] = std::stoull(kv.at(1)); } } }
  }
  void per_mapping_fields() {
    taxa.assign(taxaSet.begin(), taxaSet.end());
    std::map<std::string, int> tindex; for (size_t i = 0; i < taxa.size(); ++i) tindex[taxa[i]] = (int)i;
    // per mapping: taxon, quality, 1/nLoc (getMappingLocations, fEM.h:234-353).  nLoc(read, taxon) = sum over the taxon's contigs of
    // (len - L + 1) if len >= L, else 1 if the read has a mapping on that contig (:325-348): sorted lengths + suffix sums per taxon
    contig_tx.assign(contig_id.size(), 0); contig_len_ti.assign(contig_id.size(), -1);
    struct TaxLens { std::vector<long long> len, suffix; };
    std::vector<TaxLens> tl(taxa.size());
    for (size_t t = 0; t < taxa.size(); ++t) {
      auto it = TI.find(taxa[t]);
      if (it == TI.end()) die("Unknown taxonID '" + taxa[t] + "'; please check that your mappings file was mapped against the database now specified.");
      for (auto& c : it->second) tl[t].len.push_back((long long)c.second);
      std::sort(tl[t].len.begin(), tl[t].len.end());
      tl[t].suffix.assign(tl[t].len.size() + 1, 0);
      for (size_t i = tl[t].len.size(); i-- > 0;) tl[t].suffix[i] = tl[t].suffix[i + 1] + tl[t].len[i];
    }
    for (size_t c = 0; c < contig_id.size(); ++c) {
      contig_tx[c] = tindex.at(contig_taxon_id[c]);
      auto& m = TI.at(contig_taxon_id[c]); auto it = m.find(contig_id[c]);
      if (it != m.end()) contig_len_ti[c] = (long long)it->second;
    }
    taxon.assign(lines.size(), 0); mapq.assign(lines.size(), 0.0); inv.assign(lines.size(), 0.0);
    {
      std::vector<int> seen_c;                                     // distinct contigs of the current read
      for (size_t r = 0; r < NRD; ++r) {
        const size_t a0 = (size_t)off[r], b0 = (size_t)off[r + 1];
        const long long L = lines[a0].len;
        seen_c.clear();
        for (size_t i = a0; i < b0; ++i) if (std::find(seen_c.begin(), seen_c.end(), lines[i].contig) == seen_c.end()) seen_c.push_back(lines[i].contig);
        for (size_t i = a0; i < b0; ++i) {
          const int t = contig_tx[(size_t)lines[i].contig];
          const TaxLens& X = tl[(size_t)t];
          const size_t k0 = (size_t)(std::lower_bound(X.len.begin(), X.len.end(), L) - X.len.begin());
          long long n = X.suffix[k0] - (long long)(X.len.size() - k0) * (L - 1);
          for (int c : seen_c) if (contig_tx[(size_t)c] == t && contig_len_ti[(size_t)c] >= 0 && contig_len_ti[(size_t)c] < L) ++n;
          taxon[i] = t; mapq[i] = lines[i].mapq; inv[i] = 1 / (double)(size_t)n;
        }
      }
    }
  }
  void em() {
    const size_t NT = taxa.size(), NR = NRD;
    f.assign(NT, 1 / (double)NT);
    post.assign(taxon.size(), 0.0); best.assign(NR, 0);
    std::cout << "Starting EM..." << std::endl;
    if (need_devices) need_devices();
    run_em_sharded(devs, reduce, off, taxon, mapq, inv, NT, f, post, best);
  }
  void write_outputs() {
    Taxonomy& T = *tax;
    std::cout << "Outputting mappings with adjusted alignment qualities." << std::endl;
    std::ofstream emf(mapped + ".EM"), r2t(mapped + ".EM.reads2Taxon"), kr(mapped + ".EM.reads2Taxon.krona"), li(mapped + ".EM.lengthAndIdentitiesPerMappingUnit");
    li << "AnalysisLevel\tID\treadI\tIdentity\tLength\n";
    std::map<std::string, size_t> readsPer;
    ContigCoverage coverage;
    std::map<std::string, std::vector<double>> identsPerTaxon;     // :691, :718
    long long maxReadLen = -1;                                     // :692, :719-722
    std::thread side_files; bool unknown_written = true;
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } side_join{side_files};
    {
      // the four per-read / per-line files: ranges of reads formatted by several threads into their own buffers, written in read order
      // (4.2 M lines through std::to_string on one thread took 1.2 s); the per-taxon tallies and the coverage windows follow in read order
      std::vector<std::string> tax_nonx(taxa.size());              // getFirstNonXNode per taxon (taxonomy.h:51-74), once
      for (size_t t = 0; t < taxa.size(); ++t) tax_nonx[t] = T.first_non_x(taxa[t]);
      const size_t NTH = getenv("MM_CLASSIFY_THREADS") ? (size_t)std::min(256, std::max(1, atoi(getenv("MM_CLASSIFY_THREADS"))))
                                                       : std::max<size_t>(1, std::min<size_t>({(size_t)32, (size_t)WIDE, lines.size() / 50000 + 1}));
      std::vector<size_t> rcut(NTH + 1, NRD);
      rcut[0] = 0;
      { size_t t = 1; for (size_t r = 0; r < NRD && t < NTH; ++r) if ((uint64_t)off[r] >= (uint64_t)lines.size() * t / NTH) rcut[t++] = r; }
      struct Out { std::string em, r2, kr, li; };
      std::vector<Out> outs(NTH);
      auto fmt = [&](size_t t) {
        Out& O = outs[t];
        const size_t r0 = rcut[t], r1 = rcut[t + 1];
        if (r1 <= r0) return;
        { size_t bytes = 0; for (size_t i = (size_t)off[r0]; i < (size_t)off[r1]; ++i) bytes += lines[i].n + 5; O.em.reserve(bytes + 64); }
        char num[64];
        for (size_t r = r0; r < r1; ++r) {                         // fEM.h:684-779
          for (size_t i = (size_t)off[r]; i < (size_t)off[r + 1]; ++i) {   // the line with field 14 replaced by std::to_string(posterior) (:705)
            O.em.append(lines[i].p, (size_t)lines[i].last_space + 1);
            append_f6(O.em, post[i]);
            O.em += '\n';
          }
          const size_t b = (size_t)best[r];
          const MapLine& B = lines[b];
          const std::string& cg = contig_id[(size_t)B.contig];
          const size_t rid_len = (size_t)((const char*)memchr(B.p, ' ', B.n) - B.p);
          O.li += "EqualCoverageUnit\t"; O.li += cg; O.li += '\t';
          snprintf(num, sizeof num, "%zu\t%g\t%lld\n", r, B.ident, B.len); O.li += num;                  // :711
          O.r2.append(B.p, rid_len); O.r2 += '\t'; O.r2 += taxa[(size_t)taxon[b]]; O.r2 += '\n';
          O.kr.append(B.p, rid_len); O.kr += '\t'; O.kr += tax_nonx[(size_t)taxon[b]];
          snprintf(num, sizeof num, "\t%g\n", post[b]); O.kr += num;
        }
      };
      std::vector<std::thread> pool;
      for (size_t t = 1; t < NTH; ++t) pool.emplace_back(fmt, t);
      // meanwhile, on this thread: tallies per taxon and coverage windows, in read order (taxon and contig by index, strings only at the end)
      std::vector<size_t> readsPerIdx(taxa.size(), 0);
      std::vector<std::vector<double>> identsIdx(taxa.size());
      std::vector<ContigCoverage::Slot> cslot(contig_id.size());
      fmt(0);
      for (size_t r = 0; r < NRD; ++r) {                           // the window vectors of every contig with a best mapping (map insertions: one thread)
        const MapLine& B = lines[(size_t)best[r]];
        const size_t tx = (size_t)taxon[(size_t)best[r]];
        maxReadLen = std::max(maxReadLen, B.len);
        if (contig_len_ti[(size_t)B.contig] < 0) die("contig " + contig_id[(size_t)B.contig] + " is not listed for taxon " + taxa[tx] + " in " + db + "/taxonInfo.txt");
        ContigCoverage::Slot& sl = cslot[(size_t)B.contig];
        if (!sl.v) sl = coverage.slot(taxa[tx], contig_id[(size_t)B.contig], (size_t)contig_len_ti[(size_t)B.contig]);
      }
      {                                                            // tallies: thread k owns the taxa and the contigs with index % NT2 == k and walks the reads in order
        const size_t NT2 = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)std::max(1u, HW / 2), NRD / 20000 + 1}));
        auto tally = [&](size_t k) {
          for (size_t r = 0; r < NRD; ++r) {
            const size_t b = (size_t)best[r];
            const MapLine& B = lines[b];
            const size_t tx = (size_t)taxon[b];
            if (tx % NT2 == k) { readsPerIdx[tx]++; identsIdx[tx].push_back(B.ident); }
            if ((size_t)B.contig % NT2 == k) coverage.add(cslot[(size_t)B.contig], (size_t)contig_len_ti[(size_t)B.contig], B.start, B.stop);
          }
        };
        std::vector<std::thread> tp;
        for (size_t k = 1; k < NT2; ++k) tp.emplace_back(tally, k);
        tally(0);
        for (auto& th : tp) th.join();
      }
      for (size_t t = 0; t < taxa.size(); ++t) if (readsPerIdx[t]) { readsPer[taxa[t]] = readsPerIdx[t]; identsPerTaxon[taxa[t]] = std::move(identsIdx[t]); }
      for (auto& th : pool) th.join();
      pc.lap("c5a format");
      // the two side files only read the tallies, which are complete here: they are written beside the per-read files and the WIMP (0.1 s of their own)
      side_files = std::thread([&] {
        std::thread cov_thread([&] { coverage.write(mapped + ".EM.contigCoverage", T); });
        unknown_written = write_unknown_species(mapped + ".EM.evidenceUnknownSpecies", db, T, coverage, identsPerTaxon, maxReadLen, minReadsU);
        cov_thread.join();
      });
      auto put = [&](std::ofstream& f, std::string Out::*m) { for (auto& O : outs) f.write((O.*m).data(), (std::streamsize)(O.*m).size()); };
      std::thread w1([&] { put(r2t, &Out::r2); put(kr, &Out::kr); put(li, &Out::li); });
      put(emf, &Out::em);
      w1.join();
    }
    { std::ifstream s(mapped + ".meta.unmappedReadsLengths"); std::string ln;
      while (std::getline(s, ln)) { if (ln.empty()) continue; auto fl = split(ln, "\t"); r2t << fl.at(1) << "\t" << 0 << "\n"; kr << fl.at(1) << "\t" << 0 << "\t" << 0 << "\n"; } }
    std::map<std::string, double> fmap;
    for (size_t i = 0; i < taxa.size(); ++i) fmap[taxa[i]] = f[i];
    { const double minF = 0.9 * (1.0 / (double)st.at("ReadsMapped")); std::set<std::string> drop;   // cleanF, fEM.h:1135-1163
      for (auto& e : fmap) if (e.second < minF && !readsPer.count(e.first)) drop.insert(e.first);
      for (auto& d : drop) fmap.erase(d);
      double s = 0; for (auto& e : fmap) s += e.second; for (auto& e : fmap) e.second /= s; }
    pc.lap("c5 output files");
    write_wimp(mapped + ".EM.WIMP", T, fmap, readsPer, nTotal, nUnmapped, nTooShort);
    pc.lap("c6 WIMP");
    side_files.join();
    if (!unknown_written)
      std::cerr << "Warning: " << db << "/contigNstats_windowSize_1000.txt not found - " << mapped << ".EM.evidenceUnknownSpecies is not written." << std::endl;
    pc.lap("c8 evidence of unknown species + contig coverage");
    if (leave_now && !getenv("MM_CLI_FULL_TEARDOWN")) { emf.close(); r2t.close(); kr.close(); li.close(); pc.report(); leave_now(); finish_fast(); }   // (a GB of vectors and strings: nothing left to do with them)
  }
  // mapDirectly --then-classify: the lines are in memory with their fields parsed (LineMeta) — what read_file + tokenise produce from the file, without the
  // file: read boundaries from the batches' offsets (reads without mappings have no lines), contig IDs interned in the order of their first line
  void adopt() {
    size_t total = 0; for (const auto& P : kept->parts) total += P.n_lines;
    lines.resize(total);
    off.clear();
    std::vector<int> intern(kept->cname->size(), -1);
    size_t at = 0;
    for (const auto& P : kept->parts) {
      for (size_t r = 0; r < P.n_reads; ++r) if (P.off[r + 1] > P.off[r]) off.push_back((int64_t)at + P.off[r]);
      for (size_t i = 0; i < P.n_lines; ++i) {
        const LineMeta& m = P.meta[i];
        int& ci = intern[(size_t)m.contig];
        if (ci < 0) { ci = (int)contig_id.size(); contig_id.push_back((*kept->cname)[(size_t)m.contig]); contig_index.emplace(contig_id.back(), ci); }
        lines[at + i] = MapLine{P.text + m.beg, m.ls, m.n, ci, (long long)m.len, (size_t)m.start, (size_t)((long long)m.start + m.len - 1), m.ident, m.mapq};
      }
      at += P.n_lines;
    }
    if (off.empty()) off.push_back(0);
    if (!lines.empty()) off.push_back((int64_t)lines.size());
    NRD = off.size() - 1;
  }
  int run() {
    if (kept) adopt(); else { read_file(); tokenise(); }
    read_tables();
    pc.lap("c1 read mappings + taxonInfo");
    tax = std::make_unique<Taxonomy>(db + "/taxonomy");
    pc.lap("c2 taxonomy");
    per_mapping_fields();
    pc.lap("c3 per-mapping fields");
    em();
    pc.lap("c4 EM");
    write_outputs();
    return 0;
  }
};

int classify_one(const std::vector<Dev>& devs, EmReduce reduce, const std::string& mapped, const std::string& db, size_t minReadsU,
                 const std::function<void()>& leave_now, const std::function<void()>& need_devices, const KeptLines* kept) {
  ClassifyRun run(devs, reduce, mapped, db, minReadsU, leave_now, need_devices);
  run.kept = kept;
  return run.run();
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2 || !(std::string(argv[1]) == "index" || std::string(argv[1]) == "mapDirectly" || std::string(argv[1]) == "mapAgainstIndex" ||
                    std::string(argv[1]) == "classify" || std::string(argv[1]) == "classifyU")) {
    std::cout << "\nMetaMaps (MI355X hot path)\n\n  Simultaneous metagenomic classification and mapping.\n\nUsage:\n\n  ./metamaps mapDirectly|classify|mapAgainstIndex|index\n\n";
    return 1;
  }
  const std::string mode = argv[1];
  if (mm::env_strict()) { const std::string bad = mm::env_unknown(); if (!bad.empty()) die("unknown MM_* environment switch(es): " + bad + " (MM_STRICT_ENV is set; see INTEGRATION.md)"); }
  Options o = parse(argc, argv);
  if (mode == "mapDirectly" || mode == "index" || mode == "mapAgainstIndex") return map_mode(o, mode);
  if (mode == "classify") {
    if (!o.v.count("DB")) die("Provide path to DB.");
    if (!o.v.count("mappings")) die("Provide path to mappings.");
    const auto m0 = std::chrono::steady_clock::now();
    auto since = [&](const char* what) { if (getenv("MM_CLI_TIMING")) std::cerr << "INFO, main: " << what << " at +" << std::chrono::duration<double>(std::chrono::steady_clock::now() - m0).count() << " s\n"; };
    std::vector<Dev> devs;
    for (int p : device_list(o, false)) { Dev d; d.phys = p; devs.push_back(d); }
    // the HIP runtime and the contexts (0.1 s; up to 2 s right behind a process that gave 150 GB back) come up on a thread of their own while the
    // mappings file is read and tokenised
    std::thread ctx_thread([&] {
      std::vector<int> phys; for (auto& d : devs) phys.push_back(d.phys);
      check_devices(phys);
      for (auto& d : devs) if (mm_ctx_create(d.phys, &d.ctx) != MM_OK) die("No MI355X (gfx950) device available — this build has no CPU path");
      since("contexts created");
    });
    JoinOnExit ctx_thread_guard{ctx_thread};
    const std::function<void()> need_devices = [&] {
      if (!ctx_thread.joinable()) return;
      const auto w0 = std::chrono::steady_clock::now();
      ctx_thread.join();
      if (getenv("MM_CLI_TIMING")) std::cerr << "INFO, main: waited for the contexts " << std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() << " s\n";
    };
    // an explicit --gpus 1 also goes through RCCL (one rank); --em-host-reduce: the ranks' sums are added on the host (test hook: ranks may share a device)
    const EmReduce reduce = o.em_host ? EmReduce::Host : ((devs.size() > 1 || o.v.count("gpus") || o.v.count("devices")) ? EmReduce::Rccl : EmReduce::None);
    const size_t minReadsU = o.v.count("minreads") ? std::stoull(o.v.at("minreads")) : 10000;   // parseCmdArgs.hpp:462-471
    const std::vector<std::string> files = split(o.v.at("mappings"), ",");
    for (size_t fi = 0; fi < files.size(); ++fi) {
      if (fi + 1 == files.size()) classify_one(devs, reduce, files[fi], o.v.at("DB"), minReadsU, [&] { since("mappings file done"); }, need_devices);
      else classify_one(devs, reduce, files[fi], o.v.at("DB"), minReadsU, nullptr, need_devices);
      need_devices();
      for (auto& d : devs) mm_comm_destroy(d.ctx);
      since("mappings file done");
    }
    if (!getenv("MM_CLI_FULL_TEARDOWN")) finish_fast();
    for (auto& d : devs) mm_ctx_destroy(d.ctx);
    since("contexts destroyed");
    return 0;
  }
  die("sub-command '" + mode + "' is outside the accelerated hot path (SURVEY.md §2: Boost-archive index files / disabled upstream)");
}
