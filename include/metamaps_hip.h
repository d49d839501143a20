/*
 * metamaps_hip.h — C ABI of libmetamaps_hip.so, the MI355X (gfx950) implementation of the
 * MetaMaps mapping + EM-classification hot path.
 *
 * The reference (DiltheyLab/MetaMaps) has no FFI/plugin seam: its hot path is reached through
 * three in-process C++ interfaces (paths relative to /root/reference/src):
 *   (i)   skch::Sketch(param, maxMem, callback)            map/include/winSketch.hpp:157  — index one chunk
 *   (ii)  MapModuleOutput* skch::Map::mapModule(read)      map/include/computeMap.hpp:180 — map one read
 *   (iii) meta::doEM  per-read callback + reduction        meta/fEM.h:466, :535-600       — one EM iteration
 * Each entry point below names the interface (and the loops under it) that it replaces.  Handles are
 * opaque; buffers are caller-owned plain arrays; every call returns 0 on success or a negative
 * mm_status and leaves a message in mm_last_error().  No exceptions cross this boundary, no global
 * state.  A mm_ctx is one device + one stream + a scratch allocator; it and the objects created from it are used
 * from one host thread at a time, different contexts (one per GPU of a node, or several on one GPU) may be
 * driven from different threads concurrently, and an index may be read by mm_map_batch through any context of
 * the device it lives on.  There is no CPU fallback: mm_ctx_create fails when no gfx950 device is present.
 */
#ifndef METAMAPS_HIP_H
#define METAMAPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 6   /* 3: mm_seqset_slice/concat, mm_map_batch_reusing, mm_em_continue, mm_synth_community_species;
                            * 4: mm_sketch_batch, mm_ctx_release_cached, mm_index_dup_neighbours;
                            * 5: mm_mapping_gather, mm_comm_info, mm_seqset_fetch_range;
                            * 6: mm_index_save, mm_index_load */

typedef enum {
  MM_OK = 0,
  MM_ERR_ARG = -1,        /* bad argument / unsupported parameter value          */
  MM_ERR_DEVICE = -2,     /* HIP runtime error (message has the HIP error string) */
  MM_ERR_NOMEM = -3,      /* device or host allocation failed                     */
  MM_ERR_STATE = -4,      /* call sequence violated                               */
  MM_ERR_LIMIT = -5,      /* documented capacity limit exceeded       */
  MM_ERR_NUMERIC = -6,    /* reference would abort here (e.g. likelihood sum 0, mapWrap.h:298) */
  MM_ERR_COMM = -7        /* RCCL error                                           */
} mm_status;

typedef struct mm_ctx mm_ctx;          /* a device + streams + scratch allocator                  */
typedef struct mm_seqset mm_seqset;    /* sequences resident in HBM as packed 2-bit + exceptions  */
typedef struct mm_index mm_index;      /* reference sketch of one index chunk, resident in HBM     */
typedef struct mm_mapping mm_mapping;  /* mapping results of one read batch, resident in HBM      */
typedef struct mm_em mm_em;            /* EM state (mappings x taxa) resident in HBM              */

/* ---- context -------------------------------------------------------------------------------- */
int mm_abi_version(void);
int mm_device_count(void);                               /* visible HIP devices (0 when there is none or the runtime fails) */
int mm_ctx_create(int device_id, mm_ctx** out);
void mm_ctx_destroy(mm_ctx* ctx);
const char* mm_last_error(const mm_ctx* ctx);            /* valid until the next call on ctx       */
/* device name, CU count, total HBM bytes, free HBM bytes */
int mm_ctx_device_info(mm_ctx* ctx, char* name, size_t name_cap, int* cus, uint64_t* hbm_total, uint64_t* hbm_free);
int mm_ctx_synchronize(mm_ctx* ctx);
/* Hands the context's cached free device blocks, and the device's recycled index-scale blocks, back to the driver.  The library caches
 * what it frees (a context's own allocator; index-scale blocks per device) and gives it up by itself only when an allocation fails for
 * lack of memory; a host that has finished a phase — e.g. all chunk indexes of a --maxmemory run built, worker contexts about to start
 * beside them — calls this so that the memory is there without that detour. */
int mm_ctx_release_cached(mm_ctx* ctx);
/* raw hipStream_t the kernels are launched on (for event timing by the caller) */
void* mm_ctx_stream(mm_ctx* ctx);

/* ---- sequences -------------------------------------------------------------------------------
 * Replaces the kseq buffers handed to addMinimizers (commonFunc.hpp:92; callers winSketch.hpp:269,
 * computeMap.hpp:285).  Bytes are upper-cased exactly as makeUpperCase does (commonFunc.hpp:57);
 * A/C/G/T are packed 2 bits/base, every other byte is kept verbatim in an exception run list so that
 * hashing sees the same ASCII the reference hashes. */
int mm_seqset_create(mm_ctx* ctx, mm_seqset** out);
void mm_seqset_destroy(mm_seqset* s);
int mm_seqset_add(mm_seqset* s, const char* ascii, int64_t len);   /* host staging, keeps input order */
/* same, without the copy: the caller keeps `ascii` valid and unchanged until mm_seqset_upload has returned (a parser that
 * fills one arena per batch hands its records over this way) */
int mm_seqset_add_view(mm_seqset* s, const char* ascii, int64_t len);
int mm_seqset_upload(mm_seqset* s);                                /* pack + copy to HBM; set is then frozen.  Packs into the CONTEXT's pinned
                                                                    * staging buffer: two uploads of sets of one context must not overlap (the one-thread-
                                                                    * per-context rule above applies to this entry point too) */
/* Persistent packed form of an uploaded sequence set (2-bit bases, exception runs, lengths): what `metamaps index` stores
 * per index chunk in place of the reference's Boost archive of the sketch (createIndex, mapWrap.h:358-405;
 * winSketch.hpp:73-83).  The device index is rebuilt from it in seconds (mm_index_build), nothing derived is stored. */
int mm_seqset_save(mm_seqset* set, const char* path);
int mm_seqset_load(mm_ctx* ctx, const char* path, mm_seqset** out);
/* Sequences [first, first + count) of an uploaded set as a set of their own, and several uploaded sets of one device back to back
 * as one — device-side copies of the packed stream, nothing is packed again.  The CLI uploads the reference in bounded groups as
 * its parser delivers the contigs (the reference streams contig by contig, winSketch.hpp:242-252), concatenates them and cuts
 * the index chunks of --maxmemory (winSketch.hpp:274-329) out of the resident whole. */
int mm_seqset_slice(mm_ctx* ctx, const mm_seqset* set, int64_t first, int64_t count, mm_seqset** out);
int mm_seqset_concat(mm_ctx* ctx, const mm_seqset* const* parts, int n_parts, mm_seqset** out);
int64_t mm_seqset_count(const mm_seqset* s);
int64_t mm_seqset_total_bases(const mm_seqset* s);
int mm_seqset_lengths(const mm_seqset* s, int32_t* len_out /* [count] */);
/* read back sequence i as (upper-cased) ASCII — round-trip check for tests */
int mm_seqset_fetch(mm_seqset* s, int64_t i, char* ascii_out, int64_t cap);
/* sequences [first, first + count) one behind the other, no separators (lengths: mm_seqset_lengths); cap >= their total length */
int mm_seqset_fetch_range(mm_seqset* s, int64_t first, int64_t count, char* ascii_out, int64_t cap);

/* Device-side synthetic inputs (bench.py; DESIGN.md "Synthetic workload").  The reference is a set of
 * genomes grouped in species (strains = substituted copies of a species root), generated straight into
 * packed HBM form; reads are sampled from it with ONT/PacBio-like sub/ins/del errors. */
typedef struct {
  uint64_t seed;
  int32_t n_species;          /* species roots                                                     */
  int32_t strains_per_species;
  int32_t genome_len;         /* bases per genome (one contig per genome)                          */
  float strain_divergence;    /* per-base substitution rate of a strain w.r.t. its species root    */
  float genus_divergence;     /* roots of the same genus (4 species each) differ by this rate      */
} mm_synth_ref_params;
typedef struct {
  uint64_t seed;
  int64_t n_reads;
  int32_t read_len;           /* template length taken from the genome                              */
  float sub_rate, ins_rate, del_rate;
  float frac_random;          /* reads made of random sequence (unmappable)                          */
  int32_t n_abundant;         /* reads are drawn from this many genomes, lognormal abundances        */
  int32_t read_len_min;       /* 0: every read has read_len bases; else lengths log-uniform in [read_len_min, read_len] */
} mm_synth_read_params;
int mm_synth_reference(mm_ctx* ctx, const mm_synth_ref_params* p, mm_seqset** out);
/* The miniSeq+H-shaped community of SURVEY.md §8 D1: microbial genomes of lognormal length grouped in species of 1..12 strains
 * (substitutions at a per-strain rate + a few block insertions / deletions against the species root), species roots grouped in
 * genera; plus "human-like" contigs: `repeat_fraction` of their 256-base granules are copies (2-20 % diverged, either strand) of
 * granules of a Zipf-weighted repeat family library, and `n_fraction` of their bases are runs of N.  Contig order is shuffled.
 * contig_genome[n_contigs] (optional) receives the genome of every contig: 0 .. n_genomes-1 microbial (one contig each), n_genomes
 * for every human-like contig.  n_contigs = n_genomes + human_contigs. */
typedef struct {
  uint64_t seed;
  int32_t n_genomes, n_species, n_genera;      /* microbial genomes (= contigs), species (1..12 strains each), genera        */
  double median_len, sigma_len;                /* lognormal genome length, clipped to [min_len, max_len]                    */
  int32_t min_len, max_len;
  float strain_div_min, strain_div_max;        /* substitution rate of a strain against its species root                    */
  float genus_div_min, genus_div_max;          /* substitution rate of a species root against its genus root                */
  int32_t strain_indel_events;                 /* up to this many block insertions / deletions (50 - 5000 bases) per strain  */
  int32_t human_contigs;                       /* 0: none                                                                    */
  int64_t human_bases;                         /* total length of the human-like contigs                                     */
  float repeat_fraction, n_fraction;
  int32_t n_repeat_families;
  int64_t total_bases_target;                  /* > 0: microbial lengths are scaled so that everything sums to this          */
} mm_synth_community_params;
int mm_synth_community(mm_ctx* ctx, const mm_synth_community_params* p, mm_seqset** out, int32_t* contig_genome);
/* genome_species[n_genomes]: the species (0 .. n_species-1) of every microbial genome of the community the same parameters generate
 * (host arithmetic only: the truth labels of bench.py's and the full-size tests' species-level checks) */
int mm_synth_community_species(const mm_synth_community_params* p, int32_t* genome_species);
/* truth_genome (optional, [n_reads]) receives the source contig index or -1.  Reads are drawn from `n_abundant` contigs among
 * those long enough for the longest read (exception runs — N — of the reference read as A). */
int mm_synth_reads(mm_ctx* ctx, const mm_seqset* reference, const mm_synth_read_params* p, mm_seqset** out,
                   int32_t* truth_genome);

/* ---- A2 debug tap: winnowed minimizers of every sequence (commonFunc.hpp:92-175) -------------- */
/* offsets[count+1]; records are (hash, wpos, strand) in winnowing order.  Pass NULL arrays to query sizes. */
int mm_minimizers(mm_ctx* ctx, const mm_seqset* s, int k, int w, int64_t* offsets, uint32_t* hash, int32_t* wpos,
                  int32_t* strand, int64_t cap);

/* ---- index (replaces Sketch::build_and_store_index + computeFreqHist, winSketch.hpp:180-365, :452-494) */
typedef struct {
  int64_t n_contigs, n_entries, n_unique_hashes, n_dup_flagged;
  int64_t hbm_bytes;
} mm_index_info;
int mm_index_build(mm_ctx* ctx, const mm_seqset* contigs, int k, int w, mm_index** out);
void mm_index_destroy(mm_index* idx);
/* Persistent device index: what createIndex stores and mapAgainstIndex loads per index chunk (written mapWrap.h:388-391, read :525-529;
 * the Boost archives of winSketch.hpp:73-83), in an own versioned binary format — the index arrays as they lie in HBM (entries, occurrence
 * lists + bins, hash table, position directory, duplicate distances), the contig lengths, the occurrence histogram of the chunk (so that
 * the accumulated freqThreshold of winSketch.hpp:452-494 comes out as after a build) and the threshold that was set when it was stored.
 * A load is file -> pinned staging -> device, no kernel runs; the loaded index behaves as the built one in every entry point.
 * MM_ERR_ARG with a message for an unreadable, foreign, truncated or inconsistent file. */
int mm_index_save(mm_index* idx, const char* path);
int mm_index_load(mm_ctx* ctx, const char* path, mm_index** out);
/* --maxmemory chunk rule (Sketch::build flush test winSketch.hpp:274-329, memory model :165-178), evaluated on the
 * index of the WHOLE reference: returns the first contig of every chunk the reference would create under the
 * given limit (n_chunks == 1, first_contig[0] == 0 when everything fits or max_memory_bytes == 0).  The caller then
 * builds one mm_index per contig range, maps every read against each, and merges with mm_mapping_concat.
 * MM_ERR_LIMIT if a single contig exceeds the limit (the reference throws there, :318-322). */
int mm_index_plan_chunks(mm_ctx* ctx, const mm_index* whole, uint64_t max_memory_bytes, int32_t* first_contig, int32_t cap, int32_t* n_chunks);
int mm_index_get_info(const mm_index* idx, mm_index_info* out);
/* Occurrence-count histogram of this chunk: pairs (count, number of hashes with that count), ascending.
 * The caller accumulates it across chunks and derives freqThreshold exactly as winSketch.hpp:452-494 does
 * (mm_freq_threshold_from_hist below) — the reference never clears the histogram between chunks. */
int mm_index_freq_hist(mm_index* idx, int64_t* counts, int64_t* n_hashes, int64_t cap, int64_t* n_out);
int mm_freq_threshold_from_hist(const int64_t* counts, const int64_t* n_hashes, int64_t n, int64_t n_unique_hashes,
                                int prev_threshold);
int mm_index_set_freq_threshold(mm_index* idx, int threshold);
/* debug tap: position-ordered entries (hash, contig, wpos, strand) */
int mm_index_entries(mm_index* idx, uint32_t* hash, int32_t* contig, int32_t* wpos, int32_t* strand, int64_t cap);
/* debug tap: per entry, the distance in entries to the previous / next entry of its contig with the same hash (0: there is none),
 * saturated at 65535 — what K5 answers slidingMap.hpp:139-214's "is this hash already / still inside the window?" from */
int mm_index_dup_neighbours(mm_index* idx, int32_t* prev_dist, int32_t* next_dist, int64_t cap);

/* ---- host statistics (float math of map_stats.hpp; Boost.Math binomial restated) -------------- */
int mm_recommended_window(double p_value, int k, float pi, int min_read_len, uint64_t reference_size);   /* map_stats.hpp:226 */
double mm_estimate_pvalue(int s, int k, float pi, int min_read_len, uint64_t reference_size);             /* map_stats.hpp:179 */
int mm_min_hits_relaxed(int s, int k, float pi);                                                          /* map_stats.hpp:142 */
void mm_identity(int shared, int s, int k, float* ident, float* ident_upper /* may be NULL */);         /* computeMap.hpp:406-412 */

/* ---- mapping (replaces Map::mapModule for a whole batch: computeMap.hpp:180-538) -------------- */
typedef struct {
  int32_t k, w;
  float perc_identity;        /* --pi, default 80                                   */
  int32_t min_read_len;       /* reads shorter than max(w,k,min_read_len) are skipped (computeMap.hpp:137) */
} mm_map_params;

typedef struct {              /* one L2 mapping that passed the identity filter (base_types.hpp:133) */
  int32_t read;               /* index of the read in the seqset                      */
  int32_t ref_contig;         /* chunk-local contig id                                */
  int32_t ref_start;          /* meanOptimalPos; refEnd = ref_start + read_len - 1    */
  int32_t shared;             /* conservedSketches                                    */
  int32_t sketch;             /* sketchSize s                                         */
  int32_t strand;             /* +1 / -1                                              */
  double mapq;                /* mapping quality (mapWrap.h:215-323), filled by mm_mapping_add_qualities */
} mm_map_record;

typedef struct {
  int64_t n_reads, n_reads_long_enough, n_reads_mapped, n_mappings;
  int64_t bases_long_enough;
  /* algorithmic-traffic counters */
  int64_t sum_sketch, sum_hits, n_candidates, sum_l2_stream_entries, sum_l2_evals;
  int64_t n_ambiguous_sketch_reads;   /* reads whose duplicate-hash strands needed the std::sort tie-break */
  int64_t sum_hits_kept;              /* seed hits left after the exact run pre-filter (K3c) */
  int64_t n_l2_rebuilds;              /* window states rebuilt from scratch by the exact skip-ahead of K5 */
  int64_t n_l2_wide_redo;             /* candidates done again: by the literal full slide (reads shorter than w+k) or after a strand tie-break */
  /* device time of each stage of this batch, milliseconds, from hipEvents recorded on the ctx stream
   * around the launches; ms_l2 (the K5/K6 kernel) and ms_hit_filter (the counting+filtering K3c kernel, part of
   * ms_probe_gather) are single kernels: bench.py's roofline uses whichever is larger */
  double ms_minimizer, ms_sketch, ms_probe_gather, ms_sort_hits, ms_l1_scan, ms_l2, ms_compact, ms_total, ms_hit_filter;
  /* reads whose sketch has >= 32768 hashes (longer than ~145 kb at w = 8): mapped like every other read, but by the slow
   * K5 class that keeps its window state in global memory (informational) */
  int64_t n_reads_giant;
} mm_map_stats;

int mm_map_batch(mm_ctx* ctx, const mm_index* idx, const mm_seqset* reads, const mm_map_params* p, mm_mapping** out);
/* mm_map_batch with two stage boundaries handed to the caller: `at_stage(user, stage)` is called on the calling thread, at most once per
 * stage and not at all when the batch fails or ends before that point:
 *   stage 1  the sketches of the batch are complete (K1 + K2 have run), nothing of its seed stage (K3) is enqueued yet;
 *   stage 2  the last big kernel of the batch (K5 / K6) is enqueued; what follows are small kernels and downloads.
 * A caller that drives several contexts of one device decides with it which stages of two batches may share the GPU: bench.py lets
 * the next batch start at stage 2 (its minimizer stage takes the CUs that this batch's last kernel leaves as it drains), or — with
 * --staged-map — takes one lock for K1 + K2 and another for K3 ... K6 and swaps them at stage 1.  The callback must not call into
 * `ctx`; results are the same with or without it. */
int mm_map_batch_phased(mm_ctx* ctx, const mm_index* idx, const mm_seqset* reads, const mm_map_params* p, void (*at_stage)(void* user, int stage), void* user,
                        mm_mapping** out);
/* The same batch against another index (the next chunk of --maxmemory): minimizers and sketches of the reads — which do not depend on
 * the index — are taken from `sketch_of` (the two large read-only arrays held jointly when it belongs to `ctx`, everything copied otherwise), a mapping of THESE reads with the same k, w and minimum read length whose intermediates
 * have not been released, instead of being computed again (the reference recomputes them per chunk, computeMap.hpp:277-298).
 * Results are those of mm_map_batch. */
int mm_map_batch_reusing(mm_ctx* ctx, const mm_index* idx, const mm_seqset* reads, const mm_map_params* p, const mm_mapping* sketch_of, mm_mapping** out);
/* Minimizers and sketches of a read batch alone (K1 + K2, computeMap.hpp:277-298) — a `sketch_of` for mm_map_batch_reusing that is tied to
 * no index: a run that maps one batch against many chunk indexes one after the other, each built and dropped in turn, computes them once
 * and keeps nothing else of a chunk's mapping.  The object holds no records (the other mm_mapping_* calls see an empty mapping);
 * mm_mapping_destroy frees it. */
int mm_sketch_batch(mm_ctx* ctx, const mm_seqset* reads, const mm_map_params* p, mm_mapping** out);
void mm_mapping_destroy(mm_mapping* m);
int mm_mapping_get_stats(const mm_mapping* m, mm_map_stats* out);
/* frees everything of a batch result but its records, offsets and read lengths (what fetch, keep_best, concat, add_qualities and
 * the EM builder use); the debug taps return empty afterwards.  For callers that hold the results of many batches, e.g. while
 * the chunks of a reference larger than HBM are indexed and mapped one after the other (mapWrap.h:417-437 keeps them as files). */
int mm_mapping_release_intermediates(mm_mapping* m);
/* per-read first-record offsets [n_reads+1], records in (read, contig, position) order */
int mm_mapping_fetch(mm_mapping* m, int64_t* offsets, mm_map_record* records, int64_t cap);
/* K8: mapping qualities over the union of each read's records (mapWrap.h:215-323) */
int mm_mapping_add_qualities(mm_ctx* ctx, mm_mapping* m, const mm_seqset* reads, int k);
/* default (non --all) reporting: per read keep the records whose identity is >= best - 1.0
 * (reportReadMappings, computeMap.hpp:546-587).  Apply per index chunk, before mm_mapping_concat / add_qualities. */
int mm_mapping_keep_best(mm_ctx* ctx, mm_mapping* m, int k);
/* merge the records of several index chunks read-wise, chunk order preserved (unifyFiles, mapWrap.h:128-132), on the device.  The parts may
 * belong to any context of this process: records of a part on another device are fetched with a peer copy.  The result carries
 * records only (as after mm_mapping_release_intermediates: the debug taps return MM_ERR_STATE); its statistics are the work
 * counters and stage times summed over the parts. */
int mm_mapping_concat(mm_ctx* ctx, mm_mapping* const* parts, const int32_t* contig_base, int n_parts, mm_mapping** out);
/* the same from host-side parts, i.e. what mm_mapping_fetch returned for index chunks that live on OTHER GPUs (SURVEY §8 E1:
 * "records gathered to the read's owner"; the reference gathers them through its PREFIX.N files, mapWrap.h:417-437):
 * offsets[p][n_reads+1] and records[p][...] of chunk p, contig_base[p] = first contig of chunk p in the whole reference.
 * The result carries records only (as after mm_mapping_release_intermediates) and takes mm_mapping_add_qualities,
 * mm_mapping_fetch and mm_em_create_from_mapping. */
int mm_mapping_from_parts(mm_ctx* ctx, int64_t n_reads, const int32_t* read_len, const mm_map_params* p, int n_parts,
                          const int64_t* const* offsets, const mm_map_record* const* records, const int32_t* contig_base, mm_mapping** out);

/* the same exchange between the RANKS of ctx's communicator (mm_comm_init; one process per GPU, or one host thread per GPU of one process), device
 * to device over RCCL (ncclSend / ncclRecv over xGMI) — nothing is staged in host memory.  Collective: every rank calls it for the
 * same read batch, batches in the same order on every rank.  parts[i] = this rank's mapping of the batch against chunk chunk_id[i] (it must hold
 * every chunk c with chunk_rank[c] == its rank); n_chunks / chunk_rank[c] / contig_base[c] describe all chunks of the reference and are the same
 * on every rank.  On rank `owner`, *out receives the merged mapping (records only, chunk order; takes mm_mapping_add_qualities ...); on the other
 * ranks *out = NULL.  A one-rank communicator (or none) merges locally. */
int mm_mapping_gather(mm_ctx* ctx, int owner, int64_t n_reads, const int32_t* read_len, const mm_map_params* p, mm_mapping* const* parts, const int32_t* chunk_id,
                      int n_parts, int n_chunks, const int32_t* chunk_rank, const int32_t* contig_base, mm_mapping** out);

/* debug taps for one batch (parity tests).  Any output pointer may be NULL. */
int mm_debug_sketch(mm_mapping* m, int64_t* offsets, uint32_t* hash, int32_t* strand, int64_t cap);          /* computeMap.hpp:292-298 */
int mm_debug_hits(mm_mapping* m, int64_t* offsets, int32_t* contig, int32_t* wpos, int64_t cap);             /* :307-323, sorted :353 */
int mm_debug_candidates(mm_mapping* m, int64_t* offsets, int32_t* triples /* contig,start,end */, int64_t cap);   /* :346-386 */
int mm_debug_l2(mm_mapping* m, int64_t* per_cand /* contig, meanPos, shared, optBeg, optEnd, accepted */, int64_t cap);      /* :460-538 */
int mm_debug_min_hits(mm_mapping* m, int32_t* min_hits /* [n_reads] */);
/* measurement tap: lengths of the occurrence lists the batch's sketch hashes ask `idx` for (minimizerPosLookupIndex.find, computeMap.hpp:310):
 * hist[0] = hashes not in the index, hist[c] = lists of c entries (c < n_bins - 2), hist[n_bins - 2] = longer lists that are kept,
 * hist[n_bins - 1] = lists cut by freqThreshold (:317) */
int mm_debug_probed_lists(mm_mapping* m, const mm_index* idx, int64_t* hist, int32_t n_bins);

/* ---- EM (replaces meta::doEM's iteration, fEM.h:501-661, and the per-read likelihood fEM.h:234-373) */
/* Mappings are given read-wise: read r owns entries [read_off[r], read_off[r+1]).  For entry i:
 * l_i = f[taxon[i]] * inv_nloc[i] * mapq[i]  (fEM.h:353), p_i = l_i / sum over the read.  */
int mm_em_create(mm_ctx* ctx, int64_t n_reads, const int64_t* read_off, const int32_t* taxon, const double* mapq,
                 const double* inv_nloc, int32_t n_taxa, mm_em** out);
/* map -> classify without the text file in between: the same EM problem `classify` would read from PREFIX, built on
 * the device from the records of `m` (after mm_mapping_add_qualities): taxon = contig_taxon[contig]; the mapping
 * quality rounded to the 6 significant digits the file carries (mapWrap.h:318-320 -> fEM.h:262); 1/nLoc with
 * nLoc = sum over the taxon's contigs of (len - readLen + 1) when len >= readLen, else 1 if the read maps to that
 * contig (getMappingLocations, fEM.h:322-346). */
int mm_em_create_from_mapping(mm_ctx* ctx, const mm_mapping* m, const int32_t* contig_taxon, const int32_t* contig_len, int32_t n_contigs,
                              int32_t n_taxa, mm_em** out);
/* mappings per taxon (taxa without any start the iteration at frequency 0, fEM.h:494) */
int mm_em_taxon_counts(mm_em* em, int64_t* counts);
int mm_em_sizes(const mm_em* em, int64_t* n_reads, int64_t* n_entries, int32_t* n_taxa);
void mm_em_destroy(mm_em* em);
/* one E+M step on this rank's reads: f_partial[t] = sum of p_i, *ll_partial = sum log(sum l_i) */
int mm_em_iterate(mm_em* em, const double* f, double* f_partial, double* ll_partial);
/* same, but leaves the partial sums on the device, all-reduces them over the communicator (RCCL) and
 * returns the normalised next f and the global log-likelihood (identical on every rank) */
int mm_em_iterate_allreduce(mm_em* em, const double* f, double* f_next, double* ll);
/* The whole loop of meta::doEM (fEM.h:501-661) without a host round trip per iteration: starting from f0, iterate (E step, per-taxon
 * sums, all-reduce over the communicator if there is one, normalise) until the reference's stop rule fires (log-likelihood gain <= 1
 * and relative gain < 1e-4, from the second iteration on) or max_iter iterations.  f_out[n_taxa] = final frequencies, ll_trace
 * (optional, up to ll_cap <= 1024 values) = log-likelihood of every iteration, *n_iter = iterations done.  Every rank of the
 * communicator calls it together and gets the same result. */
int mm_em_run(mm_em* em, const double* f0, int max_iter, double* f_out, double* ll_trace, int ll_cap, int* n_iter);
/* Goes on where the previous mm_em_run / mm_em_continue on `em` ended (same f, iteration count, previous log-likelihood) for up to
 * max_iter more iterations; ll_trace receives the log-likelihoods of THESE iterations, *n_iter their number, *stopped (optional)
 * whether the stop rule has fired (then further calls do nothing).  A caller that wants every round's log-likelihood of a long
 * run (the reference prints each, fEM.h:616-634) calls mm_em_run and then mm_em_continue in slices of <= 1024 iterations. */
int mm_em_continue(mm_em* em, int max_iter, double* f_out, double* ll_trace, int ll_cap, int* n_iter, int* stopped);
/* final posteriors p_i for the current f (fEM.h:684-707) and the index of the best mapping per read (fEM.h:217) */
int mm_em_posteriors(mm_em* em, const double* f, double* post /* [n_entries] */, int64_t* best /* [n_reads] */);

/* ---- communicator (RCCL over xGMI; one process per GPU) --------------------------------------- */
#define MM_COMM_ID_BYTES 128
int mm_comm_unique_id(char id[MM_COMM_ID_BYTES]);                     /* rank 0 creates, caller broadcasts */
int mm_comm_init(mm_ctx* ctx, const char id[MM_COMM_ID_BYTES], int rank, int nranks);
/* what RCCL itself says about ctx's communicator (ncclCommCount / ncclCommUserRank); 1 / 0 without one */
int mm_comm_info(mm_ctx* ctx, int* n_ranks, int* rank);
/* Several contexts of one device (e.g. two host threads that take read batches in turn) use ONE communicator: `ctx` borrows the
 * communicator of `owner` (same device; the owner outlives it).  The caller issues the collectives of the sharing contexts in
 * the same order on every rank. */
int mm_comm_share(mm_ctx* ctx, mm_ctx* owner);
int mm_comm_allreduce_f64(mm_ctx* ctx, double* host_inout, int64_t n);   /* staging helper for small vectors */
void mm_comm_destroy(mm_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* METAMAPS_HIP_H */
