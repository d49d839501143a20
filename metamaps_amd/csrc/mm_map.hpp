// Mapping results of one read batch and every intermediate the parity taps expose.
#pragma once
#include "mm_common.hpp"
#include "mm_index.hpp"
#include "mm_minimizer.hpp"

namespace mm {
struct L2Result {
  int32_t contig, mean_pos, shared, strand, accepted, pad;
  int64_t opt_beg, opt_end;
  uint32_t n_stream, n_evals, n_rebuilds, pad2;                  // per-candidate work counters (summed by l2_stats_kernel)
};
}

struct mm_mapping {
  mm_ctx* ctx = nullptr;
  int64_t n_reads = 0, n_cand = 0, n_rec = 0;
  int smax = 0;
  mm_map_params params{};
  mm_map_stats stats{};
  std::vector<int32_t> read_len;
  std::vector<uint8_t> active;
  void (*at_stage)(void*, int) = nullptr; void* at_stage_user = nullptr;   // mm_map_batch_phased: stage 1 between K2 and K3, stage 2 once K5 is enqueued
  const mm_mapping* sketch_donor = nullptr;  // mm_map_batch_reusing: minimizers and sketches are taken from this mapping of the same reads
  bool sketch_only = false;                  // mm_sketch_batch: K1 + K2 alone, no index, no records
  // K1
  mm::MinimizerSet mz;
  // K2 (same per-read offsets as mz.off; only the first sk_n[r] slots of a read are used)
  mm::DBuf<uint32_t> sk_hash;
  mm::DBuf<uint8_t> sk_strand;
  mm::DBuf<int32_t> sk_n;
  mm::DBuf<uint8_t> amb;
  std::vector<int32_t> h_sk_n, h_min_hits;
  mm::DBuf<int32_t> min_hits, accept_min, d_read_len;
  // K3/K4
  mm::DBuf<uint64_t> read_hit_off;           // [n+1]
  std::vector<uint64_t> h_read_hit_off;
  mm::DBuf<uint64_t> hits;                   // contig<<32 | pw, sorted per read
  mm::DBuf<uint64_t> cand_off;               // [n+1]
  std::vector<uint64_t> h_cand_off;
  mm::DBuf<int32_t> cand;                    // triples contig,start,end
  mm::DBuf<int32_t> cand_read;
  // K5
  mm::DBuf<mm::L2Result> l2;
  // final
  mm::DBuf<mm_map_record> rec;
  mm::DBuf<uint64_t> rec_off;                // [n+1]
  std::vector<uint64_t> h_rec_off;
  bool has_mapq = false;
  bool released = false;                     // mm_mapping_release_intermediates: only the records are left
};

namespace mm {
void map_batch(mm_ctx* ctx, const mm_index* I, const mm_seqset* reads, const mm_map_params& P, mm_mapping* M);
void mapping_add_qualities(mm_ctx* ctx, mm_mapping* M, int k);
void probed_list_hist(mm_ctx* ctx, const mm_index* I, const mm_mapping* M, int nb, int64_t* hist);
}
