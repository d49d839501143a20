// EM state resident in HBM: mappings grouped by read (E step) and indexed by taxon (M step).
#pragma once
#include "mm_common.hpp"

struct mm_mapping;
struct mm_em {
  mm_ctx* ctx = nullptr;
  int64_t n_reads = 0, n_entries = 0;
  int32_t n_taxa = 0;
  mm::DBuf<int64_t> read_off;      // [n_reads+1]
  mm::DBuf<int32_t> taxon;         // [n_entries]
  mm::DBuf<double> mapq, inv_nloc; // [n_entries]
  mm::DBuf<int64_t> tstart, perm;  // CSR by taxon: entries of taxon t are perm[tstart[t]..tstart[t+1]) in read order
  mm::DBuf<double> post, ll_read, f, partial, block_sum;
  // device-resident loop (mm_em_run): taxa with mappings on this rank, their partial sums, loop control, log-likelihood trace
  mm::DBuf<int32_t> present; int32_t n_present = -1;
  // the per-taxon sums in fixed shape: items of <= 512 consecutive entries of one taxon (one wavefront each), the items of present taxon p
  // are [pt_item[p], pt_item[p + 1]); the resident kernel's grid, its per-workgroup log-likelihood partials and its barrier words
  mm::DBuf<int32_t> pt_item; mm::DBuf<int64_t> item_lo, item_hi; int32_t n_items = 0, n_wg = 0;
  mm::DBuf<double> item_sum, wg_ll; mm::DBuf<unsigned> bar;
  mm::DBuf<int32_t> eread;                                // read of every mapping
  mm::DBuf<int64_t> span;                                 // per read (sorted by mapping count, longest first): first and behind-last mapping
  mm::DBuf<int64_t> pos; mm::DBuf<double> post_sorted;   // pos[i]: place of entry i in taxon-sorted order (inverse of perm); the loop keeps its posteriors there
  mm::DBuf<double> local_partial, ll_trace, f_run;   // f_run: the loop's own frequencies (mm_em_iterate / mm_em_posteriors in between do not disturb mm_em_continue)
  mm::DBuf<long long> ctrl;
};

namespace mm {
void em_create(mm_ctx* ctx, int64_t n_reads, const int64_t* read_off, const int32_t* taxon, const double* mapq, const double* inv_nloc,
               int32_t n_taxa, mm_em* E);
void em_create_from_mapping(mm_ctx* ctx, const ::mm_mapping* M, const int32_t* contig_taxon, const int32_t* contig_len, int32_t n_contigs,
                            int32_t n_taxa, mm_em* E);
void em_iterate(mm_em* E, const double* f, double* f_partial, double* ll_partial);
void em_iterate_allreduce(mm_em* E, const double* f, double* f_next, double* ll);
int em_run(mm_em* E, const double* f0, int max_iter, double* f_out, double* ll_trace, int ll_cap, bool* stopped);
void em_posteriors(mm_em* E, const double* f, double* post, int64_t* best);
void comm_unique_id(char* id);
void comm_init(mm_ctx* ctx, const char* id, int rank, int nranks);
void comm_allreduce_f64(mm_ctx* ctx, double* host, int64_t n);
void comm_destroy(mm_ctx* ctx);
}
