// Device-side synthetic workload (bench.py): miniSeq+H-shaped reference and ONT/PacBio-like reads,
// generated straight into the packed HBM layout so that no 27 GB FASTA ever has to cross PCIe.
#pragma once
#include "mm_common.hpp"
namespace mm {
void synth_reference(mm_ctx* ctx, const mm_synth_ref_params& p, mm_seqset* out);
void synth_community(mm_ctx* ctx, const mm_synth_community_params& p, mm_seqset* out, int32_t* contig_genome);
void synth_community_species(const mm_synth_community_params& p, int32_t* genome_species);
void synth_reads(mm_ctx* ctx, const mm_seqset* ref, const mm_synth_read_params& p, mm_seqset* out, int32_t* truth_genome);
}
