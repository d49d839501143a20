// Reference sketch of one index chunk, resident in HBM
// (replaces skch::Sketch: winSketch.hpp:68-556; data members :102-133).
//
//   pos[N]      position-ordered minimizers {hash, pw}           == Sketch::minimizerIndex      (:129)
//   cstart[C+1] entry range of every contig in pos[]
//   occ[P]      occurrences grouped by hash, every group starting on a 64-byte boundary (padded to 8 entries)
//                                                                 == minimizerPosLookupIndex      (:119)
//               occ entry = contig<<32 | pw  (8 bytes, directly usable as an L1 seed hit / sort key)
//   occ16[P]    per occ entry (same index) the 13-bit position bin the seed-hit filter counts in:
//               ((first base of the contig in the concatenated reference + wpos) >> 13) & 8191;
//               the padding entries behind a list carry 8192 + (a number below 64): a bin no read position falls into, so a kernel that
//               counts and tests a whole 16-byte piece (the streaming seed filter) needs neither the list's length nor a mask per entry —
//               the pads land in 64 dummy counters (spread, so that they do not queue up on one LDS address) and are never "alive".
//               Kernels that mask the code to 13 bits and go by the list's count ignore them as before.
//   tab[2*cap]  open-addressing table hash -> (count, first occ), cap = 4 * tab_buckets: slot = {count<<32 | hash, start}; one 64-byte
//               line per lookup on average (uh[]/ustart[] CSR arrays only live during the build)
//   dup_bits / dup_rank / dup_dist   same-hash neighbours of the entries whose hash occurs more than once in their contig
//               (flags PW_DP / PW_DN in pos[].pw): bit j&63 of dup_bits[j>>6] marks a flagged entry, dup_rank[j>>6] counts the
//               flagged entries before that block of 64, and dup_dist[rank] = distance in entries to the previous such entry
//               (low half) and to the next one (high half), saturated at dup_sat = 65535.  K5's "is another occurrence of this hash inside
//               the window?" (slidingMap.hpp:139-214) is then one comparison instead of a scan of the window.
#pragma once
#include "mm_common.hpp"
#include <climits>
#include <algorithm>
#include <map>

struct mm_index {
  mm_ctx* ctx = nullptr;
  int k = 0, w = 0;
  int64_t n_contigs = 0, N = 0, U = 0, n_dup = 0;
  uint32_t tab_buckets = 0;                  // 4-slot buckets of tab[] (any number, not a power of two)
  int freq_threshold = INT_MAX;              // winSketch.hpp:94
  mm::DBuf<mm::Rec> pos;
  mm::DBuf<uint64_t> cstart;
  mm::DBuf<uint32_t> uh;
  mm::DBuf<uint64_t> ustart;
  mm::DBuf<uint64_t> occ;
  mm::DBuf<uint16_t> occ16;
  mm::DBuf<uint64_t> tab;
  mm::DBuf<int32_t> d_contig_len;
  // position directory: dir[dir_off[c] + b] = number of entries of contig c with wpos < (b << dir_shift), b = 0 .. (len >> dir_shift) + 1
  // (the last one is the contig's entry count): a range search of K5 starts from one directory read instead of log64(contig) rounds
  // of 64 scattered 128-byte lines each
  mm::DBuf<uint32_t> dir;
  mm::DBuf<uint64_t> dir_off;
  int dir_shift = 9;
  mm::DBuf<uint64_t> dup_bits, dup_rank;
  mm::DBuf<uint32_t> dup_dist;
  int dup_sat = 65535;                       // saturation value of the stored distances (MM_DUP_SAT lowers it: tests of the scan fall-back)
  std::vector<int32_t> contig_len;
  std::vector<uint64_t> h_cstart;
  std::map<int64_t, int64_t> hist;           // occurrence count -> number of hashes (this chunk)
  int64_t hbm_bytes() const {
    return (int64_t)(pos.bytes() + cstart.bytes() + uh.bytes() + ustart.bytes() + occ.bytes() + occ16.bytes() + tab.bytes() + d_contig_len.bytes() + dir.bytes() + dir_off.bytes() +
                     dup_bits.bytes() + dup_rank.bytes() + dup_dist.bytes());
  }
};

namespace mm {

constexpr int HF_BIN_SHIFT = 13, HF_SLOTS = 8192;              // seed-hit filter: 8192-base position bins, counted modulo 8192 bins
constexpr int HF_PAD_SLOTS = 64;                                // codes HF_SLOTS .. HF_SLOTS + 63: the padding entries of occ16[] (above)
__host__ __device__ inline uint16_t hf_pad_code(uint64_t list_start, uint64_t j) { return (uint16_t)(HF_SLOTS + (((list_start >> 3) * 5 + j) & (uint64_t)(HF_PAD_SLOTS - 1))); }
constexpr int HF_SLOT_BITS_NARROW = 13, HF_SLOT_BITS_WIDE = 15; // ... or, for long reads, modulo 32768 slots (mm_map.hip, hit_filter_kernel)

struct IndexView {
  const Rec* pos;
  const uint64_t* cstart;
  const uint64_t* occ;
  const uint16_t* occ16;
  const uint64_t* tab;
  int64_t N, U;
  uint32_t tab_buckets;
  int freq_threshold;
  const uint32_t* dir;
  const uint64_t* dir_off;
  int dir_shift;
  const uint64_t* dup_bits;
  const uint64_t* dup_rank;
  const uint32_t* dup_dist;
  int dup_sat;
};
inline IndexView make_view(const mm_index* I) {
  return IndexView{I->pos.p, I->cstart.p, I->occ.p, I->occ16.p, I->tab.p, I->N, I->U, I->tab_buckets, I->freq_threshold, I->dir.p, I->dir_off.p, I->dir_shift,
                   I->dup_bits.p, I->dup_rank.p, I->dup_dist.p, I->dup_sat};
}

// Home slot of a hash: the first slot of a 4-slot bucket (4 x 16 B = one 64-byte sector), linear probing from there.  A lookup
// reads whole sectors: at load factor <= 0.55 nearly every hash is resolved (found, or an empty slot seen) by its home sector,
// i.e. by one memory request — random requests, not bytes, are what the probe stage pays for (tools/ubench/randread).
// The table has ANY number of buckets (multiply-shift range reduction of the mixed hash: minimizer hashes are window minima, skewed
// towards small values, so the hash is multiplied by an odd constant first): a power-of-two table is up to twice as large as its
// load factor asks for — 17 GB instead of 10 for each of the chunk indexes of a --maxmemory run (docs/history.md section 7).
__host__ __device__ inline uint64_t tab_slot(uint32_t h, uint32_t buckets) {
  return (uint64_t)(uint32_t)(((uint64_t)(h * 0x9E3779B1u) * (uint64_t)buckets) >> 32) << 2;
}
__host__ __device__ inline uint64_t tab_next_sector(uint64_t slot, uint64_t slots) { slot += 4; return slot >= slots ? slot - slots : slot; }
__host__ __device__ inline uint64_t tab_next_slot(uint64_t slot, uint64_t slots) { return slot + 1 == slots ? 0 : slot + 1; }
__host__ inline uint32_t tab_buckets_for(int64_t unique_hashes) {   // load factor <= 0.55 (what the 2^30-slot table of the miniSeq+H index had)
  const int64_t slots = (int64_t)((double)unique_hashes / 0.55) + 4;
  return (uint32_t)std::max<int64_t>((slots + 3) / 4, 64);
}

// hash -> (occurrence count, first occurrence); false when the hash is not in the index
// (minimizerPosLookupIndex.find, computeMap.hpp:310).  One lane per lookup; probe_kernel has the 4-lanes-per-sector form.
__device__ inline bool index_find(const IndexView& I, uint32_t h, uint32_t* count, uint64_t* start) {
  const uint64_t slots = (uint64_t)I.tab_buckets << 2;
  uint64_t slot = tab_slot(h, I.tab_buckets);
  for (;;) {
    const uint64_t w0 = I.tab[2 * slot];
    if (w0 == 0) return false;
    if ((uint32_t)w0 == h) { *count = (uint32_t)(w0 >> 32); *start = I.tab[2 * slot + 1]; return true; }
    slot = tab_next_slot(slot, slots);
  }
}

// first entry of contig c with wpos >= p, as an ordinal into pos[]  (Sketch::searchIndex, winSketch.hpp:506)
__device__ inline int64_t index_search(const IndexView& I, int32_t c, int32_t p) {
  int64_t lo = (int64_t)I.cstart[c], hi = (int64_t)I.cstart[c + 1];
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (pw_wpos(I.pos[mid].pw) < p) lo = mid + 1; else hi = mid;
  }
  return lo;
}

void index_build(mm_ctx* ctx, const mm_seqset* contigs, int k, int w, mm_index* out);
void index_plan_chunks(mm_ctx* ctx, const mm_index* whole, uint64_t max_memory, std::vector<int32_t>& first_contig);
void index_save(const mm_index* idx, const char* path);
void index_load(mm_ctx* ctx, const char* path, mm_index* out);

}  // namespace mm
