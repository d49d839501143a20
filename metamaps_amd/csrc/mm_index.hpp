// Reference sketch of one index chunk, resident in HBM
// (replaces skch::Sketch: winSketch.hpp:68-556; data members :102-133).
//
//   pos[N]      position-ordered minimizers {hash, pw}           == Sketch::minimizerIndex      (:129)
//   cstart[C+1] entry range of every contig in pos[]
//   uh[U], ustart[U+1], occ[N]   hash -> occurrence list (CSR)    == minimizerPosLookupIndex      (:119)
//               occ entry = contig<<32 | pw  (8 bytes, directly usable as an L1 seed hit / sort key)
//   bkt[2^B+1]  first uh index per top-B-bit hash prefix (one coalesced probe + short search per lookup)
#pragma once
#include "mm_common.hpp"
#include <climits>
#include <map>

struct mm_index {
  mm_ctx* ctx = nullptr;
  int k = 0, w = 0;
  int64_t n_contigs = 0, N = 0, U = 0, n_dup = 0;
  int bkt_bits = 0;
  int freq_threshold = INT_MAX;              // winSketch.hpp:94
  mm::DBuf<mm::Rec> pos;
  mm::DBuf<uint64_t> cstart;
  mm::DBuf<uint32_t> uh;
  mm::DBuf<uint64_t> ustart;
  mm::DBuf<uint64_t> occ;
  mm::DBuf<uint64_t> bkt;
  mm::DBuf<int32_t> d_contig_len;
  std::vector<int32_t> contig_len;
  std::vector<uint64_t> h_cstart;
  std::map<int64_t, int64_t> hist;           // occurrence count -> number of hashes (this chunk)
  int64_t hbm_bytes() const {
    return (int64_t)(pos.bytes() + cstart.bytes() + uh.bytes() + ustart.bytes() + occ.bytes() + bkt.bytes() + d_contig_len.bytes());
  }
};

namespace mm {

struct IndexView {
  const Rec* pos;
  const uint64_t* cstart;
  const uint32_t* uh;
  const uint64_t* ustart;
  const uint64_t* occ;
  const uint64_t* bkt;
  int64_t N, U;
  int bkt_bits;
  int freq_threshold;
};
inline IndexView make_view(const mm_index* I) {
  return IndexView{I->pos.p, I->cstart.p, I->uh.p, I->ustart.p, I->occ.p, I->bkt.p, I->N, I->U, I->bkt_bits, I->freq_threshold};
}

// hash -> slot in uh[] or -1  (minimizerPosLookupIndex.find, computeMap.hpp:310)
__device__ inline int64_t index_find(const IndexView& I, uint32_t h) {
  uint32_t b = h >> (32 - I.bkt_bits);            // bkt_bits in [4,26]
  int64_t lo = (int64_t)I.bkt[b], hi = (int64_t)I.bkt[b + 1];
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    uint32_t v = I.uh[mid];
    if (v < h) lo = mid + 1; else hi = mid;
  }
  return (lo < (int64_t)I.bkt[b + 1] && I.uh[lo] == h) ? lo : -1;
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

}  // namespace mm
