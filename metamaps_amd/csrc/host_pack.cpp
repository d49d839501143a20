// 2-bit packing of ASCII bases on the host, 32 bases per step with AVX2 (compiled by g++, linked into libmetamaps_hip.so; mm_seq.hip calls it).
// mm_seqset_upload packs every read batch and the whole reference: at one table look-up per base (~1 ns) the 9.8 Gbp of a million 10 kb reads are
// ten CPU-seconds — more than `mapDirectly`'s whole mapping phase has under a container quota of 16 CPUs (cpu_budget.hpp), found in round 5 when the
// pools were cut to the quota and the upload of a 0.2 Gbp batch went from 8 to 18 ms.
//   code: A 0, C 1, G 2, T 3 (either case; commonFunc.hpp:57-66 upper-cases a-z before hashing, and the device turns codes back into "ACGT")
//   t = (c >> 1) & 3 gives A 0, C 1, T 2, G 3 for both cases; code = t ^ (t >> 1) swaps the last two.
// A block with any other byte is left to the caller's byte-wise path (exception runs).
#include <cstddef>
#include <cstdint>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

extern "C" int mm_host_has_avx2() {
#if defined(__x86_64__)
  static const int v = __builtin_cpu_supports("avx2") ? 1 : 0;
  return v;
#else
  return 0;
#endif
}

// packs blocks of 32 bases from p into out (two 32-bit words per block, base j of a word at bits 2j) until n_bases (a multiple of 32) are done
// or a block holds a byte that is not one of ACGTacgt; returns the number of bases packed
#if defined(__x86_64__)
#define MM_AVX2 __attribute__((target("avx2")))
#else
#define MM_AVX2
#endif
extern "C" MM_AVX2 size_t mm_pack_acgt_blocks(const uint8_t* p, size_t n_bases, uint32_t* out) {
#if defined(__x86_64__)
  const __m256i up = _mm256_set1_epi8((char)0xDF), cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
  const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1);
  const __m256i m14 = _mm256_set1_epi16(0x0401), m116 = _mm256_set1_epi32(0x00100001);
  const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  size_t j = 0;
  for (; j + 32 <= n_bases; j += 32) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + j));
    const __m256i u = _mm256_and_si256(v, up);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)), _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
    if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) break;
    const __m256i t = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);
    const __m256i code = _mm256_xor_si256(t, _mm256_and_si256(_mm256_srli_epi16(t, 1), one));
    const __m256i p16 = _mm256_maddubs_epi16(code, m14);           // c0 + 4 c1 per 16-bit lane
    const __m256i p32 = _mm256_madd_epi16(p16, m116);              // + 16 (c2 + 4 c3): four bases = one byte per 32-bit lane
    const __m256i g = _mm256_shuffle_epi8(p32, gather);            // the four bytes of each 128-bit half side by side
    out[(j >> 4)] = (uint32_t)_mm256_extract_epi32(g, 0);
    out[(j >> 4) + 1] = (uint32_t)_mm256_extract_epi32(g, 4);
  }
  return j;
#else
  (void)p; (void)n_bases; (void)out;
  return 0;
#endif
}
