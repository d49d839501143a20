// ORACLE — test infrastructure only.  Builds oracle/_ref/libref_murmur.so from the REAL
// reference header, included where it lies (never copied): /root/reference/src/common/murmur3.h.
// It is the only piece of the reference that compiles without Boost; everything above
// base_types.hpp:12 needs <boost/serialization/...>, which this image does not have.
#include <stdint.h>
#include "common/murmur3.h"
extern "C" uint32_t ref_kmer_hash(const char* s, int k) {
  char out[16];
  MurmurHash3_x64_128(s, k, 42, out);   // seed: map/include/commonFunc.hpp:33
  return *(uint32_t*)out;               // low 32 bits: commonFunc.hpp:71-81
}
extern "C" void ref_murmur3_x64_128(const void* key, int len, uint32_t seed, void* out) {
  MurmurHash3_x64_128(key, len, seed, out);
}
