// CPU harness: metamaps_amd/csrc/host_pack.cpp — 32 bases per step with AVX2 == the byte-wise definition, on plain ACGT of either case; stops at the first block
// that holds any other byte (those take mm_seqset_upload's byte-wise path with its exception runs), never writes beyond what it reports as packed.
#include "../metamaps_amd/csrc/host_pack.cpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main() {
  if (!mm_host_has_avx2()) { printf("ok (no AVX2 on this host: the byte-wise path is the only one)\n"); return 0; }
  int bad = 0;
  srand(7);
  const char* al = "ACGTacgt";
  for (int rep = 0; rep < 400; ++rep) {
    const size_t n = 32 * (size_t)(1 + rand() % 40);
    std::vector<uint8_t> s(n + 64);
    for (auto& c : s) c = (uint8_t)al[rand() & 7];
    long spoil = -1;
    if (rep % 3 == 1) { spoil = rand() % (long)n; const char odd[] = {'N', 'n', 'R', 'U', '@', 0x01, (char)0xC1, 'B', 'D', (char)0x61 - 0x20 + 1}; s[(size_t)spoil] = (uint8_t)odd[rand() % 10]; }
    std::vector<uint32_t> out(n / 16 + 4, 0xDEADBEEFu);
    const size_t done = mm_pack_acgt_blocks(s.data(), n, out.data());
    const size_t expect = spoil < 0 ? n : ((size_t)spoil / 32) * 32;
    if (done != expect) { ++bad; continue; }
    for (size_t j = 0; j < done; j += 16) {
      uint32_t w = 0;
      for (int i = 0; i < 16; ++i) { const int c = s[j + i] & 0xDF; const int k = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; w |= (uint32_t)k << (2 * i); }
      if (out[j / 16] != w) ++bad;
    }
    for (size_t j = done / 16; j < out.size(); ++j) if (out[j] != 0xDEADBEEFu) ++bad;   // nothing written behind what was packed
  }
  if (bad) { printf("%d mismatches\n", bad); return 1; }
  printf("ok\n");
  return 0;
}
