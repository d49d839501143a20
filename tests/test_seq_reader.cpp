// CPU unit test of the host program's sequence reader (metamaps_amd/csrc/host/seq_reader.hpp): records parsed block by block
// from a memory-mapped file — the scheme of the CLI's parallel block parser: block starts from MappedFile::sync, every block
// parses the records that start before the next block's limit, a block joins iff it starts where its predecessor's parse stands,
// from the first mismatch on the rest is parsed sequentially — must be the records of the sequential zlib reader, for any block
// size.  Prints one line per record (name, length, FNV-1a of the sequence) for the pytest side to compare with the oracle's reader.
#include "../metamaps_amd/csrc/host/seq_reader.hpp"
#include <cstdint>
#include <cstdio>

struct Rec { std::string name; size_t len; uint64_t h; bool operator==(const Rec& o) const { return name == o.name && len == o.len && h == o.h; } };
static uint64_t fnv(const char* p, size_t n) { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; } return h; }
static Rec rec_of(SeqFile& f) { return f.view ? Rec{f.name, f.view_len, fnv(f.view, f.view_len)} : Rec{f.name, f.seq.size(), fnv(f.seq.data(), f.seq.size())}; }

static std::vector<Rec> sequential(const std::string& path) { SeqFile f(path); std::vector<Rec> v; while (f.next()) v.push_back(rec_of(f)); return v; }

static std::vector<Rec> blocked(const MappedFile& mf, size_t blk, int* n_joined, bool* fell_back) {
  std::vector<Rec> v;
  const size_t nb = std::max<size_t>(1, (mf.size + blk - 1) / blk);
  size_t expect = 0; bool chain_ok = true, over = false;
  *n_joined = 0; *fell_back = false;
  for (size_t j = 0; j < nb && chain_ok && !over; ++j) {
    const size_t lim = std::min(mf.size, (j + 1) * blk);
    const size_t s0 = j == 0 ? 0 : mf.sync(j * blk, lim);
    if (j > 0 && s0 >= lim) { if (expect < lim) chain_ok = false; continue; }   // nothing recognised in this block
    if (j > 0 && s0 != expect) { chain_ok = false; break; }
    SeqFile f(mf.data, s0, mf.size);
    for (;;) {
      const size_t ps = f.peek_start();
      if (ps == (size_t)-1 || ps >= lim) break;
      if (!f.next()) { over = true; break; }
      v.push_back(rec_of(f));
    }
    ++*n_joined;
    expect = f.peek_start();
    if (expect == (size_t)-1) over = true;
  }
  if (!chain_ok) {
    *fell_back = true;
    SeqFile f(mf.data, expect, mf.size);
    while (f.next()) v.push_back(rec_of(f));
  }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string path = argv[1];
  std::vector<Rec> ref = sequential(path);
  if (argc > 2 && std::string(argv[2]) == "dump") { for (auto& r : ref) printf("%s %zu %016llx\n", r.name.c_str(), r.len, (unsigned long long)r.h); return 0; }
  MappedFile mf;
  if (!mf.open(path)) { printf("not mappable\n"); return 3; }
  int fails = 0;
  for (size_t blk : {(size_t)97, (size_t)1000, (size_t)4096, (size_t)30000, (size_t)1 << 20, mf.size + 1}) {
    int joined; bool fb;
    std::vector<Rec> got = blocked(mf, blk, &joined, &fb);
    const bool same = got == ref;
    printf("block %zu: %zu records, %d blocks joined, %s%s\n", blk, got.size(), joined, fb ? "sequential tail, " : "", same ? "equal" : "DIFFERENT");
    fails += !same;
  }
  printf(fails ? "FAILED\n" : "ok\n");
  return fails ? 1 : 0;
}
