// Sequence file reading for the `metamaps` host program: a kseq-compatible record reader over zlib or over a memory-mapped file,
// and the mapped file with its record-start search (used by the parallel block parser of metamaps_main.cpp).
// Host only, no device dependencies: tests/test_seq_reader.cpp checks the block parse against the sequential one on the CPU.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <iostream>
#include <string>
#include <vector>

namespace {
[[noreturn]] inline void seq_reader_die(const std::string& m) { std::cerr << m << std::endl; std::cout.flush(); fflush(nullptr); _exit(1); }   // (_exit: other threads may be inside the HIP runtime, metamaps_main.cpp die())

// FASTA/FASTQ(.gz) records with kseq's observable behaviour (common/kseq.h:170-207)
// Two sources: a (gz) file read through zlib in 1 MiB pieces, or a byte range of a memory-mapped plain file (MemView: the parallel
// block parser below).  In memory mode `rec_start` is the file offset of the header character of the record just returned, `tell`
// the offset of the first byte the next call will look at, and a record whose sequence sits on one line of a 4-line FASTQ record
// is returned as a view into the mapping (view != nullptr, seq empty) instead of a copy.
class SeqFile {
  struct Buf { const unsigned char* p; const unsigned char* data() const { return p; } } buf_{nullptr};
  gzFile fp_ = nullptr; std::vector<unsigned char> own_; size_t beg_ = 0, end_ = 0; bool eof_ = false; int pending_ = 0; size_t pending_pos_ = 0; bool mem_ = false;
  int get() {
    if (beg_ >= end_) { if (eof_) return -1; int n = gzread(fp_, own_.data(), (unsigned)own_.size()); if (n <= 0) { eof_ = true; return -1; } beg_ = 0; end_ = (size_t)n; }
    return buf_.p[beg_++];
  }
 public:
  std::string name, seq;
  const char* view = nullptr; size_t view_len = 0;               // memory mode: the sequence where it lies in the file
  size_t rec_start = 0;
  explicit SeqFile(const std::string& path) : own_(1 << 20) { fp_ = gzopen(path.c_str(), "r"); if (!fp_) seq_reader_die("Cannot open " + path); buf_.p = own_.data(); }
  SeqFile(const unsigned char* data, size_t begin, size_t size) : beg_(begin), end_(size), eof_(true), mem_(true) { buf_.p = data; }   // memory mode: parses from `begin` on
  ~SeqFile() { if (fp_) gzclose(fp_); }
  SeqFile(const SeqFile&) = delete;
  // memory mode: offset of the header character of the record the next call would return, (size_t)-1 if there is none
  size_t peek_start() const {
    if (pending_) return pending_pos_;
    for (size_t q = beg_; q < end_; ++q) if (buf_.p[q] == '>' || buf_.p[q] == '@') return q;
    return (size_t)-1;
  }
  size_t length() const { return view ? view_len : seq.size(); }
  bool next() {
    int c;
    view = nullptr; view_len = 0;
    if (!pending_) { while ((c = get()) != -1 && c != '>' && c != '@') {} if (c == -1) return false; pending_ = c; pending_pos_ = beg_ - 1; }
    rec_start = pending_pos_;
    name.clear(); seq.clear();
    bool any = false;
    while ((c = get()) != -1 && !isspace(c)) { name.push_back((char)c); any = true; }
    if (c == -1 && !any) return false;
    if (c != '\n') while (c != -1 && (c = get()) != -1 && c != '\n') {}
    if (mem_ && pending_ == '@' && c == '\n') {
      // the usual FASTQ record — sequence on one line, '+' line, as many quality characters on one line — straight from the mapping:
      // validated in place, nothing copied.  Anything else (wrapped lines, odd characters, a short quality line) takes the general path.
      const unsigned char* const base = buf_.p; const unsigned char* const fe = base + end_;
      const unsigned char* p = base + beg_;
      const unsigned char* nl = (const unsigned char*)memchr(p, '\n', (size_t)(fe - p));
      if (nl && nl > p && nl + 1 < fe && nl[1] == '+') {
        const size_t L = (size_t)(nl - p);
        unsigned char bad = 0;
        for (const unsigned char* q = p; q < nl; ++q) { const unsigned char b = *q; bad |= (unsigned char)((unsigned char)(b - 33) > 93) | (unsigned char)(b == '+') | (unsigned char)(b == '>') | (unsigned char)(b == '@'); }
        const unsigned char* pl = (const unsigned char*)memchr(nl + 1, '\n', (size_t)(fe - nl - 1));
        if (!bad && pl && (size_t)(fe - pl - 1) >= L && (pl + 1 + L == fe || pl[1 + L] == '\n')) {
          const unsigned char* ql = pl + 1;
          unsigned char qb = 0;
          for (const unsigned char* q = ql; q < ql + L; ++q) qb |= (unsigned char)((unsigned char)(*q - 33) > 94);
          if (!qb) { view = (const char*)p; view_len = L; beg_ = (size_t)(ql + L - base) + (ql + L < fe ? 1 : 0); pending_ = 0; return true; }   // (+1: kseq.h:200, below)
        }
      }
    }
    // sequence: everything up to the next '>', '+' or '@', graphic characters only — in bulk over the read buffer
    // (class table: 0 keep, 1 skip, 2 stop) instead of one call per character
    static const struct Cls { uint8_t t[256]; Cls() { for (int i = 0; i < 256; ++i) t[i] = (i == '>' || i == '+' || i == '@') ? 2 : (isgraph(i) ? 0 : 1); } } cls;
    c = -1;
    for (;;) {
      if (beg_ >= end_) { const int ch = get(); if (ch == -1) break; --beg_; }      // refill
      const unsigned char* p = buf_.data() + beg_;
      {                                                          // fast path: a whole line of plain sequence characters
        const unsigned char* const nl = (const unsigned char*)memchr(p, '\n', end_ - beg_);
        const unsigned char* const le = nl ? nl : buf_.data() + end_;
        unsigned char bad = 0;                                   // (byte-wise OR reduction: vectorises)
        for (const unsigned char* q = p; q < le; ++q) { const unsigned char b = *q; bad |= (unsigned char)((unsigned char)(b - 33) > 93) | (unsigned char)(b == '+') | (unsigned char)(b == '>') | (unsigned char)(b == '@'); }
        if (!bad && le > p) { seq.append((const char*)p, (size_t)(le - p)); beg_ = (size_t)(le - buf_.data()) + (nl ? 1 : 0); continue; }
      }
      const unsigned char* const e = p + std::min<size_t>(end_ - beg_, 16384);   // (resize zero-fills: keep the pieces small)
      const size_t old = seq.size();
      seq.resize(old + (size_t)(e - p));
      char* o = &seq[old];
      uint8_t k = 0;
      while (p < e && (k = cls.t[*p]) != 2) { *o = (char)*p; o += (k == 0); ++p; }
      seq.resize((size_t)(o - seq.data()));
      beg_ = (size_t)(p - buf_.data());
      if (p < e) { c = *p; ++beg_; break; }                       // the stop character is consumed, as get() would
    }
    pending_ = (c == '>' || c == '@') ? c : 0; pending_pos_ = beg_ - 1;
    if (c != '+') return true;
    while ((c = get()) != -1 && c != '\n') {}
    if (c == -1) { pending_ = 0; return false; }                 // the '+' line runs into the end of the file: -2 in kseq (kseq.h:196), also for an empty sequence
    size_t got = 0;                                              // qualities: as many characters in [33,127] as there are bases
    while (got < seq.size()) {
      if (beg_ >= end_) { const int ch = get(); if (ch == -1) break; --beg_; }
      const unsigned char* p = buf_.data() + beg_; const unsigned char* const e = buf_.data() + end_;
      const size_t want = seq.size() - got, avail = (size_t)(e - p);
      if (avail <= want) {                                       // the whole rest of the buffer cannot overshoot
        size_t cnt = 0;
        for (const unsigned char* q = p; q < e; ++q) cnt += (unsigned)(*q - 33u) <= 94u;
        got += cnt; beg_ = end_;
      } else {
        unsigned char bad = 0;                                   // usual case: the next `want` bytes are the quality line
        for (const unsigned char* q = p; q < p + want; ++q) bad |= (unsigned char)((unsigned char)(*q - 33) > 94);
        if (!bad) { got += want; beg_ += want; }
        else { while (p < e && got < seq.size()) { got += (unsigned)(*p - 33u) <= 94u; ++p; } beg_ = (size_t)(p - buf_.data()); }
      }
    }
    pending_ = 0;
    if (got != seq.size()) return false;                         // truncated quality string: kseq returns -2 and the callers' loops end (kseq.h:204)
    (void)get();                                                 // kseq.h:200 reads a character before it tests the count: the one behind the last quality is consumed too
    return true;                                                 // (normally the line's '\n'; a '@' that follows without a line break is lost, as in the reference)
  }
};

// A plain (not gzip) query file, memory mapped.  Parsed in blocks by several threads; what makes that exact is that the state of
// the sequential parser between two records is just a file offset: a block parser that starts on a record start the previous
// block's parser also ends on reproduces the sequential parse (SeqFile, memory mode).
struct MappedFile {
  const unsigned char* data = nullptr; size_t size = 0;
  bool open(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 4) { ::close(fd); return false; }
    unsigned char magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) { ::close(fd); return false; }   // gzip: zlib reads it
    void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) return false;
    madvise(p, (size_t)st.st_size, MADV_WILLNEED);
    data = (const unsigned char*)p; size = (size_t)st.st_size;
    return true;
  }
  ~MappedFile() { if (data) munmap((void*)data, size); }
  // the pages of [a, b) (page aligned) are not needed again: they stop counting as resident
  void drop(size_t a, size_t b) const { if (data && b > a && b <= size) madvise((void*)(data + a), b - a, MADV_DONTNEED); }
  // first offset >= x that is certainly a record start, or `limit` if none is found before it: a FASTA header ('>' at the start of
  // a line; any '>' ends a sequence for kseq), or a FASTQ record whose four lines and whose successor's first line check out
  size_t sync(size_t x, size_t limit) const {
    const bool fasta = data[0] == '>';
    for (size_t p = x; p < limit; ++p) {
      if (p > 0 && data[p - 1] != '\n') { const void* q = memchr(data + p, '\n', limit - p); if (!q) return limit; p = (size_t)((const unsigned char*)q - data); continue; }
      if (fasta) { if (data[p] == '>') return p; continue; }
      if (data[p] != '@') continue;
      size_t a = p; bool ok = true;
      const unsigned char* const fe = data + size;
      for (int rec = 0; rec < 2 && ok && a < size; ++rec) {
        if (data[a] != '@') { ok = false; break; }
        const unsigned char* l1 = (const unsigned char*)memchr(data + a, '\n', size - a);             // end of the header line
        if (!l1 || l1 + 1 >= fe) { ok = false; break; }
        const unsigned char* l2 = (const unsigned char*)memchr(l1 + 1, '\n', (size_t)(fe - l1 - 1));   // end of the sequence line
        if (!l2 || l2 + 1 >= fe || l2[1] != '+' || l2 == l1 + 1) { ok = false; break; }
        const unsigned char* l3 = (const unsigned char*)memchr(l2 + 1, '\n', (size_t)(fe - l2 - 1));   // end of the '+' line
        if (!l3) { ok = false; break; }
        const size_t L = (size_t)(l2 - l1 - 1);
        if ((size_t)(fe - l3 - 1) < L) { ok = false; break; }
        const unsigned char* e = l3 + 1 + L;                                                          // just behind the qualities
        if (e < fe && *e != '\n') { ok = false; break; }
        a = (size_t)(e - data) + 1;
      }
      if (ok) return p;
    }
    return limit;
  }
};

}  // namespace
