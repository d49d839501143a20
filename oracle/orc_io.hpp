// ORACLE — test infrastructure only (see orc_core.hpp header).
// FASTA/FASTQ(.gz) record reader with the observable behaviour of kseq_read
// (common/kseq.h:170-207): a record starts at the next '>' or '@'; the name is the header
// up to the first whitespace; sequence = every isgraph() byte until a '>', '+' or '@'
// (anywhere, not just at line start); after '+', the rest of that line is skipped,
// len(seq) quality bytes in [33,127] are consumed and one more byte behind them (kseq.h:200).
// Not reproduced: kseq holds its buffer as (signed) char, so a 0xFF byte reads as end of input.
#pragma once
#include <zlib.h>
#include <cctype>
#include <string>
#include <vector>
#include <stdexcept>

namespace orc {

class SeqReader {
  gzFile fp_ = nullptr;
  std::vector<unsigned char> buf_;
  size_t beg_ = 0, end_ = 0;
  bool eof_ = false;
  int pending_ = 0;   // header char already consumed ('>' or '@'), 0 if none
  int getc_() {
    if (beg_ >= end_) {
      if (eof_) return -1;
      int n = gzread(fp_, buf_.data(), (unsigned)buf_.size());
      if (n <= 0) { eof_ = true; return -1; }
      beg_ = 0; end_ = (size_t)n;
    }
    return buf_[beg_++];
  }
 public:
  std::string name, seq;
  explicit SeqReader(const std::string& path) : buf_(1 << 16) {
    fp_ = gzopen(path.c_str(), "r");
    if (!fp_) throw std::runtime_error("cannot open " + path);
  }
  ~SeqReader() { if (fp_) gzclose(fp_); }
  SeqReader(const SeqReader&) = delete;
  // returns sequence length, or -1 at end of file, -2 on truncated quality
  long next() {
    int c;
    if (pending_ == 0) {
      while ((c = getc_()) != -1 && c != '>' && c != '@') {}
      if (c == -1) return -1;
      pending_ = c;
    }
    name.clear(); seq.clear();
    // header: name up to whitespace, then rest of line is the comment
    bool any = false;
    while ((c = getc_()) != -1 && !isspace(c)) { name.push_back((char)c); any = true; }
    if (c == -1 && !any) return -1;
    if (c != '\n') while (c != -1 && (c = getc_()) != -1 && c != '\n') {}
    while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@')
      if (isgraph(c)) seq.push_back((char)c);
    pending_ = (c == '>' || c == '@') ? c : 0;
    if (c != '+') return (long)seq.size();
    while ((c = getc_()) != -1 && c != '\n') {}
    if (c == -1) return -2;
    size_t got = 0;
    // kseq.h:200 reads the character BEFORE it tests the count: the character behind the last quality (normally the line's
    // '\n'; a '@' that follows without a line break is lost) is consumed too — pinned against the real kseq.h, tests/test_ref_host.py
    while ((c = getc_()) != -1 && got < seq.size())
      if (c >= 33 && c <= 127) ++got;
    pending_ = 0;
    if (got != seq.size()) return -2;
    return (long)seq.size();
  }
};

}  // namespace orc
