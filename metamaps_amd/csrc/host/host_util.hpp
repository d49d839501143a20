// Two small pieces of meta/util.h that `classify` depends on, kept in a header of their own so that tests/test_host_util.cpp can
// hold them against the reference's own functions (oracle/_ref/ref_host, built from /root/reference/src/meta/util.h).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace {

std::vector<std::string> split(const std::string& in, const std::string& d) {   // meta/util.h:80
  std::vector<std::string> out;
  if (in.empty()) return out;
  if (d.empty()) { for (char ch : in) out.emplace_back(1, ch); return out; }   // util.h:88-95
  size_t s = 0, p;
  while ((p = in.find(d, s)) != std::string::npos) { out.push_back(in.substr(s, p - s)); s = p + d.size(); }
  out.push_back(in.substr(s));
  return out;
}

// overlap of two closed intervals as computed by meta/util.h:118-172
size_t iv_overlap_big_small(size_t bigL, size_t bigR, size_t smL, size_t smR) {
  if (bigL <= smL && bigR >= smR) return smR - smL + 1;
  if (smL >= bigL && smL <= bigR) return bigR - smL + 1;
  if (smR >= bigL && smR <= bigR) return smR - bigL + 1;
  return 0;
}
size_t iv_overlap(size_t aL, size_t aR, size_t bL, size_t bR) {
  return (aR - aL + 1 > bR - bL + 1) ? iv_overlap_big_small(aL, aR, bL, bR) : iv_overlap_big_small(bL, bR, aL, aR);
}

}  // namespace
