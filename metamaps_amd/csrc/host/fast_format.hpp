// printf-free "%g" (6 significant digits: what operator<<(float/double) prints, computeMap.hpp:565-581, mapWrap.h:318-320) and "%f" (std::to_string,
// fEM.h:705) for the text the host program writes by the million lines.  Both produce exactly glibc's characters: the decimal digits are the
// value times a power of ten — an exact double for |power| <= 22, so the product is off by at most one rounding — rounded to an integer, and
// that rounding is only trusted when the product is further from a tie than any such error could reach; everything else (ties, huge or tiny
// magnitudes, non-finite values) goes through snprintf.  tests/test_fast_format.cpp holds both against snprintf on 10^7 values.
#pragma once
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

inline double ff_pow10(int n) {                                   // exact for 0 <= n <= 22
  static const double t[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  return t[n];
}

// appends snprintf("%g", x)
inline void append_g6(std::string& out, double x) {
  char buf[40];
  if (x == 0 || !std::isfinite(x)) { const int n = snprintf(buf, sizeof buf, "%g", x); out.append(buf, (size_t)n); return; }
  const double a = std::fabs(x);
  int e = (int)std::floor(std::log10(a));
  for (int tries = 0; tries < 3; ++tries) {
    const int sh = 5 - e;                                          // a * 10^sh in [1e5, 1e6)
    if (sh > 22 || sh < -22) break;
    const double v = sh >= 0 ? a * ff_pow10(sh) : a / ff_pow10(-sh);
    if (v < 1e5) { --e; continue; }
    if (v >= 1e6) { ++e; continue; }
    const double fl = std::floor(v), fr = v - fl;
    if (std::fabs(fr - 0.5) < 1e-6) break;                         // too close to a tie for the product's own rounding: let printf decide on the exact value
    unsigned q = (unsigned)fl + (fr > 0.5 ? 1u : 0u);
    if (q == 1000000u) { q = 100000u; ++e; }
    char d[6];
    for (int i = 5; i >= 0; --i) { d[i] = (char)('0' + q % 10); q /= 10; }
    int nd = 6; while (nd > 1 && d[nd - 1] == '0') --nd;           // %g drops trailing zeros
    char* p = buf;
    if (x < 0) *p++ = '-';
    if (e < -4 || e >= 6) {                                        // d.ddddde±XX
      *p++ = d[0];
      if (nd > 1) { *p++ = '.'; memcpy(p, d + 1, (size_t)nd - 1); p += nd - 1; }
      *p++ = 'e'; int ee = e; if (ee < 0) { *p++ = '-'; ee = -ee; } else *p++ = '+';
      if (ee >= 100) { *p++ = (char)('0' + ee / 100); ee %= 100; }
      *p++ = (char)('0' + ee / 10); *p++ = (char)('0' + ee % 10);
    } else if (e >= 0) {                                           // e + 1 integer digits
      const int ni = e + 1;
      memcpy(p, d, (size_t)std::min(ni, nd)); p += std::min(ni, nd);
      for (int i = nd; i < ni; ++i) *p++ = '0';
      if (nd > ni) { *p++ = '.'; memcpy(p, d + ni, (size_t)(nd - ni)); p += nd - ni; }
    } else {                                                       // 0.000ddd
      *p++ = '0'; *p++ = '.';
      for (int i = 0; i < -e - 1; ++i) *p++ = '0';
      memcpy(p, d, (size_t)nd); p += nd;
    }
    out.append(buf, (size_t)(p - buf));
    return;
  }
  const int n = snprintf(buf, sizeof buf, "%g", x);
  out.append(buf, (size_t)n);
}

// strtod of a text append_g6 wrote (digits, optional fraction, optional e+-XX), without strtod: at most 15 significant digits and a decimal
// exponent within +-22 make mantissa and power of ten exact doubles, whose product or quotient is then correctly rounded (Clinger's fast path) —
// the same double strtod returns.  false: not of that shape (the caller takes strtod).
inline bool parse_g6_text(const char* s, size_t n, double* out) {
  size_t i = 0; bool neg = false;
  if (i < n && s[i] == '-') { neg = true; ++i; }
  unsigned long long m = 0; int nd = 0, frac = 0; bool any = false;
  for (; i < n && s[i] >= '0' && s[i] <= '9'; ++i) { m = m * 10 + (unsigned)(s[i] - '0'); if (m) ++nd; any = true; }
  if (i < n && s[i] == '.') { ++i; for (; i < n && s[i] >= '0' && s[i] <= '9'; ++i) { m = m * 10 + (unsigned)(s[i] - '0'); if (m) ++nd; ++frac; any = true; } }
  if (!any || nd > 15) return false;
  int e = 0;
  if (i < n && (s[i] == 'e' || s[i] == 'E')) {
    ++i; bool eneg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { eneg = s[i] == '-'; ++i; }
    if (i >= n) return false;
    for (; i < n && s[i] >= '0' && s[i] <= '9'; ++i) { e = e * 10 + (s[i] - '0'); if (e > 400) return false; }
    if (eneg) e = -e;
  }
  if (i != n) return false;
  const int e10 = e - frac;
  if (e10 > 22 || e10 < -22) return false;
  const double v = e10 >= 0 ? (double)m * ff_pow10(e10) : (double)m / ff_pow10(-e10);
  *out = neg ? -v : v;
  return true;
}

// appends snprintf("%f", x) (= std::to_string(x)); the fast path covers [0, 1], the range of a posterior
inline void append_f6(std::string& out, double x) {
  if (x >= 0 && x <= 1 && !std::signbit(x)) {
    const double v = x * 1e6, fl = std::floor(v), fr = v - fl;
    if (std::fabs(fr - 0.5) > 1e-6) {
      unsigned long long q = (unsigned long long)fl + (fr > 0.5 ? 1 : 0);   // 0 .. 1000000
      char b[8]; b[0] = (char)('0' + q / 1000000); q %= 1000000; b[1] = '.';
      for (int i = 7; i >= 2; --i) { b[i] = (char)('0' + q % 10); q /= 10; }
      out.append(b, 8);
      return;
    }
  }
  char num[400]; const int n = snprintf(num, sizeof num, "%f", x); out.append(num, (size_t)n);
}

// appends a non-negative integer in decimal
inline void append_uint(std::string& out, unsigned long long v) {   // two digits per division, one append
  static const char pairs[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                              "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
  char b[24]; char* e = b + sizeof b; char* p = e;
  if (v <= 0xffffffffull) {                                       // (32-bit divisions: lengths, positions, counts)
    uint32_t u = (uint32_t)v;
    while (u >= 100) { const uint32_t q = u / 100, r = u - q * 100; u = q; p -= 2; p[0] = pairs[2 * r]; p[1] = pairs[2 * r + 1]; }
    if (u >= 10) { p -= 2; p[0] = pairs[2 * u]; p[1] = pairs[2 * u + 1]; } else *--p = (char)('0' + u);
  } else {
    while (v >= 100) { const unsigned long long q = v / 100; const unsigned r = (unsigned)(v - q * 100); v = q; p -= 2; p[0] = pairs[2 * r]; p[1] = pairs[2 * r + 1]; }
    if (v >= 10) { p -= 2; p[0] = pairs[2 * v]; p[1] = pairs[2 * v + 1]; } else *--p = (char)('0' + v);
  }
  out.append(p, (size_t)(e - p));
}

inline void append_int(std::string& out, long long v) { if (v < 0) { out += '-'; append_uint(out, 0ull - (unsigned long long)v); } else append_uint(out, (unsigned long long)v); }

}  // namespace
