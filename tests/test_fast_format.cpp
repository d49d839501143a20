// CPU harness: metamaps_amd/csrc/host/fast_format.hpp against snprintf.  Prints "ok <n checked>" or the first mismatch.
#include "../metamaps_amd/csrc/host/fast_format.hpp"
#include <cstdint>
#include <cstdlib>
#include <random>

static long n_fast_parse = 0;
static int check_g(double x) {
  char ref[64]; snprintf(ref, sizeof ref, "%g", x);
  std::string s; append_g6(s, x);
  if (s != ref) { printf("MISMATCH %%g %.17g: fast '%s' printf '%s'\n", x, s.c_str(), ref); return 1; }
  // the text read back (what `classify` parses with strtod): parse_g6_text either declines or returns strtod's double, bit for bit
  double back = 0;
  if (parse_g6_text(s.data(), s.size(), &back)) {
    ++n_fast_parse;
    const double want = strtod(s.c_str(), nullptr);
    if (memcmp(&back, &want, sizeof back) != 0) { printf("MISMATCH parse '%s': fast %.17g strtod %.17g\n", s.c_str(), back, want); return 1; }
  }
  return 0;
}
static int check_f(double x) {
  char ref[400]; snprintf(ref, sizeof ref, "%f", x);
  std::string s; append_f6(s, x);
  if (s != ref) { printf("MISMATCH %%f %.17g: fast '%s' printf '%s'\n", x, s.c_str(), ref); return 1; }
  return 0;
}
int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 2000000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> u01(0, 1), ue(-320, 320);
  long cnt = 0;
  const double fixed[] = {0.0, -0.0, 1.0, 100.0, 99.99995, 99.999949999, 0.5, 0.1, 1e-5, 9.999995e-5, 0.0001, 0.00001234565, 123456.5, 1234565, 999999.5, 1e6, 1e22, 1e23, 1e-22, 1e-23,
                          85.5956, 82.8927, 84.4464, 0.501122, 1e-300, 1e300, 4.9e-324, 0.0000005, 0.9999995, 0.99999949, 2.5e-7, 1.5e-6, 0.1234565, 0.1234575, 88.49335, 1e100, 1e-100,
                          (double)(float)84.44644, (double)(float)99.99999};
  for (double x : fixed) { if (check_g(x) || check_g(-x) || check_f(x) || check_f(-x)) return 1; cnt += 4; }
  for (long i = 0; i < n; ++i) {
    const double a = u01(rng);                                   // posteriors, mapping qualities
    const double b = std::pow(10.0, ue(rng) / 16.0) * u01(rng);  // 1e-20 .. 1e20
    const double c = (double)(float)(80.0 + 20.0 * u01(rng));     // identities (float valued)
    const double d = std::ldexp(1.0 + (double)(rng() >> 12) * 0x1p-52, (int)(rng() % 2000) - 1000);   // the whole exponent range
    const double t6 = (double)(rng() % 2000000) / 2000000.0;      // ties and near-ties of the sixth decimal
    const double g6 = ((double)(100000 + rng() % 900000) + 0.5) * std::pow(10.0, (double)((int)(rng() % 20) - 14));   // ties of the sixth significant digit
    if (check_g(a) || check_f(a) || check_g(b) || check_g(c) || check_g(d) || check_f(t6) || check_g(t6) || check_g(g6)) return 1;
    cnt += 8;
  }
  // integers (append_int / append_uint against %lld / %llu): every digit count, the powers of ten and their neighbours, the extremes
  {
    auto check_i = [](long long v) { char ref[32]; snprintf(ref, sizeof ref, "%lld", v); std::string s = "x"; append_int(s, v); if (s != std::string("x") + ref) { printf("MISMATCH int %lld: '%s'\n", v, s.c_str()); return 1; } return 0; };
    auto check_u = [](unsigned long long v) { char ref[32]; snprintf(ref, sizeof ref, "%llu", v); std::string s; append_uint(s, v); if (s != ref) { printf("MISMATCH uint %llu: '%s'\n", v, s.c_str()); return 1; } return 0; };
    unsigned long long p10 = 1;
    for (int d = 0; d < 20; ++d) { for (long long dl = -2; dl <= 2; ++dl) { const unsigned long long v = p10 + (unsigned long long)dl; if (check_u(v) || (v <= 0x7fffffffffffffffull && (check_i((long long)v) || check_i(-(long long)v)))) return 1; cnt += 3; } if (d < 19) p10 *= 10; }
    if (check_u(0) || check_u(~0ull) || check_u(0xffffffffull) || check_u(0x100000000ull) || check_i(0) || check_i(-1) || check_i(-0x7fffffffffffffffLL - 1) || check_i(0x7fffffffffffffffLL)) return 1;
    for (long i = 0; i < n; ++i) { const int bits = 1 + (int)(rng() % 64); const unsigned long long v = rng() >> (64 - bits); if (check_u(v) || check_i((long long)v)) return 1; cnt += 2; }
  }
  if (n_fast_parse < cnt / 8) { printf("parse_g6_text declined too often: %ld of %ld\n", n_fast_parse, cnt); return 1; }
  printf("ok %ld\n", cnt);
  return 0;
}
