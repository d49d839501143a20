// CPU harness for tests/test_ref_host.py: the host program's split / interval overlap (metamaps_amd/csrc/host/host_util.hpp) and the
// oracle's (oracle/orc_post.hpp), printed in the format of oracle/_ref/ref_host (the reference's own meta/util.h functions).
//   test_host_util host|oracle split DELIM < lines      test_host_util host|oracle overlap < "a b c d" lines
#include "../metamaps_amd/csrc/host/host_util.hpp"
#include "../oracle/orc_post.hpp"
#include <cstdio>
#include <iostream>

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const bool host = std::string(argv[1]) == "host";
  if (std::string(argv[2]) == "split" && argc >= 4) {
    std::string ln;
    while (std::getline(std::cin, ln)) {
      const std::vector<std::string> v = host ? split(ln, argv[3]) : orc::split(ln, argv[3]);
      std::string j;
      for (size_t i = 0; i < v.size(); ++i) { if (i) j += '\x1f'; j += v[i]; }
      printf("%zu\t%s\n", v.size(), j.c_str());
    }
    return 0;
  }
  if (std::string(argv[2]) == "overlap") {
    unsigned long long a, b, c, d;
    while (std::cin >> a >> b >> c >> d) printf("%zu\n", host ? iv_overlap(a, b, c, d) : orc::overlap(a, b, c, d));
    return 0;
  }
  return 2;
}
