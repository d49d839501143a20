// ORACLE — test infrastructure only.  Builds oracle/_ref/ref_host (a command line tool: a crash of the reference code on
// hostile input then costs a test case, not the test process) from the REAL reference headers, included where they lie
// (never copied): /root/reference/src/common/kseq.h (the FASTA/FASTQ reader under every sequence the reference ever sees:
// winSketch.hpp:245-252, computeMap.hpp:123-134, mapWrap.h:107-114) and /root/reference/src/meta/util.h (split / overlap, used by
// classify: fEM.h:256, :771, :1349).  Both compile without Boost; kseq needs zlib only.  util.h relies on its includer for
// <map>, <set>, <fstream> and <algorithm> (in the reference they arrive through taxonomy.h:12-15): standard headers, given here.
#include <zlib.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include "common/kseq.h"
KSEQ_INIT(gzFile, gzread)                                         // as winSketch.hpp:27 / computeMap.hpp:34 instantiate it
#include "meta/util.h"

static uint64_t fnv(const char* p, size_t n) { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; } return h; }

// ref_host kseq FILE              every record the reference's read loop sees (`while ((len = kseq_read(seq)) >= 0)`,
//                                  computeMap.hpp:123): one line "name len fnv1a(seq)" per record, then "END code" with the return
//                                  value that ended the loop (-1 end of file, -2 truncated quality)
// ref_host split DELIM < lines     util.h split() of every input line: "n<TAB>piece<US>piece..." per line (US = 0x1f)
// ref_host overlap < "a b c d"     util.h overlap() of the closed intervals [a,b], [c,d] per line (its asserts are live: a < b, c < d)
int main(int argc, char** argv) {
  if (argc >= 3 && std::string(argv[1]) == "kseq") {
    gzFile fp = gzopen(argv[2], "r");
    if (!fp) return 2;
    kseq_t* seq = kseq_init(fp);
    long len;
    while ((len = kseq_read(seq)) >= 0) printf("%s %ld %016llx\n", seq->name.s, len, (unsigned long long)fnv(seq->seq.s, (size_t)len));
    printf("END %ld\n", len);
    kseq_destroy(seq);
    gzclose(fp);
    return 0;
  }
  if (argc >= 3 && std::string(argv[1]) == "split") {
    std::string ln;
    while (std::getline(std::cin, ln)) {
      std::vector<std::string> v = split(ln, argv[2]);
      printf("%zu\t%s\n", v.size(), join(v, "\x1f").c_str());
    }
    return 0;
  }
  if (argc >= 2 && std::string(argv[1]) == "overlap") {
    unsigned long long a, b, c, d;
    while (std::cin >> a >> b >> c >> d) printf("%zu\n", overlap(a, b, c, d));
    return 0;
  }
  return 2;
}
