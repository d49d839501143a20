// ORACLE — test infrastructure only (see orc_core.hpp header).
// Command-line front end with the reference's sub-commands and flags
// (map/include/parseCmdArgs.hpp:33-117, :255-505; map/mash_map.cpp:257-317) so that the
// product CLI and this restatement can be run on the same inputs and their files compared.
#include "orc_post.hpp"
#include <chrono>
#include <sys/stat.h>

using namespace orc;

static uint64_t file_size(const std::string& f) {                // commonFunc.hpp:211-231
  struct stat st; if (stat(f.c_str(), &st) != 0) { std::cerr << "Cannot open " << f << " for size determination.\n"; exit(1); }
  return (uint64_t)st.st_size;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::cerr << "usage: metamaps_oracle mapDirectly|classify ...\n"; return 1; }
  std::string mode = argv[1];
  std::map<std::string, std::string> opt; bool all = false;
  static const std::map<std::string, std::string> alias{{"-r", "reference"}, {"-q", "query"}, {"-o", "output"},
      {"-k", "kmer"}, {"-p", "pval"}, {"-w", "window"}, {"-m", "minReadLen"}, {"-t", "threads"}, {"--mm", "maxmemory"},
      {"--pi", "perc_identity"}};
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--all") { all = true; continue; }
    std::string key = alias.count(a) ? alias.at(a) : (a.rfind("--", 0) == 0 ? a.substr(2) : a);
    if (i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; return 1; }
    opt[key] = argv[++i];
  }
  auto t0 = std::chrono::steady_clock::now();
  if (mode == "mapDirectly") {
    if (!opt.count("reference")) { std::cerr << "Provide reference file (s)\n"; return 1; }
    if (!opt.count("query")) { std::cerr << "Provide query file (s)\n"; return 1; }
    if (!opt.count("output")) { std::cerr << "Provide output file\n"; return 1; }
    Params P;
    P.refSize = file_size(opt["reference"]);
    P.maxMem = opt.count("maxmemory") ? (uint64_t)(std::pow(1024, 3) * std::stoull(opt["maxmemory"])) : 0;
    P.k = opt.count("kmer") ? std::stoi(opt["kmer"]) : 16;
    P.pval = opt.count("pval") ? std::stod(opt["pval"]) : 1e-3;
    P.minReadLen = opt.count("minReadLen") ? std::stoi(opt["minReadLen"]) : 1000;
    P.pi = opt.count("perc_identity") ? std::stof(opt["perc_identity"]) : 80;
    if (opt.count("maxmemory-bytes")) P.maxMem = std::stoull(opt["maxmemory-bytes"]);   // test hook: sub-GiB limits
    P.reportAll = all;
    P.threads = opt.count("threads") ? std::max(1, std::stoi(opt["threads"])) : 1;
    if (opt.count("window")) {                                   // parseCmdArgs.hpp:363-374
      P.w = std::stoi(opt["window"]);
      int s = P.minReadLen * 2 / P.w;
      P.pval = estimate_pvalue(s, P.k, P.alphabet, P.pi, P.minReadLen, P.refSize);
    } else {
      P.w = recommended_window(P.pval, P.k, P.alphabet, P.pi, P.minReadLen, P.refSize);
    }
    auto qs = split(opt["query"], ","), os = split(opt["output"], ",");
    if (qs.size() != os.size()) { std::cerr << "Please specify an equal number of input and output files\n"; return 1; }
    MapCounters C;
    // the reference indexes once and maps every query file per chunk (mapWrap.h:417-430); with one
    // query file this is the same as the loop below, with several the index is simply rebuilt.
    for (size_t i = 0; i < qs.size(); ++i) map_directly(P, opt["reference"], qs[i], os[i], &C);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "{\"oracle\":\"mapDirectly\",\"w\":" << P.w << ",\"reads\":" << C.reads << ",\"bases\":" << C.bases
              << ",\"sketch\":" << C.sketch << ",\"hits\":" << C.hits << ",\"cands\":" << C.cands << ",\"stream\":"
              << C.stream << ",\"evals\":" << C.evals << ",\"mappings\":" << C.maps << ",\"chunks\":" << C.chunks << ",\"threads\":" << P.threads << ",\"map_seconds\":" << C.map_seconds << ",\"seconds\":" << sec << "}\n";
  } else if (mode == "classify") {
    if (!opt.count("DB")) { std::cerr << "Provide path to DB.\n"; return 1; }
    if (!opt.count("mappings")) { std::cerr << "Provide path to mappings.\n"; return 1; }
    for (auto& m : split(opt["mappings"], ",")) {
      EMTrace tr = do_em(m, opt["DB"], true, opt.count("minreads") ? std::stoull(opt["minreads"]) : 10000);   // parseCmdArgs.hpp:462-471
      std::cerr << "{\"oracle\":\"classify\",\"iterations\":" << tr.ll.size() << ",\"ll\":[";
      for (size_t i = 0; i < tr.ll.size(); ++i) { char b[64]; snprintf(b, sizeof b, "%s%.17g", i ? "," : "", tr.ll[i]); std::cerr << b; }
      double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::cerr << "],\"seconds\":" << sec << "}\n";
    }
  } else { std::cerr << "unknown sub-command " << mode << "\n"; return 1; }
  return 0;
}
