// Every MM_* environment switch the library and the drop-in CLI read, in ONE place (round-4 review: 68 switches, 14 of them documented).  None is
// needed to run the product: defaults are what bench.py and the tests' main paths use.  Kinds:
//   user     a knob a deployment may turn (sizes of batches, worker counts, progress output)
//   tuning   a measured threshold between two code paths whose results are identical either way
//   test     a hook that forces a rarely taken — or a cross-check — path so that the tests can hold it against the default path / the oracle;
//            results are identical, speed is not
//   debug    traces and timing aids (stderr output, early kernel exits that RETURN NO RESULTS: never set outside tools/)
// MM_STRICT_ENV=1 makes mm_ctx_create (and the CLI at start) refuse any MM_* variable that is not in this table: a misspelt switch is an
// error instead of a silent default.  tests/test_env_table.py holds the table against the sources and INTEGRATION.md against the table.
#pragma once
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <string>

extern "C" char** environ;

namespace mm {

struct EnvSwitch { const char* name; const char* dflt; const char* kind; const char* what; };

inline const EnvSwitch* env_table(size_t* n) {
  static const EnvSwitch T[] = {
    {"MM_STRICT_ENV", "unset", "user", "refuse MM_* variables that are not in this table (mm_ctx_create returns MM_ERR_ARG, the CLI exits 1)"},
    {"MM_CPU_BUDGET", "min(hardware threads, affinity mask, cgroup CPU quota)", "user", "CPUs the process may keep busy: what every thread pool of the library and the CLI is sized from (cpu_budget.hpp)"},
    {"MM_SYNC", "block when the CPU budget is <= 32, else spin", "user", "how a host thread waits for its stream: \"spin\" (hipStreamSynchronize) or \"block\" (sleep on an event created with hipEventBlockingSync)"},
    // ---- CLI (host/metamaps_main.cpp, host/*.hpp)
    {"MM_CLI_WORKERS", "4", "user", "worker contexts per GPU that take read batches in turn (= --workers-per-gpu)"},
    {"MM_CLI_BATCH_READS", "100000", "user", "reads per batch handed to a worker"},
    {"MM_CLI_BATCH_MBASES", "256", "user", "... or this many million bases, whichever comes first"},
    {"MM_CLI_MAP_SLOTS", "2", "tuning", "batches per device that may be inside their mapping section at the same time"},
    {"MM_CLI_BLOCK_BYTES", "128 MiB", "tuning", "block size of the block-parallel FASTQ/FASTA parser over an mmap-ed query file"},
    {"MM_CLI_REF_BLOCK_BYTES", "min(group, 256 MiB)", "tuning", "the same for the reference FASTA"},
    {"MM_CLI_REF_GROUP_BASES", "2^30", "test", "bases per upload group of the reference (small groups: the concat path on small inputs)"},
    {"MM_CLI_REF_SEQUENTIAL", "unset", "test", "reference through the sequential kseq-style reader instead of the block parser"},
    {"MM_CLI_NO_MMAP", "unset", "test", "query and reference files through the sequential reader (what .gz and pipes always take)"},
    {"MM_CLI_NO_PREWARM", "unset", "test", "worker contexts come up with their first batch instead of beside the index build"},
    {"MM_SF_GRID", "the device's CU count", "debug", "resident workgroups of the streaming seed filter (measurement aid: how K3 scales with the CUs at work)"},
    {"MM_CLI_LATE_READER", "unset", "debug", "the query reader starts when the index is built instead of beside the build (measurement aid)"},
    {"MM_CLI_NO_SKETCH_REUSE", "unset", "test", "chunk-major runs recompute minimizers and sketches per chunk (mm_map_batch instead of mm_map_batch_reusing)"},
    {"MM_CLI_NO_HUGE", "unset", "test", "no transparent-huge-page arena for host blocks >= 4 MiB (huge_new.hpp)"},
    {"MM_CLI_FULL_TEARDOWN", "unset", "test", "destroy every object and run static destructors at exit instead of _exit after the last file is closed"},
    {"MM_CLASSIFY_THREADS", "32", "user", "host threads that tokenise the mappings file and format classify's outputs"},
    {"MM_EM_MAX_ITER", "unbounded", "test", "cap on EM iterations in classify (the reference has none)"},
    {"MM_EM_SLICE", "1024", "test", "EM iterations per mm_em_run / mm_em_continue call of classify"},
    {"MM_CLI_TIMING", "unset", "debug", "phase laps of the CLI on stderr (bench.py's e2e legs parse them)"},
    {"MM_CLI_FORMAT_TRACE", "unset", "debug", "per-thread times of the text formatting of a batch on stderr"},
    // ---- library: allocator (mm_common.hpp)
    {"MM_DEVICE_BYTES_CAP", "0 (off)", "test", "the library behaves as if every device had this many bytes: allocations beyond fail, mm_ctx_device_info reports it (placement tests)"},
    {"MM_INDEX_SCALE_MB", "8192", "test", "blocks from this size on are index-scale (pooled per device); a few MB exercise the pool on small inputs"},
    {"MM_NO_SLABS", "unset", "test", "worker buffers are not cut out of pooled index-scale blocks"},
    {"MM_NO_POOL_RESCUE", "unset", "test", "round-3 behaviour: a request the driver refuses hands the whole pool back instead of being served from it"},
    {"MM_RETURN_INDEX_BLOCKS", "unset", "test", "index-scale blocks go back to the driver when released instead of into the device's pool"},
    {"MM_INDEX_PRETRIM", "unset", "test", "a device-filling index build hands the pool back before it starts (round-4 mid-round behaviour)"},
    {"MM_INDEX_NO_PRETRIM", "unset", "test", "... and does not even trim the context's own cache"},
    {"MM_ALLOC_TRACE", "unset", "debug", "every block that comes from the driver, with its cost, on stderr"},
    {"MM_ALLOC_NO_MID_HEADROOM", "unset", "test", "device buffers of 256 KiB .. 64 MiB are asked for without the quarter of headroom that lets later, slightly larger batches reuse them (round 5 behaviour)"},
    {"MM_CTX_TRACE", "unset", "debug", "phases of mm_ctx_create (HIP initialisation, stream, allocator) on stderr"},
    {"MM_HOST_TIMING", "unset", "debug", "host-side sections of mm_map_batch and of the index build on stderr"},
    {"MM_PACK_SCALAR", "unset", "test", "mm_seqset_upload packs bases with the byte-table loop only (cross-check of the AVX2 path, host_pack.cpp)"},
    // ---- library: index build (mm_index.hip)
    {"MM_INDEX_PART_MAX", "2^31 - 2^24 entries", "test", "entries per partition of the hash sort (small values: the partitioned sort + merge on small inputs)"},
    {"MM_DUP_SAT", "65535", "test", "saturation value of the stored same-hash neighbour distances (small values: K5's scan fall-back)"},
    // ---- library: mapping (mm_map.hip, mm_seq.hip)
    {"MM_SKETCH_BITONIC", "unset", "test", "K2 for sketches beyond the LDS radix sort: bitonic network instead of one segmented device sort"},
    {"MM_EAGER_TIEBREAK", "unset", "test", "resolve every duplicate-hash strand by the reference's std::sort order up front instead of only where a strand vote reads one"},
    {"MM_FORCE_AMB_REDO", "unset", "test", "every read with an unresolved strand goes through the redo path"},
    {"MM_NO_HIT_FILTER", "unset", "test", "K3 without the exact seed-hit pre-filter: every hit of every kept list reaches the sort (parity tests of the raw hit list)"},
    {"MM_NO_FUSED_FILTER", "unset", "test", "K3 as probe_kernel + two-pass hit_filter_kernel for every read (what reads that do not fit the fused kernel take)"},
    {"MM_SF_ONESHOT", "unset", "test", "the fused seed filter as one read per workgroup instead of the resident streaming form"},
    {"MM_HF_STAGE_CAP", "auto", "test", "capacity of a read's survivor stage (tiny values: the overflow / re-filter path)"},
    {"MM_HF_WIDE_FROM", "13000", "tuning", "sketch size from which the two-pass filter counts in 32768 position slots instead of 8192"},
    {"MM_HITS_BITONIC", "unset", "test", "K4 hit sort: bitonic / LDS radix per read only, never the segmented device sort"},
    {"MM_SEGSORT_FROM", "auto", "tuning", "hits per read from which K4 uses the segmented device sort"},
    {"MM_L1_SERIAL", "unset", "test", "K4's L1 merge loop as the literal one-thread-per-read loop (cross-check of the wavefront form)"},
    {"MM_L2_FULL", "unset", "test", "K5 evaluates every window (the literal slide) instead of the exact skip-ahead"},
    {"MM_L2_NO_CODES", "unset", "test", "K5 classifies streamed entries again in every pass instead of parking their codes"},
    {"MM_L2_NO_DENSE", "unset", "test", "no dense path for sketches >= 13000 hashes (they take the long-read classes of l2_kernel)"},
    {"MM_L2_DENSE_FROM", "13000", "tuning", "sketch size from which K5 runs as l2_codes_kernel + l2_dense_kernel"},
    {"MM_L2_DENSE_NO_STOP", "unset", "test", "dense path without its early stop (every window evaluated)"},
    {"MM_L2_NO_SMALL_GROUPS", "unset", "test", "groups of fewer than three candidates also run as four-wave workgroups (one launch instead of two)"},
    {"MM_L2_HOST_GROUPS", "unset", "test", "candidate groups formed on the host instead of by l2_group_kernel"},
    {"MM_L2_NO_GROUP_SORT", "unset", "test", "K5 workgroups in read order instead of the order of their candidates' positions"},
    {"MM_L2_GROUP_SORT_MIN", "2048", "tuning", "groups from which the launch order is sorted"},
    {"MM_L2_XCD_ORDER", "unset", "tuning", "deal the position-sorted workgroup list out per XCD (measured: slower, DESIGN.md section 7)"},
    {"MM_L2_V1", "unset", "test", "K5's 10 kb class through l2_kernel (rank codes per entry) instead of the zone kernel l2z_kernel (mm_l2z.hpp): the cross-check and the A/B"},
    {"MM_L2_V1_LONG", "unset", "test", "the long-read classes (sketches of 3 073 .. 13 000 hashes) through l2_kernel, the 10 kb class through the zone kernel"},
    {"MM_L2_SMALL_QLDS", "unset", "tuning", "zone kernel, two-wave workgroups (groups of one or two candidates): the sketch in LDS as in the four-wave workgroups instead of searched in global memory"},
    {"MM_L2_NO_FUSE", "unset", "test", "zone kernel without the band predicted from L1's seed-hit count: its masks always come from a second pass over the stream"},
    {"MM_L2_NO_RANGES", "unset", "test", "the zone kernel's waves search their candidates' index ranges themselves (until round 6) instead of taking them from l2_ranges_kernel (cross-check)"},
    {"MM_L2Z_DBG", "unset", "debug", "zone kernel writes pivot-minus-estimate and its pass count INSTEAD OF RESULTS (tools/l2z_pivot_hist.py; 2: against the predicted estimate)"},
    {"MM_L2_ONE_STREAM", "unset", "test", "the two launches of K5's 10 kb class one behind the other on the context's stream instead of side by side (auxiliary stream)"},
    {"MM_L2_NO_SLOTS", "unset", "test", "K5 scratch indexed by wave number of the launch instead of per-XCD slots taken and given back"},
    {"MM_L2_SLOTS", "auto (resident waves)", "tuning", "number of K5 scratch slots"},
    {"MM_L2_STOP", "0", "debug", "K5 leaves after phase n WITHOUT RESULTS (tools/l2_stop.py)"},
    {"MM_L2_PHASES", "unset", "debug", "K5 phase clocks into its counters (tools/l2_long_phases.py; the zone kernel has them in a build with -DL2Z_CLOCKS only)"},
    {"MM_MZ_DBG", "0", "debug", "K1 leaves after step n WITHOUT RESULTS (tools/stage_ms.py)"},
    {"MM_SF_DBG", "0", "debug", "fused seed filter leaves after phase n WITHOUT RESULTS (tools/sf_dbg.py)"},
    {"MM_HF_DBG", "0", "debug", "two-pass filter leaves after phase n WITHOUT RESULTS"},
    {"MM_SF_PROF", "unset", "debug", "cycle counts per phase of the streaming seed filter on stderr"},
    // ---- library: EM and exchange (mm_post.hip, mm_api.hip)
    {"MM_EM_FORCE_COLLECTIVE", "unset", "test", "a one-rank communicator keeps P1-P3' | ncclAllReduce | finalize instead of the plain loop (how one GPU drives the multi-rank path)"},
    {"MM_EM_RESIDENT", "unset", "tuning", "EM as one resident kernel with grid barriers instead of one launch per phase (measured: slower, DESIGN.md section 4)"},
    {"MM_EM_SPLIT", "unset", "test", "with MM_EM_RESIDENT: the phases as launches all the same (cross-check); 2: P2 and P3 in one launch (measured slower than a launch per phase)"},
    {"MM_EM_GRID", "256 (128 resident)", "tuning", "workgroups of the EM kernels"},
    {"MM_EM_BARRIER_TICKS", "2 s", "test", "time-out of the resident kernel's grid barrier before it hands the run to the launch path"},
    {"MM_CLI_FORMAT_PART", "10000", "test", "mapping records per formatter thread of a batch (the text of a batch is formatted in up to eight parts and joined)"},
    {"MM_CLI_CLASSIFY_FROM_FILE", "unset", "test", "mapDirectly --then-classify reads PREFIX back and tokenises it (as `classify` does) instead of taking the lines it has just formatted from memory"},
    {"MM_EM_ORDER", "file", "debug", "\"count\": reads by mapping count for the thread-per-read E step (measurement aid)"},
    {"MM_EM_DBG", "0", "debug", "E-step variants of the round-4 measurements (tools/em_latency.py)"},
    {"MM_EM_PROF", "unset", "debug", "phase clocks of the resident EM kernel on stderr"},
    {"MM_GATHER_SELF_SEND", "unset", "test", "mm_mapping_gather sends the owner's own parts through ncclSend / ncclRecv too (one-GPU test of the exchange)"},
  };
  *n = sizeof T / sizeof T[0];
  return T;
}

// names of MM_* variables in the environment that the table does not know ("" when all are known)
inline std::string env_unknown() {
  size_t n = 0; const EnvSwitch* T = env_table(&n);
  std::string bad;
  for (char** e = environ; e && *e; ++e) {
    if (strncmp(*e, "MM_", 3) != 0) continue;
    const char* eq = strchr(*e, '=');
    const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
    // switches of the repository's own Python / shell side (bench.py, tests/, tools/): read there, not by the product — a fuzz campaign or an
    // A/B run (MM_LIB_PATH) must be combinable with strict mode.  tests/test_env_table.py holds this list against *.py and *.sh.
    static const char* const outside[] = {"MM_BENCH_", "MM_FUZZ_", "MM_TEST_", "MM_CLI_FUZZ_CASES", "MM_LIB_PATH"};
    bool ours = false;
    for (const char* pfx : outside) ours = ours || name.rfind(pfx, 0) == 0;
    if (ours) continue;
    bool known = false;
    for (size_t i = 0; i < n && !known; ++i) known = name == T[i].name;
    if (!known) bad += (bad.empty() ? "" : ", ") + name;
  }
  return bad;
}
inline bool env_strict() { const char* e = getenv("MM_STRICT_ENV"); return e && *e && strcmp(e, "0") != 0; }

}  // namespace mm
