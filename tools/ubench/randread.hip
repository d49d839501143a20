// Microbenchmark: random-access ceiling of HBM on one GPU.  Each group of G lanes reads one random, contiguous,
// 8*G-byte-aligned piece of a large buffer (G=1: isolated 8-byte words; G=16: 128-byte lines ...), U independent
// requests in flight per lane.  Prints requests/s and bytes/s.   hipcc --offload-arch=gfx950 -O3 randread.hip
// Usage: randread [GiB | <MiB>m]   (a buffer below 256 MiB measures the Infinity Cache, below 32 MiB the L2);
// the ALIGNED=false variants start each piece at any 8-byte offset (what an occurrence list does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int G, int U, bool ALIGNED = true>
__global__ void __launch_bounds__(256) rr(const uint64_t* __restrict__ buf, uint64_t n_units, int iters, uint64_t* __restrict__ sink) {
  const uint64_t gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const int sub = threadIdx.x % G;
  uint64_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint64_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const uint64_t unit = mix(gid * 1315423911ULL + (uint64_t)it * U + u) % n_units; v[u] = ALIGNED ? buf[unit * G + sub] : buf[mix(unit + 77) % (n_units * G - G) + sub]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc == 0x1234567) sink[0] = acc;
}
template <int G, int U, bool ALIGNED = true>
void run(const uint64_t* buf, uint64_t words, uint64_t* sink, int blocks, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  rr<G, U, ALIGNED><<<blocks, 256>>>(buf, words / G, 2, sink);
  CK(hipEventRecord(a));
  rr<G, U, ALIGNED><<<blocks, 256>>>(buf, words / G, iters, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double req = (double)blocks * 256 / G * iters * U;
  printf("G=%2d (%4d B contiguous%s) U=%d blocks=%d: %.2f G requests/s, %.2f TB/s\n", G, 8 * G, ALIGNED ? "" : ", unaligned", U, blocks, req / ms / 1e6, req * 8 * G / ms / 1e9);
}
int main(int argc, char** argv) {
  uint64_t words = 40ull << 27;
  if (argc > 1) { const std::string a = argv[1]; words = a.back() == 'm' ? (uint64_t)atoll(a.c_str()) << 17 : (uint64_t)atoll(a.c_str()) << 27; }
  printf("buffer %.1f MiB\n", words * 8 / 1048576.0);
  uint64_t *buf, *sink;
  CK(hipMalloc(&buf, words * 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(buf, 1, words * 8));
  const int blocks = 256 * 8 * 4, iters = 64;
  run<1, 8>(buf, words, sink, blocks, iters);
  run<2, 8>(buf, words, sink, blocks, iters);
  run<4, 8>(buf, words, sink, blocks, iters);
  run<8, 8>(buf, words, sink, blocks, iters);
  run<8, 8, false>(buf, words, sink, blocks, iters);
  run<16, 8>(buf, words, sink, blocks, iters);
  run<16, 8, false>(buf, words, sink, blocks, iters);
  run<16, 2>(buf, words, sink, blocks, iters * 2);
  run<64, 8>(buf, words, sink, blocks, iters);
  run<64, 2>(buf, words, sink, blocks, iters * 2);
  return 0;
}
