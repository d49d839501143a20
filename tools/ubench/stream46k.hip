// Microbenchmark: every wave streams one random, contiguous region of R bytes (the K5 candidate pattern: 46 KB),
// B wave-loads of 512 B in flight per wait, optionally with LDS allocated to limit occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int B>
__global__ void __launch_bounds__(256) st(const uint64_t* __restrict__ buf, uint64_t words, int chunks, uint64_t* __restrict__ sink, uint64_t align_mask) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63;
  const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t start = (mix(wid) % (words - (uint64_t)chunks * 64 - 64)) & align_mask;
  const uint64_t* p = buf + start;
  uint64_t acc = 0;
  for (int c = 0; c < chunks; c += B) {
    uint64_t v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) v[i] = p[(uint64_t)(c + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < B; ++i) acc += __popcll(__ballot((v[i] & 0xff) == 1));
  }
  if (acc == 0x1234567) { sink[0] = acc; lds[0] = 1; }
}
template <int B>
void run(const uint64_t* buf, uint64_t words, uint64_t* sink, int waves, int chunks, int lds_bytes, uint64_t align_mask = ~7ull) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  if (lds_bytes > 65536) CK(hipFuncSetAttribute((const void*)st<B>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  st<B><<<waves / 4, 256, lds_bytes>>>(buf, words, chunks, sink, align_mask);
  CK(hipEventRecord(a));
  st<B><<<waves / 4, 256, lds_bytes>>>(buf, words, chunks, sink, align_mask);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("B=%2d chunks=%d (%d KB/wave) lds=%d KB/WG align %d B: %.2f ms, %.2f TB/s\n", B, chunks, chunks / 2, lds_bytes >> 10, (int)(8 * (~align_mask + 1)), ms, (double)waves * chunks * 512 / ms / 1e9);
}
int main(int argc, char** argv) {
  const uint64_t gib = argc > 1 ? atoll(argv[1]) : 44;
  const uint64_t words = gib << 27;
  uint64_t *buf, *sink;
  CK(hipMalloc(&buf, words * 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(buf, 1, words * 8));
  const int waves = 380000 / 4 * 4;
  run<8>(buf, words, sink, waves, 96, 0);
  run<8>(buf, words, sink, waves, 96, 39 * 1024);
  run<4>(buf, words, sink, waves, 96, 39 * 1024);
  run<16>(buf, words, sink, waves, 96, 39 * 1024);
  run<1>(buf, words, sink, waves, 96, 39 * 1024);
  run<8>(buf, words, sink, waves, 96, 78 * 1024);
  run<8>(buf, words, sink, waves, 32, 39 * 1024);
  run<8>(buf, words, sink, waves, 96, 39 * 1024, ~0ull);      // starts at any 8-byte offset (the K5 candidates)
  run<8>(buf, words, sink, waves, 96, 39 * 1024, ~15ull);     // 128-byte aligned
  return 0;
}
