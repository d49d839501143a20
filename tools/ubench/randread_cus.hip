// Microbenchmark: does the rate of random 64-byte requests follow the number of CUs that issue them (a CU has a fixed number of requests in
// flight and the latency does the rest), or is it the memory side's?  randread.hip's G = 8 case (8 lanes x 8 B) and the seed filter's shape
// (4 lanes x 16 B) on streams restricted to the first N CUs (hipExtStreamCreateWithCUMask), at several numbers of waves per CU and loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 randread_cus.hip -o randread_cus;  ./randread_cus [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// 4 lanes x 16 bytes = one 64-byte sector, U loads in flight per lane; persistent: every workgroup loops `iters` times
template <int U>
__global__ void __launch_bounds__(256) rr16(const ulonglong2* __restrict__ buf, uint64_t n_sectors, int iters, uint64_t* __restrict__ sink) {
  const uint64_t gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
  const int sub = threadIdx.x & 3;
  uint64_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    ulonglong2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const uint64_t s = mix(gid * 1315423911ULL + (uint64_t)it * U + u) % n_sectors; v[u] = buf[s * 4 + sub]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y;
  }
  if (acc == 0x1234567) sink[0] = acc;
}
template <int U>
void run(hipStream_t st, int cus, int blocks_per_cu, const ulonglong2* buf, uint64_t sectors, uint64_t* sink, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = cus * blocks_per_cu;
  rr16<U><<<blocks, 256, 0, st>>>(buf, sectors, 2, sink);
  CK(hipEventRecord(a, st));
  rr16<U><<<blocks, 256, 0, st>>>(buf, sectors, iters, sink);
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double req = (double)blocks * 64 * iters * U;
  printf("  CUs %3d  waves/CU %2d  loads in flight per lane %2d (%5d requests per CU): %6.2f G requests/s = %5.3f per CU per ns, %.2f TB/s\n", cus, blocks_per_cu * 4, U,
         blocks_per_cu * 64 * U, req / ms / 1e6, req / ms / 1e6 / cus, req * 64 / ms / 1e9);
}
int main(int argc, char** argv) {
  const uint64_t gib = argc > 1 ? (uint64_t)atoll(argv[1]) : 90;
  const uint64_t sectors = (gib << 30) / 64;
  printf("buffer %llu GiB, 64-byte requests as 4 lanes x 16 B\n", (unsigned long long)gib);
  ulonglong2* buf; uint64_t* sink;
  CK(hipMalloc(&buf, sectors * 64)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(buf, 1, sectors * 64));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int total = p.multiProcessorCount;
  for (int cus : {total, total / 2, total / 4, total / 8}) {
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    // every (total / cus)-th CU: the enabled CUs are spread over the XCDs and shader engines whatever the numbering is
    for (int i = 0; i < total; i += total / cus) mask[i / 32] |= 1u << (i % 32);
    hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    for (int bpc : {8, 4, 2}) { run<8>(st, cus, bpc, buf, sectors, sink, 48); run<2>(st, cus, bpc, buf, sectors, sink, 192); }
    run<11>(st, cus, 4, buf, sectors, sink, 48);
    CK(hipStreamDestroy(st));
  }
  return 0;
}
