// Microbenchmark: can the SCALAR memory path (s_load_dwordx16 = one 64-byte sector per request, through the scalar data cache) answer random requests
// beside the vector path?  K3 is bound by ~190e6 answered 64-byte requests per second and CU on the vector path whatever level answers them
// (tools/ubench/randread: 49e9/s device-wide from HBM, 81e9/s from L2): if the scalar path has request capacity of its own, a table look-up — one
// sector per hash — could go there.    hipcc --offload-arch=gfx950 -O3 scalar_randread.hip -o scalar_randread ;  scalar_randread [GiB | <MiB>m]
// Three kernels: vector only (4 lanes x 16 B per sector, 8 in flight per lane), scalar only (U sectors in flight per wave), both in one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t v16u __attribute__((ext_vector_type(16)));
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ inline uint64_t uni64(uint64_t v) { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32; }

template <int US, int UV>   // US scalar sectors and UV vector sectors (per 4-lane group) in flight per wave and round
__global__ void __launch_bounds__(256) mixed(const ulonglong2* __restrict__ buf, uint64_t n_sectors, int iters, uint64_t* __restrict__ sink) {
  const uint64_t wave = uni64(((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6);
  const uint64_t grp = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
  const int sub = threadIdx.x & 3;
  uint64_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    v16u sr[US > 0 ? US : 1];
    ulonglong2 vr[UV > 0 ? UV : 1];
    // (the loads and the wait for them are ONE asm statement: the compiler must not touch — or re-use — the destination registers while a load is in flight)
    const char* sp[4] = {nullptr, nullptr, nullptr, nullptr};
#pragma unroll
    for (int u = 0; u < US; ++u) sp[u] = reinterpret_cast<const char*>(buf) + uni64(mix(wave * 2654435761ULL + (uint64_t)it * 64 + u) % n_sectors) * 64;
#pragma unroll
    for (int u = 0; u < UV; ++u) { const uint64_t sec = mix(grp * 1315423911ULL + (uint64_t)it * 64 + 32 + u) % n_sectors; vr[u] = buf[sec * 4 + sub]; }
    if (US == 1) asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(sr[0]) : "s"(sp[0]) : "memory");
    if (US == 2) asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(sr[0]), "=&s"(sr[US > 1 ? 1 : 0]) : "s"(sp[0]), "s"(sp[1]) : "memory");
    if (US == 4) asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %5, 0x0\n\ts_load_dwordx16 %2, %6, 0x0\n\ts_load_dwordx16 %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                              : "=&s"(sr[0]), "=&s"(sr[US > 1 ? 1 : 0]), "=&s"(sr[US > 2 ? 2 : 0]), "=&s"(sr[US > 3 ? 3 : 0]) : "s"(sp[0]), "s"(sp[1]), "s"(sp[2]), "s"(sp[3]) : "memory");
#pragma unroll
    for (int u = 0; u < US; ++u) acc += sr[u][0] + sr[u][15];
#pragma unroll
    for (int u = 0; u < UV; ++u) acc += vr[u].x;
  }
  if (acc == 0x1234567) sink[0] = acc;
}
template <int US, int UV>
void run(const ulonglong2* buf, uint64_t sectors, uint64_t* sink, int blocks, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  mixed<US, UV><<<blocks, 256>>>(buf, sectors, 2, sink);
  CK(hipEventRecord(a));
  mixed<US, UV><<<blocks, 256>>>(buf, sectors, iters, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double waves = (double)blocks * 4, sreq = waves * iters * US, vreq = waves * 16 * iters * UV;
  printf("scalar %d + vector %d in flight per wave: %.2f G scalar sectors/s + %.2f G vector sectors/s = %.2f G/s (%.2f ms)\n", US, UV, sreq / ms / 1e6, vreq / ms / 1e6, (sreq + vreq) / ms / 1e6, ms);
}
int main(int argc, char** argv) {
  uint64_t bytes = 40ull << 30;
  if (argc > 1) { const std::string a = argv[1]; bytes = a.back() == 'm' ? (uint64_t)atoll(a.c_str()) << 20 : (uint64_t)atoll(a.c_str()) << 30; }
  printf("buffer %.1f MiB\n", bytes / 1048576.0);
  ulonglong2* buf; uint64_t* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(buf, 1, bytes));
  const uint64_t sectors = bytes / 64;
  const int blocks = 256 * 8, iters = 256;
  run<0, 8>(buf, sectors, sink, blocks, iters);
  run<1, 0>(buf, sectors, sink, blocks, iters * 4);
  run<2, 0>(buf, sectors, sink, blocks, iters * 4);
  run<4, 0>(buf, sectors, sink, blocks, iters * 4);
  run<2, 8>(buf, sectors, sink, blocks, iters);
  run<4, 8>(buf, sectors, sink, blocks, iters);
  return 0;
}
