// Microbenchmark: what one CU's LDS does per cycle for the seed filter's access shapes — 1024 threads (16 waves) of one workgroup per CU, every
// wave issuing the same instruction kind in a loop: ds_add_u32 to random words of a 33 KB counter array (all 64 lanes, and a third of them),
// the same to ONE address per wave-instruction, ds_read_b32 of random words, ds_read_b128 / ds_write_b128 of consecutive 16-byte pieces.
// Prints cycles per wave-instruction per CU (s_memtime around the loop, slowest wave of the workgroup).
//   hipcc --offload-arch=gfx950 -O3 lds_rates.hip -o lds_rates && ./lds_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
constexpr int WORDS = 8256, ITERS = 256, UNROLL = 8;
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long* __restrict__ out, uint32_t* __restrict__ sink) {
  __shared__ __align__(16) uint32_t cnt[WORDS];
  __shared__ __align__(16) uint4 big[6144];
  const int tid = threadIdx.x;
  for (int i = tid; i < WORDS; i += 1024) cnt[i] = 0;
  for (int i = tid; i < 6144; i += 1024) big[i] = make_uint4(i, i, i, i);
  __syncthreads();
  uint32_t idx[UNROLL];
  for (int u = 0; u < UNROLL; ++u) idx[u] = mix32(tid * 9781u + u * 77u + blockIdx.x) % WORDS;
  uint32_t acc = 0; uint4 acc4 = make_uint4(0, 0, 0, 0);
  const bool active = MODE == 1 ? (tid % 3 == 0) : true;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t a = (idx[u] + (uint32_t)it * 97u) % WORDS;
      if (MODE == 0) atomicAdd(&cnt[a], 1u);
      if (MODE == 1) { if (active) atomicAdd(&cnt[a], 1u); }
      if (MODE == 2) atomicAdd(&cnt[(tid >> 6) * 64 + u], 1u);                 // one address per wave-instruction
      if (MODE == 3) acc += cnt[a];
      if (MODE == 4) { const uint4 v = big[(tid + (it * UNROLL + u) * 1024) % 6144]; acc4.x ^= v.x; acc4.y ^= v.y; acc4.z ^= v.z; acc4.w ^= v.w; }
      if (MODE == 5) big[(tid + (it * UNROLL + u) * 1024) % 6144] = make_uint4(a, a, a, a);
      if (MODE == 6) acc += (cnt[a >> 5] >> (a & 31)) & 1u;                     // the alive test's shape: 258 words
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (acc + acc4.x + acc4.y + acc4.z + acc4.w + cnt[tid] + big[tid].x == 0x12345678u) sink[0] = acc;
}
template <int MODE>
void run(const char* what, int blocks, unsigned long long* out, uint32_t* sink) {
  k<MODE><<<blocks, 1024>>>(out, sink);
  k<MODE><<<blocks, 1024>>>(out, sink);
  CK(hipDeviceSynchronize());
  unsigned long long h[1024]; CK(hipMemcpy(h, out, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
  const double per = s / blocks / ((double)ITERS * UNROLL * 16);      // cycle-counter ticks per wave-instruction (16 waves share the CU's LDS)
  printf("%-62s %7.2f ticks of s_memtime per wave-instruction per CU (100 MHz ticks x 24 = %.1f cycles at 2.4 GHz)\n", what, per, per * 24);
}
int main() {
  unsigned long long* out; uint32_t* sink; CK(hipMalloc(&out, 1024 * 8)); CK(hipMalloc(&sink, 4));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount;
  printf("%d workgroups of 1024 threads (one per CU), %d wave-instructions per wave\n", blocks, ITERS * UNROLL);
  run<0>("ds_add_u32, 64 lanes, random words of 33 KB", blocks, out, sink);
  run<1>("ds_add_u32, every third lane, random words", blocks, out, sink);
  run<2>("ds_add_u32, 64 lanes, ONE word", blocks, out, sink);
  run<3>("ds_read_b32, 64 lanes, random words", blocks, out, sink);
  run<6>("ds_read_b32, 64 lanes, random words of 1 KB (alive test)", blocks, out, sink);
  run<4>("ds_read_b128, consecutive pieces", blocks, out, sink);
  run<5>("ds_write_b128, consecutive pieces", blocks, out, sink);
  return 0;
}
