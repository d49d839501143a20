// Microbenchmark: issue cost of the integer VALU instructions K1's hashing is made of, in cycles per wave instruction on one SIMD
// (8 independent chains per lane, 4 waves per SIMD, every CU busy; 4.0 = full rate for a wave64 instruction on a 16-lane SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4096, CH = 8;
#define BODY(NAME, DECL, OP)                                                                         \
  __global__ void __launch_bounds__(256) NAME(uint64_t* out, uint32_t seed) {                        \
    DECL                                                                                              \
    for (int it = 0; it < ITER; ++it) {                                                               \
      _Pragma("unroll") for (int c = 0; c < CH; ++c) { OP }                                           \
    }                                                                                                 \
    uint64_t acc = 0;                                                                                 \
    for (int c = 0; c < CH; ++c) acc += (uint64_t)a[c];                                               \
    if (acc == 0x123456789abcull) out[0] = acc;                                                       \
  }
#define DECL32 uint32_t a[CH]; const uint32_t m = seed | 1u; for (int c = 0; c < CH; ++c) a[c] = threadIdx.x * 2654435761u + c + seed;
#define DECL64 uint64_t a[CH]; const uint32_t m = seed | 1u; for (int c = 0; c < CH; ++c) a[c] = (uint64_t)(threadIdx.x * 2654435761u + c) * 0x9E3779B97F4A7C15ull + seed;
BODY(k_add32, DECL32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(m));)
BODY(k_mul_lo, DECL32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(m));)
BODY(k_mul_hi, DECL32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "v"(m));)
BODY(k_mul_u24, DECL32, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(m));)
BODY(k_mad_u24, DECL32, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[c]) : "v"(m));)
BODY(k_mad_u64_u32, DECL64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"((uint32_t)a[c]), "v"(m) : "vcc");)
BODY(k_lshl_add_u64, DECL64, asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(a[c]));)
BODY(k_lshlrev_b64, DECL64, asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[c]));)
BODY(k_alignbit, DECL32, asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[c]) : "v"(m));)
BODY(k_perm, DECL32, asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(m));)
BODY(k_add3, DECL32, asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(m));)
BODY(k_xor, DECL32, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(m));)
BODY(k_cmp_u64, DECL64, asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[c]) : "v"((uint64_t)m), "v"((uint32_t)a[c]), "v"(m) : "vcc");)
BODY(k_fma_f64, DECL64, asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[c]));)
BODY(k_mul64, DECL64, a[c] = a[c] * 0x87c37b91114253d5ull + (uint64_t)m;)
template <typename K> void run(const char* name, K kern, uint64_t* out, int cus, double clk_ghz, double per_iter_ops) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<cus * 4, 256>>>(out, 12345u);                          // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  CK(hipEventRecord(e0));
  kern<<<cus * 4, 256>>>(out, 12345u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_instr_per_simd = 4.0 * ITER * CH * per_iter_ops;
  printf("%-16s %8.3f ms  %6.2f cycles per wave instruction (at %.2f GHz)\n", name, ms, ms * 1e-3 * clk_ghz * 1e9 / wave_instr_per_simd, clk_ghz);
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const double ghz = p.clockRate * 1e-6;
  uint64_t* out; CK(hipMalloc(&out, 64));
  printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
  run("v_add_u32", k_add32, out, p.multiProcessorCount, ghz, 1);
  run("v_xor_b32", k_xor, out, p.multiProcessorCount, ghz, 1);
  run("v_add3_u32", k_add3, out, p.multiProcessorCount, ghz, 1);
  run("v_alignbit_b32", k_alignbit, out, p.multiProcessorCount, ghz, 1);
  run("v_perm_b32", k_perm, out, p.multiProcessorCount, ghz, 1);
  run("v_mul_u32_u24", k_mul_u24, out, p.multiProcessorCount, ghz, 1);
  run("v_mad_u32_u24", k_mad_u24, out, p.multiProcessorCount, ghz, 1);
  run("v_mul_lo_u32", k_mul_lo, out, p.multiProcessorCount, ghz, 1);
  run("v_mul_hi_u32", k_mul_hi, out, p.multiProcessorCount, ghz, 1);
  run("v_mad_u64_u32", k_mad_u64_u32, out, p.multiProcessorCount, ghz, 1);
  run("v_lshl_add_u64", k_lshl_add_u64, out, p.multiProcessorCount, ghz, 1);
  run("v_lshlrev_b64", k_lshlrev_b64, out, p.multiProcessorCount, ghz, 1);
  run("v_cmp_lt_u64+cnd", k_cmp_u64, out, p.multiProcessorCount, ghz, 2);
  run("v_fma_f64", k_fma_f64, out, p.multiProcessorCount, ghz, 1);
  run("u64 * const + u", k_mul64, out, p.multiProcessorCount, ghz, 1);
  return 0;
}
