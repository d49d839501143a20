#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
template <int IPT>
__global__ void __launch_bounds__(256) k(const uint32_t* in, int n, uint32_t* ok, uint16_t* ov) {
  using Sort = rocprim::block_radix_sort<uint32_t, 256, IPT, uint16_t>;
  __shared__ typename Sort::storage_type st;
  uint32_t key[IPT]; uint16_t val[IPT];
  const int t = threadIdx.x;
  for (int i = 0; i < IPT; ++i) { int idx = t * IPT + i; key[i] = idx < n ? in[idx] : 0xffffffffu; val[i] = (uint16_t)idx; }
  Sort().sort(key, val, st);
  for (int i = 0; i < IPT; ++i) { ok[t * IPT + i] = key[i]; ov[t * IPT + i] = val[i]; }
}
int main() {
  const int IPT = 16, N = 256 * IPT, n = 2200;
  std::vector<uint32_t> h(N); for (int i = 0; i < N; ++i) h[i] = (uint32_t)(rand() % 1000) * 4000000u;
  uint32_t *d, *ok; uint16_t* ov;
  hipMalloc(&d, N * 4); hipMalloc(&ok, N * 4); hipMalloc(&ov, N * 2);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  k<IPT><<<1, 256>>>(d, n, ok, ov);
  std::vector<uint32_t> rk(N); std::vector<uint16_t> rv(N);
  hipMemcpy(rk.data(), ok, N * 4, hipMemcpyDeviceToHost); hipMemcpy(rv.data(), ov, N * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 1; i < n; ++i) { if (rk[i - 1] > rk[i]) ++bad; if (rk[i - 1] == rk[i] && rv[i - 1] > rv[i]) ++bad; }
  for (int i = 0; i < n; ++i) if (rk[i] != h[rv[i]]) ++bad;
  printf("bad=%d first keys %u %u %u vals %u %u %u\n", bad, rk[0], rk[1], rk[2], rv[0], rv[1], rv[2]);
  return 0;
}
