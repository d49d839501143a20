// Device-wide exclusive prefix sum (uint32 in -> uint64 out), three launches, no library.
// HBM-bound: reads the input twice, writes the output once.
#pragma once
#include "mm_common.hpp"

namespace mm {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;                      // per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ inline uint64_t wave_incl_scan_u64(uint64_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, total in *total
__device__ inline uint64_t block_excl_scan_u64(uint64_t v, uint64_t* total) {
  __shared__ uint64_t wsum[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint64_t inc = wave_incl_scan_u64(v);
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_THREADS / 64; ++i) { if (i < wid) base += wsum[i]; tot += wsum[i]; }
  __syncthreads();
  if (total) *total = tot;
  return base + inc - v;
}

static __global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const uint32_t* __restrict__ in, int64_t n, uint64_t* __restrict__ tile_sum) {
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t j = base + (int64_t)i * SCAN_THREADS + threadIdx.x;
    if (j < n) s += in[j];
  }
  uint64_t tot;
  block_excl_scan_u64(s, &tot);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// single block: in-place exclusive scan of tile sums (any length), grand total to *grand
static __global__ void __launch_bounds__(SCAN_THREADS) scan_tiles_kernel(uint64_t* __restrict__ tile_sum, int64_t ntiles, uint64_t* __restrict__ grand) {
  uint64_t carry = 0;
  for (int64_t base = 0; base < ntiles; base += SCAN_THREADS) {
    int64_t j = base + threadIdx.x;
    uint64_t v = j < ntiles ? tile_sum[j] : 0;
    uint64_t tot;
    uint64_t ex = block_excl_scan_u64(v, &tot);
    if (j < ntiles) tile_sum[j] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && grand) *grand = carry;
}

static __global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const uint32_t* __restrict__ in, int64_t n, const uint64_t* __restrict__ tile_sum,
                                                                  uint64_t* __restrict__ out) {
  // thread t owns SCAN_ITEMS consecutive items so that the output order is the input order
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { int64_t j = base + i; v[i] = j < n ? in[j] : 0u; s += v[i]; }
  uint64_t ex = block_excl_scan_u64(s, nullptr) + tile_sum[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { int64_t j = base + i; if (j < n) out[j] = ex; ex += v[i]; }
}

// out[0..n) = exclusive prefix of in[0..n); out[n] = total (out must hold n+1 entries). Returns nothing; async on `st`.
inline void exclusive_scan_u32_u64(const uint32_t* in, int64_t n, uint64_t* out, DBuf<uint64_t>& tile_tmp, hipStream_t st) {
  if (n <= 0) { MM_HIP(hipMemsetAsync(out, 0, sizeof(uint64_t), st)); return; }
  int64_t ntiles = ceil_div(n, SCAN_TILE);
  if ((int64_t)tile_tmp.n < ntiles) tile_tmp.alloc((size_t)ntiles);
  scan_reduce_kernel<<<dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st>>>(in, n, tile_tmp.p);
  MM_KERNEL_CHECK();
  scan_tiles_kernel<<<dim3(1), dim3(SCAN_THREADS), 0, st>>>(tile_tmp.p, ntiles, out + n);
  MM_KERNEL_CHECK();
  scan_apply_kernel<<<dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st>>>(in, n, tile_tmp.p, out);
  MM_KERNEL_CHECK();
}

}  // namespace mm
