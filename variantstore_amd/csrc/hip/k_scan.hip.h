// k_scan.hip.h -- exclusive scans over the regions of a batch.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_image.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Exclusive scan of a uint64 array (three launches; sizes here are <= a few 1e7).
// ---------------------------------------------------------------------------
constexpr int kScanBlock = 256, kScanItems = 8, kScanTile = kScanBlock * kScanItems;

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t* total) {
  __shared__ uint64_t wsum[kScanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint64_t incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  uint64_t woff = 0, tot = 0;
  for (int w = 0; w < kScanBlock / 64; ++w) {
    if (w < wid) woff += wsum[w];
    tot += wsum[w];
  }
  __syncthreads();
  *total = tot;
  return woff + incl - v;
}

template <typename T>
__global__ void __launch_bounds__(kScanBlock) k_scan_tile_sums(const T* in, uint64_t n, uint64_t* tile_sums) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint64_t s = 0;
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += in[base + i];
  uint64_t tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single block: tile_sums -> exclusive prefix in place; writes the grand total to out[n]
__global__ void __launch_bounds__(kScanBlock) k_scan_spine(uint64_t* tile_sums, uint64_t ntiles, uint64_t* grand_total) {
  uint64_t carry = 0;
  for (uint64_t base = 0; base < ntiles; base += kScanBlock) {
    const uint64_t i = base + threadIdx.x;
    const uint64_t v = i < ntiles ? tile_sums[i] : 0;
    uint64_t tot;
    const uint64_t ex = block_exclusive_scan(v, &tot);
    if (i < ntiles) tile_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *grand_total = carry;
}

template <typename T>
__global__ void __launch_bounds__(kScanBlock) k_scan_apply(const T* in, uint64_t n, const uint64_t* tile_sums, uint64_t* out) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint64_t loc[kScanItems];
  uint64_t s = 0;
  for (int i = 0; i < kScanItems; ++i) {
    loc[i] = base + i < n ? (uint64_t)in[base + i] : 0;
    s += loc[i];
  }
  uint64_t tot;
  uint64_t ex = block_exclusive_scan(s, &tot) + tile_sums[blockIdx.x];
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += loc[i];
  }
}

// ---------------------------------------------------------------------------
// Both offset arrays of a batch -- var_begin (slots) and car_base (padded arena entries) -- in ONE pass over the
// regions: three launches instead of six.  The grand totals also go to `totals` (mapped host memory).
// ---------------------------------------------------------------------------
struct Scan2 { uint64_t a, c; };

__device__ __forceinline__ Scan2 block_exclusive_scan2(Scan2 v, Scan2* total) {
  __shared__ Scan2 wsum[kScanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  Scan2 incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t ta = __shfl_up(incl.a, d, 64), tc = __shfl_up(incl.c, d, 64);
    if (lane >= d) { incl.a += ta; incl.c += tc; }
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  Scan2 woff{0, 0}, tot{0, 0};
  for (int w = 0; w < kScanBlock / 64; ++w) {
    if (w < wid) { woff.a += wsum[w].a; woff.c += wsum[w].c; }
    tot.a += wsum[w].a; tot.c += wsum[w].c;
  }
  __syncthreads();
  *total = tot;
  return Scan2{woff.a + incl.a - v.a, woff.c + incl.c - v.c};
}

__global__ void __launch_bounds__(kScanBlock) k_scan2_tile_sums(const uint64_t* nvar, const uint64_t* ncar, uint64_t n, Scan2* tile_sums) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  Scan2 s{0, 0};
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) { s.a += nvar[base + i]; s.c += ncar[base + i]; }
  Scan2 tot;
  block_exclusive_scan2(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single block: tile sums -> exclusive prefixes in place; grand totals to the two [n] entries and to totals[0..1]
__global__ void __launch_bounds__(kScanBlock) k_scan2_spine(Scan2* tile_sums, uint64_t ntiles, uint64_t* var_end, uint64_t* car_end,
                                                            uint64_t* totals) {
  Scan2 carry{0, 0};
  for (uint64_t base = 0; base < ntiles; base += kScanBlock) {
    const uint64_t i = base + threadIdx.x;
    const Scan2 v = i < ntiles ? tile_sums[i] : Scan2{0, 0};
    Scan2 tot;
    const Scan2 ex = block_exclusive_scan2(v, &tot);
    if (i < ntiles) tile_sums[i] = Scan2{carry.a + ex.a, carry.c + ex.c};
    carry.a += tot.a; carry.c += tot.c;
  }
  if (threadIdx.x == 0) {
    *var_end = carry.a; *car_end = carry.c;
    if (totals) { totals[0] = carry.a; totals[1] = carry.c; }
  }
}

__global__ void __launch_bounds__(kScanBlock) k_scan2_apply(const uint64_t* nvar, const uint64_t* ncar, uint64_t n, const Scan2* tile_sums,
                                                            uint64_t* var_begin, uint64_t* car_base) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  Scan2 loc[kScanItems];
  Scan2 s{0, 0};
  for (int i = 0; i < kScanItems; ++i) {
    loc[i] = base + i < n ? Scan2{nvar[base + i], ncar[base + i]} : Scan2{0, 0};
    s.a += loc[i].a; s.c += loc[i].c;
  }
  Scan2 tot;
  Scan2 ex = block_exclusive_scan2(s, &tot);
  const Scan2 ts = tile_sums[blockIdx.x];
  ex.a += ts.a; ex.c += ts.c;
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) { var_begin[base + i] = ex.a; car_base[base + i] = ex.c; }
    ex.a += loc[i].a; ex.c += loc[i].c;
  }
}

// ---------------------------------------------------------------------------
// Up to kScanSmallMax elements: ONE launch.  A walking batch of ten thousand regions is a string of a dozen dependent launches of a
// few microseconds each, and its three scans were nine of them.  Up to four blocks of 1024 threads, a tile of 4096 elements each;
// a block adds up the tiles in front of its own BY ITSELF (at most 12 coalesced loads per thread, all in flight at once) instead of
// waiting for anybody -- the same trade as k_t6_mid's -- and scans its tile with one block scan.  in and out must not overlap.
// (Measured before this form, round 5: one block for everything with 16 consecutive elements per thread -- its strided accesses
//  took 8 - 21 us per launch -- and one wave per segment with a wave scan per 64 elements -- coalesced, but ten rounds of ~120
//  dependent cross-lane moves each: 10 - 26 us.)
// ---------------------------------------------------------------------------
constexpr uint32_t kScanSmallBlock = 1024, kScanSmallItems = 4, kScanSmallTile = kScanSmallBlock * kScanSmallItems, kScanSmallMax = 4 * kScanSmallTile;
constexpr uint32_t kScanSmallPre = (kScanSmallMax - kScanSmallTile) / kScanSmallBlock;   // loads per thread that cover every tile in front of the last

// out[0..n) = exclusive prefix of in, out[n] = total
template <typename T>
__global__ void __launch_bounds__(kScanSmallBlock) k_scan_small(const T* __restrict__ in, uint32_t n, uint64_t* __restrict__ out) {
  __shared__ uint64_t wsum[kScanSmallBlock / 64], wpre[kScanSmallBlock / 64];
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, tile0 = blockIdx.x * kScanSmallTile, base = tile0 + threadIdx.x * kScanSmallItems;
  uint64_t pre = 0, v[kScanSmallItems];
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallPre; ++k) {
    const uint32_t i = k * kScanSmallBlock + threadIdx.x;
    pre += i < tile0 ? (uint64_t)in[i] : 0;
  }
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallItems; ++k) v[k] = base + k < n ? (uint64_t)in[base + k] : 0;
  uint64_t own = 0;
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallItems; ++k) own += v[k];
  uint64_t incl = own;
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up(incl, d, 64);
    pre += __shfl_xor(pre, d, 64);
    if (lane >= (uint32_t)d) incl += t;
  }
  if (lane == 63) { wsum[wid] = incl; wpre[wid] = pre; }
  __syncthreads();
  uint64_t ex = incl - own;
  for (uint32_t w = 0; w < kScanSmallBlock / 64; ++w) ex += wpre[w] + (w < wid ? wsum[w] : 0);
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallItems; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanSmallBlock - 1) out[n] = ex;   // (elements past n count as zero)
}
// both offset arrays of a batch (k_scan2_*): prefixes, the totals at [n] and in `totals` (mapped host memory; NULL: not wanted)
__global__ void __launch_bounds__(kScanSmallBlock) k_scan2_small(const uint64_t* __restrict__ nvar, const uint64_t* __restrict__ ncar, uint32_t n,
                                                                 uint64_t* __restrict__ var_begin, uint64_t* __restrict__ car_base, uint64_t* totals) {
  __shared__ Scan2 wsum[kScanSmallBlock / 64], wpre[kScanSmallBlock / 64];
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, tile0 = blockIdx.x * kScanSmallTile, base = tile0 + threadIdx.x * kScanSmallItems;
  Scan2 pre{0, 0};
  uint64_t va[kScanSmallItems], vc[kScanSmallItems];
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallPre; ++k) {
    const uint32_t i = k * kScanSmallBlock + threadIdx.x;
    const bool live = i < tile0;
    pre.a += live ? nvar[i] : 0; pre.c += live ? ncar[i] : 0;
  }
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallItems; ++k) {
    const bool live = base + k < n;
    va[k] = live ? nvar[base + k] : 0; vc[k] = live ? ncar[base + k] : 0;
  }
  Scan2 own{0, 0};
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallItems; ++k) { own.a += va[k]; own.c += vc[k]; }
  Scan2 incl = own;
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t ta = __shfl_up(incl.a, d, 64), tc = __shfl_up(incl.c, d, 64);
    pre.a += __shfl_xor(pre.a, d, 64); pre.c += __shfl_xor(pre.c, d, 64);
    if (lane >= (uint32_t)d) { incl.a += ta; incl.c += tc; }
  }
  if (lane == 63) { wsum[wid] = incl; wpre[wid] = pre; }
  __syncthreads();
  Scan2 ex{incl.a - own.a, incl.c - own.c};
  for (uint32_t w = 0; w < kScanSmallBlock / 64; ++w) {
    const Scan2 p = wpre[w], t = wsum[w];
    ex.a += p.a + (w < wid ? t.a : 0); ex.c += p.c + (w < wid ? t.c : 0);
  }
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallItems; ++k) {
    if (base + k < n) { var_begin[base + k] = ex.a; car_base[base + k] = ex.c; }
    ex.a += va[k]; ex.c += vc[k];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanSmallBlock - 1) {
    var_begin[n] = ex.a; car_base[n] = ex.c;
    if (totals) { totals[0] = ex.a; totals[1] = ex.c; }
  }
}

}  // namespace vsamd
