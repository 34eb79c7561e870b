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

}  // namespace vsamd
