// k_digest.hip.h -- order-independent digest of a result.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_image.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Order-independent digest of a result: sum over kept variants of a 64-bit mix
// of (region, pos, ref bases, alt bases, every carrier word with its rank).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
// pass 1, one wave per TABLE row: the carrier part of the row's hash (shared rows: once for all regions reporting them)
__global__ void __launch_bounds__(256) k_digest_rows(DevResult r, uint64_t* row_hash) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t a = wave; a < r.A; a += nwaves) {
    const VariantRow v = row_load(r.rows, a);
    uint64_t h = 0;
    const uint32_t cnt = row_count(v);
    const uint32_t* car32 = reinterpret_cast<const uint32_t*>(r.carriers) + v.car_begin;
    const uint16_t* car16 = reinterpret_cast<const uint16_t*>(r.carriers) + v.car_begin;
    for (uint32_t k = lane; k < cnt; k += 64) {   // the digest is defined over the 32-bit form of a carrier word
      const uint32_t c = r.car_width == 2 ? ((uint32_t)(car16[k] & 0x1FFFu) | ((uint32_t)(car16[k] >> 13) << 29)) : car32[k];
      h += mix64(((uint64_t)c << 32) | k);
    }
    for (int d = 32; d >= 1; d >>= 1) h += __shfl_down(h, d, 64);
    if (lane == 0) row_hash[a] = h;
  }
}
// pass 2, one wave per region: every reported (region, row) pair adds mix(row hash + mix(region, pos, ref bases, alt bases))
__global__ void __launch_bounds__(256) k_digest(DevImage im, DevResult r, const uint64_t* row_hash, uint64_t* digest) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= r.Q) return;
  const uint64_t n = r.q_nvar[q], a0 = r.var_begin[q];
  uint64_t acc = 0;
  for (uint64_t j = threadIdx.x & 63; j < n; j += 64) {
    const VariantRow v = row_load(r.rows, a0 + j);
    if (row_dropped(v)) continue;
    uint64_t s = mix64((uint32_t)q * 0x9E3779B97F4A7C15ULL + v.pos);
    for (uint32_t i = 0; i < v.ref_len; ++i) s = mix64(s ^ (im.seq_codes[v.ref_off + i] + 1));
    s = mix64(s ^ 0xABCDEFULL);
    for (uint32_t i = 0; i < v.alt_len; ++i) s = mix64(s ^ (im.seq_codes[v.alt_off + i] + 1));
    acc += mix64(row_hash[a0 + j] + s);   // the row's hash is mixed once more so carriers are tied to their variant
  }
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd((unsigned long long*)digest, (unsigned long long)acc);
}

}  // namespace vsamd
