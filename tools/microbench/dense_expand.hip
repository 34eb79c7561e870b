// dense_expand.hip -- the DENSE phase of the carrier expansion, stand-alone: the production formulation against the
// "equal output shares" formulation VERDICT r5 asked for (next #1), on the bench cohort's own distribution of dense variants.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/dense_expand tools/microbench/dense_expand.hip
//   tools/microbench/dense_expand [variants = 200000] [only this formulation: 0 .. 3]
//
// Input: `variants` class bit rows of 2504 samples (40 words, bit 0 = "ref" clear), carrier fraction 1 - (1 - af)^2 with af
// log-uniform over [0.137, 1] -- the bench generator's variants of more than 640 carriers (AF = 10^-3U) -- one genotype group
// word per 8 carriers, arena ranges on group boundaries.  Every kernel expands every variant into 16-bit carrier words
// (id | gt << 13), 8 per 16-byte group, and is checked against a host expansion.
//
//   slices   THE PRODUCTION CODE: expand_task<.., DENSE only> of k_expand.hip.h, included as it is (two rounds of 20-bit slices per lane,
//            ids peeled in pairs into an LDS list at prefix-sum positions, 16-byte groups copied out with the genotypes merged)
//   shares   equal OUTPUT shares: lane L produces the carriers [L k, (L + 1) k), k = 8 ceil(groups / 64): prefix sums of the 40 word
//            popcounts in LDS, bisection for the lane's first word, select inside the word by binary descent, then k ids peeled from
//            the bit stream in registers (refill from the LDS copy of the row when a 32-bit window runs dry), four dwords per group
//            merged with the group's genotype word and stored straight from registers -- no id list in LDS
//   shares+t the same with the finished groups transposed through LDS (ds_write_b128 / ds_read_b128) so that a wave's stores are
//            1 KiB-contiguous like the production copy-out
//   mbcnt    VERDICT r5's other axis: one pass per 64-bit row word with exec = the word, v_mbcnt ranks, ds_write_b16 into an LDS id list,
//            lane-per-group copy-out (round 1's dense path, here with 16-bit ids and group genotype words)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include "../../variantstore_amd/csrc/hip/kernels.hip.h"

using namespace vsamd;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t kSamples = 2504, kWpc = 40;

// ---- the equal-output-shares formulation of one dense variant (wave-uniform cnt / cls / gt0 / cb) ----
// lds: [0, 64) word prefixes, [64, 64 + 2 * 66) the row as dwords (one zero word behind it), then (TRANSPOSE) the finished groups
template <bool TRANSPOSE>
__device__ __forceinline__ void expand_equal_shares(const DevImage& im, uint4* __restrict__ arena_groups, uint32_t* lds, uint32_t lane, uint64_t mine,
                                                    uint32_t cnt, uint64_t gt0, uint64_t cb, uint32_t m_both) {
  uint32_t* s_pre = lds;
  uint32_t* s_rowd = lds + 64;
  uint4* s_groups = reinterpret_cast<uint4*>(lds + 64 + 136);
  const uint32_t G = (cnt + 7) >> 3, m = (G + 63) >> 6;          // groups of the variant, groups per lane
  const uint32_t pc = (uint32_t)__popcll(mine);
  const uint32_t incl = wave_inclusive_scan(pc);
  s_pre[lane] = lane < kWpc ? incl - pc : cnt;                    // (entries behind the row: never "at or below" a carrier index)
  reinterpret_cast<uint64_t*>(s_rowd)[lane] = mine;              // lanes >= wpc hold 0: the zero words behind the row
  if (lane < 2) reinterpret_cast<uint64_t*>(s_rowd)[64 + lane] = 0;
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");   // (the LDS writes above are read by other lanes of the wave below)
  const uint32_t g_first = lane * m;                              // this lane's groups: [g_first, g_first + m) below G
  const uint32_t t0 = g_first * 8;
  const bool live = g_first < G;
  // the word that holds carrier t0: the last one whose prefix is <= t0
  uint32_t w = 0;
#pragma unroll
  for (uint32_t step = 32; step; step >>= 1)
    if (s_pre[w + step] <= t0) w += step;
  uint32_t r = t0 - s_pre[w];                                     // rank of the lane's first carrier inside that word
  const uint32_t lo = s_rowd[2 * w], hi = s_rowd[2 * w + 1];
  const uint32_t c_lo = __popc(lo);
  uint32_t wi = 2 * w, cur = lo;
  if (r >= c_lo) { r -= c_lo; cur = hi; wi += 1; }
  // clear the r lowest set bits of `cur` (binary descent to the position of the r-th one)
  {
    uint32_t x = cur, pos = 0;
#pragma unroll
    for (uint32_t s = 16; s; s >>= 1) {
      const uint32_t c = __popc(x & ((1u << s) - 1u));
      const bool ge = r >= c;
      r -= ge ? c : 0u;
      x >>= ge ? s : 0u;
      pos += ge ? s : 0u;
    }
    cur &= ~0u << pos;
  }
  if (!live) cur = 0;
  uint32_t base = wi * 32;
  // genotype words of the lane's groups (TRANSPOSE: fetched per pass below instead)
  const uint32_t* __restrict__ gtw = im.gt_groups + (gt0 >> 3);
  uint4* __restrict__ dst = arena_groups + (cb >> 3);
  for (uint32_t j = 0; j < m; ++j) {
    const uint32_t g = g_first + j;
    const bool on = g < G;
    const uint32_t nk = on ? (cnt - 8 * g < 8 ? cnt - 8 * g : 8u) : 0u;   // ids of this group (the last group of a variant may be short)
    uint32_t gw = 0;
    if (!TRANSPOSE && on) gw = gtw[g];
    uint32_t id[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
      const bool want = i < nk;
      for (uint32_t guard = 0; guard < 2 * kWpc && __ballot(want && cur == 0); ++guard) {   // a window ran dry in some lane: the next dword of the row
        if (want && cur == 0) { ++wi; cur = s_rowd[wi < 131 ? wi : 131]; base += 32; }
      }
      const uint32_t b = __builtin_ctz(cur | 0x80000000u);
      id[i] = want ? base + b : 0u;
      if (want) cur &= cur - 1;
    }
    uint4 v{id[0] | (id[1] << 16), id[2] | (id[3] << 16), id[4] | (id[5] << 16), id[6] | (id[7] << 16)};
    if (TRANSPOSE) { if (on) s_groups[g] = v; }
    else if (on) store_group_nt(dst + g, merge_group16(v, gw, m_both));
  }
  if (TRANSPOSE) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    for (uint32_t g = lane; g < G; g += 64)                        // lane per group, 1 KiB-contiguous stores
      store_group_nt(dst + g, merge_group16(s_groups[g], gtw[g], m_both));
  }
}

// ---- bit per lane: one pass per 64-bit row word with exec = the word; a lane whose bit is set ranks itself with v_mbcnt and drops its
//      16-bit id into an LDS list at its carrier index; lane per group copy-out as in the production code (fixed 40 passes whatever the
//      bits: no data-dependent loop, no idle lanes in a peel -- and one ds_write_b16 wave-instruction per row word) ----
__device__ __forceinline__ void expand_mbcnt(const DevImage& im, uint4* __restrict__ arena_groups, uint32_t* lds, uint32_t lane, uint64_t mine,
                                             uint32_t cnt, uint64_t gt0, uint64_t cb, uint32_t m_both) {
  uint16_t* ids16 = reinterpret_cast<uint16_t*>(lds);
  uint32_t at = 0;
#pragma unroll 4
  for (uint32_t w = 0; w < kWpc; ++w) {
    const uint64_t word = wave_bcast64(mine, (int)w);
    if (__builtin_amdgcn_inverse_ballot_w64(word)) {
      const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(word >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)word, at));
      ids16[pos] = (uint16_t)(w * 64 + lane);
    }
    at += (uint32_t)__popcll(word);
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  const uint32_t G = (cnt + 7) >> 3;
  const uint32_t* __restrict__ gtw = im.gt_groups + (gt0 >> 3);
  uint4* __restrict__ dst = arena_groups + (cb >> 3);
  for (uint32_t g = lane; g < G; g += 64)
    store_group_nt(dst + g, merge_group16(*reinterpret_cast<const uint4*>(ids16 + 8 * g), gtw[g], m_both));
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

template <int ALG, uint32_t K>
__global__ void __launch_bounds__(256) k_dense(DevImage im, void* arena, const uint32_t* __restrict__ v_cnt, const uint64_t* __restrict__ v_gt0,
                                               const uint64_t* __restrict__ v_cb, uint32_t nvar, uint32_t lds_words_per_wave) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  uint32_t* lds_wave = &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave];
  const uint64_t first = wave * K;
  if (first >= nvar) return;
  if (ALG == 0) {
    uint32_t cnt = 0, cls = 0;
    uint64_t gt0 = 0, cb = 0;
    if (lane < K && first + lane < nvar) { cnt = v_cnt[first + lane]; cls = (uint32_t)(first + lane); gt0 = v_gt0[first + lane]; cb = v_cb[first + lane]; }
    expand_task<false, true, false, false, true>(im, arena, lds_wave, lane, cnt, cls, gt0, cb, 0u, slice_gt_words(im.num_samples));
  } else {
    uint32_t m_both = 0xE000E000u;
    asm volatile("" : "+s"(m_both));
    const uint32_t n = nvar - first < K ? (uint32_t)(nvar - first) : K;
    uint64_t next = lane < kWpc ? im.class_rows[first * kWpc + lane] : 0ull;
    for (uint32_t t = 0; t < n; ++t) {
      uint64_t mine = next;
      if (t + 1 < n) next = lane < kWpc ? im.class_rows[(first + t + 1) * kWpc + lane] : 0ull;   // the next row is on its way while this one is expanded
      if (lane == 0) mine &= ~1ull;
      const uint32_t cnt = v_cnt[first + t];
      if (ALG == 3) expand_mbcnt(im, reinterpret_cast<uint4*>(arena), lds_wave, lane, mine, cnt, v_gt0[first + t], v_cb[first + t], m_both);
      else expand_equal_shares<ALG == 2>(im, reinterpret_cast<uint4*>(arena), lds_wave, lane, mine, cnt, v_gt0[first + t], v_cb[first + t], m_both);
    }
  }
}

int main(int argc, char** argv) {
  const uint32_t nvar = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000u;
  constexpr uint32_t K = 16;
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::vector<uint64_t> rows((size_t)nvar * kWpc, 0);
  std::vector<uint32_t> cnt(nvar);
  std::vector<uint64_t> gt0(nvar), cb(nvar);
  uint64_t pool = 0;
  for (uint32_t v = 0; v < nvar; ++v) {
    uint32_t c = 0;
    do {
      const double af = std::pow(10.0, std::log10(0.137) * U(rng));   // log-uniform over [0.137, 1]
      const double d = 1.0 - (1.0 - af) * (1.0 - af);
      c = 0;
      for (uint32_t w = 0; w < kWpc; ++w) rows[(size_t)v * kWpc + w] = 0;
      for (uint32_t s = 1; s <= kSamples - 1; ++s)
        if (U(rng) < d) { rows[(size_t)v * kWpc + (s >> 6)] |= 1ull << (s & 63); ++c; }
    } while (c <= 640);
    cnt[v] = c; gt0[v] = pool; cb[v] = pool;
    pool += (c + 7) & ~7u;
  }
  std::vector<uint32_t> gtw(pool / 8 + 16);
  for (auto& x : gtw) x = (uint32_t)rng();
  // host expansion
  std::vector<uint16_t> want(pool, 0);
  for (uint32_t v = 0; v < nvar; ++v) {
    uint64_t k = 0;
    for (uint32_t s = 1; s < kSamples; ++s)
      if ((rows[(size_t)v * kWpc + (s >> 6)] >> (s & 63)) & 1) {
        const uint32_t W = gtw[(gt0[v] + k) >> 3], i = (uint32_t)(k & 7);
        const uint32_t g3 = (i & 1) ? (W >> (16 + 3 * (i >> 1))) & 7u : (W >> (3 * (i >> 1))) & 7u;
        want[cb[v] + k] = (uint16_t)(s | (g3 << 13));
        ++k;
      }
  }
  DevImage im{};
  im.num_samples = kSamples; im.wpc = kWpc; im.use_bv = 1; im.list_max = 640;
  uint64_t* d_rows; uint32_t* d_gt; uint32_t* d_cnt; uint64_t *d_gt0, *d_cb; uint16_t* d_arena;
  CHECK(hipMalloc(&d_rows, rows.size() * 8 + 4096)); CHECK(hipMalloc(&d_gt, gtw.size() * 4)); CHECK(hipMalloc(&d_cnt, nvar * 4));
  CHECK(hipMalloc(&d_gt0, nvar * 8)); CHECK(hipMalloc(&d_cb, nvar * 8)); CHECK(hipMalloc(&d_arena, pool * 2 + 4096));
  CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_gt, gtw.data(), gtw.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_cnt, cnt.data(), nvar * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_gt0, gt0.data(), nvar * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_cb, cb.data(), nvar * 8, hipMemcpyHostToDevice));
  im.class_rows = d_rows; im.gt_groups = d_gt;
  const double bytes = (double)nvar * kWpc * 8 + (double)pool / 8 * 4 + (double)pool * 2;
  printf("%u dense variants of %u samples, %.1f carriers each on average, %.3f GB per launch (rows + genotype words + arena), %u variants per wave\n",
         nvar, kSamples, (double)pool / nvar, bytes / 1e9, K);
  const unsigned blocks = (unsigned)(((nvar + K - 1) / K + 3) / 4);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<uint16_t> got(pool);
  const char* names[4] = {"slices (production expand_task)", "equal output shares, stores from registers", "equal output shares, groups transposed through LDS",
                          "bit per lane: exec = row word, v_mbcnt ranks, LDS id list"};
  const int only = argc > 2 ? atoi(argv[2]) : -1;
  for (int alg = 0; alg < 4; ++alg) {
    if (only >= 0 && alg != only) continue;
    const uint32_t lds_words = alg == 0 ? slice_lds_words(kSamples) : (alg == 1 ? 64 + 136 : (alg == 2 ? 64 + 136 + 4 * 320 : 4 * 320));
    const size_t lds_bytes = (size_t)lds_words * 4 * 4;
    CHECK(hipMemset(d_arena, 0, pool * 2));
    float best = 1e9f, sum = 0;
    const int reps = 7;
    for (int it = 0; it < reps + 1; ++it) {
      CHECK(hipEventRecord(e0));
      if (alg == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense<0, K>), dim3(blocks), dim3(256), lds_bytes, 0, im, (void*)d_arena, d_cnt, d_gt0, d_cb, nvar, lds_words);
      else if (alg == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense<1, K>), dim3(blocks), dim3(256), lds_bytes, 0, im, (void*)d_arena, d_cnt, d_gt0, d_cb, nvar, lds_words);
      else if (alg == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense<2, K>), dim3(blocks), dim3(256), lds_bytes, 0, im, (void*)d_arena, d_cnt, d_gt0, d_cb, nvar, lds_words);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense<3, K>), dim3(blocks), dim3(256), lds_bytes, 0, im, (void*)d_arena, d_cnt, d_gt0, d_cb, nvar, lds_words);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (it) { sum += ms; best = ms < best ? ms : best; }
    }
    CHECK(hipMemcpy(got.data(), d_arena, pool * 2, hipMemcpyDeviceToHost));
    uint64_t bad = 0;
    for (uint32_t v = 0; v < nvar; ++v)
      for (uint32_t k = 0; k < cnt[v]; ++k) bad += got[cb[v] + k] != want[cb[v] + k];
    printf("%-52s %.4f ms (best %.4f)  %.0f GB/s  %.1f ns per variant per CU  LDS %zu B per wave  %s\n", names[alg], sum / reps, best, bytes / (sum / reps * 1e-3) / 1e9,
           sum / reps * 1e6 / ((double)nvar / 256.0), lds_bytes / 4, bad ? "MISMATCH" : "ok");
    if (bad) printf("   %llu carrier words differ from the host expansion\n", (unsigned long long)bad);
  }
  return 0;
}
