// k_expand.hip.h -- carrier expansion (expand_task, k_fill_carriers, k_fill_sites) -- the dominant kernel.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_rows.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Carrier expansion (expand_task: k_fill_carriers, k_query_small, k_query_server).  One wave owns CH consecutive
// variant slots (64; 4 in latency launches); every lane first gathers the parameters of "its" slot, then the wave
// works through the task:
//
//  cohorts of at most 4032 samples with class rows (WIDE=false, use_bv) -- the main case:
//    listed  (<= list_max = 640 carriers): LANE PER GROUP of 8 carriers from the class's decoded 16-bit id list; the
//            groups of all listed variants of the task form one list, a lane finds its variant by bisection over the
//            64 offsets (LDS) and produces one finished 16-byte arena group.
//    denser  WAVE PER VARIANT, two rounds of half a row: every lane peels its own ceil(wpc / 2) bits into a 16-bit id
//            list in LDS at its prefix-sum position; the complete groups leave in 128-byte-aligned blocks, one
//            16-byte store per lane (8 carriers of 16 bits: id | gt << 13), genotypes merged from the raw nibble
//            stream on the way out.  Rows are requested two variants ahead, nibbles one.
//  explicit-id cohorts: LANE PER GROUP for every variant (ids from the carrier pool).
//  cohorts above 4032 samples with class rows (WIDE=true): the list path with 32-bit entries and 32-bit carrier words
//    (id | gt << 29); denser variants keep the round-1 row code -- medium ones lane per row word with an LDS id
//    list, dense ones bit per lane (exec = row word, v_mbcnt rank) through a 512-entry LDS ring; rows wider than one
//    wave take the out-of-line generic path.
// ---------------------------------------------------------------------------
constexpr uint32_t kFillChunk = 64;          // variant slots per wave task, throughput launches
constexpr uint32_t kFillChunkDense = 16;     // throughput launches over few, carrier-heavy variants (type-4 batches)
constexpr uint32_t kFillChunkSmall = 4;      // latency launches (a handful of regions): more waves per region
constexpr uint32_t kRingWords = 512;             // per wave: output ring of the dense path (flushed 1 KiB at a time)
// per-wave LDS = gt_words (one genotype byte per carrier, sized from the cohort) + kRingWords, passed at launch
constexpr uint32_t kMidMax = 640;            // <= this many carriers: ids are staged in LDS and copied out coalesced
// Slice path (cohorts of at most 4032 samples): per-wave LDS = row staging + raw genotype nibbles + 16-bit id list
constexpr uint32_t kRowWords = 132;          // 65 x uint64 (the row and one zero word behind it), padded
// layout: [raw nibbles][id list; the row staging aliases its start -- the slices are cut before the list is written]
__host__ __device__ inline uint32_t slice_gt_words(uint32_t n_samples) {
  uint32_t b = 16 + (n_samples + 32) / 2;   // the bias, then the nibbles of one variant starting anywhere in a 16-byte group
  b = (b + 15) & ~15u;
  if (b < 1024 + 16) b = 1024 + 16;         // the first 1 KiB is written by all lanes
  return b / 4;
}
constexpr uint32_t kListWindow = 64;         // the id list is laid out by arena position modulo 64 entries (128 bytes)
__host__ __device__ inline uint32_t slice_ids_words(uint32_t n_samples) {
  // one round of the slice path: half a row (64 lanes x ceil(wpc / 2) bits) behind the alignment window, plus the
  // incomplete group carried over from the first round
  const uint32_t wpc = (n_samples + 63) / 64, round_bits = 64 * ((wpc + 1) / 2);
  const uint32_t w = ((kListWindow + round_bits + 16 + 7) & ~7u) / 2;
  return w < kRowWords ? kRowWords : w;
}
__host__ __device__ inline uint32_t slice_lds_words(uint32_t n_samples) {
  const uint32_t w = slice_gt_words(n_samples) + slice_ids_words(n_samples);
  return w < 384 ? 384 : w;                  // the sparse phase keeps 6 x 64 words at the start of the region
}
constexpr uint32_t kMidIdsAt = 256;          // medium path: ids live at word 256.. (genotype bytes need < 1 KiB there)

// one 16-byte arena group, written once and not read again by this kernel
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_group_nt(uint4* p, uint4 v) {
  __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(p));
}

// (Round 5, tried and removed: non-temporal LOADS for the read-once streams -- genotype nibbles, site rows and their parameters -- so that
//  they would not push the shared class rows and decoded lists out of the L2 / MALL: 0.559 -> 0.569 ms.)
__device__ __forceinline__ uint4 ld_stream16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

__device__ __forceinline__ uint4 load_u128_unaligned(const uint32_t* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

__device__ __forceinline__ uint64_t wave_bcast64(uint64_t v, int src_lane) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src_lane);
  const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src_lane);
  return ((uint64_t)hi << 32) | lo;
}

// 32 packed genotype nibbles (one uint4) -> 32 bytes in LDS, nibble order preserved.
__device__ __forceinline__ void stage_unpacked(uint8_t* dst, uint4 n) {
  uint32_t in[4] = {n.x, n.y, n.z, n.w};
  uint32_t o[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t a = in[i] & 0x07070707u, b = (in[i] >> 4) & 0x07070707u;
    o[2 * i] = __builtin_amdgcn_perm(b, a, 0x05010400u);      // a0 b0 a1 b1
    o[2 * i + 1] = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
  }
  reinterpret_cast<uint4*>(dst)[0] = uint4{o[0], o[1], o[2], o[3]};
  reinterpret_cast<uint4*>(dst)[1] = uint4{o[4], o[5], o[6], o[7]};
}

// Generic (slow) expansion of one variant by a whole wave: any row width, genotype
// nibbles read straight from global memory.  Kept out of line so that its loads do
// not force memory waits into the tuned loops of k_fill_carriers.
__device__ __noinline__ void expand_generic(const uint64_t* row, uint32_t wpc, const uint8_t* gtp, uint64_t gt0,
                                            uint32_t* out, uint32_t lane) {
  uint32_t base = 0;
  for (uint32_t wb = 0; wb < wpc; wb += 64) {
    uint64_t mine = (wb + lane < wpc) ? row[wb + lane] : 0ULL;
    if (wb == 0 && lane == 0) mine &= ~1ULL;
    uint64_t nz = __ballot(mine != 0);
    while (nz) {
      const int w = __builtin_ctzll(nz);
      nz &= nz - 1;
      const uint64_t word = wave_bcast64(mine, w);
      if ((word >> lane) & 1) {
        const uint32_t k = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(word >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)word, 0));
        const uint64_t c = gt0 + k;
        const uint32_t nib = (gtp[c >> 1] >> ((c & 1) * 4)) & 7u;
        out[k] = ((wb + w) * 64 + lane) | (nib << 29);
      }
      base += __popcll(word);
    }
  }
}

// One finished 16-byte arena group of 16-bit carrier words from four id pairs and the group's genotype word (DevImage::gt_groups:
// genotype k at bit 3 (k / 2) + 16 (k & 1)): W << (13 - 3 j) puts pair j's two genotypes at bits 13-15 and 29-31.
__device__ __forceinline__ uint4 merge_group16(uint4 ids, uint32_t W, uint32_t m_both) {
  uint4 v;
  v.x = and_or(W << 13, m_both, ids.x);
  v.y = and_or(W << 10, m_both, ids.y);
  v.z = and_or(W << 7, m_both, ids.z);
  v.w = and_or(W << 4, m_both, ids.w);
  return v;
}
// The genotypes of one dense variant, requested as up to two 16-byte loads per lane: raw nibbles from the 16-byte-aligned byte
// below its first carrier (WIDE), or group words from the 4-word-aligned word below its first group (a vertex's carriers start
// on a multiple of 8 pool records: gt0 % 8 == 0).
template <bool WIDE>
__device__ __forceinline__ void request_genotypes(const DevImage& im, uint64_t gt0_v, uint32_t cnt_v, uint32_t lane, uint4& q0, uint4& q1) {
  if constexpr (WIDE) {
    const uint8_t* __restrict__ gtp = im.gt_nibbles;
    const uint64_t b0 = (gt0_v >> 1) & ~15ULL;                        // aligned byte base
    const uint64_t need = ((gt0_v + cnt_v + 1) >> 1) - b0;            // bytes that hold this variant's nibbles
    if ((uint64_t)lane * 16 < need) q0 = *reinterpret_cast<const uint4*>(gtp + b0 + lane * 16);
    if ((uint64_t)lane * 16 + 1024 < need) q1 = *reinterpret_cast<const uint4*>(gtp + b0 + 1024 + lane * 16);
  } else {
    const uint64_t w0 = gt0_v >> 3, wb = w0 & ~3ULL;
    const uint32_t need = (uint32_t)(w0 - wb) + (cnt_v + 7) / 8;      // words from the aligned base on
    if (lane * 4 < need) q0 = *reinterpret_cast<const uint4*>(im.gt_groups + wb + lane * 4);
    if (lane * 4 + 256 < need) q1 = *reinterpret_cast<const uint4*>(im.gt_groups + wb + 256 + lane * 4);
  }
}

// Expansion of one task: the lanes hold (cnt, cls, gt0, cb) of up to 64 variant slots (cnt == 0: nothing to do for the
// lane) and the wave writes their carrier words into the arena.  Shared by k_fill_carriers (slots whose headers an
// earlier kernel wrote) and k_query_small (single-launch latency path, slots read straight from the site table).
// `lds_wave` is the wave's LDS block (slice_lds_words / gt_words + kRingWords words).
// `ablate_arg` is a profiling aid of tuning builds (TUNE: bit0 skip listed/sparse, bit1 skip medium, bit2 skip dense).
// WIDE=false is instantiated for cohorts of at most 4032 samples (<= 63 row words): every variant then
// fits the staged paths and the out-of-line generic call -- whose calling convention costs registers and
// one wave of occupancy -- is compiled out.
// EARLY_NIB: request the first dense variant's genotype nibbles before the list phase too (latency launches: one task
// per wave and nothing to overlap with; throughput launches request them afterwards to stay within 64 registers).
// LISTED / DENSE: instantiations that take only one of the two regimes (the split form: k_fill_sites2 without the dense
// variants, k_fill_dense with nothing else) compile the other one out.
template <bool WIDE, bool EARLY_NIB, bool TUNE = false, bool LISTED = true, bool DENSE = true>
__device__ __forceinline__ void expand_task(const DevImage& im, void* arena, uint32_t* lds_wave, uint32_t lane, uint32_t cnt, uint32_t cls,
                                            uint64_t gt0, uint64_t cb, uint32_t ablate_arg, uint32_t gt_words, unsigned long long* tstat = nullptr) {
  const uint32_t ablate = TUNE ? ablate_arg : 0u;   // production instantiations carry no ablation tests
  // tuning builds (option fill_stats): device-clock ticks (10 ns) per phase of the task, summed over the waves
  const uint64_t t_enter = (TUNE && tstat) ? wall_clock64() : 0;
  const uint32_t wpc = im.wpc;
  const uint64_t* __restrict__ class_rows = im.class_rows;
  const uint8_t* __restrict__ gtp = im.gt_nibbles;
  // carrier word in the arena: 16 bits when every sample id fits 13 bits (the non-WIDE instantiation), else 32
  using CT = typename std::conditional<WIDE, uint32_t, uint16_t>::type;
  CT* __restrict__ carriers = reinterpret_cast<CT*>(arena);
  uint32_t m_both = 0xE000E000u;   // genotype fields of the two 16-bit carrier words in a dword (in an SGPR: VOP3 takes no literals on gfx9)
  asm volatile("" : "+s"(m_both));
  const bool explicit_ids = !im.use_bv;   // sparse cohorts: sample ids stored per carrier instead of class rows
  if (explicit_ids) {
    // Explicit-id cohorts (somatic-like: a handful of carriers per variant, ids in the carrier pool): LANE PER GROUP of
    // 8 carriers for every variant whatever its size, exactly like the list path below -- the groups of the task form
    // one list, a lane finds its variant by bisection, loads the 8 ids (32 bytes of car_sid) and their 32 genotype bits
    // and stores one finished group (16 bytes of 16-bit words, or 32 bytes of 32-bit words above 4032 samples).  The
    // last group of a variant reads up to 7 ids of the next one: they land in the padding the range owns.
    uint32_t* s_off = lds_wave;
    const uint32_t c = cnt && !(ablate & 1) ? (cnt + kCarAlign - 1) / kCarAlign : 0u;
    const uint32_t incl = wave_inclusive_scan(c);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    if (total) {
      uint64_t* s_gt0 = reinterpret_cast<uint64_t*>(s_off + 128);
      uint64_t* s_cb = reinterpret_cast<uint64_t*>(s_off + 256);
      s_off[lane] = incl - c;
      s_gt0[lane] = gt0;
      s_cb[lane] = cb;
      const uint32_t* __restrict__ gt32 = reinterpret_cast<const uint32_t*>(gtp);
      for (uint32_t e = lane; e < total; e += 64) {
        uint32_t L = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1)
          if (s_off[L + step] <= e) L += step;
        const uint32_t k8 = (e - s_off[L]) * kCarAlign;
        const uint64_t g = s_gt0[L] + k8;                        // carrier record of the group's first entry
        uint4 ia, ib;
        __builtin_memcpy(&ia, im.car_sid + g, 16);
        __builtin_memcpy(&ib, im.car_sid + g + 4, 16);
        uint2 nw;                                                 // (the explicit-id pool is not padded: an unaligned window of the nibble stream)
        __builtin_memcpy(&nw, gt32 + (g >> 3), 8);
        const uint32_t n = __builtin_amdgcn_alignbit(nw.y, nw.x, ((uint32_t)g & 7u) * 4);
        const uint32_t id[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
        CT* dst = carriers + (s_cb[L] + k8);
        if constexpr (WIDE) {
          uint4 lo, hi;
          lo.x = id[0] | (((n >> 0) & 7u) << 29); lo.y = id[1] | (((n >> 4) & 7u) << 29);
          lo.z = id[2] | (((n >> 8) & 7u) << 29); lo.w = id[3] | (((n >> 12) & 7u) << 29);
          hi.x = id[4] | (((n >> 16) & 7u) << 29); hi.y = id[5] | (((n >> 20) & 7u) << 29);
          hi.z = id[6] | (((n >> 24) & 7u) << 29); hi.w = id[7] | (((n >> 28) & 7u) << 29);
          store_group_nt(reinterpret_cast<uint4*>(dst), lo);
          store_group_nt(reinterpret_cast<uint4*>(dst) + 1, hi);
        } else {
          // (every word of car_sid is a valid sample id < 4032 or zero padding: 13 bits, nothing to mask)
          uint32_t m_lo = 0xE000u, m_hi = 0xE0000000u;
          asm volatile("" : "+s"(m_lo), "+s"(m_hi));
          uint4 v;
          const uint32_t p0 = id[0] | (id[1] << 16), p1 = id[2] | (id[3] << 16), p2 = id[4] | (id[5] << 16), p3 = id[6] | (id[7] << 16);
          v.x = and_or(n << 25, m_hi, and_or(n << 13, m_lo, p0));
          v.y = and_or(n << 17, m_hi, and_or(n << 5, m_lo, p1));
          v.z = and_or(n << 9, m_hi, and_or(n >> 3, m_lo, p2));
          v.w = and_or(n << 1, m_hi, and_or(n >> 11, m_lo, p3));
          store_group_nt(reinterpret_cast<uint4*>(dst), v);
        }
      }
    }
    return;
  }

  const bool lists = true;   // (explicit-id cohorts returned above) every class of at most list_max carriers has a decoded id list
  const uint32_t list_max = im.list_max;

  // ---------------- denser variants (wave per variant, below): their first loads are requested NOW, so that the
  //                  list phase runs in the shadow of that memory latency ----------------
  uint64_t dmask = DENSE ? __ballot(cnt > list_max && !explicit_ids) : 0ull;
  uint64_t word_cur = 0, word_n1 = 0, word_n2 = 0;   // bit rows of the current dense variant and of the next two
  uint4 nq0 = {0, 0, 0, 0}, nq1 = {0, 0, 0, 0};      // raw genotype nibbles of the current one (then of the next)
  if (dmask) {
    const int t0 = __builtin_ctzll(dmask);
    const uint32_t cls_0 = __builtin_amdgcn_readlane(cls, t0), cnt_0 = __builtin_amdgcn_readlane(cnt, t0);
    const uint64_t gt0_0 = wave_bcast64(gt0, t0);
    if (lane < wpc) word_cur = class_rows[(uint64_t)cls_0 * wpc + lane];
    if (!lists || EARLY_NIB) request_genotypes<WIDE>(im, gt0_0, cnt_0, lane, nq0, nq1);
    const uint64_t d1 = dmask & (dmask - 1);
    if (d1) {
      const uint32_t cls_1 = __builtin_amdgcn_readlane(cls, __builtin_ctzll(d1));
      if (lane < wpc) word_n1 = class_rows[(uint64_t)cls_1 * wpc + lane];
    }
  }

  // Cohorts of at most 4032 samples with class rows: every variant of at most list_max carriers is expanded from its
  // class's decoded 16-bit id list, LANE PER GROUP of 8 carriers (= one 16-byte arena group; every variant's arena
  // range and every list start on a group boundary and own their padding).  The groups of all such variants of the
  // task form one list: a DPP prefix sum over the group counts gives every variant its slice, a lane takes entry e,
  // finds its variant by bisection over the 64 offsets (LDS), loads the 8 ids (one 16-byte load) and the 32
  // genotype bits that go with them (one 8-byte load of the nibble pool), and stores one finished 16-byte group.
  // No bit row is read, nothing is staged, no lane idles: a rare variant is one group, a 640-carrier one is 80.
  // Two entries per lane and pass, so that four independent loads are in flight per lane.
  if constexpr (WIDE) {
    // the same with 32-bit list entries and 32-bit carrier words (id | gt << 29): a group is two loads and two stores
    uint32_t* s_off = lds_wave;
    const bool sp = LISTED && cnt > 0 && cnt <= list_max && !(ablate & 1);
    const uint32_t c = sp ? (cnt + kCarAlign - 1) / kCarAlign : 0u;
    const uint32_t incl = LISTED ? wave_inclusive_scan(c) : 0u;
    const uint32_t total = LISTED ? __builtin_amdgcn_readlane(incl, 63) : 0u;
    if (total) {
      uint32_t* s_idb = s_off + 64;
      uint64_t* s_gt0 = reinterpret_cast<uint64_t*>(s_off + 128);
      uint64_t* s_cb = reinterpret_cast<uint64_t*>(s_off + 256);
      s_off[lane] = incl - c;
      s_idb[lane] = cls;      // listed variants: group index of the class's list (DevImage::v_src)
      s_gt0[lane] = gt0;
      s_cb[lane] = cb;
      const uint4* __restrict__ list_groups = reinterpret_cast<const uint4*>(im.cls_list_ids);
      const uint32_t* __restrict__ gt32 = reinterpret_cast<const uint32_t*>(gtp);
      for (uint32_t e = lane; e < total; e += 64) {
        uint32_t L = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1)
          if (s_off[L + step] <= e) L += step;
        const uint32_t k = e - s_off[L];
        const uint64_t g = s_gt0[L] + (uint64_t)k * kCarAlign;
        const uint4 ia = list_groups[2 * ((uint64_t)s_idb[L] + k)], ib = list_groups[2 * ((uint64_t)s_idb[L] + k) + 1];
        const uint32_t n = gt32[g >> 3];   // (g is a multiple of 8: eight nibbles, one aligned word)
        uint4 lo, hi;
        lo.x = ia.x | (((n >> 0) & 7u) << 29); lo.y = ia.y | (((n >> 4) & 7u) << 29);
        lo.z = ia.z | (((n >> 8) & 7u) << 29); lo.w = ia.w | (((n >> 12) & 7u) << 29);
        hi.x = ib.x | (((n >> 16) & 7u) << 29); hi.y = ib.y | (((n >> 20) & 7u) << 29);
        hi.z = ib.z | (((n >> 24) & 7u) << 29); hi.w = ib.w | (((n >> 28) & 7u) << 29);
        uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(arena) + (s_cb[L] + (uint64_t)k * kCarAlign));
        store_group_nt(dst, lo);
        store_group_nt(dst + 1, hi);
      }
    }
  } else {
    uint32_t* s_off = lds_wave;   // aliases the genotype staging area
    const bool sp = LISTED && cnt > 0 && cnt <= list_max && !(ablate & 1);
    const uint32_t c = sp ? (cnt + kCarAlign - 1) / kCarAlign : 0u;
    const uint32_t incl = LISTED ? wave_inclusive_scan(c) : 0u;
    const uint32_t total = LISTED ? __builtin_amdgcn_readlane(incl, 63) : 0u;
    if (total) {
      uint32_t* s_idb = s_off + 64;
      uint64_t* s_gt0 = reinterpret_cast<uint64_t*>(s_off + 128);
      uint64_t* s_cb = reinterpret_cast<uint64_t*>(s_off + 256);
      s_off[lane] = incl - c;
      s_idb[lane] = cls;      // listed variants: group index of the class's list (DevImage::v_src)
      s_gt0[lane] = gt0;
      s_cb[lane] = cb;
      const uint4* __restrict__ list_groups = reinterpret_cast<const uint4*>(im.cls_list16);
      const uint32_t* __restrict__ gt_groups = im.gt_groups;
      uint4* __restrict__ arena_groups = reinterpret_cast<uint4*>(arena);
      for (uint32_t e0 = lane; e0 < total; e0 += 128) {
        const uint32_t e1 = e0 + 64;
        const bool two = e1 < total;
        uint32_t L0 = 0, L1 = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) {
          if (s_off[L0 + step] <= e0) L0 += step;
          if (s_off[L1 + step] <= e1) L1 += step;
        }
        const uint32_t k0 = e0 - s_off[L0], k1 = e1 - s_off[L1];   // group within its variant
        const uint64_t w0 = (s_gt0[L0] >> 3) + k0, w1 = (s_gt0[L1] >> 3) + k1;   // the groups' genotype words (a vertex's records start on a multiple of 8)
        const uint4 iw0 = list_groups[(uint64_t)s_idb[L0] + k0];
        const uint32_t gw0 = gt_groups[w0];
        uint4 iw1 = {0, 0, 0, 0};
        uint32_t gw1 = 0;
        if (two) {
          iw1 = list_groups[(uint64_t)s_idb[L1] + k1];
          gw1 = gt_groups[w1];
        }
        const uint64_t dst0 = (s_cb[L0] >> 3) + k0, dst1 = (s_cb[L1] >> 3) + k1;   // arena ranges start on group boundaries
        store_group_nt(&arena_groups[dst0], merge_group16(iw0, gw0, m_both));
        if (two) store_group_nt(&arena_groups[dst1], merge_group16(iw1, gw1, m_both));
      }
    }
  }

  // The parameter and list phases are the kernel's memory-bound part, the dense phase its VALU- and LDS-bound part (round 5:
  // profiles/r05_exp_split_pmc.json).  Callers that mix the two (k_fill_sites2) run the former at priority 3 so that a wave's
  // loads are issued ahead of other waves' peel loops; from here on the wave is an ordinary one again (0.536 -> 0.522 ms on a
  // fast box, no difference on a slow one: profiles/r05_exp_priorities_box_*.txt; the opposite order costs 1.5 %).
  __builtin_amdgcn_s_setprio(0);
  if (TUNE && tstat) {   // (tstat: three words of the CALLER's registers -- list phase, dense phase, dense variants)
    __builtin_amdgcn_s_waitcnt(0);   // (the list phase's loads have returned; its stores are on their way)
    tstat[0] = wall_clock64() - t_enter; tstat[2] = (unsigned long long)__popcll(dmask);
  }
  const uint64_t t_lists = (TUNE && tstat) ? wall_clock64() : 0;
  if (dmask == 0) return;
  if (lists && !EARLY_NIB) {   // the first dense variant's nibbles (8 registers) are requested after the list phase: its peak register
                 // demand decides how many waves a SIMD holds
    const int t0 = __builtin_ctzll(dmask);
    const uint32_t cnt_0 = __builtin_amdgcn_readlane(cnt, t0);
    const uint64_t gt0_0 = wave_bcast64(gt0, t0);
    request_genotypes<WIDE>(im, gt0_0, cnt_0, lane, nq0, nq1);
  }
  // Per-wave LDS block: the genotype staging area (raw nibbles; cohorts above 4032 samples: one byte per carrier),
  // the id list of the slice path (the medium path of wide cohorts keeps its ids at word 256..) and, for wide
  // cohorts, the output ring.
  uint8_t* gt_lds = reinterpret_cast<uint8_t*>(lds_wave);
  uint32_t* ids_lds = lds_wave + kMidIdsAt;
  uint32_t* ring = lds_wave + gt_words;
  while (dmask) {
    const int t = __builtin_ctzll(dmask);
    dmask &= dmask - 1;
    const uint32_t cnt_t = __builtin_amdgcn_readlane(cnt, t);
    const uint32_t cls_t = __builtin_amdgcn_readlane(cls, t);
    const uint64_t gt0_t = wave_bcast64(gt0, t);
    const uint64_t cb_t = wave_bcast64(cb, t);
    const uint64_t b0 = (gt0_t >> 1) & ~15ULL;
    const uint32_t nshift = (uint32_t)(gt0_t - 2 * b0);               // staged index of carrier 0
    const bool staged = (uint64_t)nshift + cnt_t <= gt_words * 4;     // fits the staging block
    // stage this variant's genotypes (fetched during the previous variant)
    if (WIDE) {   // one byte per carrier
      stage_unpacked(gt_lds + lane * 32, nq0);
      if (nshift + cnt_t > 2048) stage_unpacked(gt_lds + 2048 + lane * 32, nq1);
    } else {      // the genotype words of the variant's groups from the 4-word-aligned word below the first one
      *reinterpret_cast<uint4*>(gt_lds + lane * 16) = nq0;
      if (lane * 4 + 256 < ((uint32_t)(gt0_t >> 3) & 3u) + (cnt_t + 7) / 8) *reinterpret_cast<uint4*>(gt_lds + 1024 + lane * 16) = nq1;
    }
    // request the next variant's nibbles and the row of the one after it before expanding this one (rows are the
    // random 320-byte reads of this kernel: two of them stay in flight per wave)
    if (WIDE) { nq0 = uint4{0, 0, 0, 0}; nq1 = uint4{0, 0, 0, 0}; }   // (the slice path never reads nibbles it did not load)
    word_n2 = 0;
    if (dmask) {
      const int tn = __builtin_ctzll(dmask);
      const uint64_t gt0_n = wave_bcast64(gt0, tn);
      const uint32_t cnt_n = __builtin_amdgcn_readlane(cnt, tn);
      request_genotypes<WIDE>(im, gt0_n, cnt_n, lane, nq0, nq1);
      const uint64_t d2 = dmask & (dmask - 1);
      if (d2) {
        const uint32_t cls_2 = __builtin_amdgcn_readlane(cls, __builtin_ctzll(d2));
        if (lane < wpc) word_n2 = class_rows[(uint64_t)cls_2 * wpc + lane];
      }
    }
    const uint64_t word_this = word_cur;
    word_cur = word_n1; word_n1 = word_n2;   // the queue advances here: every `continue` below leaves it consistent
    if constexpr (WIDE) {
      if (!staged || wpc > 64) {
        // rows wider than one wave or more than 4096 staged genotypes: generic path
        expand_generic(class_rows + (uint64_t)cls_t * wpc, wpc, gtp, gt0_t, carriers + cb_t, lane);
        continue;
      }
    }
    uint64_t mine = word_this;
    if (lane == 0) mine &= ~1ULL;  // bit 0 is "ref"
    if ((ablate & 2) && cnt_t <= kMidMax) continue;
    if ((ablate & 4) && cnt_t > kMidMax) continue;
    if constexpr (!WIDE) {
      // ---- slice path: the row is expanded in TWO rounds of 64 x sb bits (sb = ceil(wpc / 2) <= 32): in a round
      //      every lane owns sb consecutive bits, peels them into a 16-bit id list in LDS at its prefix-sum position,
      //      then the complete 16-byte groups of the list leave in 128-byte-aligned blocks, one store per lane,
      //      genotypes merged from the raw nibble stream on the way out; the (< 8) ids of the last, incomplete group
      //      move to the front of the list and the second round continues behind them.  The list therefore holds
      //      half a row at most -- the per-wave LDS block is what limits this kernel's occupancy. ----
      const uint32_t* gw_lds = reinterpret_cast<const uint32_t*>(gt_lds);
      uint16_t* ids16 = reinterpret_cast<uint16_t*>(gt_lds + slice_gt_words(im.num_samples) * 4);
      uint64_t* rowq = reinterpret_cast<uint64_t*>(ids16);                      // [65], dead before the list is written
      rowq[lane] = mine;
      if (lane == 0) rowq[64] = 0;
      const uint32_t sb = (wpc + 1) >> 1;                                       // bits per lane and round
      const uint32_t smask = sb >= 32 ? 0xFFFFFFFFu : (1u << sb) - 1u;
      const uint32_t* rowd = reinterpret_cast<const uint32_t*>(rowq);
      const uint32_t bp0 = sb * lane, bp1 = bp0 + 64 * sb;                      // first bit of the lane's slice per round
      uint32_t bits0 = __builtin_amdgcn_alignbit(rowd[(bp0 >> 5) + 1], rowd[bp0 >> 5], bp0 & 31u) & smask;
      uint32_t bits1 = __builtin_amdgcn_alignbit(rowd[(bp1 >> 5) + 1], rowd[bp1 >> 5], bp1 & 31u) & smask;
      const uint32_t a1k = (uint32_t)(cb_t & (kListWindow - 1));   // offset of the variant inside its 128-byte line
      uint16_t* g1k = carriers + (cb_t - a1k);            // that block's base: g1k[a1k + k] is carrier k
      // staged genotype word of the group at list index q8 (a multiple of 8, as a1k is): (q8 + D) / 8
      uint32_t D = ((uint32_t)(gt0_t >> 3) & 3u) * 8u - a1k;
      uint32_t pos = a1k;                                 // list index of the round's first carrier
      const bool fits = a1k + cnt_t + 16u <= slice_ids_words(im.num_samples) * 2u;   // (entries of the list's LDS block)
      uint32_t done8 = a1k;                               // groups below this list index have been written
#pragma unroll
      for (int round = 0; round < 2; ++round) {
        uint32_t bits = round ? bits1 : bits0;
        const uint32_t idb = round ? bp1 : bp0;
        const uint32_t pc = __popc(bits);
        uint32_t incl = wave_inclusive_scan(pc);
        asm volatile("" : "+v"(incl));   // keeps the six fused DPP adds (the compiler otherwise re-associates them into ~20)
        const uint32_t end = pos + __builtin_amdgcn_readlane(incl, 63);
        uint32_t j = pos + incl - pc;                     // list index of this lane's first carrier of the round
        // (Tried in round 5 and removed: peeling by NIBBLE through a 16-entry table -- four 16-bit positions per 8-byte store at
        //  the lane's cursor, ceil(sb / 4) steps whatever the popcount.  A store that is only 2-byte aligned is legal on
        //  gfx950 but the LDS takes it one LANE at a time: 64 cycles per wave-instruction against 5.6 for ds_write_b16 and
        //  7.2 for an aligned ds_write_b64 (tools/microbench/lds_store.hip); the kernel went from 0.51 to 0.59 ms.)
#ifdef VS_PEEL_SINGLE   // (rounds 2-4: one id, one ds_write_b16 per iteration -- as many iterations as the fullest lane has bits)
        while (bits) {
          ids16[j++] = (uint16_t)(idb + __builtin_ctz(bits));
          bits &= bits - 1;
        }
#else
        // Two ids per iteration, one aligned ds_write_b32: the LDS store pipeline takes 5.6 cycles per ds_write_b16 and 6.5 per
        // ds_write_b32 (tools/microbench/lds_store.hip) and is ~70 % busy in this phase (profiles/r05_exp_split_pmc.json), so
        // pairs halve what the peel asks of it, and the loop's scalar bookkeeping with it.  A lane whose cursor is odd writes its
        // first id alone, a lane with one bit left its last.
        {
          if ((j & 1u) && bits) {
            ids16[j++] = (uint16_t)(idb + __builtin_ctz(bits));
            bits &= bits - 1;
          }
          uint32_t* pair = reinterpret_cast<uint32_t*>(ids16 + j);     // (j is even here for every lane that has bits left)
          uint32_t idbpk = idb * 0x10001u;
          asm volatile("" : "+v"(idbpk));                               // (kept out of the loop: the compiler otherwise rebuilds it per iteration)
          uint32_t t = bits & (bits - 1);
          while (t) {                                                   // at least two bits left
            const uint32_t w = ((uint32_t)__builtin_ctz(bits) | ((uint32_t)__builtin_ctz(t) << 16)) + idbpk;
            *pair++ = w;
            bits = t & (t - 1);
            t = bits & (bits - 1);
          }
          if (bits) *reinterpret_cast<uint16_t*>(pair) = (uint16_t)(idb + __builtin_ctz(bits));
        }
#endif
        // copy-out: lane q of a pass owns list entries 8q..8q+7 (one 16-byte store); their nibbles are 32 consecutive
        // bits of the stream.  Round 0 writes complete groups only, round 1 everything (the range owns its padding).
        // A variant whose whole list fits the LDS block (most dense ones: up to ~1290 of 2504 samples) is copied out ONCE, after the
        // second round -- three passes of 512 entries where two rounds took two each (round 5: -0.6 %).
        const uint32_t flush = round ? ((end + 7u) & ~7u) : (fits ? done8 : (end & ~7u));
        for (uint32_t q8 = done8 + lane * 8; q8 < flush; q8 += 512) {
          const uint4 iw = *reinterpret_cast<const uint4*>(ids16 + q8);
          // two carriers per word: id | gt << 13 in each half; the group's genotype word gives all four pairs by one shift and one
          // v_and_or_b32 each (round 5; a nibble stream cost two of each and an unaligned window per group)
          const uint32_t gw = gw_lds[(q8 + D) >> 3];
          store_group_nt(reinterpret_cast<uint4*>(g1k + q8), merge_group16(iw, gw, m_both));   // a1k is a multiple of 8 and the range owns its padding (pad_car)
        }
        if (round == 0) {
          // rebase: the incomplete group [flush, end) moves down by a whole number of 128-byte lines
          const uint32_t o = flush & ~(kListWindow - 1);
          if (o) {
            if (lane == 0) *reinterpret_cast<uint4*>(ids16 + (flush - o)) = *reinterpret_cast<const uint4*>(ids16 + flush);
            g1k += o;
            D += o;
          }
          pos = end - o;
          done8 = flush - o;
        }
      }
    } else if (cnt_t <= kMidMax) {
      const uint32_t a0 = (uint32_t)(cb_t & 63);        // offset of the variant inside its first aligned block
      uint32_t* gbase = carriers + (cb_t - a0);         // that block's base: gbase[a0 + k] is carrier k
      const uint32_t endpos = a0 + cnt_t;
      // ---- medium density: lane per row word, ids staged in LDS, coalesced copy-out ----
      const uint32_t pc = __popcll(mine);
      uint32_t incl = pc;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
      }
      uint32_t k = incl - pc;
      const uint32_t idbase = lane * 64;
      while (mine) {
        const uint32_t bit = __builtin_ctzll(mine);
        mine &= mine - 1;
        ids_lds[k] = idbase + bit;
        ++k;
      }
      // copy-out in 256-byte-aligned blocks of the arena
      for (uint32_t pos = lane; pos < endpos; pos += 64)
        if (pos >= a0) gbase[pos] = ids_lds[pos - a0] | ((uint32_t)gt_lds[nshift + pos - a0] << 29);
    } else {
      // ---- dense: bit per lane, two row words per step.  The lanes whose bit is set (exec mask =
      //      the word itself) rank themselves with v_mbcnt and drop id|gt into a 512-entry LDS ring
      //      indexed by arena position; the ring leaves 1 KiB at a time as one 16-byte store per lane
      //      on a 1 KiB-aligned arena block (aligned full stores run at twice the rate of partial ones,
      //      tools/microbench/write_bw.hip) ----
      const uint32_t a1k = (uint32_t)(cb_t & 255);        // offset of the variant inside its 1 KiB block
      uint32_t* g1k = carriers + (cb_t - a1k);            // that block's base: g1k[a1k + k] is carrier k
      const uint32_t end1k = a1k + cnt_t;
      const uint32_t gtoff = nshift - a1k;                // staged genotype index = arena position + gtoff
      uint32_t bpos = a1k;                                // arena position of the step's first carrier
      uint32_t nfl = 0;                                   // 256-entry blocks already written
      for (uint32_t w = 0; w < wpc; w += 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint64_t word = (w + i < wpc) ? wave_bcast64(mine, w + i) : 0ULL;
          if (__builtin_amdgcn_inverse_ballot_w64(word)) {
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(word >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)word, bpos));
            ring[pos & 511u] = ((w + i) * 64 + lane) | ((uint32_t)gt_lds[pos + gtoff] << 29);
          }
          bpos += __popcll(word);
        }
        while (nfl < (bpos >> 8)) {                        // a complete 256-entry block is ready
          const uint32_t p4 = nfl * 256 + lane * 4;
          const uint4 v = *reinterpret_cast<const uint4*>(&ring[p4 & 511u]);
          if (p4 >= a1k) *reinterpret_cast<uint4*>(g1k + p4) = v;
          else if (p4 + 4 > a1k) {                         // the variant starts inside this lane's quad
            const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) if (p4 + j >= a1k) g1k[p4 + j] = e[j];
          }
          ++nfl;
        }
      }
      for (uint32_t p4 = nfl * 256 + lane * 4; p4 < end1k; p4 += 256) {   // tail (at most 2 passes)
        const uint4 v = *reinterpret_cast<const uint4*>(&ring[p4 & 511u]);
        if (p4 >= a1k && p4 + 4 <= end1k) *reinterpret_cast<uint4*>(g1k + p4) = v;
        else {
          const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) if (p4 + j >= a1k && p4 + j < end1k) g1k[p4 + j] = e[j];
        }
      }
    }
  }
  if (TUNE && tstat) tstat[1] = wall_clock64() - t_lists;
}

template <bool WIDE, uint32_t CH, bool TUNE>
__global__ void __launch_bounds__(256) k_fill_carriers(DevImage im, DevResult r, uint32_t ablate, uint32_t gt_words) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t A = r.A;
  const uint64_t nchunks = (A + CH - 1) / CH;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  const uint32_t lds_words_per_wave = WIDE ? gt_words + kRingWords : slice_lds_words(im.num_samples);
  // one task per wave (the launch covers every task): with no loop around it the compiler has no lane-dependent
  // invariants to keep alive, and the hardware's block scheduler balances the tenfold spread of task costs
  if (wave < nchunks) {
    const uint64_t a = wave * CH + lane;
    uint32_t cnt = 0, cls = 0;
    uint64_t gt0 = 0, cb = 0;
    if (a < A && lane < CH) {   // read once
      const uint4 y = reinterpret_cast<const uint4*>(r.rows + a)[1];   // {alt_len, count | dropped, car_begin}
      cnt = y.y & ~kRowDropped;
      cb = ((uint64_t)y.w << 32) | y.z;
      cls = __builtin_nontemporal_load(&r.r_class[a]);
      gt0 = __builtin_nontemporal_load(&r.r_gt0[a]);
      if (cls == kNone) cnt = 0;   // the row shares another row's list (k_t4_claim): nothing to expand here
    }
    expand_task<WIDE, false, TUNE>(im, r.carriers, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], lane, cnt, cls, gt0, cb, ablate, gt_words);
  }
}

// The same expansion over the UNIQUE sites of a batch whose carrier lists are shared: the slot parameters come straight
// from the site table (sequential reads, each site once), the arena offset from k_unique_sites.
template <bool WIDE, uint32_t CH, bool TUNE>
__global__ void __launch_bounds__(256) k_fill_sites(DevImage im, DevResult r, const uint32_t* u_site, uint64_t U, uint32_t ablate, uint32_t gt_words) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nchunks = (U + CH - 1) / CH;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  const uint32_t lds_words_per_wave = WIDE ? gt_words + kRingWords : slice_lds_words(im.num_samples);
  if (wave < nchunks) {
    const uint64_t u = wave * CH + lane;
    uint32_t cnt = 0, cls = 0;
    uint64_t gt0 = 0, cb = 0;
    if (u < U && lane < CH) {
      const uint32_t g = __builtin_nontemporal_load(&u_site[u]);
      const uint4 y = reinterpret_cast<const uint4*>(r.rows + u)[1];   // {alt_len, count | dropped, car_begin}: the shared rows are table rows [0, U)
      cnt = y.y & ~kRowDropped;
      cb = ((uint64_t)y.w << 32) | y.z;
      cls = im.s_class[g];
      gt0 = im.s_gt0[g];
    }
    expand_task<WIDE, false, TUNE>(im, r.carriers, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], lane, cnt, cls, gt0, cb, ablate, gt_words);
  }
}

// Round 4: the same expansion WRITES the shared rows as well (k_share_rows2's work, on the 16 lanes that fetch the task's
// parameters anyway): the lanes find the runs of covered sites their rows lie in (shared_row_run), a lane turns its row
// number into a site with that run's record, copies the static site row with the list offset rebased, and takes
// count / source handle / genotype offset of the same site for the expansion.  No row kernel, no site index written
// and read back, no second read of the rows.
// One task of the shared expansion: rows [u_first, u_first + CH) of the shared table -- their rows and their lists.
template <bool WIDE, uint32_t CH, bool TUNE, bool DENSE>
__device__ __forceinline__ void fill_sites_task(const DevImage& im, const DevResult& r, const RunRec* __restrict__ runs, const uint32_t* __restrict__ coarse, uint64_t n_runs,
                                                uint64_t U, uint64_t u_first, uint32_t lane, uint32_t* lds_wave, uint32_t ablate, uint32_t gt_words,
                                                unsigned long long* tstat, uint64_t task, uint64_t t_start) {
  __builtin_amdgcn_s_setprio(3);   // (back to 0 where expand_task's dense phase begins)
  const RowDelta d = shared_row_run(runs, coarse, n_runs, u_first, lane, lane < CH ? lane : 0u);
  const uint64_t u = u_first + lane;
  uint32_t cnt = 0, cls = 0;
  uint64_t gt0 = 0, cb = 0;
  if (u < U && lane < CH) {
    const uint32_t g = (uint32_t)(u + d.dg);
    const uint4* src = reinterpret_cast<const uint4*>(im.s_row + g);
    const uint4 x = ld_stream16(src);
    uint4 y = ld_stream16(src + 1);
    cls = im.s_class[g];
    gt0 = im.s_gt0[g];
    cb = (((uint64_t)y.w << 32) | y.z) + d.dc;
    y.z = (uint32_t)cb; y.w = (uint32_t)(cb >> 32);
    uint4* dst = reinterpret_cast<uint4*>(r.rows + u);
    dst[0] = x; dst[1] = y;
    cnt = y.y & ~kRowDropped;
  }
  unsigned long long ph[3] = {0, 0, 0}, t_params = 0;
  if (TUNE && tstat) {
    __builtin_amdgcn_s_waitcnt(0);   // the task's parameters are in registers, its rows are on their way
    t_params = wall_clock64() - t_start;
  }
  expand_task<WIDE, false, TUNE, true, DENSE>(im, r.carriers, lds_wave, lane, cnt, cls, gt0, cb, ablate, gt_words, (TUNE && tstat) ? ph : nullptr);
  if (TUNE && tstat && lane == 0)   // one 16-byte record per task, written once at the end: {parameters + rows, list phase, dense phase, whole task | dense variants << 24}
    reinterpret_cast<uint4*>(tstat)[task] = uint4{(uint32_t)t_params, (uint32_t)ph[0], (uint32_t)ph[1], (uint32_t)(wall_clock64() - t_start) | ((uint32_t)ph[2] << 24)};
}

template <bool WIDE, uint32_t CH, bool TUNE, bool DENSE = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_fill_sites2(DevImage im, DevResult r, const RunRec* __restrict__ runs, const uint32_t* __restrict__ coarse, uint64_t n_runs,
                                                     uint64_t U, uint32_t ablate, uint32_t gt_words, unsigned long long* tstat, const PlanDev* pd) {
  if (pd) {   // speculative batch (k_rows.hip.h: PlanDev): the launch covers the rows that were ALLOCATED; the plan's record says how many exist
    if (pd->refused) return;
    U = pd->U; n_runs = pd->n_runs;
  }
  const uint64_t t_start = (TUNE && tstat) ? wall_clock64() : 0;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  const uint32_t lds_words_per_wave = WIDE ? gt_words + kRingWords : slice_lds_words(im.num_samples);
  const uint64_t u_first = wave * CH;
  if (u_first < U)
    fill_sites_task<WIDE, CH, TUNE, DENSE>(im, r, runs, coarse, n_runs, U, u_first, lane, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], ablate, gt_words, tstat, wave, t_start);
}

// Tuning builds only (option fill_mode = 2): the DENSE variants of a shared batch in a launch of their own, so that a profiler
// sees the two regimes of the expansion apart -- round 5's counters: the lists + rows kernel moves its 1.34 GB in 0.21 ms
// (6.3 TB/s: at the part's ceiling), the dense kernel its 1.33 GB in 0.38 - 0.40 ms with the VALU 60 - 65 % and the LDS
// pipeline ~70 % busy (profiles/r05_exp_split_pmc.json); one after the other they take 0.59 ms where the mixed kernel takes
// 0.51, so the mixed kernel stays.  (Also tried and removed: resident waves pulling tasks from eight counters, 0.545 - 0.56
// against 0.505 ms -- the dispatcher is not what holds the kernel back.)  The index keeps the list
// of its dense sites (more than list_max carriers: DevImage::dense_site); a wave owns K consecutive entries of that list,
// its lanes look their sites up in the batch's runs -- covered or not, which row, which arena offset -- and the wave expands
// the covered ones one after the other, rows requested two variants ahead and nibbles one, K deep instead of the two or
// three a mixed task holds.  k_fill_sites2<..., DENSE = false> writes the rows and everything else.
#ifdef VS_TUNING
template <bool WIDE, uint32_t K>
__global__ void __launch_bounds__(256) k_fill_dense(DevImage im, DevResult r, const RunRec* __restrict__ runs, uint64_t n_runs, uint64_t U, uint32_t gt_words) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  const uint32_t lds_words_per_wave = WIDE ? gt_words + kRingWords : slice_lds_words(im.num_samples);
  const uint64_t d_first = wave * K;
  if (d_first >= im.n_dense) return;
  uint32_t cnt = 0, cls = 0;
  uint64_t gt0 = 0, cb = 0;
  if (lane < K && d_first + lane < im.n_dense) {
    const uint32_t g = im.dense_site[d_first + lane];
    uint64_t lo = 0, hi = n_runs;                // the last run that starts at or before site g (runs are in site order as well as in row order)
    while (hi - lo > 1) {
      const uint64_t m = (lo + hi) >> 1;
      if (runs[m].u_start + runs[m].dg <= g) lo = m; else hi = m;
    }
    const uint4* rp = reinterpret_cast<const uint4*>(runs + lo);
    const uint4 a = rp[0], b = rp[1];
    const uint64_t u_start = ((uint64_t)a.y << 32) | a.x, dg = ((uint64_t)a.w << 32) | a.z, dc = ((uint64_t)b.y << 32) | b.x;
    const uint64_t u_end = lo + 1 < n_runs ? runs[lo + 1].u_start : U;
    if (n_runs && g >= u_start + dg && g - dg < u_end) {
      const uint4 y = reinterpret_cast<const uint4*>(im.s_row + g)[1];
      cnt = y.y & ~kRowDropped;
      cb = (((uint64_t)y.w << 32) | y.z) + dc;
      cls = im.s_class[g];
      gt0 = im.s_gt0[g];
    }
  }
  if (__ballot(cnt != 0) == 0) return;
  expand_task<WIDE, true, false, false, true>(im, r.carriers, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], lane, cnt, cls, gt0, cb, 0u, gt_words);
}
#endif

}  // namespace vsamd
