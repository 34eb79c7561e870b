// k_points.hip.h -- point queries (types 1 and 7).
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_sites.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Point queries (types 1 and 7).  A single next_variant_in_ref(pos) call with an empty `vars`
// walks the ref path from find(pos) and stops at the first node with a reportable branch, so its
// answer is the branch list of ONE ref-path slot: the first slot >= slot(find(pos)) whose sites
// carry anybody (always-dropped sites have s_ncar == 0, reportable ones >= 1).  The arena prefix at the slots' first
// sites (rp_carpre) is monotone in the slot, so that slot is found by a galloping search from the start slot.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t slot_of_find(const DevImage& im, uint64_t pos) {  // Index::find, index.h:119-133
  uint64_t rf;
  if (pos >= im.ref_length) rf = im.R - 1;
  else { const uint32_t k = rank1(im, pos); rf = k == 0 ? 0 : k - 1; }
  return im.rank_to_slot[rf];
}

// first slot m >= s0 with a reportable branch: rp_carpre[m] == rp_carpre[s0] < rp_carpre[m + 1] (rp_carpre = the arena
// prefix at every slot's first site).  The next variant is a few slots away, so the search gallops from s0 (probes at
// +1, +2, +4, ...: the first ones share a cache line) and bisects inside the last stride -- a bisection over the whole
// path read two dependent tables at each of its 23 levels: 11 KB of HBM traffic per query by the counters.
__device__ __forceinline__ uint32_t next_valid_slot(const DevImage& im, uint32_t s0) {
  const uint32_t P = (uint32_t)im.P;
  if (s0 >= P) return P;
  const uint64_t base = im.rp_carpre[s0];
  uint32_t lo = s0, hi = s0, step = 1;
  while (true) {
    hi = (P - s0 > step) ? s0 + step : P;
    if (im.rp_carpre[hi] > base) break;
    if (hi == P) return P;
    lo = hi;
    step = step < 0x80000000u ? step << 1 : 0xFFFFFFFFu;
  }
  while (lo + 1 < hi) {   // rp_carpre[lo] == base < rp_carpre[hi]
    const uint32_t m = lo + ((hi - lo) >> 1);
    if (im.rp_carpre[m] > base) hi = m; else lo = m;
  }
  return lo;
}

__device__ __forceinline__ uint64_t first_reported_pos(const DevImage& im, uint32_t s) {  // vars[0].var_pos of the call
  uint32_t g = im.rp_cand_prefix[s];
  while (im.s_ncar[g] == 0) ++g;
  return im.s_pos[g];
}

// mode 1: closest_var, mode 7: samples_has_var.  regions[2q] = pos.
__global__ void __launch_bounds__(256) k_point_bounds(DevImage im, DevResult r, uint32_t mode) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  const uint64_t pos = r.regions[2 * q];
  const uint32_t P = (uint32_t)im.P;
  uint8_t fl = 0;
  uint32_t chosen = P;
  const uint32_t s = next_valid_slot(im, slot_of_find(im, pos));
  if (mode == 7) {
    chosen = s;
    if (s == P) fl = kRegionNotFound;
  } else if (s < P) {  // query.h:451-465
    const uint64_t next_var_pos = first_reported_pos(im, s);
    const int cur_pos = (int)(uint32_t)(pos - (next_var_pos - pos));
    chosen = s;
    if (cur_pos > 0) {
      const uint32_t s2 = next_valid_slot(im, slot_of_find(im, (uint64_t)cur_pos));
      // s2 == P would be prev_var[0] of an empty vector in the reference; next_var is kept then
      if (s2 < P && first_reported_pos(im, s2) != next_var_pos) chosen = s2;
    }
  } else {             // query.h:466-473: step back one position at a time until a call finds something
    const int cur_pos = (int)(uint32_t)(pos - 1);
    if (cur_pos > 0) {
      const uint32_t s2 = next_valid_slot(im, slot_of_find(im, (uint64_t)cur_pos));
      if (s2 < P) chosen = s2;
      else if (im.s_carpre[im.G] == 0) fl = kRegionNotFound;  // reaches cur_pos == 1: returns false
      else {
        // The first call that finds something is the one at the largest position whose find() slot is not
        // beyond Z, the last slot with a reportable branch; it reports the first such slot from there on
        // (slots between two find() images are the zero-length dummy nodes' successors, so that need not be Z).
        const uint64_t total = im.s_carpre[im.G];
        uint32_t lo = 0, hi = P - 1;
        while (lo < hi) {
          const uint32_t m = lo + ((hi - lo) >> 1);
          if (im.s_carpre[im.rp_cand_prefix[m + 1]] >= total) hi = m; else lo = m + 1;
        }
        const uint32_t Z = lo;
        uint64_t rlo = 0, rhi = im.R - 1;   // largest rank whose first slot is <= Z (rank 0 maps to slot 0)
        while (rlo < rhi) {
          const uint64_t m = rlo + ((rhi - rlo + 1) >> 1);
          if (im.rank_to_slot[m] <= Z) rlo = m; else rhi = m - 1;
        }
        chosen = next_valid_slot(im, im.rank_to_slot[rlo]);
      }
    }  // else: the loop is not entered, vars stays empty and the call returns true
  }
  uint32_t g0 = 0, g1 = 0;
  if (chosen < P) {
    g0 = im.rp_cand_prefix[chosen]; g1 = im.rp_cand_prefix[chosen + 1];
    const uint32_t lo = im.rp_sus_prefix[chosen], hi = im.rp_sus_prefix[chosen + 1];
    for (uint32_t k = lo; k < hi; ++k) {
      const uint32_t pv = im.sus_prev[k];
      if (pv == kNone || pv >= g0) { fl |= kRegionSlow; break; }
    }
  }
  r.q_flags[q] = fl;
  r.q_g0[q] = g0;
  r.q_nvar[q] = g1 - g0;
  r.q_ncar[q] = im.s_carpre[g1] - im.s_carpre[g0];
}

// samples_has_var: keep the first reported variant whose (ref, var_pos, alt) equals the query's
// (query.h:802-803); everything else of the slot is dropped.  One thread per query; strings are the
// caller's bytes, compared with the decoded sequence characters (get_sequence, variant_graph.h:1261-1268).
__global__ void __launch_bounds__(64) k_has_var_filter(DevImage im, DevResult r, const uint8_t* chars, const uint64_t* str_off) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  const uint64_t pos = r.regions[2 * q];
  const uint8_t* ref = chars + str_off[2 * q];
  const uint64_t ref_len = str_off[2 * q + 1] - str_off[2 * q];
  const uint8_t* alt = chars + str_off[2 * q + 1];
  const uint64_t alt_len = str_off[2 * q + 2] - str_off[2 * q + 1];
  const uint64_t a0 = r.var_begin[q], n = r.q_nvar[q];
  bool found = false;
  uint64_t kept_car = 0;
  for (uint64_t j = 0; j < n; ++j) {
    const uint64_t a = a0 + j;
    const VariantRow v = row_load(r.rows, a);
    if (row_dropped(v)) continue;
    bool match = !found && v.pos == pos && v.ref_len == ref_len && v.alt_len == alt_len;
    if (match) {
      const char dec[8] = {'A', 'C', 'T', 'G', 'N', 5, 5, 5};  // map_int, util.cc:32-41
      for (uint64_t i = 0; match && i < ref_len; ++i) match = (uint8_t)dec[im.seq_codes[v.ref_off + i] & 7] == ref[i];
      for (uint64_t i = 0; match && i < alt_len; ++i) match = (uint8_t)dec[im.seq_codes[v.alt_off + i] & 7] == alt[i];
    }
    if (match) { found = true; kept_car = row_count(v); }
    else r.rows[a].count_flags = kRowDropped;
  }
  r.var_count[q] = found ? 1 : 0;
  r.q_ncar[q] = kept_car;
  if (!found) r.q_flags[q] |= kRegionNotFound;
}

}  // namespace vsamd
