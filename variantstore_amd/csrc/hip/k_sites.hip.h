// k_sites.hip.h -- site table built when an index is opened; region bounds.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_image.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Site table: one thread per ref-path slot.  Output position of a slot's j-th
// branch is rp_cand_prefix[slot] + j, so no inter-lane communication is needed.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_sites(DevImage im, uint64_t slot_begin, uint64_t slot_end) {
  const uint64_t i = slot_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slot_end) return;
  const uint32_t it = im.rp_vid[i], succ = im.rp_vid[i + 1];
  uint32_t g = im.rp_cand_prefix[i];
  const uint32_t it_ridx = im.v_ridx[it], it_len = im.v_len[it];
  const uint32_t e1 = im.row_ptr[it + 1];
  for (uint32_t e = im.row_ptr[it]; e < e1; ++e) {
    const uint32_t b = im.col[e];
    if (b == succ) continue;
    const uint32_t ncar = im.v_ncar[b];
    uint32_t pos = 0, ro = 0, rl = 0, ao = 0, al = 0, fl = 0;
    const uint32_t succ_ridx = im.v_ridx[succ];
    if (ncar == 0) {
      fl = kSiteAlwaysDrop;  // get_samples() false: var_pos never written in the reference
    } else if (im.v_ridx[b] != 0) {  // deletion, query.h:336-350
      if (succ_ridx == 0) fl = kSiteAlwaysDrop;
      pos = succ_ridx; ro = im.v_off[succ]; rl = im.v_len[succ];
    } else {
      uint32_t nri = im.v_nri[b];
      if (nri == kNone) nri = it_ridx;  // "consecutive mutation": sample keeps *it's ref entry
      if (nri == it_ridx + it_len) {    // insertion, query.h:369-376
        pos = nri - 1; ao = im.v_off[b]; al = im.v_len[b];
      } else {                          // substitution, query.h:377-392
        if (succ_ridx == 0) fl = kSiteAlwaysDrop;
        pos = succ_ridx; ro = im.v_off[succ]; rl = im.v_len[succ];
        ao = im.v_off[b]; al = im.v_len[b];
      }
    }
    im.s_pos[g] = pos; im.s_ref_off[g] = ro; im.s_ref_len[g] = rl; im.s_alt_off[g] = ao; im.s_alt_len[g] = al;
    im.s_vid[g] = b; im.s_ncar[g] = (fl & kSiteAlwaysDrop) ? 0u : ncar; im.s_flags[g] = fl;
    im.s_class[g] = im.v_src[b]; im.s_gt0[g] = im.v_car_begin[b];
    ++g;
  }
}

// per-slot copies of the two site-table prefixes (one memory level less in every region-bounds computation)
__global__ void __launch_bounds__(256) k_slot_prefixes(DevImage im, uint64_t* rp_carpre, uint64_t* rp_kpre) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > im.P) return;
  const uint32_t g = im.rp_cand_prefix[i];
  rp_carpre[i] = im.s_carpre[g];
  rp_kpre[i] = im.s_kpre[g];
}

// The static site rows (DevImage::s_row): what a row of the variant table says about site g when its carrier list lies at
// s_carpre[g] -- a batch copies the record and rebases the list offset (k_share_rows2, k_fill_sites2, emit_region).
__global__ void __launch_bounds__(256) k_build_site_rows(DevImage im, VariantRow* rows) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= im.G) return;
  row_store(rows, g, im.s_pos[g], im.s_ref_off[g], im.s_ref_len[g], im.s_alt_off[g], im.s_alt_len[g], im.s_ncar[g],
            (im.s_flags[g] & kSiteAlwaysDrop) != 0, im.s_carpre[g]);
}

// per-slot and per-rank records of the region bounds (DevImage::rp_rec, rk_rec)
__global__ void __launch_bounds__(256) k_slot_records(DevImage im, uint4* rec) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > im.P) return;
  const uint64_t c = im.rp_carpre[i], k = im.rp_kpre[i];
  rec[2 * i] = uint4{im.rp_cand_prefix[i], im.rp_sus_prefix[i], (uint32_t)c, (uint32_t)(c >> 32)};
  rec[2 * i + 1] = uint4{(uint32_t)k, (uint32_t)(k >> 32), 0u, 0u};
}
__global__ void __launch_bounds__(256) k_rank_records(DevImage im, uint2* rec) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > im.R) return;
  rec[r] = uint2{r < im.R ? im.idx_pos[r] : 0u, im.rank_to_slot[r]};
}

// nearest earlier site with the same (pos, alt); positions are sorted up to an
// off-by-one (an insertion reports end-1, everything else end), so the backward
// scan stops at the first site whose pos < p-1.
__global__ void __launch_bounds__(256) k_mark_dups(DevImage im) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= im.G) return;
  uint32_t res = kNone;
  if (!(im.s_flags[g] & kSiteAlwaysDrop)) {
    const uint32_t p = im.s_pos[g], ao = im.s_alt_off[g], al = im.s_alt_len[g];
    for (uint64_t k = 0; k < g && k < 65536; ++k) {
      const uint64_t i = g - 1 - k;
      if (im.s_flags[i] & kSiteAlwaysDrop) continue;
      const uint32_t pi = im.s_pos[i];
      if (pi + 1 < p) break;
      if (pi == p && im.s_alt_len[i] == al && seq_equal(im, im.s_alt_off[i], ao, al)) { res = (uint32_t)i; break; }
    }
  }
  im.s_dup_prev[g] = res;
}

// ---------------------------------------------------------------------------
// Region bounds: one thread per region.
// ---------------------------------------------------------------------------
struct RegionBounds {
  uint32_t g0, g1;   // site range [g0, g1)
  uint8_t flags;
  uint64_t pre0;     // arena prefix at g0 (s_carpre[g0]), padded arena entries and reported carriers of the range --
  uint64_t npad;     //   read through per-slot copies (rp_carpre / rp_kpre) at the same memory level as g0 and g1,
  uint64_t nkept;    //   not one level later through the site table
};
// Index::is_empty (index.h:150-166), Index::find(x) (index.h:119-133) and the stop rule of the walk (query.h:312).
// Written for memory-level parallelism: both ranks are requested together, every table is read on a clamped index
// whether or not the reference's early-outs fire (they select the result at the end), so a region costs three
// dependent memory levels -- ranks; select + slots; branch, dedup and arena prefixes of the two slots.
__device__ __forceinline__ RegionBounds region_bounds_of(const DevImage& im, uint64_t x, uint64_t y) {
  const RankLoads lx = rank1_issue(im, x), ly = rank1_issue(im, y - 1);   // y == 0 wraps and is clamped: x < y fails then
  const uint32_t rx = rank1_finish(lx), ry = rank1_finish(ly);
  const uint32_t R = (uint32_t)im.R, P = (uint32_t)im.P;
  const bool invalid = x < 1;                                            // the reference aborts (index.h:151-154)
  // rank level: {select(rank + 1), first slot} records; find(x) = rank(x) - 1 is the neighbour of rank(x): one line
  uint64_t rf = (x >= im.ref_length) ? (uint64_t)R - 1 : (uint64_t)(rx ? rx - 1 : 0);
  if (rf > (uint64_t)R - 1) rf = (uint64_t)R - 1;
  const uint2 kx = im.rk_rec[rx < R ? rx : R - 1], kf = im.rk_rec[rf], ky = im.rk_rec[ry < R ? ry : R];
  const uint64_t sel = kx.x;                                             // select(rank(x) + 1)
  // is_empty: x beyond the reference, select past the last one (defined as empty), or no node start in (.., y]
  const bool empty = x > im.ref_length || rx >= R || !(sel - 1 <= y);
  // find(x): rank(x) >= 1 for every x >= 1 because a node starts at index 1
  const uint32_t s0 = kf.y;
  // first slot whose node ends at or after y stops the walk; node ends tile the reference, so that is the slot before
  // the first start >= y (rank_to_slot[R] == P)
  const uint32_t s1raw = ky.y;
  uint32_t s1 = s1raw ? s1raw - 1 : 0;
  if (s1 < s0) s1 = s0;
  if (s1 > P) s1 = P;
  // slot level: one 32-byte record per end {first site, suspicious sites before it, arena prefix, carrier prefix}
  const uint4 a0 = im.rp_rec[2 * (uint64_t)s0], a1 = im.rp_rec[2 * (uint64_t)s0 + 1], b0 = im.rp_rec[2 * (uint64_t)s1], b1 = im.rp_rec[2 * (uint64_t)s1 + 1];
  uint32_t g0 = a0.x, g1 = b0.x;
  // can the "already seen" rule fire inside [g0,g1)?  (g0, g1 are slot boundaries: the list range is tabulated)
  uint32_t lo = a0.y, hi = b0.y;
  uint64_t pre0 = ((uint64_t)a0.w << 32) | a0.z, npad = (((uint64_t)b0.w << 32) | b0.z) - pre0;
  uint64_t nkept = (((uint64_t)b1.y << 32) | b1.x) - (((uint64_t)a1.y << 32) | a1.x);
  const bool walk = !invalid && !empty && x < y;
  if (!walk) { g0 = 0; g1 = 0; lo = 0; hi = 0; pre0 = 0; npad = 0; nkept = 0; }
  uint8_t fl = invalid ? kRegionInvalid : (empty ? kRegionEmpty : 0);
  for (uint32_t k = lo; k < hi; ++k) {
    const uint32_t pv = im.sus_prev[k];
    if (pv == kNone || pv >= g0) { fl |= kRegionSlow; break; }
  }
  return RegionBounds{g0, g1, fl, pre0, npad, nkept};
}

__device__ __forceinline__ void region_bounds(const DevImage& im, const DevResult& r, uint64_t q) {
  const RegionBounds b = region_bounds_of(im, r.regions[2 * q], r.regions[2 * q + 1]);
  r.q_flags[q] = b.flags;
  r.q_g0[q] = b.g0;
  r.q_nvar[q] = b.g1 - b.g0;
  r.q_ncar[q] = b.npad;
}

__global__ void __launch_bounds__(256) k_region_bounds(DevImage im, DevResult r) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < r.Q) region_bounds(im, r, q);
}

}  // namespace vsamd
