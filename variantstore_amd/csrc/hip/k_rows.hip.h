// k_rows.hip.h -- the variant table: private rows, shared rows and carrier lists of a sorted batch, the duplicate rule.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_sites.hip.h"
#include "k_scan.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Variant headers: one wave per region, lanes stride the region's site range.
// ---------------------------------------------------------------------------
// Private rows of region q: its site range copied into the table at var_begin[q].  PARAMS: also the per-row parameters
// k_fill_carriers reads (source handle, genotype offset) -- a batch with shared lists expands from the site table instead.
template <bool PARAMS>
__device__ __forceinline__ void emit_region(const DevImage& im, const DevResult& r, uint64_t q, uint32_t lane) {
  const uint64_t n = r.q_nvar[q], a0 = r.var_begin[q], cb = r.car_base[q];
  const uint32_t g0 = r.q_g0[q];
  const uint64_t rebase = cb - im.s_carpre[g0];   // a static site row carries s_carpre[g] as its list offset
  uint32_t kept = 0;
  for (uint64_t j = lane; j < n; j += 64) {
    const uint64_t a = a0 + j;
    const uint32_t g = g0 + (uint32_t)j;
    const uint4* src = reinterpret_cast<const uint4*>(im.s_row + g);   // one 32-byte record per site (round 3: eight SoA reads)
    const uint4 x = src[0];
    uint4 y = src[1];
    kept += y.y & ~kRowDropped;
    const uint64_t at = (((uint64_t)y.w << 32) | y.z) + rebase;
    y.z = (uint32_t)at; y.w = (uint32_t)(at >> 32);
    uint4* dst = reinterpret_cast<uint4*>(r.rows + a);
    dst[0] = x; dst[1] = y;
    if (PARAMS) {
      r.r_class[a] = im.s_class[g];
      r.r_gt0[a] = im.s_gt0[g];
    }
  }
  kept = wave_inclusive_scan(kept);
  if (lane == 63 && !(r.q_flags[q] & kRegionSlow)) { r.var_count[q] = n; r.q_ncar[q] = kept; }
}

template <bool PARAMS>
__global__ void __launch_bounds__(256) k_emit_headers(DevImage im, DevResult r) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= r.Q) return;
  emit_region<PARAMS>(im, r, q, threadIdx.x & 63);
}

// ---------------------------------------------------------------------------
// Shared carrier lists.  The regions of a batch arrive sorted (the reference's driver sorts them, commands.cc:91) and
// overlap -- 100 k regions of 10 kb cover chr1 four times over -- so most sites are reported by several regions of the
// same batch.  A site's carrier list is then expanded ONCE into the arena and every region that reports the site
// points its row at it (the way REF / ALT strings are (offset, length) references into the sequence pool):
//   E_prev[q]   = largest site end among the regions before q         (exclusive prefix max)
//   new part    = [max(g0, E_prev), g1): the sites no earlier region covers -- the part of the arena region q OWNS
//   arena_new   = exclusive prefix sum of the new parts' padded carrier counts: where the new part starts
//   car_base[q] = arena position of site g0 = arena_new[q] - (carpre[E_prev] - carpre[g0]) when g0 lies in covered
//                 ground (sites [g0, E_prev) are contiguous there: the region that reached E_prev starts at or before g0)
// so a row's carrier offset keeps its form car_base[q] + carpre[g] - carpre[g0].  Needs g0 ascending over the
// regions with any site; a batch that is not reports so (PlanTotals::not_sorted) and is sorted on the device first.
//
// THE PLAN of a batch is four short launches (rounds 2-3 took bounds + five scan launches):
//   k_t6_bounds   region bounds (or bounds from gathered records) + the {max end, max start} of every tile
//   k_t6_mid      every tile reduces the tiles before it by itself (a few thousand words at most: no spine launch),
//                 E_prev per region, what each region adds to the batch, the tile sums of that
//   k_t6_totals   one block: the tile sums become their exclusive prefix in place, the batch's totals go into mapped host
//                 memory, sequence word last -- the host spins on it
//   k_t6_apply    the per-region arrays (regions under the duplicate rule get their private rows' place behind the shared
//                 table: every block reads the totals), the run records of k_share_rows2 / k_fill_sites2, the list of
//                 slow regions.  No host attention; it has to fit the holes an expansion in flight leaves (<= 64 VGPRs).
// A thread owns `items` consecutive regions (1 up to a million regions per batch), so the number of tiles stays
// within what a block reduces by itself whatever the batch.
// ---------------------------------------------------------------------------
constexpr int kPlanBlock = 256;
constexpr uint32_t kPlanMaxTiles = 4096;
// Round 6: the prefix maximum carries the ARENA PREFIX at the largest site end with it (c1 = s_carpre[g1], which the bounds kernel has from
// the slot record anyway): E_prev and s_carpre[E_prev] then come out of the same scan, and what a region adds to the batch (share_new)
// follows from three numbers the plan already holds -- k_t6_mid and k_t6_apply no longer look anything up in the site table (57 MB of
// scattered 128-byte lines per 100 k regions in round 5: four s_carpre / s_kpre look-ups per region and pass), they stream their arrays.
struct ShareMax { uint32_t g1, g0; uint64_t c1; };
__device__ __forceinline__ ShareMax smax(ShareMax a, ShareMax b) {
  return ShareMax{a.g1 >= b.g1 ? a.g1 : b.g1, a.g0 > b.g0 ? a.g0 : b.g0, a.g1 >= b.g1 ? a.c1 : b.c1};
}
__device__ __forceinline__ ShareMax block_exclusive_max(ShareMax v, ShareMax* total) {
  __shared__ ShareMax wmx[kPlanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  ShareMax incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t a = __shfl_up(incl.g1, d, 64), b = __shfl_up(incl.g0, d, 64);
    const uint64_t c = __shfl_up(incl.c1, d, 64);
    if (lane >= d) incl = smax(incl, ShareMax{a, b, c});
  }
  if (lane == 63) wmx[wid] = incl;
  __syncthreads();
  ShareMax woff{0, 0, 0}, tot{0, 0, 0};
  for (int w = 0; w < kPlanBlock / 64; ++w) {
    if (w < wid) woff = smax(woff, wmx[w]);
    tot = smax(tot, wmx[w]);
  }
  __syncthreads();
  *total = tot;
  const uint32_t pa = __shfl_up(incl.g1, 1, 64), pb = __shfl_up(incl.g0, 1, 64);
  const uint64_t pc = __shfl_up(incl.c1, 1, 64);
  return lane ? smax(woff, ShareMax{pa, pb, pc}) : woff;
}
// What the bounds kernel leaves per region for the rest of the plan, in arrays the plan's last pass overwrites with their final content:
//   q_ncar[q]    padded arena entries of the region's site range (final in q_car_len)
//   car_base[q]  arena prefix at its first site, s_carpre[g0]     (final: the region's arena offset)
//   q_car_len[q] carriers its rows report, s_kpre[g1] - s_kpre[g0] (final in q_ncar)
__device__ __forceinline__ ShareMax share_elem(const DevResult& r, uint64_t q) {   // {end, start, arena prefix at the end}; zeros without sites
  const uint32_t nv = (uint32_t)r.q_nvar[q];
  return nv ? ShareMax{r.q_g0[q] + nv, r.q_g0[q], r.car_base[q] + r.q_ncar[q]} : ShareMax{0, 0, 0};
}
// rows reported (all regions), newly covered sites, their arena entries, private rows (regions under the duplicate rule), such
// regions, runs (maximal stretches of covered sites: a region starts one when no earlier region reaches its first site)
struct Scan5 { uint64_t a, u, c, p, s, r; };
__device__ __forceinline__ Scan5 operator+(Scan5 x, Scan5 y) { return Scan5{x.a + y.a, x.u + y.u, x.c + y.c, x.p + y.p, x.s + y.s, x.r + y.r}; }
__device__ __forceinline__ Scan5 wave_shfl_up5(Scan5 v, int d) {
  return Scan5{__shfl_up(v.a, d, 64), __shfl_up(v.u, d, 64), __shfl_up(v.c, d, 64), __shfl_up(v.p, d, 64), __shfl_up(v.s, d, 64), __shfl_up(v.r, d, 64)};
}
__device__ __forceinline__ Scan5 wave_shfl_xor5(Scan5 v, int d) {
  return Scan5{__shfl_xor(v.a, d, 64), __shfl_xor(v.u, d, 64), __shfl_xor(v.c, d, 64), __shfl_xor(v.p, d, 64), __shfl_xor(v.s, d, 64), __shfl_xor(v.r, d, 64)};
}
__device__ __forceinline__ Scan5 block_exclusive_scan5(Scan5 v, Scan5* total) {
  __shared__ Scan5 wsum[kPlanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  Scan5 incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const Scan5 t = wave_shfl_up5(incl, d);
    if (lane >= d) incl = incl + t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  Scan5 woff{0, 0, 0, 0, 0, 0}, tot{0, 0, 0, 0, 0, 0};
  for (int w = 0; w < kPlanBlock / 64; ++w) {
    if (w < wid) woff = woff + wsum[w];
    tot = tot + wsum[w];
  }
  __syncthreads();
  *total = tot;
  return Scan5{woff.a + incl.a - v.a, woff.u + incl.u - v.u, woff.c + incl.c - v.c, woff.p + incl.p - v.p, woff.s + incl.s - v.s, woff.r + incl.r - v.r};
}
// what region q adds to the batch, given the largest site end before it
struct ShareNew { uint32_t ns; uint64_t n_new, arena_new, back, rback, pre_ns; };
// c0 = s_carpre[g0], npad = s_carpre[g1] - c0, ce = s_carpre[e_prev]: the first site the region is the first to cover is g0, e_prev or g1
__device__ __forceinline__ ShareNew share_new(uint32_t g0, uint32_t nv, uint32_t e_prev, uint64_t c0, uint64_t npad, uint64_t ce) {
  ShareNew o{g0, 0, 0, 0, 0, c0};
  if (!nv) return o;
  const uint32_t g1 = g0 + nv;
  const uint64_t c1 = c0 + npad;
  if (e_prev <= g0) { o.ns = g0; o.pre_ns = c0; }
  else if (e_prev < g1) { o.ns = e_prev; o.pre_ns = ce; }
  else { o.ns = g1; o.pre_ns = c1; }
  o.n_new = g1 - o.ns;
  o.arena_new = c1 - o.pre_ns;
  o.back = e_prev > g0 ? ce - c0 : 0;        // arena distance from site g0 to where the covered ground ends
  o.rback = e_prev > g0 ? e_prev - g0 : 0;   // the same in rows
  return o;
}

// What the plan leaves in mapped host memory: the host sizes the table and the arena from it and launches the rest.
struct PlanTotals {
  uint64_t rows;        // rows of the variant table: shared + private
  uint64_t arena;       // arena entries
  uint64_t shared_rows; // U: sites the batch covers
  uint64_t not_sorted;  // the regions were not sorted by first site: nothing below is meaningful
  uint64_t reported;    // rows over all regions (a shared row once per region reporting it)
  uint64_t n_slow;      // regions under the duplicate rule
  uint64_t n_runs;      // maximal stretches of covered sites (RunRec)
  uint64_t seq;         // written last, system-scope release: the host spins on it
};

// A SPECULATIVE batch (round 6: its table and arena were sized from the handle's previous batch and nothing waits for the plan's totals
// on the host) hands the totals to the kernels behind the plan in device memory: k_t6_totals writes this record, k_t6_slow and
// k_fill_sites2 read it instead of kernel arguments -- and return at once when the plan REFUSED the batch (not sorted, or rows or arena
// beyond what was allocated): the host learns that from the mailbox the first time it asks for the result's sizes and redoes the batch
// with the exact sizes (engine.hip: result_sizes), as the walking batches do (WalkAdmit).
struct PlanDev { uint64_t U, n_runs, n_slow; uint32_t refused, pad_; };

// SRC 0: bounds from (x, y); 1: from gathered records (k_bounds_from_records); 2: the per-region arrays are already there
// (a batch sorted on the device works on permuted copies of them)
struct RecordBounds { uint32_t g0, nv; uint8_t fl; uint64_t npad, pre0, nkept; };
__device__ __forceinline__ RecordBounds record_bounds(const DevImage& im, const uint64_t* recs, uint64_t q) {
  const uint64_t w1 = recs[4 * q + 1], w2 = recs[4 * q + 2];
  uint32_t g0 = (uint32_t)w1, nsites = (uint32_t)w2;
  uint8_t fl = (uint8_t)((w1 >> 32) & (kRegionEmpty | kRegionInvalid | kRegionNotFound | kRegionEndless));
  if ((uint64_t)g0 + nsites > im.G) { g0 = 0; nsites = 0; fl = kRegionInvalid; }
  if ((w1 >> 40) & 1) fl |= kRegionSlow;   // the producing rank dropped rows: the literal rule runs again here
  const uint64_t pre0 = im.s_carpre[g0];
  return RecordBounds{g0, nsites, fl, im.s_carpre[g0 + nsites] - pre0, pre0, im.s_kpre[g0 + nsites] - im.s_kpre[g0]};
}
// `recs`: SRC 1 the gathered records; SRC 0 the caller's regions when they are in device memory (NULL: already copied into
// the result) -- the kernel that reads them anyway keeps the result's copy, and block 0 clears the plan's status word:
// two operations fewer on the plan's stream than a copy and a memset in front of it (each is a launch: ~5 us on either side).
template <int SRC>
__device__ __forceinline__ void plan_bounds(const DevImage& im, const DevResult& r, const uint64_t* recs, uint32_t items, ShareMax* tile_max, uint32_t* status) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *status = 0;
  const uint64_t base = ((uint64_t)blockIdx.x * kPlanBlock + threadIdx.x) * items;
  ShareMax m{0, 0, 0};
  for (uint32_t i = 0; i < items; ++i) {
    const uint64_t q = base + i;
    if (q >= r.Q) break;
    uint32_t g0, nv;
    uint64_t c1;
    if (SRC == 0) {
      uint64_t x, y;
      if (recs) {
        const ulonglong2 xy = *reinterpret_cast<const ulonglong2*>(recs + 2 * q);
        x = xy.x; y = xy.y;
        *reinterpret_cast<ulonglong2*>(const_cast<uint64_t*>(r.regions) + 2 * q) = xy;
      } else { x = r.regions[2 * q]; y = r.regions[2 * q + 1]; }
      const RegionBounds b = region_bounds_of(im, x, y);
      g0 = b.g0; nv = b.g1 - b.g0; c1 = b.pre0 + b.npad;
      r.q_flags[q] = b.flags; r.q_g0[q] = g0; r.q_nvar[q] = nv; r.q_ncar[q] = b.npad; r.car_base[q] = b.pre0; r.q_car_len[q] = b.nkept;
    } else if (SRC == 1) {
      const RecordBounds b = record_bounds(im, recs, q);
      g0 = b.g0; nv = b.nv; c1 = b.pre0 + b.npad;
      r.q_flags[q] = b.fl; r.q_g0[q] = g0; r.q_nvar[q] = nv; r.q_ncar[q] = b.npad; r.car_base[q] = b.pre0; r.q_car_len[q] = b.nkept;
    } else {
      g0 = r.q_g0[q]; nv = (uint32_t)r.q_nvar[q]; c1 = r.car_base[q] + r.q_ncar[q];
    }
    if (nv) m = smax(m, ShareMax{g0 + nv, g0, c1});
  }
  ShareMax tot;
  block_exclusive_max(m, &tot);
  if (threadIdx.x == 0) tile_max[blockIdx.x] = tot;
}
template <int SRC>
__global__ void __launch_bounds__(kPlanBlock) k_t6_bounds(DevImage im, DevResult r, const uint64_t* recs, uint32_t items, ShareMax* tile_max, uint32_t* status) {
  plan_bounds<SRC>(im, r, recs, items, tile_max, status);
}
// per region: E_prev (kept for the last pass); per tile: the sums
__device__ __forceinline__ void plan_mid(const DevImage& im, const DevResult& r, const ShareMax* tile_max, uint32_t items, uint32_t* e_prev, uint64_t* e_prev_c,
                                         Scan5* tile_sums, uint32_t* status) {
  __shared__ ShareMax red[kPlanBlock / 64];
  ShareMax pm{0, 0, 0};   // the tiles before this one
  for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kPlanBlock) pm = smax(pm, tile_max[t]);
  for (int d = 32; d >= 1; d >>= 1) pm = smax(pm, ShareMax{(uint32_t)__shfl_xor(pm.g1, d, 64), (uint32_t)__shfl_xor(pm.g0, d, 64), (uint64_t)__shfl_xor(pm.c1, d, 64)});
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pm;
  __syncthreads();
  pm = smax(smax(red[0], red[1]), smax(red[2], red[3]));
  const uint64_t base = ((uint64_t)blockIdx.x * kPlanBlock + threadIdx.x) * items;
  ShareMax m{0, 0, 0};
  for (uint32_t i = 0; i < items && base + i < r.Q; ++i) m = smax(m, share_elem(r, base + i));
  ShareMax tot;
  ShareMax ex = smax(block_exclusive_max(m, &tot), pm);
  Scan5 s{0, 0, 0, 0, 0, 0};
  for (uint32_t i = 0; i < items && base + i < r.Q; ++i) {
    const uint64_t q = base + i;
    const ShareMax el = share_elem(r, q);
    const uint32_t nv = el.g1 - el.g0;
    if (nv && el.g0 < ex.g0) *status = 1;          // a region that starts before an earlier one: not sorted
    e_prev[q] = ex.g1; e_prev_c[q] = ex.c1;
    const ShareNew w = share_new(el.g0, nv, ex.g1, el.c1 - r.q_ncar[q], r.q_ncar[q], ex.c1);
    s.a += nv; s.u += w.n_new; s.c += w.arena_new;
    if (r.q_flags[q] & kRegionSlow) { s.p += nv; s.s += 1; }
    if (nv && ex.g1 <= el.g0) s.r += 1;           // no earlier region reaches its first site: a run of covered sites starts here
    ex = smax(ex, el);
  }
  Scan5 t5;
  block_exclusive_scan5(s, &t5);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = t5;
}
__global__ void __launch_bounds__(kPlanBlock) k_t6_mid(DevImage im, DevResult r, const ShareMax* tile_max, uint32_t items, uint32_t* e_prev, uint64_t* e_prev_c,
                                                       Scan5* tile_sums, uint32_t* status) {
  plan_mid(im, r, tile_max, items, e_prev, e_prev_c, tile_sums, status);
}
// Rows of the shared table are in site order, so inside a RUN -- a maximal stretch of covered sites -- row number and
// site index differ by a constant, and so do a list's arena offset and the site table's arena prefix: one record per
// run turns "row u" into "site g and its arena offset" (k_share_rows2, k_fill_sites2): g = u + dg, car_begin =
// s_carpre[g] + dc.  A batch that covers the chromosome has a handful of runs; one of short scattered regions as many
// as regions.  coarse[k] = the run that owns row k * kCoarseRows (written when there are more than 64 runs).
struct RunRec { uint64_t u_start, dg, dc, pad_; };
struct RowDelta { uint64_t dg, dc; };
constexpr uint32_t kCoarseRows = 64;
// what k_t6_apply scans per region (the rows reported are the totals' business alone): 8 registers where Scan5 takes 12
struct Scan4 { uint64_t u, c, p; uint32_t s, r; };
__device__ __forceinline__ Scan4 block_exclusive_scan4(Scan4 v) {
  __shared__ Scan4 wsum[kPlanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  Scan4 incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t tu = __shfl_up(incl.u, d, 64), tc = __shfl_up(incl.c, d, 64), tp = __shfl_up(incl.p, d, 64);
    const uint32_t ts = __shfl_up(incl.s, d, 64), tr = __shfl_up(incl.r, d, 64);
    if (lane >= d) { incl.u += tu; incl.c += tc; incl.p += tp; incl.s += ts; incl.r += tr; }
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  Scan4 o{incl.u - v.u, incl.c - v.c, incl.p - v.p, incl.s - v.s, incl.r - v.r};
  for (int w = 0; w < kPlanBlock / 64 - 1; ++w)
    if (w < wid) { const Scan4 t = wsum[w]; o.u += t.u; o.c += t.c; o.p += t.p; o.s += t.s; o.r += t.r; }
  return o;
}
// tile_pre: k_t6_totals' exclusive prefix of the tile sums, the batch's totals at [ntiles] -- a block reads its own entry and the
// last one (two uniform loads) where round 4's kernel reduced all the tiles by itself in twenty-four registers.
template <bool RESIDENT>
__device__ __forceinline__ void plan_apply(const DevImage& im, const DevResult& r, const uint32_t* e_prev, const uint64_t* e_prev_c, const Scan5* tile_pre, uint32_t ntiles, uint32_t items,
                                           RunRec* runs, uint32_t* coarse, uint32_t* slow_list,
                                           const uint32_t* status, uint64_t resident_entries) {
  const uint64_t U = tile_pre[ntiles].u, all_p = tile_pre[ntiles].p, all_c = tile_pre[ntiles].c;
  const bool many_runs = tile_pre[ntiles].r > 64;
  const uint64_t base = ((uint64_t)blockIdx.x * kPlanBlock + threadIdx.x) * items;
  Scan4 s{0, 0, 0, 0, 0};
  for (uint32_t i = 0; i < items && base + i < r.Q; ++i) {
    const uint64_t q = base + i;
    const uint32_t nv = (uint32_t)r.q_nvar[q], g0 = r.q_g0[q], ep = e_prev[q];
    const ShareNew w = share_new(g0, nv, ep, r.car_base[q], r.q_ncar[q], e_prev_c[q]);
    s.u += w.n_new; s.c += w.arena_new;
    if (r.q_flags[q] & kRegionSlow) { s.p += nv; s.s += 1; }
    if (nv && ep <= g0) s.r += 1;
  }
  Scan4 ex = block_exclusive_scan4(s);
  {
    const Scan5 pre = tile_pre[blockIdx.x];
    ex.u += pre.u; ex.c += pre.c; ex.p += pre.p; ex.s += (uint32_t)pre.s; ex.r += (uint32_t)pre.r;
  }
  const bool sorted = *status == 0;   // a batch that is not sorted is planned again after the sort: its per-region inputs stay as the bounds left them
  for (uint32_t i = 0; sorted && i < items && base + i < r.Q; ++i) {
    const uint64_t q = base + i;
    const uint32_t nv = (uint32_t)r.q_nvar[q], g0 = r.q_g0[q], ep = e_prev[q];
    const bool slow = (r.q_flags[q] & kRegionSlow) != 0;
    const uint64_t c0 = r.car_base[q], npad = r.q_ncar[q], nkept = r.q_car_len[q];   // (what the bounds left: overwritten with the final values below)
    const ShareNew w = share_new(g0, nv, ep, c0, npad, e_prev_c[q]);   // (worked out again: registers held across the scan are not free)
    const bool run_start = nv && ep <= g0;
    if (run_start) runs[ex.r] = RunRec{ex.u, (uint64_t)g0 - ex.u, RESIDENT ? 0 : ex.c - w.pre_ns, 0};   // (a run starts at the region's first site: ns == g0)
    if (many_runs && w.n_new) {   // the region's own run, for every kCoarseRows-th row it is the first to cover
      const uint32_t rid = ex.r + (run_start ? 1u : 0u) - 1u;
      for (uint64_t m = (ex.u + kCoarseRows - 1) & ~(uint64_t)(kCoarseRows - 1); m < ex.u + w.n_new; m += kCoarseRows) coarse[m / kCoarseRows] = rid;
    }
    // a region's rows: its range of the shared table -- or, under the duplicate rule (its drops are its own), a private copy behind it
    r.var_begin[q] = slow ? U + ex.p : ex.u - w.rback;
    if (RESIDENT) {   // the lists ARE the index's arena: a site's list lies at s_carpre[g] (s_carpre[0] == 0: a region without sites)
      r.car_base[q] = nv ? c0 : 0;
      r.q_car_len[q] = nv ? npad : 0;
    } else {
      r.car_base[q] = ex.c - w.back;
      r.q_car_len[q] = npad;                         // the region's own padded arena extent
    }
    if (slow) slow_list[ex.s] = (uint32_t)q;
    else {                                           // (dedup_region sets these for the others)
      r.var_count[q] = nv;
      r.q_ncar[q] = nkept;
    }
    ex.u += w.n_new; ex.c += w.arena_new;
    if (slow) { ex.p += nv; ex.s += 1; }
    if (run_start) ex.r += 1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint64_t arena = RESIDENT ? resident_entries : all_c;
    r.var_begin[r.Q] = U + all_p; r.car_base[r.Q] = arena;
  }
}
// ONE block right behind k_t6_mid (round 5): the tile sums become their exclusive prefix in place (the batch's totals at
// [ntiles]) and the totals go to the host, which sizes table and arena and enqueues the expansion while k_t6_apply writes the
// per-region arrays nobody on the host is waiting for.
__global__ void __launch_bounds__(kPlanBlock) k_t6_totals(Scan5* tile_sums, uint32_t ntiles, const uint32_t* status, PlanTotals* totals_host, uint64_t seq,
                                                          uint64_t resident_entries, uint32_t resident, PlanDev* pd, uint64_t cap_rows, uint64_t cap_arena) {
  Scan5 carry{0, 0, 0, 0, 0, 0};
  for (uint32_t t0 = 0; t0 < ntiles; t0 += kPlanBlock) {
    const uint32_t t = t0 + threadIdx.x;
    const Scan5 v = t < ntiles ? tile_sums[t] : Scan5{0, 0, 0, 0, 0, 0};
    Scan5 tot;
    const Scan5 ex = block_exclusive_scan5(v, &tot);
    if (t < ntiles) tile_sums[t] = carry + ex;
    carry = carry + tot;
  }
  if (threadIdx.x == 0) {
    tile_sums[ntiles] = carry;
    totals_host->rows = carry.u + carry.p; totals_host->arena = resident ? resident_entries : carry.c; totals_host->shared_rows = carry.u;
    totals_host->not_sorted = *status;
    totals_host->reported = carry.a; totals_host->n_slow = carry.s; totals_host->n_runs = carry.r;
    if (pd) {   // speculative batch: the verdict and what the kernels behind the plan need, on the device
      const uint64_t arena = resident ? resident_entries : carry.c;
      pd->U = carry.u; pd->n_runs = carry.r; pd->n_slow = carry.s;
      pd->refused = (*status || carry.u + carry.p > cap_rows || arena > cap_arena) ? 1u : 0u;
    }
    __hip_atomic_store(&totals_host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
template <bool RESIDENT>
__global__ void __launch_bounds__(kPlanBlock) k_t6_apply(DevImage im, DevResult r, const uint32_t* e_prev, const uint64_t* e_prev_c, const Scan5* tile_pre, uint32_t ntiles,
                                                         uint32_t items, RunRec* runs, uint32_t* coarse, uint32_t* slow_list,
                                                         const uint32_t* status, uint64_t resident_entries) {
  plan_apply<RESIDENT>(im, r, e_prev, e_prev_c, tile_pre, ntiles, items, runs, coarse, slow_list, status, resident_entries);
}
// (The three steps as ONE launch with hand-made grid barriers between them -- 512 resident blocks, a growing counter,
//  agent-scope release / acquire around it -- was built and measured in round 4: 0.19 ms against 0.05 ms for the three
//  launches.  A barrier across the eight XCDs costs an L2 write-back and invalidate per block: ~70 us each, an order
//  of magnitude more than the launch gap it replaces.  Removed.)
// end of a batch: one word into mapped host memory behind everything else on the stream (the host spins on it: the
// runtime's completion wait costs tens of microseconds more)
__global__ void k_post_done(uint64_t* flag, uint64_t seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the same with up to three device words for the host (a scan's grand total, an overflow flag): dst[0..2] = *a, *b, *c
// (NULL: 0), dst[3] = seq last -- instead of a staged device-to-host copy and a stream synchronisation
__global__ void k_post_words(uint64_t* dst, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t seq) {
  dst[0] = a ? *a : 0; dst[1] = b ? *b : 0; dst[2] = c ? *c : 0;
  __hip_atomic_store(&dst[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// A batch that is NOT sorted by first site is sorted here -- a counting sort over the site index: histogram of the first
// sites (regions without sites count as site 0), exclusive scan over the G + 1 buckets (the engine's scan kernels),
// scatter through per-bucket cursors -- and then runs through the same kernels on sorted copies of its per-region
// arrays; k_permute_out hands every region's outcome back to its place in the caller's order.  Rows and lists are
// shared either way: the table is in site order whatever the order of the regions.
__device__ __forceinline__ uint32_t sort_key(const DevResult& r, uint64_t q) { return r.q_nvar[q] ? r.q_g0[q] : 0u; }
__global__ void __launch_bounds__(256) k_sort_hist(DevResult r, uint32_t* count) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < r.Q) atomicAdd(&count[sort_key(r, q)], 1u);
}
__global__ void __launch_bounds__(256) k_sort_scatter(DevResult r, unsigned long long* cursor, uint32_t* perm) {   // cursor: the scanned histogram, consumed
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < r.Q) perm[atomicAdd(&cursor[sort_key(r, q)], 1ull)] = (uint32_t)q;
}
// s.X[i] = r.X[perm[i]] for what the bounds kernel left per region
__global__ void __launch_bounds__(256) k_permute_in(DevResult r, DevResult s, const uint32_t* perm, uint64_t* s_regions) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= r.Q) return;
  const uint32_t q = perm[i];
  s_regions[2 * i] = r.regions[2 * q]; s_regions[2 * i + 1] = r.regions[2 * q + 1];
  s.q_flags[i] = r.q_flags[q]; s.q_g0[i] = r.q_g0[q]; s.q_nvar[i] = r.q_nvar[q]; s.q_ncar[i] = r.q_ncar[q];
  s.car_base[i] = r.car_base[q]; s.q_car_len[i] = r.q_car_len[q];   // (what the bounds left for the plan: share_elem)
}
// r.X[perm[i]] = s.X[i] for what the batch computed per region (s: the sorted working copy, r: the caller's order)
__global__ void __launch_bounds__(256) k_permute_out(DevResult s, DevResult r, const uint32_t* perm) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == r.Q) { r.var_begin[i] = s.var_begin[i]; r.car_base[i] = s.car_base[i]; }
  if (i >= r.Q) return;
  const uint32_t q = perm[i];
  r.q_flags[q] = s.q_flags[i]; r.q_ncar[q] = s.q_ncar[i]; r.var_begin[q] = s.var_begin[i]; r.car_base[q] = s.car_base[i];
  r.q_car_len[q] = s.q_car_len[i]; r.var_count[q] = s.var_count[i];
}

// Resident carrier lists, private rows: a region's lists ARE the arena range of its sites -- car_base = s_carpre[g0]
// whatever the scans made of it.
__global__ void __launch_bounds__(256) k_resident_bases(DevImage im, DevResult r, uint64_t arena_entries) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q == r.Q) r.car_base[q] = arena_entries;
  if (q >= r.Q) return;
  const uint64_t n = r.q_nvar[q];
  const uint32_t g0 = n ? r.q_g0[q] : 0u;
  const uint64_t pre = im.s_carpre[g0];
  r.car_base[q] = pre;
  r.q_car_len[q] = im.s_carpre[g0 + n] - pre;
}

// ---------------------------------------------------------------------------
// The shared rows, ROW-CENTRIC (round 4): a lane per row of the shared table.  With the record of the run a row lies
// in, the row is ONE 32-byte load of the static site row (DevImage::s_row: the row as a resident list would have
// it), one add, one 32-byte store.  The time follows the rows, not the regions (round 3: one wave per region, eight
// SoA reads per row).
//
// shared_row_run: the lanes load 64 run records together -- all of them when the batch has at most 64 runs (one load,
// the same addresses in every wave: cache hits), else the 64 from the run that owns the kCoarseRows-row block of the
// wave's first row on (coarse index) -- and every lane counts the runs that start at or below its own row by bisection
// over the lanes (six ds_bpermute).  A lane whose row lies beyond those 64 runs bisects the run table on its own.
// ---------------------------------------------------------------------------
__device__ __forceinline__ RowDelta shared_row_run(const RunRec* __restrict__ runs, const uint32_t* __restrict__ coarse, uint64_t n_runs, uint64_t u_first,
                                                   uint32_t lane, uint32_t row_in_wave) {
  const uint64_t base = n_runs <= 64 ? 0 : coarse[u_first / kCoarseRows];   // runs[base].u_start <= u_first
  const uint64_t i = base + lane;
  uint4 a{~0u, ~0u, 0, 0}, b{0, 0, 0, 0};
  if (i < n_runs) { const uint4* p = reinterpret_cast<const uint4*>(runs + i); a = p[0]; b = p[1]; }
  const uint64_t u_start = ((uint64_t)a.y << 32) | a.x;   // (~0 beyond the table)
  const uint64_t me = u_first + row_in_wave;
  uint32_t cnt = 0;                                       // runs of the window that start at or below my row, among the first 63
#pragma unroll
  for (uint32_t step = 32; step; step >>= 1) {
    const uint64_t v = __shfl(u_start, (int)(cnt + step - 1), 64);
    if (v <= me) cnt += step;
  }
  // (the 64th record is fetched by EVERY lane before the test: a shuffle under `cnt == 63 && ...` runs with only those lanes active, and a
  //  lane that is switched off -- lane 63 itself, whose row in a task of fewer than 64 rows is the task's first -- hands over 0, which made
  //  the 64th run "start at or below" any row: rows of the 64th run's site for the last row of 63 single-row runs.  Found in round 6 by the
  //  VCF-text check on thousands of 7-base regions; batches of long overlapping regions have a handful of runs and never came here.)
  const uint64_t u_start_63 = __shfl(u_start, 63, 64);
  if (cnt == 63 && u_start_63 <= me) cnt = 64;
  const int src = cnt ? (int)cnt - 1 : 0;
  RowDelta d;
  d.dg = ((uint64_t)(uint32_t)__shfl((int)a.w, src, 64) << 32) | (uint32_t)__shfl((int)a.z, src, 64);
  d.dc = ((uint64_t)(uint32_t)__shfl((int)b.y, src, 64) << 32) | (uint32_t)__shfl((int)b.x, src, 64);
  if (cnt == 64 && base + 64 < n_runs) {                  // maybe beyond the window (many short runs): plain bisection
    uint64_t l2 = base + 63, h2 = n_runs;                 // runs[l2].u_start <= me < runs[h2].u_start (h2 == n_runs: +inf)
    while (h2 - l2 > 1) {
      const uint64_t m = (l2 + h2) >> 1;
      if (runs[m].u_start <= me) l2 = m; else h2 = m;
    }
    d.dg = runs[l2].dg; d.dc = runs[l2].dc;
  }
  return d;
}
// rows only (resident carrier lists; results whose expansion runs on another stream): u_site receives the site of every row
__global__ void __launch_bounds__(256) k_share_rows2(DevImage im, DevResult r, const RunRec* __restrict__ runs, const uint32_t* __restrict__ coarse, uint64_t n_runs,
                                                     uint64_t U, uint32_t* u_site) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t u_first = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
  if (u_first >= U) return;
  const RowDelta d = shared_row_run(runs, coarse, n_runs, u_first, lane, lane);
  const uint64_t u = u_first + lane;
  if (u >= U) return;
  const uint32_t g = (uint32_t)(u + d.dg);
  const uint4* src = reinterpret_cast<const uint4*>(im.s_row + g);
  uint4 x = src[0], y = src[1];
  const uint64_t cb = (((uint64_t)y.w << 32) | y.z) + d.dc;
  y.z = (uint32_t)cb; y.w = (uint32_t)(cb >> 32);
  uint4* dst = reinterpret_cast<uint4*>(r.rows + u);
  dst[0] = x; dst[1] = y;
  if (u_site) u_site[u] = g;
}

// The reference's "only add var if not seen before" rule (query.h:397-414),
// literally, for the regions flagged by k_region_bounds.  One thread per region.
__device__ __forceinline__ void dedup_region(const DevImage& im, const DevResult& r, uint64_t q) {
  const uint64_t a0 = r.var_begin[q], n = r.q_nvar[q];
  uint64_t kept = 0, back = 0, kept_car = 0;
  VariantRow vb{};   // the last row kept (vars.back())
  for (uint64_t j = 0; j < n; ++j) {
    const uint64_t a = a0 + j;
    VariantRow v = row_load(r.rows, a);
    if (row_dropped(v)) { if (row_count(v)) { v.count_flags = kRowDropped; r.rows[a].count_flags = v.count_flags; } continue; }
    const uint64_t p = v.pos;
    const uint32_t ao = v.alt_off, al = v.alt_len;
    bool push = true;
    if (kept >= 1) {
      const bool same_back = vb.pos == p && vb.alt_len == al && seq_equal(im, vb.alt_off, ao, al);
      if (same_back) push = false;
      else if (kept > 1 && vb.pos == p) {
        for (uint64_t i = back + 1; i-- > a0;) {
          const VariantRow w = row_load(r.rows, i);
          if (row_dropped(w)) continue;
          if (w.pos < p) break;
          if (w.pos == p && w.alt_len == al && seq_equal(im, w.alt_off, ao, al)) { push = false; break; }
        }
      }
    }
    if (push) { kept++; back = a; vb = v; kept_car += row_count(v); }
    else r.rows[a].count_flags = kRowDropped;   // dropped: no carriers reported
  }
  r.var_count[q] = kept;
  r.q_ncar[q] = kept_car;
}

__global__ void __launch_bounds__(64) k_dedup_slow(DevImage im, DevResult r) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q || !(r.q_flags[q] & kRegionSlow)) return;
  dedup_region(im, r, q);
}

// Regions under the duplicate rule: a private copy of their rows behind the shared table (their drops are their own),
// then the literal rule.  One wave per such region, from the list k_t6_apply left.
__global__ void __launch_bounds__(256) k_t6_slow(DevImage im, DevResult r, const uint32_t* slow_list, uint64_t n, const PlanDev* pd) {
  if (pd) {   // speculative batch: the count from the plan's record; nothing to do for a batch the plan refused
    if (pd->refused) return;
    n = pd->n_slow;
  }
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t i = wave; i < n; i += nwaves) {
    const uint64_t q = slow_list[i];
    emit_region<false>(im, r, q, lane);
    __threadfence();   // the rows of the other lanes, before lane 0 reads them back
    if (lane == 0) dedup_region(im, r, q);
  }
}

}  // namespace vsamd
