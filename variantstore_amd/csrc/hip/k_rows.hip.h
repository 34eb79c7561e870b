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
  const uint64_t pre0 = im.s_carpre[g0];
  uint32_t kept = 0;
  for (uint64_t j = lane; j < n; j += 64) {
    const uint64_t a = a0 + j;
    const uint32_t g = g0 + (uint32_t)j;
    const uint32_t cnt = im.s_ncar[g];
    kept += cnt;
    row_store(r.rows, a, im.s_pos[g], im.s_ref_off[g], im.s_ref_len[g], im.s_alt_off[g], im.s_alt_len[g], cnt,
              (im.s_flags[g] & kSiteAlwaysDrop) != 0, cb + (im.s_carpre[g] - pre0));
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
// regions with any site; a batch that is not reports so (status) and takes the private-list path.
// ---------------------------------------------------------------------------
// (the scans over the regions keep 2 items per thread: their per-item work is a chain of dependent site-table reads, and
//  100 k regions in tiles of 2048 would be 49 blocks on a 256-CU part)
constexpr int kShareItems = 2, kShareTile = kScanBlock * kShareItems;
struct ShareMax { uint32_t g1, g0; };
__device__ __forceinline__ ShareMax smax(ShareMax a, ShareMax b) { return ShareMax{a.g1 > b.g1 ? a.g1 : b.g1, a.g0 > b.g0 ? a.g0 : b.g0}; }
__device__ __forceinline__ ShareMax block_exclusive_max(ShareMax v, ShareMax* total) {
  __shared__ ShareMax wmx[kScanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  ShareMax incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t a = __shfl_up(incl.g1, d, 64), b = __shfl_up(incl.g0, d, 64);
    if (lane >= d) incl = smax(incl, ShareMax{a, b});
  }
  if (lane == 63) wmx[wid] = incl;
  __syncthreads();
  ShareMax woff{0, 0}, tot{0, 0};
  for (int w = 0; w < kScanBlock / 64; ++w) {
    if (w < wid) woff = smax(woff, wmx[w]);
    tot = smax(tot, wmx[w]);
  }
  __syncthreads();
  *total = tot;
  const uint32_t pa = __shfl_up(incl.g1, 1, 64), pb = __shfl_up(incl.g0, 1, 64);
  return lane ? smax(woff, ShareMax{pa, pb}) : woff;
}
__device__ __forceinline__ ShareMax share_elem(const DevResult& r, uint64_t q) {   // {end, start} of a region's site range; {0, 0} without sites
  const uint32_t nv = (uint32_t)r.q_nvar[q];
  return nv ? ShareMax{r.q_g0[q] + nv, r.q_g0[q]} : ShareMax{0, 0};
}
__global__ void __launch_bounds__(kScanBlock) k_share_tile_max(DevResult r, ShareMax* tile_max) {
  const uint64_t base = (uint64_t)blockIdx.x * kShareTile + (uint64_t)threadIdx.x * kShareItems;
  ShareMax m{0, 0};
  for (int i = 0; i < kShareItems; ++i)
    if (base + i < r.Q) m = smax(m, share_elem(r, base + i));
  ShareMax tot;
  block_exclusive_max(m, &tot);
  if (threadIdx.x == 0) tile_max[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(kScanBlock) k_share_spine_max(ShareMax* tile_max, uint64_t ntiles) {   // exclusive prefix max, in place
  ShareMax carry{0, 0};
  for (uint64_t base = 0; base < ntiles; base += kScanBlock) {
    const uint64_t i = base + threadIdx.x;
    const ShareMax v = i < ntiles ? tile_max[i] : ShareMax{0, 0};
    ShareMax tot;
    const ShareMax ex = block_exclusive_max(v, &tot);
    if (i < ntiles) tile_max[i] = smax(carry, ex);
    carry = smax(carry, tot);
  }
}
struct Scan4 { uint64_t a, u, c, p; };   // rows reported (all regions), newly covered sites, their arena entries, private rows (regions under the duplicate rule)
__device__ __forceinline__ Scan4 block_exclusive_scan4(Scan4 v, Scan4* total) {
  __shared__ Scan4 wsum[kScanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  Scan4 incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t ta = __shfl_up(incl.a, d, 64), tu = __shfl_up(incl.u, d, 64), tc = __shfl_up(incl.c, d, 64), tp = __shfl_up(incl.p, d, 64);
    if (lane >= d) { incl.a += ta; incl.u += tu; incl.c += tc; incl.p += tp; }
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  Scan4 woff{0, 0, 0, 0}, tot{0, 0, 0, 0};
  for (int w = 0; w < kScanBlock / 64; ++w) {
    if (w < wid) { woff.a += wsum[w].a; woff.u += wsum[w].u; woff.c += wsum[w].c; woff.p += wsum[w].p; }
    tot.a += wsum[w].a; tot.u += wsum[w].u; tot.c += wsum[w].c; tot.p += wsum[w].p;
  }
  __syncthreads();
  *total = tot;
  return Scan4{woff.a + incl.a - v.a, woff.u + incl.u - v.u, woff.c + incl.c - v.c, woff.p + incl.p - v.p};
}
// what region q adds to the batch, given the largest site end before it
struct ShareNew { uint32_t ns; uint64_t n_new, arena_new, back, rback; };
__device__ __forceinline__ ShareNew share_new(const DevImage& im, uint32_t g0, uint32_t nv, uint32_t e_prev) {
  ShareNew o{g0, 0, 0, 0, 0};
  if (!nv) return o;
  const uint32_t g1 = g0 + nv;
  o.ns = e_prev > g0 ? (e_prev < g1 ? e_prev : g1) : g0;
  o.n_new = g1 - o.ns;
  const uint64_t c0 = im.s_carpre[g0];
  o.arena_new = im.s_carpre[g1] - im.s_carpre[o.ns];
  o.back = e_prev > g0 ? im.s_carpre[e_prev] - c0 : 0;   // arena distance from site g0 to where the covered ground ends
  o.rback = e_prev > g0 ? e_prev - g0 : 0;               // the same in rows
  return o;
}
// per element: E_prev (kept for the last pass) and the tile sums
__global__ void __launch_bounds__(kScanBlock) k_share_mid(DevImage im, DevResult r, const ShareMax* tile_max, uint32_t* e_prev, Scan4* tile_sums,
                                                         uint32_t* status) {
  const uint64_t base = (uint64_t)blockIdx.x * kShareTile + (uint64_t)threadIdx.x * kShareItems;
  ShareMax loc[kShareItems], m{0, 0};
  for (int i = 0; i < kShareItems; ++i) {
    loc[i] = base + i < r.Q ? share_elem(r, base + i) : ShareMax{0, 0};
    m = smax(m, loc[i]);
  }
  ShareMax tot;
  ShareMax ex = smax(block_exclusive_max(m, &tot), tile_max[blockIdx.x]);
  Scan4 s{0, 0, 0, 0};
  for (int i = 0; i < kShareItems; ++i) {
    if (base + i < r.Q) {
      const uint32_t nv = (uint32_t)r.q_nvar[base + i];
      if (nv && loc[i].g0 < ex.g0) *status = 1;          // a region that starts before an earlier one: not sorted
      e_prev[base + i] = ex.g1;
      const ShareNew w = share_new(im, loc[i].g0, nv, ex.g1);
      s.a += nv; s.u += w.n_new; s.c += w.arena_new;
      if (r.q_flags[base + i] & kRegionSlow) s.p += nv;
    }
    ex = smax(ex, loc[i]);
  }
  Scan4 t4;
  block_exclusive_scan4(s, &t4);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = t4;
}
// totals: {rows of the table (shared + private), arena entries, shared rows, not-sorted flag, rows reported over all regions}
// (totals lie in mapped host memory; totals[5] = seq is written last, with a system-scope release: the host spins on it
//  instead of synchronising the stream)
__global__ void __launch_bounds__(kScanBlock) k_share_spine_sum(Scan4* tile_sums, uint64_t ntiles, DevResult r, uint64_t* u_begin, uint64_t* totals,
                                                               const uint32_t* status, uint64_t seq) {
  Scan4 carry{0, 0, 0, 0};
  for (uint64_t base = 0; base < ntiles; base += kScanBlock) {
    const uint64_t i = base + threadIdx.x;
    const Scan4 v = i < ntiles ? tile_sums[i] : Scan4{0, 0, 0, 0};
    Scan4 tot;
    const Scan4 ex = block_exclusive_scan4(v, &tot);
    if (i < ntiles) tile_sums[i] = Scan4{carry.a + ex.a, carry.u + ex.u, carry.c + ex.c, carry.p + ex.p};
    carry.a += tot.a; carry.u += tot.u; carry.c += tot.c; carry.p += tot.p;
  }
  if (threadIdx.x == 0) {
    r.var_begin[r.Q] = carry.u + carry.p; r.car_base[r.Q] = carry.c; u_begin[r.Q] = carry.u;
    totals[0] = carry.u + carry.p; totals[1] = carry.c; totals[2] = carry.u; totals[3] = *status; totals[4] = carry.a;
    __hip_atomic_store(&totals[5], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void __launch_bounds__(kScanBlock) k_share_apply(DevImage im, DevResult r, const uint32_t* e_prev, const Scan4* tile_sums, uint32_t* new_start,
                                                           uint64_t* u_begin, uint64_t* arena_new) {
  const uint64_t base = (uint64_t)blockIdx.x * kShareTile + (uint64_t)threadIdx.x * kShareItems;
  const uint64_t U = u_begin[r.Q];   // (written by the spine kernel before this launch)
  ShareNew loc[kShareItems];
  uint32_t nvs[kShareItems];
  bool slow[kShareItems];
  Scan4 s{0, 0, 0, 0};
  for (int i = 0; i < kShareItems; ++i) {
    loc[i] = ShareNew{0, 0, 0, 0, 0}; nvs[i] = 0; slow[i] = false;
    if (base + i < r.Q) {
      nvs[i] = (uint32_t)r.q_nvar[base + i];
      slow[i] = (r.q_flags[base + i] & kRegionSlow) != 0;
      loc[i] = share_new(im, r.q_g0[base + i], nvs[i], e_prev[base + i]);
      s.a += nvs[i]; s.u += loc[i].n_new; s.c += loc[i].arena_new; s.p += slow[i] ? nvs[i] : 0;
    }
  }
  Scan4 tot;
  Scan4 ex = block_exclusive_scan4(s, &tot);
  const Scan4 ts = tile_sums[blockIdx.x];
  ex.a += ts.a; ex.u += ts.u; ex.c += ts.c; ex.p += ts.p;
  for (int i = 0; i < kShareItems; ++i) {
    if (base + i < r.Q) {
      const uint64_t q = base + i;
      u_begin[q] = ex.u; arena_new[q] = ex.c; new_start[q] = loc[i].ns;
      // a region's rows: its range of the shared table -- or, under the duplicate rule (its drops are its own), a private copy behind it
      r.var_begin[q] = slow[i] ? U + ex.p : ex.u - loc[i].rback;
      r.car_base[q] = ex.c - loc[i].back;
      r.q_car_len[q] = r.q_ncar[q];                  // the region's own padded arena extent
      if (!slow[i]) {                                // (dedup_region sets these for the others)
        const uint32_t g0 = r.q_g0[q];
        r.var_count[q] = nvs[i];
        r.q_ncar[q] = im.s_kpre[g0 + nvs[i]] - im.s_kpre[g0];
      }
    }
    ex.a += nvs[i]; ex.u += loc[i].n_new; ex.c += loc[i].arena_new; ex.p += slow[i] ? nvs[i] : 0;
  }
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

// Resident carrier lists: a region's lists ARE the arena range of its sites -- car_base = s_carpre[g0] whatever the
// scans made of it (and the start of the region's new part likewise, for k_share_rows).
__global__ void __launch_bounds__(256) k_resident_bases(DevImage im, DevResult r, const uint32_t* new_start, uint64_t* arena_new, uint64_t arena_entries) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q == r.Q) r.car_base[q] = arena_entries;
  if (q >= r.Q) return;
  const uint64_t n = r.q_nvar[q];
  const uint32_t g0 = n ? r.q_g0[q] : 0u;
  const uint64_t pre = im.s_carpre[g0];
  r.car_base[q] = pre;
  r.q_car_len[q] = im.s_carpre[g0 + n] - pre;
  if (arena_new) arena_new[q] = im.s_carpre[new_start[q]];
}

// The shared rows: every region writes the rows of the sites it is the first to cover (one wave per region), and the
// site index beside them for the expansion; regions under the duplicate rule also get their private copy.
__global__ void __launch_bounds__(256) k_share_rows(DevImage im, DevResult r, const uint32_t* new_start, const uint64_t* u_begin, const uint64_t* arena_new,
                                                    uint32_t* u_site) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= r.Q) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t u0 = u_begin[q], n_new = u_begin[q + 1] - u0;
  if (n_new) {
    const uint32_t ns = new_start[q];
    const uint64_t cb0 = arena_new[q], pre = im.s_carpre[ns];
    for (uint64_t j = lane; j < n_new; j += 64) {
      const uint32_t g = ns + (uint32_t)j;
      row_store(r.rows, u0 + j, im.s_pos[g], im.s_ref_off[g], im.s_ref_len[g], im.s_alt_off[g], im.s_alt_len[g], im.s_ncar[g],
                (im.s_flags[g] & kSiteAlwaysDrop) != 0, cb0 + (im.s_carpre[g] - pre));
      u_site[u0 + j] = g;   // (the expansion takes source handle and genotype offset from the site table: writing them here as well cost more than the look-up)
    }
  }
  if (r.q_flags[q] & kRegionSlow) emit_region<false>(im, r, q, lane);
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

}  // namespace vsamd
