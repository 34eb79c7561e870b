// k_sample_coords.hip.h -- sample-coordinate queries (types 2, 3, 5), result totals, batched find.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_walk.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Sample-coordinate queries (types 2, 3 and 5).  They need the per-carrier `index` of
// sample_info (variantgraphvertex.proto:12) -- DevImage::car_index.
// ---------------------------------------------------------------------------

// get_sample_from_vertex_if_exists(v, sample, out) -> out.index (variant_graph.h:1296-1339).  The s_info
// entry of a sample is found by its position in the class's ascending id list (bit-vector mode) or by a
// linear search (explicit ids); the carrier pool holds the non-ref entries in s_info order.
// (ridx, cls: the vertex's ref index and class, which the caller already holds in a walk record)
__device__ __forceinline__ bool sample_entry_rec(const DevImage& im, uint32_t v, uint32_t ridx, uint32_t cls, uint32_t sid, uint32_t& index) {
  if (sid == 0) {
    if (!ridx) return false;
    index = ridx;
    return true;
  }
  if (im.use_bv) {
    const uint64_t* row = im.class_rows + (uint64_t)cls * im.wpc;
    const uint32_t w = sid >> 6, bit = sid & 63;
    const uint64_t word = row[w];
    if (!((word >> bit) & 1)) return false;
    uint32_t rank = __popcll(word & ((1ULL << bit) - 1));
    if (im.class_cum) rank += im.class_cum[(uint64_t)cls * im.wpc + w];
    else for (uint32_t i = 0; i < w; ++i) rank += __popcll(row[i]);
    rank -= (uint32_t)(row[0] & 1);  // the ref entry is not part of the pool
    index = im.car_index[im.v_car_begin[v] + rank];
    return true;
  }
  // (explicit ids: the FIRST record with the sample's id, eight records per round trip -- see BitRow::holds_explicit)
  const uint64_t b = im.v_car_begin[v];
  const uint32_t n = im.v_ncar[v];
  for (uint32_t k = 0; k < n; k += 8) {
    uint4 p, q;
    __builtin_memcpy(&p, im.car_sid + b + k, 16);
    __builtin_memcpy(&q, im.car_sid + b + k + 4, 16);
    const uint32_t m = n - k;
    const uint32_t hits = (uint32_t)(p.x == sid) | ((uint32_t)(p.y == sid && m > 1) << 1) | ((uint32_t)(p.z == sid && m > 2) << 2) |
                          ((uint32_t)(p.w == sid && m > 3) << 3) | ((uint32_t)(q.x == sid && m > 4) << 4) | ((uint32_t)(q.y == sid && m > 5) << 5) |
                          ((uint32_t)(q.z == sid && m > 6) << 6) | ((uint32_t)(q.w == sid && m > 7) << 7);
    if (hits) { index = im.car_index[b + k + (uint32_t)__builtin_ctz(hits)]; return true; }
  }
  return false;
}
__device__ __forceinline__ bool sample_entry(const DevImage& im, uint32_t v, uint32_t sid, uint32_t& index) {
  return sample_entry_rec(im, v, im.v_ridx[v], im.use_bv ? im.v_class[v] : 0u, sid, index);
}

// ones before each word of every class row (DevImage::class_cum), once when an index with sample coordinates is opened
__global__ void __launch_bounds__(256) k_class_cum(const uint64_t* class_rows, uint64_t n_rows, uint32_t wpc, uint16_t* cum) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_rows) return;
  uint32_t acc = 0;
  for (uint32_t w = 0; w < wpc; ++w) { cum[c * wpc + w] = (uint16_t)acc; acc += __popcll(class_rows[c * wpc + w]); }
}

// get_neighbor_vertex (variant_graph.h:1402-1451): first out-neighbour holding the sample, else the ref
// neighbour with the smallest ref index; 0 = none (the path iterator is done)
__device__ __forceinline__ uint32_t next_on_path(const DevImage& im, uint32_t cur, uint32_t sid) {
  uint32_t nxt = 0, min_idx = 0xFFFFFFFFu;
  const uint4 wc = im.w_vertex[2 * (uint64_t)cur];   // {row_begin, degree, ..}
  for (uint32_t e = wc.x; e < wc.x + wc.y; ++e) {
    const WalkEdge ed = walk_edge(im, e);
    if (sid != 0 && record_has_sample(im, ed.nbr, ed.ridx, ed.cls, sid)) return ed.nbr;
    if (ed.ridx && min_idx > ed.ridx) { nxt = ed.nbr; min_idx = ed.ridx; }
  }
  return nxt;
}

// get_prev_vertex_with_sample (query.h:57-113) including the sample-coordinate output
__device__ __forceinline__ uint32_t prev_vertex_with_sample(const DevImage& im, uint64_t pos, uint32_t sid, uint64_t& ref_pos,
                                                            uint64_t& sample_pos) {
  uint64_t rank;  // find(pos, rank), index.h:135-148 (rank is left unset for pos == 0 there: defined as 0)
  if (pos >= im.ref_length) rank = im.R - 1;
  else { const uint32_t k = rank1(im, pos); rank = k == 0 ? 0 : k - 1; }
  uint32_t v_find = 0;
  while (true) {
    const uint32_t v = im.rp_vid[im.rank_to_slot[rank == 0 ? 0 : rank - 1]];  // Index::previous
    if (rank <= 1) { ref_pos = 1; v_find = v; sample_pos = im.v_ridx[v]; break; }
    bool found = false;
    const uint4 wv = im.w_vertex[2 * (uint64_t)v];   // {row_begin, degree, ..}
    for (uint32_t e = wv.x; e < wv.x + wv.y; ++e) {
      const WalkEdge ed = walk_edge(im, e);
      if (ed.ridx) ref_pos = ed.ridx;
      uint32_t idx;
      if (sample_entry_rec(im, ed.nbr, ed.ridx, ed.cls, sid, idx)) { v_find = ed.nbr; found = true; sample_pos = idx; }
      rank = rank ? rank - 1 : 0;  // unsigned wrap in the reference: clamped
    }
    if (found) break;
  }
  return v_find;
}

// the backward search of query.h:213-218 / :507-512; false when the reference would loop forever
__device__ __forceinline__ bool rewind_to_sample_pos(const DevImage& im, uint64_t x, uint32_t sid, uint32_t& closest_v,
                                                     uint64_t& ref_pos, uint64_t& sample_pos) {
  closest_v = prev_vertex_with_sample(im, x, sid, ref_pos, sample_pos);
  uint64_t guard = 0;
  while (sample_pos >= x && closest_v > 0) {
    const uint64_t pos = ref_pos, before_ref = ref_pos, before_sample = sample_pos;
    const uint32_t before_v = closest_v;
    closest_v = prev_vertex_with_sample(im, pos, sid, ref_pos, sample_pos);
    if (ref_pos == before_ref && sample_pos == before_sample && closest_v == before_v) return false;
    if (++guard > 4 * im.V + 64) return false;
  }
  return true;
}

// ---- the same searches with the per-sample event and hold rows of query type 4 (DevImage::t4_events / t4_hold) ----
// get_prev_vertex_with_sample as in walk_start_search<true>: ranks whose node has no out-neighbour holding the sample
// (clear event bit) are counted down without being read; the candidate's edge records come from the walk blob, "holds
// the sample" from the hold row, and only the vertex that is found pays the look-up of its sample-coordinate index.
constexpr uint32_t kSerialHopAfter = 4;   // skipped ranks before a one-lane backward search goes over to the event row's set bits
template <bool WANT_INDEX = true>   // (type 2 never reads sample_pos: no look-up of the found vertex's index)
__device__ __forceinline__ uint32_t prev_vertex_with_sample_ev(const DevImage& im, uint64_t pos, uint32_t sid, BitRow& ev, BitRow& hold,
                                                               uint64_t& ref_pos, uint64_t& sample_pos) {
  uint64_t rank;
  if (pos >= im.ref_length) rank = im.R - 1;
  else { const uint32_t k = rank1(im, pos); rank = k == 0 ? 0 : k - 1; }
  const uint64_t ref_pos_in = ref_pos;
  const uint64_t rank_in = rank;
  bool jumped = false;
  uint32_t skipped = 0;
  while (true) {
    uint2 back = im.rk_back[rank == 0 ? 0 : rank - 1];
    if (rank <= 1) { const uint32_t v = im.rp_vid[back.x]; ref_pos = 1; sample_pos = im.v_ridx[v]; return v; }
    if (!ev.bit(back.x)) {
      rank = rank > back.y ? rank - back.y : 0;
      jumped = true;
      if (++skipped < kSerialHopAfter || rank <= 1 || !im.slot_rank) continue;
      // A long search (a sample with few variants): from here on by the SET BITS of the sample's event row, highest slot
      // first.  A set bit is a candidate when its slot is the first of its rank (the only node of a rank the chain looks
      // at) and that rank is on the chain from the start rank (ancestor labels, DevImage::rk_anc).
      const uint32_t tin0 = im.rk_anc[rank_in - 1].x;
      const uint32_t s_top = im.rk_back[rank - 1].x;
      const uint32_t sh = ev.sh, c_top = s_top >> sh;   // (coarse rows: a bit stands for 2^sh slots, every one of them a candidate)
      uint32_t hit = kNone;
      for (int64_t wi = c_top >> 6; wi >= 0 && hit == kNone; --wi) {
        uint64_t word = ev.row[wi];
        if ((uint32_t)wi == (c_top >> 6) && (c_top & 63) != 63) word &= (1ULL << ((c_top & 63) + 1)) - 1;
        while (word && hit == kNone) {
          const uint32_t bpos = 63u - (uint32_t)__builtin_clzll(word);
          word &= ~(1ULL << bpos);
          const uint32_t c = (uint32_t)wi * 64u + bpos;
          for (uint32_t j = 1u << sh; j-- > 0;) {
            const uint32_t k = (c << sh) + j;
            if (k > s_top || k >= im.P) continue;
            const uint32_t r = im.slot_rank[k];
            if (r < 1) continue;                    // (the chain stops at rank <= 1 before it would look there)
            if (im.rk_back[r].x != k) continue;     // not the first slot of its rank
            const uint2 an = im.rk_anc[r];
            if (!(an.x <= tin0 && tin0 - an.x < an.y)) continue;
            // the literal test of this node; if no neighbour holds the sample the search goes on below it
            bool f2 = false;
            const uint32_t deg = im.rk_back[r].y, rbk = im.blob_of_slot[k] + 1;
            for (uint32_t e = rbk; e < rbk + deg; ++e)
              if (hold.bit_n(im.wblob[2 * (uint64_t)e].x, im.wblob[2 * (uint64_t)e + 1].w)) { f2 = true; break; }
            if (f2) { hit = r; break; }
          }
        }
      }
      if (hit == kNone) { rank = 0; continue; }   // nothing below: the head of the path
      rank = (uint64_t)hit + 1;                    // the chain rank that looks at this node: the loop's literal test takes it from here
      back = im.rk_back[hit];
    }
    bool found = false, had_ref = false;
    uint32_t fv = 0, fr = 0, fc = 0;
    const uint32_t rb0 = im.blob_of_slot[back.x] + 1;   // the edge records follow the slot's header
    for (uint32_t e = rb0; e < rb0 + back.y; ++e) {
      const uint4 a = im.wblob[2 * (uint64_t)e];
      if (a.y) { ref_pos = a.y; had_ref = true; }
      if (hold.bit_n(a.x, im.wblob[2 * (uint64_t)e + 1].w)) { fv = a.x; fr = a.y; fc = a.z; found = true; }
    }
    rank = rank > back.y ? rank - back.y : 0;
    if (found) {
      if (!had_ref && jumped) {   // ref_pos would be an earlier (skipped) node's: the literal search knows
        ref_pos = ref_pos_in;
        return prev_vertex_with_sample(im, pos, sid, ref_pos, sample_pos);
      }
      if (WANT_INDEX) {
        uint32_t idx = 0;
        (void)sample_entry_rec(im, fv, fr, im.use_bv ? fc : 0u, sid, idx);
        sample_pos = idx;
      }
      return fv;
    }
  }
}
__device__ __forceinline__ bool rewind_to_sample_pos_ev(const DevImage& im, uint64_t x, uint32_t sid, BitRow& ev, BitRow& hold, uint32_t& closest_v,
                                                        uint64_t& ref_pos, uint64_t& sample_pos) {
  closest_v = prev_vertex_with_sample_ev(im, x, sid, ev, hold, ref_pos, sample_pos);
  uint64_t guard = 0;
  while (sample_pos >= x && closest_v > 0) {
    const uint64_t pos = ref_pos, before_ref = ref_pos, before_sample = sample_pos;
    const uint32_t before_v = closest_v;
    closest_v = prev_vertex_with_sample_ev(im, pos, sid, ev, hold, ref_pos, sample_pos);
    if (ref_pos == before_ref && sample_pos == before_sample && closest_v == before_v) return false;
    if (++guard > 4 * im.V + 64) return false;
  }
  return true;
}

// Capacities of the single recording walk of type 5: branch sites of the reference range [x, y) widened by the
// region's own length (the sample's coordinates are shifted against the reference's by its net indel length).
__device__ __forceinline__ void walk_caps_sc_region(const DevImage& im, const DevResult& r, uint64_t q) {
  const uint64_t x = r.regions[2 * q], y = r.regions[2 * q + 1];
  const uint64_t margin = (y > x ? y - x : 0) + 256;
  const uint64_t lo = x > margin + 1 ? x - margin : 1, hi = (y > x ? y : x) + margin;
  const uint32_t s0 = slot_of_find(im, lo), s1 = slot_of_find(im, hi);
  r.q_nvar[q] = (s1 >= s0 ? (uint64_t)(im.rp_cand_prefix[s1 + 1] - im.rp_cand_prefix[s0]) : 0) + 8;
}
__global__ void __launch_bounds__(256) k_walk_caps_sc(DevImage im, DevResult r) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < r.Q) walk_caps_sc_region(im, r, q);
}
// The first kernel of a type-4 / type-5 batch whose regions and sample ids arrive in DEVICE memory: the result's copies of both,
// the range check of the ids (`bad`: zero when the batch starts) and the capacities -- one launch where two copies, a
// memset and two kernels stood.  src_ids NULL: one sample for the whole batch, checked by the host.
template <bool SC>
__global__ void __launch_bounds__(256) k_walk_setup(DevImage im, DevResult r, const uint64_t* __restrict__ src_regions, const uint32_t* __restrict__ src_ids,
                                                    uint32_t* dsids, uint32_t num_samples, uint64_t* bad) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  uint64_t* reg = const_cast<uint64_t*>(r.regions);
  reg[2 * q] = src_regions[2 * q]; reg[2 * q + 1] = src_regions[2 * q + 1];
  if (src_ids) {
    const uint32_t sid = src_ids[q];
    dsids[q] = sid;
    if (sid >= num_samples) *bad = 1;
  }
  if (SC) walk_caps_sc_region(im, r, q);
  else region_bounds(im, r, q);
}

// Query type 5.  MODE as in k_sample_walk (0 count, 1 emit, 2 record once).
// SUB lanes per region, all of them running the SAME walk (round 4): a region's walk is one serial chain of dependent
// look-ups, so the only parallelism is across regions -- and with one lane per region a batch of 100 k regions is 1,563
// waves (1.5 per SIMD: nothing hides a memory latency) that each run 64 different chains one after the other (64-way
// divergence).  With SUB = 8 the lanes of a group issue the same addresses (one request), a wave diverges 8 ways, and
// the batch is 12,500 waves: every SIMD holds eight.  Lane 0 of a group writes.
constexpr uint32_t kScGroup = 8;
// One region's walk, start to end, by one lane.  MODE as in k_sample_walk (0 count, 1 emit, 2 record once); `lead`: this lane writes.
template <int MODE>
__device__ __forceinline__ void sc_walk_region(const DevImage& im, const DevResult& r, const WalkScratch& ws, uint64_t q, uint32_t sid, uint64_t x, uint64_t y,
                                               bool lead, uint8_t& fl, uint64_t& nvar, uint64_t& ncar, uint64_t& ncar_kept) {
  constexpr bool EMIT = MODE == 1;
  fl = 0; nvar = 0; ncar = 0; ncar_kept = 0;
  uint64_t ref_pos = 0, sample_pos = 0;
  uint32_t closest_v = 0;
  // With the per-sample event and hold rows of query type 4 (+ the break bits of the sequence queries): the backward
  // searches skip ranks without a holder, the walk jumps over uneventful runs of ref-path slots (nothing is reported
  // there; ref_pos, sample_pos and cur_ref advance with the path), and only a vertex that holds the sample pays the
  // look-up of its sample-coordinate index.
  const bool fast = im.t4_events != nullptr && im.seq_breaks != nullptr && sid != 0;
  BitRow ev = sample_event_row(im, sid, fast);   // the sample's own events: the searches, the walk's jumps
  BitRow hold = sample_hold_row(im, sid, fast);
  BitRow brk{im.seq_breaks, kNone, 0};
  const bool rewound = fast ? rewind_to_sample_pos_ev(im, x, sid, ev, hold, closest_v, ref_pos, sample_pos)
                            : rewind_to_sample_pos(im, x, sid, closest_v, ref_pos, sample_pos);
  if (!rewound) fl = kRegionEndless;
  else if (fast) {
    closest_v = im.rp_vid[slot_of_find(im, ref_pos)];
    if (im.v_ridx[closest_v]) {
      const uint64_t seq_len = ref_pos - im.v_ridx[closest_v];
      ref_pos = im.v_ridx[closest_v];
      sample_pos -= seq_len;
    }
    uint32_t cur = closest_v, cur_ref_v = kNone;
    const uint4 v0 = im.w_vertex[2 * (uint64_t)cur], v1 = im.w_vertex[2 * (uint64_t)cur + 1];
    uint32_t rbeg = im.blob_row[cur], deg = v0.y, ridx = v0.z, len = v1.x, cls = v1.y, ncar_v = v1.z, slot1 = v1.w;
    const uint32_t last_slot = (uint32_t)im.P - 1;
    const uint64_t a0 = EMIT ? r.var_begin[q] : 0;
    const uint64_t cb = EMIT ? r.car_base[q] : 0;
    while (true) {
      if (sample_pos >= y) break;
      if (slot1 && ref_pos == ridx) {
        const uint32_t s0 = slot1 - 1;
        const uint32_t lim = s0 + 1024 < last_slot ? s0 + 1024 : last_slot;
        // (the sample's OWN events.  An irregular slot -- its LAST ref neighbour is not its successor -- sets ref_pos and cur_ref
        //  for the NEXT vertex only, and those are read only if that vertex is reported, i.e. holds the sample: then the slot
        //  has an out-neighbour holding the sample and is an event of its own.  Unlike type 4 the loop stops on sample_pos, which
        //  advances by lengths alone.  What else could break a run -- the smallest-index ref neighbour is not the successor, ref
        //  indexes that do not continue -- is in the break bits.)
        uint32_t k = s0 < lim ? ev.next(s0, lim) : s0;
        if (k > s0) { const uint32_t kb = brk.next_wide(s0, lim); k = kb < k ? kb : k; }
        if (k > s0) {
          const uint64_t h = im.blob_of_slot[k];
          const uint4 ra = im.wblob[2 * h], rb = im.wblob[2 * h + 1];   // header of slot k
          sample_pos += (uint64_t)ra.z - ridx;
          ref_pos = ra.z;
          cur = rb.w; rbeg = ra.x; deg = ra.y; ridx = ra.z; len = rb.x; cls = rb.y; ncar_v = rb.z; slot1 = k + 1;
          cur_ref_v = cur;   // (the last ref neighbour of the node before: not read before this node's step overwrites it)
          continue;
        }
      }
      uint64_t next_ref_pos = ref_pos + len;
      uint32_t next_ref_v = kNone, nxt = 0, min_idx = 0xFFFFFFFFu;
      uint32_t n_rbeg = 0, n_deg = 0, n_ridx = 0, n_len = 0, n_cls = 0, n_ncar = 0, n_slot1 = 0;
      bool by_sample = false;
      for (uint32_t e = rbeg; e < rbeg + deg; ++e) {
        const uint4 a = im.wblob[2 * (uint64_t)e];
        if (a.y) { next_ref_pos = a.y; next_ref_v = a.x; }   // the last ref neighbour
        if (!by_sample) {
          const uint4 b = im.wblob[2 * (uint64_t)e + 1];
          const bool holds = hold.bit_n(a.x, b.w);
          if (holds || (a.y && min_idx > a.y)) {
            nxt = a.x; n_rbeg = a.w; n_deg = b.x; n_ridx = a.y; n_len = b.z; n_cls = a.z; n_ncar = b.w; n_slot1 = b.y;
            if (holds) by_sample = true; else min_idx = a.y;
          }
        }
      }
      uint32_t sidx = 0;
      if (sample_pos > x && hold.bit_n(cur, ncar_v) && sample_entry_rec(im, cur, ridx, im.use_bv ? cls : 0u, sid, sidx)) {
        uint64_t pos;
        uint32_t ro, rl, ao, al;
        if (ref_pos == next_ref_pos) {        // insertion
          pos = ref_pos; ro = 0; rl = 0; ao = im.v_off[cur]; al = len;
        } else if (ridx) {                    // deletion: ref = sequence of find(ref_pos - 1)
          const uint32_t fv = im.rp_vid[slot_of_find(im, ref_pos - 1)];
          pos = sidx; ro = im.v_off[fv]; rl = im.v_len[fv]; ao = 0; al = 0;
        } else {                              // substitution: ref = sequence of the previous step's last ref neighbour
          pos = sidx; ro = 0; rl = 0; ao = im.v_off[cur]; al = len;
          if (cur_ref_v != kNone) { ro = im.v_off[cur_ref_v]; rl = im.v_len[cur_ref_v]; }
        }
        const uint32_t c = ncar_v;
        if (EMIT && lead) {
          const uint64_t a = a0 + nvar;
          row_store(r.rows, a, (uint32_t)pos, ro, rl, ao, al, c, false, cb + ncar);
          r.r_class[a] = im.v_src[cur]; r.r_gt0[a] = im.v_car_begin[cur];
        }
        if (MODE == 2 && lead) {
          const uint64_t s0 = ws.cap_begin[q];
          if (nvar < ws.cap_begin[q + 1] - s0) {
            const uint64_t s = s0 + nvar;
            ws.pos[s] = pos; ws.cur[s] = cur; ws.ro[s] = ro; ws.rl[s] = rl; ws.ao[s] = ao; ws.al[s] = al;
          } else *ws.overflow = 1;
        }
        nvar++; ncar += pad_car(c); ncar_kept += c;
      }
      cur_ref_v = next_ref_v;
      ref_pos = next_ref_pos;
      sample_pos += len;
      if (nxt == 0) break;
      cur = nxt; rbeg = n_rbeg; deg = n_deg; ridx = n_ridx; len = n_len; cls = n_cls; ncar_v = n_ncar; slot1 = n_slot1;
    }
  } else {
    closest_v = im.rp_vid[slot_of_find(im, ref_pos)];
    if (im.v_ridx[closest_v]) {
      const uint64_t seq_len = ref_pos - im.v_ridx[closest_v];
      ref_pos = im.v_ridx[closest_v];
      sample_pos -= seq_len;
    }
    uint32_t cur = closest_v;
    uint32_t cur_ref_v = kNone;
    bool done = false;
    const uint64_t a0 = EMIT ? r.var_begin[q] : 0;
    const uint64_t cb = EMIT ? r.car_base[q] : 0;
    while (!done) {
      if (sample_pos >= y) break;
      const WalkVertex wc = walk_vertex(im, cur);   // one record per vertex, one per neighbour; ONE pass over the edges
      const uint32_t l = wc.len;
      uint64_t next_ref_pos = ref_pos + l;
      uint32_t next_ref_v = kNone;                  // the last ref neighbour (its sequence becomes cur_ref)
      uint32_t nxt = 0, min_idx = 0xFFFFFFFFu;      // get_neighbor_vertex (next_on_path) in the same pass
      bool nxt_by_sample = false;
      for (uint32_t e = wc.row_begin; e < wc.row_begin + wc.deg; ++e) {
        const WalkEdge ed = walk_edge(im, e);
        if (ed.ridx) { next_ref_pos = ed.ridx; next_ref_v = ed.nbr; }
        if (!nxt_by_sample) {
          if (sid != 0 && record_has_sample(im, ed.nbr, ed.ridx, ed.cls, sid)) { nxt = ed.nbr; nxt_by_sample = true; }
          else if (ed.ridx && min_idx > ed.ridx) { nxt = ed.nbr; min_idx = ed.ridx; }
        }
      }
      uint32_t sidx = 0;
      if (sample_pos > x && sample_entry_rec(im, cur, wc.ridx, wc.cls, sid, sidx)) {
        uint64_t pos;
        uint32_t ro, rl, ao, al;
        if (ref_pos == next_ref_pos) {        // insertion
          pos = ref_pos; ro = 0; rl = 0; ao = wc.off; al = l;
        } else if (wc.ridx) {                 // deletion: ref = sequence of find(ref_pos - 1)
          const uint32_t fv = im.rp_vid[slot_of_find(im, ref_pos - 1)];
          pos = sidx; ro = im.v_off[fv]; rl = im.v_len[fv]; ao = 0; al = 0;
        } else {                              // substitution: ref = sequence of the previous step's last ref neighbour
          pos = sidx; ro = 0; rl = 0; ao = wc.off; al = l;
          if (cur_ref_v != kNone) { const WalkVertex wr = walk_vertex(im, cur_ref_v); ro = wr.off; rl = wr.len; }
        }
        const uint32_t c = wc.ncar;
        if (EMIT && lead) {
          const uint64_t a = a0 + nvar;
          row_store(r.rows, a, (uint32_t)pos, ro, rl, ao, al, c, false, cb + ncar);
          r.r_class[a] = im.v_src[cur]; r.r_gt0[a] = im.v_car_begin[cur];
        }
        if (MODE == 2 && lead) {
          const uint64_t s0 = ws.cap_begin[q];
          if (nvar < ws.cap_begin[q + 1] - s0) {
            const uint64_t s = s0 + nvar;
            ws.pos[s] = pos; ws.cur[s] = cur; ws.ro[s] = ro; ws.rl[s] = rl; ws.ao[s] = ao; ws.al[s] = al;
          } else *ws.overflow = 1;
        }
        nvar++; ncar += pad_car(c); ncar_kept += c;
        // the insertion branch clears cur_ref before it is copied into the variant (query.h:564-566)
      }
      cur_ref_v = next_ref_v;
      ref_pos = next_ref_pos;
      sample_pos += l;
      if (nxt == 0) done = true;
      cur = nxt;
    }
  }
}

template <int MODE, uint32_t SUB = kScGroup>
__global__ void __launch_bounds__(SUB > 1 ? 256 : 64) k_sample_walk_sc(DevImage im, DevResult r, const uint32_t* sid_per_region, WalkScratch ws) {
  constexpr bool EMIT = MODE == 1;
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
  const bool lead = (threadIdx.x % SUB) == 0;
  if (MODE == 2 && walk_void(r, ws, q, lead)) return;
  if (q >= r.Q) return;
  const uint32_t sid = sid_per_region[q];
  const uint64_t x = r.regions[2 * q], y = r.regions[2 * q + 1];
  uint8_t fl;
  uint64_t nvar, ncar, ncar_kept;
  sc_walk_region<MODE>(im, r, ws, q, sid, x, y, lead, fl, nvar, ncar, ncar_kept);
  if (!lead) return;
  if (!EMIT) { r.q_flags[q] = fl; r.q_g0[q] = 0; r.q_nvar[q] = nvar; r.q_ncar[q] = ncar; }
  else { r.var_count[q] = nvar; r.q_ncar[q] = ncar_kept; }
}

// Query types 2 and 3: the sequence of a sample over [x, y).  The walk produces the list of (pool offset,
// length) pieces; seg_begin / byte_begin are the exclusive scans of the counting pass.
struct DevSeqResult {
  uint64_t Q;
  const uint64_t* regions;
  const uint32_t* sids;
  uint8_t* q_flags;
  uint64_t *q_nseg, *q_nbytes;      // [Q] counting pass
  uint64_t *seg_begin, *byte_begin; // [Q+1]
  uint32_t *seg_src, *seg_len;      // [nseg]
  uint64_t* seg_dst;                // [nseg] byte offset in chars (relative to the region's first byte when `relative`)
  uint8_t* chars;
  uint64_t* overflow;               // single-walk mode: set when a region outgrew its piece capacity (2 / 3: the batch was refused, seq_void)
  uint32_t relative, pad_;          // single-walk mode: pieces sit at seg_begin[q] .. + q_nseg[q], seg_begin = capacities' scan
  WalkAdmit admit;                  // the piece list was sized from the previous batch: does this one fit (k_walk.hip.h)
};

// Per-region records of a SEQUENCE result (query types 2 / 3) for the collective, in the format of k_pack_regions:
//   {region_base + q, region flags << 32, pieces, bytes of the sequence}
__global__ void __launch_bounds__(256) k_pack_seq_regions(DevSeqResult r, uint64_t* dst, uint64_t region_base) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  dst[4 * q + 0] = region_base + q;
  dst[4 * q + 1] = (uint64_t)r.q_flags[q] << 32;
  dst[4 * q + 2] = r.q_nseg[q] & 0xFFFFFFFFULL;
  dst[4 * q + 3] = r.byte_begin[q + 1] - r.byte_begin[q];
}

// a batch whose piece list was sized from the previous batch and does not fit (verdict 2, or 3 for a sample id out of range:
// admit_verdict) records nothing: no pieces, no bytes
__device__ __forceinline__ bool seq_void(const DevSeqResult& r, uint64_t q, bool writer) {
  if (!r.overflow) return false;
  const uint32_t verdict = admit_verdict(r.admit);
  if (!verdict) return false;
  if (writer) {
    if (q < r.Q) { r.q_nseg[q] = 0; r.q_nbytes[q] = 0; }
    *r.overflow = verdict;
  }
  return true;
}

// PASS 0 counts, PASS 1 writes the pieces at their scanned places (second walk), PASS 2 is the single walk: pieces go
// to the region's slice of a capacity-sized list with byte offsets relative to the region's first byte.
struct SeqSink {
  uint64_t nseg, nbytes;
  bool lead;   // this lane writes for its region (lane 0 of the group that walks it)
};
template <int PASS>
__device__ __forceinline__ void seq_append(const DevSeqResult& r, SeqSink& s, uint64_t seg0, uint64_t byte0, uint32_t off,
                                           uint64_t len, uint64_t cap) {
  if (len == 0) return;
  if (PASS == 1 || (PASS == 2 && s.nseg < cap)) {
    if (s.lead) { r.seg_src[seg0 + s.nseg] = off; r.seg_len[seg0 + s.nseg] = (uint32_t)len; r.seg_dst[seg0 + s.nseg] = byte0 + s.nbytes; }
  } else if (PASS == 2 && s.lead) *r.overflow = 1;
  s.nseg++; s.nbytes += len;
}

// the window logic of query.h:160-177 / :236-247 on (off, l) instead of a std::string.
// Returns 0 continue, 1 stop, 2 std::out_of_range (uncaught in the reference).
template <int PASS>
__device__ __forceinline__ int seq_window(const DevSeqResult& r, SeqSink& s, uint64_t seg0, uint64_t byte0, uint64_t cap, bool& record,
                                          uint32_t off, uint64_t l, uint64_t cur, uint64_t next, uint64_t x, uint64_t y) {
  if (record && next < y) {
    seq_append<PASS>(r, s, seg0, byte0, off, l, cap);
  } else if (record && next >= y) {
    const uint64_t n = y - cur;  // substr(0, n): n may have wrapped, it is clipped to the string
    seq_append<PASS>(r, s, seg0, byte0, off, n < l ? n : l, cap);
    return 1;
  } else if (next >= x && next < y) {
    record = true;
    const uint64_t p = x - cur;
    if (p > l) return 2;
    seq_append<PASS>(r, s, seg0, byte0, off + (uint32_t)p, l - p, cap);
  } else if (next >= x && next >= y) {
    const uint64_t p = x - cur;
    if (p > l) return 2;
    const uint64_t n = y - x;
    seq_append<PASS>(r, s, seg0, byte0, off + (uint32_t)p, n < l - p ? n : l - p, cap);
    return 1;
  }
  return 0;
}

// One region's walk, start to end, by one lane (the reference's loop with jumps over uneventful runs when the sample has
// event rows): returns the region's flag, the pieces go through `s`.
template <int MODE, int PASS>
__device__ __forceinline__ uint8_t seq_walk_region(const DevImage& im, const DevSeqResult& r, uint64_t x, uint64_t y, uint32_t sid,
                                                   uint64_t seg0, uint64_t byte0, uint64_t cap, SeqSink& s) {
  uint8_t fl = 0;
  uint64_t ref_pos = 0, sample_pos = 0;
  uint32_t cur = 0;
  bool ok = true;
  // With the event and hold rows of query type 4 (and the break bits of device_image.hpp) the search jumps over ranks
  // without a neighbour holding the sample, and the walk turns every uneventful run of ref-path slots into ONE piece.
  // (not for y < x: the window's `y - x` then wraps and is clipped to the VERTEX it is applied to -- a merged run would clip differently)
  const bool fast = im.t4_events != nullptr && im.seq_breaks != nullptr && sid != 0 && y >= x;
  BitRow ev = sample_event_row(im, sid, fast);   // the sample's own events: the searches, the walk's jumps
  BitRow hold = sample_hold_row(im, sid, fast);
  BitRow brk{im.seq_breaks, kNone, 0};
  if (MODE == 2) cur = fast ? prev_vertex_with_sample_ev<false>(im, x, sid, ev, hold, ref_pos, sample_pos) : prev_vertex_with_sample(im, x, sid, ref_pos, sample_pos);
  else ok = fast ? rewind_to_sample_pos_ev(im, x, sid, ev, hold, cur, ref_pos, sample_pos) : rewind_to_sample_pos(im, x, sid, cur, ref_pos, sample_pos);
  if (!ok) fl = kRegionEndless;
  else if (fast) {
    // state: cur and its record {first edge record in the blob, degree, ref index, sequence offset, length, ref-path slot + 1}
    const uint4 v0 = im.w_vertex[2 * (uint64_t)cur], v1 = im.w_vertex[2 * (uint64_t)cur + 1];
    uint32_t rbeg = im.blob_row[cur], deg = v0.y, ridx = v0.z, off = v0.w, len = v1.x, slot1 = v1.w;
    const uint32_t last_slot = (uint32_t)im.P - 1;
    bool record = false;
    while (true) {
      // On a ref-path node (type 2: in step with it, ref_pos == its index): up to the next slot k with an event for this
      // sample or a break bit, the literal loop appends node after node -- consecutive in the pool and in the
      // coordinate, the path successor taken every time -- which seq_window sees as ONE vertex of their total length.
      if (slot1 && (MODE != 2 || ref_pos == ridx)) {
        const uint32_t s0 = slot1 - 1;
        const uint32_t lim = s0 + 1024 < last_slot ? s0 + 1024 : last_slot;   // (bounded look-ahead: a clear slot `lim` is as good a place to land)
        // (the sample's OWN events: an irregular slot -- its LAST ref neighbour is not its successor -- is nothing to a sequence
        //  walk, which follows the FIRST ref neighbour and the smallest-index one; what could break a run here is in the break bits)
        uint32_t k = s0 < lim ? ev.next(s0, lim) : s0;
        if (k > s0) { const uint32_t kb = brk.next_wide(s0, lim); k = kb < k ? kb : k; }
        if (k > s0) {
          const uint64_t h = im.blob_of_slot[k];
          const uint4 ra = im.wblob[2 * h], rb = im.wblob[2 * h + 1];   // header of slot k
          const uint64_t run = (uint64_t)ra.z - ridx;                   // bases of slots [s0, k)
          int st;
          if (MODE == 2) {
            st = seq_window<PASS>(r, s, seg0, byte0, cap, record, off, run, ref_pos, (uint64_t)ra.z, x, y);
            ref_pos = ra.z;
          } else {
            st = seq_window<PASS>(r, s, seg0, byte0, cap, record, off, run, sample_pos, sample_pos + run, x, y);
            sample_pos += run;
          }
          if (st == 2) { fl = kRegionInvalid; break; }
          if (st == 1) break;
          cur = rb.w; rbeg = ra.x; deg = ra.y; ridx = ra.z; off = ra.w; len = rb.x; slot1 = k + 1;
          continue;
        }
      }
      // the literal iteration (query.h:143-180 / :228-250) over the blob's edge records
      uint64_t next_ref_pos = ref_pos + len;
      bool have_ref = false, by_sample = false;
      uint32_t nxt = 0, min_idx = 0xFFFFFFFFu, n_rbeg = 0, n_deg = 0, n_ridx = 0, n_len = 0, n_slot1 = 0;
      for (uint32_t e = rbeg; e < rbeg + deg; ++e) {
        const uint4 a = im.wblob[2 * (uint64_t)e];
        if (a.y && !have_ref) { next_ref_pos = a.y; have_ref = true; }   // the FIRST ref neighbour (query.h:150-153)
        if (!by_sample) {                                                 // get_neighbor_vertex
          const uint4 b = im.wblob[2 * (uint64_t)e + 1];
          const bool holds = hold.bit_n(a.x, b.w);
          if (holds || (a.y && min_idx > a.y)) {
            nxt = a.x; n_rbeg = a.w; n_deg = b.x; n_ridx = a.y; n_len = b.z; n_slot1 = b.y;
            if (holds) by_sample = true; else min_idx = a.y;
          }
        }
      }
      int st;
      if (MODE == 2) {
        st = seq_window<PASS>(r, s, seg0, byte0, cap, record, off, len, ref_pos, next_ref_pos, x, y);
        ref_pos = next_ref_pos;
      } else {
        st = seq_window<PASS>(r, s, seg0, byte0, cap, record, off, len, sample_pos, sample_pos + len, x, y);
        sample_pos += len;
      }
      if (st == 2) { fl = kRegionInvalid; break; }
      if (st == 1 || nxt == 0) break;
      cur = nxt; rbeg = n_rbeg; deg = n_deg; ridx = n_ridx; len = n_len; slot1 = n_slot1;
      off = im.v_off[cur];
    }
  } else {
    bool record = false, done = false;
    while (!done) {
      const WalkVertex wc = walk_vertex(im, cur);
      const uint32_t off = wc.off;
      const uint64_t l = wc.len;
      int st;
      if (MODE == 2) {
        uint64_t next_ref_pos = ref_pos + l;
        for (uint32_t e = wc.row_begin; e < wc.row_begin + wc.deg; ++e) {
          const uint32_t nr = walk_edge(im, e).ridx;
          if (nr) { next_ref_pos = nr; break; }  // the FIRST ref neighbour here (query.h:150-153)
        }
        st = seq_window<PASS>(r, s, seg0, byte0, cap, record, off, l, ref_pos, next_ref_pos, x, y);
        ref_pos = next_ref_pos;
      } else {
        const uint64_t next_sample_pos = sample_pos + l;
        st = seq_window<PASS>(r, s, seg0, byte0, cap, record, off, l, sample_pos, next_sample_pos, x, y);
        sample_pos = next_sample_pos;
      }
      if (st == 2) { fl = kRegionInvalid; break; }
      if (st == 1) break;
      const uint32_t nxt = next_on_path(im, cur, sid);
      if (nxt == 0) done = true;
      cur = nxt;
    }
  }
  return fl;
}

// (SUB lanes per region running the same walk, lane 0 writing: see k_sample_walk_sc)
template <int MODE, int PASS, uint32_t SUB = kScGroup>
__global__ void __launch_bounds__(SUB > 1 ? 256 : 64) k_sample_seq(DevImage im, DevSeqResult r) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
  if (PASS == 2 && seq_void(r, q, (threadIdx.x % SUB) == 0)) return;
  if (q >= r.Q) return;
  if (PASS == 1 && r.q_flags[q]) return;
  const uint32_t sid = r.sids[q];
  const uint64_t x = r.regions[2 * q], y = r.regions[2 * q + 1];
  const uint64_t seg0 = PASS ? r.seg_begin[q] : 0, byte0 = PASS == 1 ? r.byte_begin[q] : 0;
  const uint64_t cap = PASS == 2 ? r.seg_begin[q + 1] - seg0 : 0;
  SeqSink s{0, 0, (threadIdx.x % SUB) == 0};
  const uint8_t fl = seq_walk_region<MODE, PASS>(im, r, x, y, sid, seg0, byte0, cap, s);
  if (PASS != 1 && s.lead) {
    r.q_flags[q] = fl;
    r.q_nseg[q] = fl ? 0 : s.nseg;
    r.q_nbytes[q] = fl ? 0 : s.nbytes;
  }
}

// ---------------------------------------------------------------------------
// Query types 2 and 3, cooperative (round 4): SUB lanes per region.
// One lane per region walks a chain of ~90 dependent look-ups (a sample has ~11 events in a 10 kb region of the chr1
// cohort, each costs the jump to it, the step off the path and the step back); a batch is 1,563 waves and the kernel's
// time is the longest chain's.  Here the chain is cut: after the backward search and the head (redundant in the group,
// same addresses) the group takes the region's EVENTS -- set bits of the sample's event row and of the break row -- SUB
// at a time, one per lane, and every lane walks its own EPISODE: arrive at the event's slot as a jump would (in step),
// step literally until the walk stands on the ref path again (type 2: in step with it).  An episode's pieces stay in
// registers.  Then the hand-over, in order and in registers: the first episode at or after the slot the chain stands on
// is the one the serial walk would run next; its lane puts the uneventful run up to its slot and its own pieces through
// the reference's window logic (seq_window: the `record` flag, clipping at x and y, the stop) with the running position
// and counts it receives, writes the pieces, and hands the end state on.  Episodes the chain jumps over are dropped.
// When fewer than SUB events lie in the SUB words a round looks at, the next lane lands on the first slot behind them
// (a jump to a slot without an event is a literal step like any other: the one-lane walk does the same at its
// look-ahead limit), so every round advances.  Pieces differ from the one-lane walk's in where runs are cut, the bytes
// do not.  Anything that does not fit -- an episode of more than kSeqEpSteps steps, a walk that never finds the path,
// positions beyond 32 bits -- sends the region through the one-lane walk (seq_walk_region), in this kernel.
// Not for coarse event rows (explicit-id cohorts: one to three events per region, bits that do not name slots).
// ---------------------------------------------------------------------------
constexpr uint32_t kSeqEpSteps = 6;
struct SeqSt { uint32_t cur, rbeg, deg, ridx, off, len, slot1; };   // a vertex of the walk: first edge record, degree, ref index, sequence, ref-path slot + 1
// the out-edges of one literal iteration (query.h:143-180 / :228-250): the FIRST ref neighbour's index; get_neighbor_vertex
// (variant_graph.h:1402-1451) with the record of the vertex it names (n.off is not part of an edge record)
__device__ __forceinline__ uint32_t seq_step_edges(const DevImage& im, BitRow& hold, const SeqSt& st, uint64_t& next_ref_pos, SeqSt& n) {
  bool have_ref = false, by_sample = false;
  uint32_t nxt = 0, min_idx = 0xFFFFFFFFu;
  for (uint32_t e = st.rbeg; e < st.rbeg + st.deg; ++e) {
    const uint4 a = im.wblob[2 * (uint64_t)e];
    if (a.y && !have_ref) { next_ref_pos = a.y; have_ref = true; }
    if (!by_sample) {
      const uint4 b = im.wblob[2 * (uint64_t)e + 1];
      const bool holds = hold.bit_n(a.x, b.w);
      if (holds || (a.y && min_idx > a.y)) {
        nxt = a.x; n.cur = a.x; n.rbeg = a.w; n.deg = b.x; n.ridx = a.y; n.len = b.z; n.slot1 = b.y;
        if (holds) by_sample = true; else min_idx = a.y;
      }
    }
  }
  return nxt;
}
__device__ __forceinline__ SeqSt seq_slot_state(const DevImage& im, uint32_t k) {   // the node of ref-path slot k from its header record
  const uint64_t h = im.blob_of_slot[k];
  const uint4 ra = im.wblob[2 * h], rb = im.wblob[2 * h + 1];
  return SeqSt{rb.w, ra.x, ra.y, ra.z, ra.w, rb.x, k + 1};
}

template <int MODE, uint32_t SUB>
__global__ void __launch_bounds__(256) k_sample_seq_coop(DevImage im, DevSeqResult r) {
  static_assert(MODE == 2 || MODE == 3, "query type");
  static_assert(SUB == 8 || SUB == 16, "group width");
  constexpr uint32_t kGroupMask = (1u << SUB) - 1u;
  const uint32_t lane = threadIdx.x & 63, l = lane & (SUB - 1), gbase = lane & (64 - SUB);
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
  if (seq_void(r, q, l == 0)) return;
  const bool live = q < r.Q;
  uint32_t sid = 0;
  uint64_t x = 0, y = 0, seg0 = 0, cap = 0;
  if (live) { sid = r.sids[q]; x = r.regions[2 * q]; y = r.regions[2 * q + 1]; seg0 = r.seg_begin[q]; cap = r.seg_begin[q + 1] - seg0; }
  const bool fast = live && im.t4_events != nullptr && im.seq_breaks != nullptr && sid != 0 && y >= x && im.t4_ev_shift == 0;
  // group-uniform state
  bool serial = live && !fast, busy = false, record = false;
  SeqSink s{0, 0, l == 0};
  uint8_t fl = 0;
  uint64_t pos = 0;                                   // type 2: ref_pos, type 3: sample_pos
  uint32_t cur_slot = 0, off_cur = 0, ridx_cur = 0;   // the ref-path node the chain stands on: slot, sequence offset, ref index
  BitRow ev = sample_event_row(im, sid, fast), hold = sample_hold_row(im, sid, fast);
  const uint32_t last_slot = (uint32_t)im.P - 1;
  // (the backward searches run redundantly in the group -- one-lane code, same addresses.  The group-parallel window search
  //  of the type-4 kernel, group_search_prev, was tried here: the same kernel times.)
  if (fast) {
    uint64_t ref_pos = 0, sample_pos = 0;
    uint32_t cur = 0;
    bool ok = true;
    if (MODE == 2) cur = prev_vertex_with_sample_ev<false>(im, x, sid, ev, hold, ref_pos, sample_pos);
    else ok = rewind_to_sample_pos_ev(im, x, sid, ev, hold, cur, ref_pos, sample_pos);
    if (!ok) fl = kRegionEndless;
    else {
      // ---- head: literal iterations from the start vertex until the walk stands on the path (redundant in the group, lane 0 writes) ----
      const uint4 v0 = im.w_vertex[2 * (uint64_t)cur], v1 = im.w_vertex[2 * (uint64_t)cur + 1];
      SeqSt st{cur, im.blob_row[cur], v0.y, v0.z, v0.w, v1.x, v1.w};
      uint32_t steps = 0;
      while (true) {
        if (st.slot1 && (MODE != 2 || ref_pos == st.ridx)) { busy = true; cur_slot = st.slot1 - 1; off_cur = st.off; ridx_cur = st.ridx; break; }
        if (++steps > 64) { serial = true; break; }
        uint64_t next_ref_pos = ref_pos + st.len;
        SeqSt n{};
        const uint32_t nxt = seq_step_edges(im, hold, st, next_ref_pos, n);
        int w;
        if (MODE == 2) { w = seq_window<2>(r, s, seg0, 0, cap, record, st.off, st.len, ref_pos, next_ref_pos, x, y); ref_pos = next_ref_pos; }
        else { w = seq_window<2>(r, s, seg0, 0, cap, record, st.off, st.len, sample_pos, sample_pos + st.len, x, y); sample_pos += st.len; }
        if (w == 2) { fl = kRegionInvalid; break; }
        if (w == 1 || nxt == 0) break;
        st = n;
        st.off = im.v_off[st.cur];
      }
      pos = MODE == 2 ? ref_pos : sample_pos;
    }
  }
  // ---- episodes, SUB events of a group at a time ----
  while (__any(busy)) {
    // the events at or after cur_slot and below the last slot in SUB words of the two rows, one word per lane
    const uint32_t w0 = cur_slot >> 6, wi = w0 + l;
    uint64_t word = 0;
    if (busy && ((uint64_t)wi << 6) < last_slot) {
      word = ev.row[wi] | im.seq_breaks[wi];
      if (l == 0) word &= ~0ULL << (cur_slot & 63);
      if (wi == (last_slot >> 6)) word &= (1ULL << (last_slot & 63)) - 1;
    }
    const uint32_t pc = (uint32_t)__popcll(word), incl = group_inclusive_scan<SUB>(l, pc);
    const uint32_t total = (uint32_t)__shfl((int)incl, (int)gbase + (int)SUB - 1, 64);
    uint32_t j = 0;                                   // the word holding this lane's event: #words whose inclusive count is <= l
#pragma unroll
    for (int t = 0; t < (int)SUB; ++t) j += (uint32_t)__shfl((int)incl, (int)gbase + t, 64) <= l ? 1u : 0u;
    const int src = (int)gbase + (int)(j < SUB ? j : SUB - 1);
    const uint64_t wj = shfl64(word, src);
    const uint32_t excl_j = (uint32_t)__shfl((int)(incl - pc), src, 64);
    const uint64_t chunk_end64 = ((uint64_t)(w0 + SUB)) << 6;
    const uint32_t chunk_end = chunk_end64 < last_slot ? (uint32_t)chunk_end64 : last_slot;
    bool have = busy && l < total;
    uint32_t slot = 0;
    if (have) slot = ((w0 + j) << 6) + select_bit(wj, l - excl_j);
    else if (busy && l == total) { have = true; slot = chunk_end; }   // the landing slot behind the round's events
    // ---- this lane's episode ----
    uint32_t p_off[kSeqEpSteps], p_len[kSeqEpSteps], p_next[kSeqEpSteps];   // (type 2: a piece's own position is the one before it's next, the first one's the slot's index)
    uint32_t n_p = 0, ep_end = 0, e_off = 0, e_ridx = 0, k_ridx = 0;
    bool ep_term = false, ep_ovf = false;
    if (have) {
      SeqSt es = seq_slot_state(im, slot);
      k_ridx = es.ridx;
      uint64_t e_ref = es.ridx;
      while (true) {
        uint64_t next_ref_pos = e_ref + es.len;
        SeqSt n{};
        const uint32_t nxt = seq_step_edges(im, hold, es, next_ref_pos, n);
        if (next_ref_pos >> 32) { ep_ovf = true; break; }
#pragma unroll
        for (uint32_t t = 0; t < kSeqEpSteps; ++t)
          if (t == n_p) { p_off[t] = es.off; p_len[t] = es.len; if (MODE == 2) p_next[t] = (uint32_t)next_ref_pos; }
        ++n_p;
        e_ref = next_ref_pos;
        if (nxt == 0) { ep_term = true; break; }
        es = n;
        es.off = im.v_off[es.cur];
        if (es.slot1 && (MODE != 2 || e_ref == es.ridx)) { ep_end = es.slot1 - 1; e_ridx = es.ridx; e_off = es.off; break; }
        if (n_p >= kSeqEpSteps) { ep_ovf = true; break; }
      }
      if (!ep_term && !ep_ovf && ep_end <= slot) ep_ovf = true;   // (a walk that does not advance: one-lane walk)
    }
    // ---- the chain of hand-overs ----
    bool gdone = !busy;
#pragma unroll 1
    for (int t = 0; t < (int)SUB; ++t) {
      const bool cand = !gdone && have && slot >= cur_slot;
      const uint32_t gb = (uint32_t)((__ballot(cand) >> gbase) & kGroupMask);
      const int i = gb ? (int)gbase + __builtin_ctz(gb) : (int)gbase;
      int stt = 0;                                    // 0 on, 1 the walk is over, 2 std::out_of_range, 3 one-lane walk
      uint64_t pos_l = pos;
      bool rec_l = record;
      SeqSink s_l{s.nseg, s.nbytes, true};
      if (gb && (int)lane == i) {
        if (ep_ovf) stt = 3;
        else {
          if (slot > cur_slot) {                      // the uneventful run [cur_slot, slot): one piece
            const uint64_t run = (uint64_t)k_ridx - ridx_cur;
            if (MODE == 2) stt = seq_window<2>(r, s_l, seg0, 0, cap, rec_l, off_cur, run, (uint64_t)ridx_cur, (uint64_t)k_ridx, x, y);
            else { stt = seq_window<2>(r, s_l, seg0, 0, cap, rec_l, off_cur, run, pos_l, pos_l + run, x, y); pos_l += run; }
          }
          uint32_t p_at = k_ridx;
#pragma unroll
          for (uint32_t t2 = 0; t2 < kSeqEpSteps; ++t2)
            if (stt == 0 && t2 < n_p) {
              if (MODE == 2) { stt = seq_window<2>(r, s_l, seg0, 0, cap, rec_l, p_off[t2], p_len[t2], (uint64_t)p_at, (uint64_t)p_next[t2], x, y); p_at = p_next[t2]; }
              else { stt = seq_window<2>(r, s_l, seg0, 0, cap, rec_l, p_off[t2], p_len[t2], pos_l, pos_l + p_len[t2], x, y); pos_l += p_len[t2]; }
            }
          if (stt == 0 && ep_term) stt = 1;          // no next vertex: the path iterator is done
        }
      }
      const int g_stt = __shfl(stt, i, 64);
      const uint64_t g_nseg = shfl64(s_l.nseg, i), g_nbytes = shfl64(s_l.nbytes, i), g_pos = shfl64(pos_l, i);
      const bool g_rec = __shfl((int)rec_l, i, 64) != 0;
      const uint32_t g_end = (uint32_t)__shfl((int)ep_end, i, 64), g_off = (uint32_t)__shfl((int)e_off, i, 64), g_ridx = (uint32_t)__shfl((int)e_ridx, i, 64);
      if (!gdone) {
        if (!gb) gdone = true;                        // no event left in this round
        else {
          s.nseg = g_nseg; s.nbytes = g_nbytes; pos = g_pos; record = g_rec;
          if (g_stt == 0) { cur_slot = g_end; off_cur = g_off; ridx_cur = g_ridx; }
          else {
            gdone = true; busy = false;
            if (g_stt == 2) fl = kRegionInvalid;
            if (g_stt == 3) serial = true;
          }
        }
      }
      if (!__any(!gdone)) break;
    }
  }
  // ---- regions without event rows, and fallbacks: the one-lane walk (redundant in the group; lane 0 writes) ----
  if (__any(serial)) {
    if (serial) {
      s = SeqSink{0, 0, l == 0};
      fl = seq_walk_region<MODE, 2>(im, r, x, y, sid, seg0, 0, cap, s);
    }
  }
  if (live && l == 0) {
    r.q_flags[q] = fl;
    r.q_nseg[q] = fl ? 0 : s.nseg;
    r.q_nbytes[q] = fl ? 0 : s.nbytes;
  }
}

// ---------------------------------------------------------------------------
// Query type 5, cooperative (round 4): the recording walk of get_sample_var_in_sample with SUB lanes per region, built
// like k_sample_seq_coop -- search and head redundant in the group, the region's events (sample's event row | break row)
// taken SUB at a time, one episode per lane, hand-over in order.  An episode notes, per literal step, the bases walked
// before it, and resolves the rows of the vertices that hold the sample (position, REF, ALT: what the one-lane walk
// writes) without knowing where in the sample's coordinates it is; the hand-over, which knows (the running sample_pos),
// applies the reference's two tests to each step -- stop at sample_pos >= y, report only beyond x (query.h:520-575) --
// and writes the rows that pass.  Episodes of more than kScEpSteps steps or kScEpRows rows: one-lane walk (sc_walk_region).
// ---------------------------------------------------------------------------
constexpr uint32_t kScEpSteps = 6, kScEpRows = 2;
struct ScSt { uint32_t cur, rbeg, deg, ridx, len, cls, ncar, slot1; };
struct ScRow { uint32_t pos, cur, ro, rl, ao, al, c, step; };
// the out-edges of one literal iteration (query.h:530-545): the LAST ref neighbour; get_neighbor_vertex with its record
__device__ __forceinline__ uint32_t sc_step_edges(const DevImage& im, BitRow& hold, const ScSt& st, uint64_t& next_ref_pos, uint32_t& next_ref_v, ScSt& n) {
  uint32_t nxt = 0, min_idx = 0xFFFFFFFFu;
  bool by_sample = false;
  for (uint32_t e = st.rbeg; e < st.rbeg + st.deg; ++e) {
    const uint4 a = im.wblob[2 * (uint64_t)e];
    if (a.y) { next_ref_pos = a.y; next_ref_v = a.x; }
    if (!by_sample) {
      const uint4 b = im.wblob[2 * (uint64_t)e + 1];
      const bool holds = hold.bit_n(a.x, b.w);
      if (holds || (a.y && min_idx > a.y)) {
        nxt = a.x; n.cur = a.x; n.rbeg = a.w; n.deg = b.x; n.ridx = a.y; n.len = b.z; n.cls = a.z; n.ncar = b.w; n.slot1 = b.y;
        if (holds) by_sample = true; else min_idx = a.y;
      }
    }
  }
  return nxt;
}
// the row of a vertex that holds the sample (query.h:547-575), wherever the walk is in the sample's coordinates
__device__ __forceinline__ bool sc_resolve_row(const DevImage& im, const ScSt& st, uint32_t sid, uint64_t ref_pos, uint64_t next_ref_pos, uint32_t cur_ref_v, ScRow& row) {
  uint32_t sidx = 0;
  if (!sample_entry_rec(im, st.cur, st.ridx, im.use_bv ? st.cls : 0u, sid, sidx)) return false;
  row.cur = st.cur; row.c = st.ncar;
  if (ref_pos == next_ref_pos) {        // insertion
    row.pos = (uint32_t)ref_pos; row.ro = 0; row.rl = 0; row.ao = im.v_off[st.cur]; row.al = st.len;
  } else if (st.ridx) {                 // deletion: ref = sequence of find(ref_pos - 1)
    const uint32_t fv = im.rp_vid[slot_of_find(im, ref_pos - 1)];
    row.pos = sidx; row.ro = im.v_off[fv]; row.rl = im.v_len[fv]; row.ao = 0; row.al = 0;
  } else {                              // substitution: ref = sequence of the previous step's last ref neighbour
    row.pos = sidx; row.ro = 0; row.rl = 0; row.ao = im.v_off[st.cur]; row.al = st.len;
    if (cur_ref_v != kNone) { row.ro = im.v_off[cur_ref_v]; row.rl = im.v_len[cur_ref_v]; }
  }
  return true;
}

template <uint32_t SUB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) k_sample_walk_sc_coop(DevImage im, DevResult r, const uint32_t* sid_per_region, WalkScratch ws) {
  static_assert(SUB == 8 || SUB == 16, "group width");
  constexpr uint32_t kGroupMask = (1u << SUB) - 1u;
  const uint32_t lane = threadIdx.x & 63, l = lane & (SUB - 1), gbase = lane & (64 - SUB);
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
  if (walk_void(r, ws, q, l == 0)) return;
  const bool live = q < r.Q;
  uint32_t sid = 0;
  uint64_t x = 0, y = 0, s0 = 0, scap = 0;
  if (live) { sid = sid_per_region[q]; x = r.regions[2 * q]; y = r.regions[2 * q + 1]; s0 = ws.cap_begin[q]; scap = ws.cap_begin[q + 1] - s0; }
  const bool fast = live && im.t4_events != nullptr && im.seq_breaks != nullptr && sid != 0 && im.t4_ev_shift == 0;
  // group-uniform state
  bool serial = live && !fast, busy = false;
  uint8_t fl = 0;
  uint64_t nvar = 0, ncar = 0, ncar_kept = 0;
  uint64_t pos = 0;                               // sample_pos where the chain stands
  uint32_t cur_slot = 0, ridx_cur = 0;            // the ref-path node the chain stands on, in step: slot, ref index
  BitRow ev = sample_event_row(im, sid, fast), hold = sample_hold_row(im, sid, fast);
  const uint32_t last_slot = (uint32_t)im.P - 1;
  auto put_row = [&](const ScRow& row, uint64_t& nv, uint64_t& nc) {   // (the lane that calls it writes)
    if (nv < scap) {
      const uint64_t s = s0 + nv;
      ws.pos[s] = row.pos; ws.cur[s] = row.cur; ws.ro[s] = row.ro; ws.rl[s] = row.rl; ws.ao[s] = row.ao; ws.al[s] = row.al;
    } else *ws.overflow = 1;
    nv++; nc += pad_car(row.c);
  };
  if (fast) {
    uint64_t ref_pos = 0, sample_pos = 0;
    uint32_t closest_v = 0;
    if (!rewind_to_sample_pos_ev(im, x, sid, ev, hold, closest_v, ref_pos, sample_pos)) fl = kRegionEndless;
    else {
      closest_v = im.rp_vid[slot_of_find(im, ref_pos)];
      if (im.v_ridx[closest_v]) {
        const uint64_t seq_len = ref_pos - im.v_ridx[closest_v];
        ref_pos = im.v_ridx[closest_v];
        sample_pos -= seq_len;
      }
      // ---- head: literal iterations until the walk is in step with the path (redundant in the group, lane 0 writes) ----
      const uint4 v0 = im.w_vertex[2 * (uint64_t)closest_v], v1 = im.w_vertex[2 * (uint64_t)closest_v + 1];
      ScSt st{closest_v, im.blob_row[closest_v], v0.y, v0.z, v1.x, v1.y, v1.z, v1.w};
      uint32_t cur_ref_v = kNone, steps = 0;
      while (true) {
        if (sample_pos >= y) break;
        if (st.slot1 && ref_pos == st.ridx) { busy = true; cur_slot = st.slot1 - 1; ridx_cur = st.ridx; break; }
        if (++steps > 64 || (ref_pos >> 32)) { serial = true; break; }
        uint64_t next_ref_pos = ref_pos + st.len;
        uint32_t next_ref_v = kNone;
        ScSt n{};
        const uint32_t nxt = sc_step_edges(im, hold, st, next_ref_pos, next_ref_v, n);
        ScRow row{};
        if (sample_pos > x && hold.bit_n(st.cur, st.ncar) && sc_resolve_row(im, st, sid, ref_pos, next_ref_pos, cur_ref_v, row)) {
          uint64_t nv = nvar, nc = ncar;
          if (l == 0) put_row(row, nv, nc);
          nvar++; ncar += pad_car(row.c);
        }
        cur_ref_v = next_ref_v; ref_pos = next_ref_pos; sample_pos += st.len;
        if (nxt == 0) break;
        st = n;
      }
      pos = sample_pos;
    }
  }
  // ---- episodes, SUB events of a group at a time ----
  while (__any(busy)) {
    const uint32_t w0 = cur_slot >> 6, wi = w0 + l;
    uint64_t word = 0;
    if (busy && ((uint64_t)wi << 6) < last_slot) {
      word = ev.row[wi] | im.seq_breaks[wi];
      if (l == 0) word &= ~0ULL << (cur_slot & 63);
      if (wi == (last_slot >> 6)) word &= (1ULL << (last_slot & 63)) - 1;
    }
    const uint32_t pc = (uint32_t)__popcll(word), incl = group_inclusive_scan<SUB>(l, pc);
    const uint32_t total = (uint32_t)__shfl((int)incl, (int)gbase + (int)SUB - 1, 64);
    uint32_t j = 0;
#pragma unroll
    for (int t = 0; t < (int)SUB; ++t) j += (uint32_t)__shfl((int)incl, (int)gbase + t, 64) <= l ? 1u : 0u;
    const int src = (int)gbase + (int)(j < SUB ? j : SUB - 1);
    const uint64_t wj = shfl64(word, src);
    const uint32_t excl_j = (uint32_t)__shfl((int)(incl - pc), src, 64);
    const uint64_t chunk_end64 = ((uint64_t)(w0 + SUB)) << 6;
    const uint32_t chunk_end = chunk_end64 < last_slot ? (uint32_t)chunk_end64 : last_slot;
    bool have = busy && l < total;
    uint32_t slot = 0;
    if (have) slot = ((w0 + j) << 6) + select_bit(wj, l - excl_j);
    else if (busy && l == total) { have = true; slot = chunk_end; }   // the landing slot behind the round's events
    // ---- this lane's episode ----
    uint32_t p_cum[kScEpSteps];                     // bases walked before each step
    ScRow rows[kScEpRows];
    uint32_t n_s = 0, n_rows = 0, ep_cum = 0, ep_end = 0, e_ridx = 0, k_ridx = 0;
    bool ep_term = false, ep_ovf = false;
    if (have) {
      const uint64_t h = im.blob_of_slot[slot];
      const uint4 ra = im.wblob[2 * h], rb = im.wblob[2 * h + 1];   // header of the slot
      ScSt es{rb.w, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, slot + 1};
      k_ridx = es.ridx;
      uint64_t e_ref = es.ridx;
      uint32_t cur_ref_v = es.cur;
      while (true) {
        uint64_t next_ref_pos = e_ref + es.len;
        uint32_t next_ref_v = kNone;
        ScSt n{};
        const uint32_t nxt = sc_step_edges(im, hold, es, next_ref_pos, next_ref_v, n);
        if ((next_ref_pos >> 32) || ((uint64_t)ep_cum + es.len) >> 31) { ep_ovf = true; break; }
        ScRow row{};
        if (hold.bit_n(es.cur, es.ncar) && sc_resolve_row(im, es, sid, e_ref, next_ref_pos, cur_ref_v, row)) {
          row.step = n_s;
#pragma unroll
          for (uint32_t t = 0; t < kScEpRows; ++t) if (t == n_rows) rows[t] = row;
          if (n_rows >= kScEpRows) { ep_ovf = true; break; }
          ++n_rows;
        }
#pragma unroll
        for (uint32_t t = 0; t < kScEpSteps; ++t) if (t == n_s) p_cum[t] = ep_cum;
        ++n_s;
        ep_cum += es.len;
        cur_ref_v = next_ref_v; e_ref = next_ref_pos;
        if (nxt == 0) { ep_term = true; break; }
        es = n;
        if (es.slot1 && e_ref == es.ridx) { ep_end = es.slot1 - 1; e_ridx = es.ridx; break; }
        if (n_s >= kScEpSteps) { ep_ovf = true; break; }
      }
      if (!ep_term && !ep_ovf && ep_end <= slot) ep_ovf = true;   // (a walk that does not advance: one-lane walk)
    }
    // ---- the chain of hand-overs ----
    bool gdone = !busy;
#pragma unroll 1
    for (int t = 0; t < (int)SUB; ++t) {
      const bool cand = !gdone && have && slot >= cur_slot;
      const uint32_t gb = (uint32_t)((__ballot(cand) >> gbase) & kGroupMask);
      const int i = gb ? (int)gbase + __builtin_ctz(gb) : (int)gbase;
      int stt = 0;                                  // 0 on, 1 the walk is over, 3 one-lane walk
      uint64_t pos_l = pos, nv_l = nvar, nc_l = ncar;
      if (gb && (int)lane == i) {
        if (ep_ovf) stt = 3;
        else {
          if (slot > cur_slot) {                    // the jump over the uneventful run [cur_slot, slot): the loop's test, then the bases
            if (pos_l >= y) stt = 1;
            else pos_l += (uint64_t)k_ridx - ridx_cur;
          }
#pragma unroll
          for (uint32_t t2 = 0; t2 < kScEpSteps; ++t2)
            if (stt == 0 && t2 < n_s) {
              const uint64_t sp = pos_l + p_cum[t2];
              if (sp >= y) stt = 1;
              else if (sp > x) {
#pragma unroll
                for (uint32_t t3 = 0; t3 < kScEpRows; ++t3)
                  if (t3 < n_rows && rows[t3].step == t2) put_row(rows[t3], nv_l, nc_l);
              }
            }
          if (stt == 0) { pos_l += ep_cum; if (ep_term) stt = 1; }
        }
      }
      const int g_stt = __shfl(stt, i, 64);
      const uint64_t g_pos = shfl64(pos_l, i), g_nv = shfl64(nv_l, i), g_nc = shfl64(nc_l, i);
      const uint32_t g_end = (uint32_t)__shfl((int)ep_end, i, 64), g_ridx = (uint32_t)__shfl((int)e_ridx, i, 64);
      if (!gdone) {
        if (!gb) gdone = true;
        else {
          pos = g_pos; nvar = g_nv; ncar = g_nc;
          if (g_stt == 0) { cur_slot = g_end; ridx_cur = g_ridx; }
          else {
            gdone = true; busy = false;
            if (g_stt == 3) serial = true;
          }
        }
      }
      if (!__any(!gdone)) break;
    }
  }
  // ---- regions without event rows, and fallbacks: the one-lane walk (redundant in the group; lane 0 writes) ----
  if (__any(serial)) {
    if (serial) sc_walk_region<2>(im, r, ws, q, sid, x, y, l == 0, fl, nvar, ncar, ncar_kept);
  }
  if (live && l == 0) { r.q_flags[q] = fl; r.q_g0[q] = 0; r.q_nvar[q] = nvar; r.q_ncar[q] = ncar; }
}

// Piece capacity of a region for the single walk: twice the ref-path slots plus branch sites of the (for sample
// coordinates: generously widened) reference range, plus slack.  Too small a guess only costs the fallback.
__device__ __forceinline__ void seq_caps_region(const DevImage& im, const DevSeqResult& r, uint64_t q, uint32_t sample_coordinates) {
  const uint64_t x = r.regions[2 * q], y = r.regions[2 * q + 1];
  const uint64_t margin = sample_coordinates ? (y > x ? y - x : 0) + 256 : 0;
  const uint64_t lo = x > margin + 1 ? x - margin : 1, hi = (y > x ? y : x) + margin;
  const uint32_t s0 = slot_of_find(im, lo), s1 = slot_of_find(im, hi);
  const uint64_t slots = s1 >= s0 ? (uint64_t)(s1 - s0) + 1 : 1;
  const uint64_t sites = s1 >= s0 ? (uint64_t)(im.rp_cand_prefix[s1 + 1] - im.rp_cand_prefix[s0]) : 0;
  r.q_nseg[q] = 2 * (slots + sites) + 8;
}
__global__ void __launch_bounds__(256) k_seq_caps(DevImage im, DevSeqResult r, uint32_t sample_coordinates) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < r.Q) seq_caps_region(im, r, q, sample_coordinates);
}
// the same for a batch of query types 2 / 3 whose regions and sample ids arrive in device memory, fused with the copies and the
// range check of the ids (k_walk_setup)
__global__ void __launch_bounds__(256) k_seq_setup(DevImage im, DevSeqResult r, const uint64_t* __restrict__ src_regions, const uint32_t* __restrict__ src_ids,
                                                   uint32_t num_samples, uint64_t* bad, uint32_t sample_coordinates) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  uint64_t* reg = const_cast<uint64_t*>(r.regions);
  uint32_t* ids = const_cast<uint32_t*>(r.sids);
  reg[2 * q] = src_regions[2 * q]; reg[2 * q + 1] = src_regions[2 * q + 1];
  const uint32_t sid = src_ids[q];
  ids[q] = sid;
  if (sid >= num_samples) *bad = 1;
  seq_caps_region(im, r, q, sample_coordinates);
}

// Decode the pieces into characters (map_int, util.cc:32-41: codes 0..4 -> "ACTGN", anything else -> char 5).  One wave
// per region, 64 piece descriptors at a time; the pieces of a region lie back to back in the output, so the wave works
// through the OUTPUT in aligned 16-byte groups, one group per lane: a lane finds the piece its group starts in by
// bisection over the 64 destination offsets (LDS), and -- when the group lies inside one piece, as all but one or two
// per piece do -- decodes it with one (unaligned) 16-byte load of codes, four v_perm_b32 with the eight characters as
// the byte table and one aligned 16-byte store; groups across a piece boundary go byte by byte.
__device__ __forceinline__ uint8_t decode_base(uint8_t c) { return (uint8_t)(0x0505054E47544341ULL >> (8 * (c & 7))); }
__global__ void __launch_bounds__(256) k_copy_segments(DevImage im, DevSeqResult r) {
  __shared__ uint32_t s_all[4][3][65];
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= r.Q) return;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t* s_dst = s_all[threadIdx.x >> 6][0];
  uint32_t* s_src = s_all[threadIdx.x >> 6][1];
  uint32_t* s_len = s_all[threadIdx.x >> 6][2];
  const uint64_t s0 = r.seg_begin[q], s1 = r.relative ? s0 + r.q_nseg[q] : r.seg_begin[q + 1];
  const uint64_t dst0 = r.relative ? r.byte_begin[q] : 0;
  const uint8_t* __restrict__ codes = im.seq_codes;
  for (uint64_t base = s0; base < s1; base += 64) {
    const uint64_t mine = base + lane;
    uint32_t src = 0, len = 0;
    uint64_t dst = 0;
    if (mine < s1) { src = r.seg_src[mine]; len = r.seg_len[mine]; dst = dst0 + r.seg_dst[mine]; }
    const uint32_t cnt = (uint32_t)((s1 - base) < 64 ? (s1 - base) : 64);
    const uint64_t b0 = wave_bcast64(dst, 0);                                   // first output byte of this batch of pieces
    const uint64_t b1 = wave_bcast64(dst, cnt - 1) + __builtin_amdgcn_readlane(len, cnt - 1);
    s_dst[lane] = lane < cnt ? (uint32_t)(dst - b0) : 0xFFFFFFFFu;
    s_src[lane] = src; s_len[lane] = len;
    if (lane == 0) s_dst[64] = 0xFFFFFFFFu;
    __builtin_amdgcn_wave_barrier();
    const uint64_t g0 = b0 & ~15ULL;                                            // (r.chars is 256-byte aligned: offsets align like addresses)
    for (uint64_t dbase = g0; dbase < b1; dbase += 16ULL * 64) {               // 64 groups = 1 KiB of output per pass
      const uint64_t d = dbase + 16ULL * lane;
      bool odd = false;                                                          // my group is not wholly inside one piece
      if (d < b1) {
        const uint64_t lo = d > b0 ? d : b0, hi = d + 16 < b1 ? d + 16 : b1;
        const uint32_t rel = (uint32_t)(lo - b0);
        uint32_t p = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1)
          if (s_dst[p + step] <= rel) p += step;
        const uint32_t pd = s_dst[p], ps = s_src[p], pl = s_len[p];
        if (hi - lo == 16 && rel + 16 <= pd + pl) {
          uint4 c;
          __builtin_memcpy(&c, codes + ps + (rel - pd), 16);
          uint4 o;
          o.x = __builtin_amdgcn_perm(0x0505054Eu, 0x47544341u, c.x & 0x07070707u);
          o.y = __builtin_amdgcn_perm(0x0505054Eu, 0x47544341u, c.y & 0x07070707u);
          o.z = __builtin_amdgcn_perm(0x0505054Eu, 0x47544341u, c.z & 0x07070707u);
          o.w = __builtin_amdgcn_perm(0x0505054Eu, 0x47544341u, c.w & 0x07070707u);
          typedef unsigned int u32x4_chars_t __attribute__((ext_vector_type(4)));   // written once, not read again here: non-temporal
          __builtin_nontemporal_store(u32x4_chars_t{o.x, o.y, o.z, o.w}, reinterpret_cast<u32x4_chars_t*>(r.chars + d));
        } else odd = true;
      }
      // groups across a piece boundary (and the ragged first / last one): four of them per pass, one BYTE per lane --
      // sixteen lanes per group, each finds its byte's piece by itself
      uint64_t m = __ballot(odd);
      while (m) {
        uint64_t mm = m;
        for (uint32_t t = 0; t < (lane >> 4); ++t) mm &= mm - 1;               // the (lane / 16)-th odd group of this pass
        if (mm) {
          const uint64_t b = dbase + 16ULL * (uint32_t)__builtin_ctzll(mm) + (lane & 15u);
          if (b >= b0 && b < b1) {
            const uint32_t rel = (uint32_t)(b - b0);
            uint32_t p = 0;
#pragma unroll
            for (uint32_t step = 32; step; step >>= 1)
              if (s_dst[p + step] <= rel) p += step;
            r.chars[b] = decode_base(codes[s_src[p] + (rel - s_dst[p])]);
          }
        }
        for (int t = 0; t < 4 && m; ++t) m &= m - 1;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Totals of a result without copying it: {variants reported, their carriers, their REF + ALT bases}, one wave per region
__global__ void __launch_bounds__(256) k_result_totals(DevResult r, unsigned long long* out) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const bool live = q < r.Q;
  const uint64_t n = live ? r.q_nvar[q] : 0, a0 = live ? r.var_begin[q] : 0;
  unsigned long long nv = 0, nc = 0, nb = 0;
  for (uint64_t j = threadIdx.x & 63; j < n; j += 64) {
    const VariantRow v = row_load(r.rows, a0 + j);
    if (row_dropped(v)) continue;
    nv += 1; nc += row_count(v); nb += (uint64_t)v.ref_len + v.alt_len;
  }
  for (int d = 32; d >= 1; d >>= 1) { nv += __shfl_down(nv, d, 64); nc += __shfl_down(nc, d, 64); nb += __shfl_down(nb, d, 64); }
  __shared__ unsigned long long part[3][4];   // one atomic triple per block, not per wave: the three words are one hot line
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = nv; part[1][threadIdx.x >> 6] = nc; part[2][threadIdx.x >> 6] = nb; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long t = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
    if (t) atomicAdd(out + threadIdx.x, t);
  }
}

// sample ids handed over in device memory are checked here (a host array is checked by the host before anything is launched)
__global__ void __launch_bounds__(256) k_check_sample_ids(const uint32_t* sids, uint64_t n, uint32_t num_samples, uint64_t* bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && sids[i] >= num_samples) *bad = 1;
}

// Index::find batched (index.h:119-133)
__global__ void __launch_bounds__(256) k_find(DevImage im, const uint64_t* pos, uint64_t n, uint32_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t p = pos[i];
  if (p < 1) { out[i] = kNone; return; }
  uint64_t rf = (p >= im.ref_length) ? im.R - 1 : (uint64_t)rank1(im, p);
  if (p < im.ref_length) rf = rf == 0 ? 0 : rf - 1;
  out[i] = im.rp_vid[im.rank_to_slot[rf]];
}

}  // namespace vsamd
