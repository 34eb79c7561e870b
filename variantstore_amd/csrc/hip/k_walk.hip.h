// k_walk.hip.h -- query type 4: literal walk, event bitmaps, hold rows, cooperative walk, list claims, hit-list records.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_rows.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Query type 4: get_sample_var_in_ref (query.h:618-729) with its start search
// get_prev_vertex_with_sample (query.h:57-113).  One thread per region walks the
// sample's path literally (get_neighbor_vertex, variant_graph.h:1402-1451) over the
// CSR + vertex table; EMIT=false counts, EMIT=true writes variant headers.  The
// carriers of each reported vertex are expanded afterwards by k_fill_carriers.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool vertex_has_sample(const DevImage& im, uint32_t v, uint32_t sid) {
  if (im.use_bv) return (im.class_rows[(uint64_t)im.v_class[v] * im.wpc + (sid >> 6)] >> (sid & 63)) & 1;
  if (sid == 0) return im.v_ridx[v] != 0;
  const uint64_t b = im.v_car_begin[v];
  for (uint32_t i = 0; i < im.v_ncar[v]; ++i)
    if (im.car_sid[b + i] == sid) return true;
  return false;
}

// ---------------------------------------------------------------------------
// Event bitmaps of query type 4 (DevImage::t4_events + the global t4_irr), built once when an index is opened.
// One wave per tile of 64 consecutive ref-path slots: lane j ORs the class rows of slot j's node and of its
// out-neighbours one 64-sample word at a time, a 64 x 64 bit transpose through 64 ballots turns "samples of a slot"
// into "slots of a sample", and lane t stores the tile's word of sample w * 64 + t.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool slot_is_irregular(const DevImage& im, uint64_t j) {
  // the walk's "last ref neighbour" (next_ref_pos / cur_ref of query.h:640-667) must be the path successor, and there
  // must be one; anything else is walked literally
  if (j + 1 >= im.P) return true;
  const uint32_t v = im.rp_vid[j], succ = im.rp_vid[j + 1];
  uint32_t last_ref = kNone;
  for (uint32_t e = im.row_ptr[v]; e < im.row_ptr[v + 1]; ++e)
    if (im.v_ridx[im.col[e]]) last_ref = im.col[e];
  return last_ref != succ;
}

__global__ void __launch_bounds__(256) k_build_events(DevImage im, uint64_t* events, uint64_t* irr_row) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t tile = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t ntiles = (im.P + 63) >> 6;
  if (tile >= ntiles) return;
  const uint64_t j = tile * 64 + lane;
  const bool valid = j < im.P;
  const uint64_t irr = __ballot(valid && slot_is_irregular(im, j));
  if (lane == 0) irr_row[tile] = irr;   // the sample-independent part, once (round 4: it used to be OR-ed into every sample's row)
  const uint32_t v = valid ? im.rp_vid[j] : 0;
  const uint32_t e0 = valid ? im.row_ptr[v] : 0, e1 = valid ? im.row_ptr[v + 1] : 0;
  const uint32_t wpc = im.wpc;
  for (uint32_t w = 0; w < wpc; ++w) {
    uint64_t word = 0;
    if (valid) {
      word = im.class_rows[(uint64_t)im.v_class[v] * wpc + w];
      for (uint32_t e = e0; e < e1; ++e) word |= im.class_rows[(uint64_t)im.v_class[im.col[e]] * wpc + w];
    }
    uint64_t mine = 0;
#pragma unroll 8
    for (uint32_t b = 0; b < 64; ++b) {
      const uint64_t m = __ballot((word >> b) & 1);
      if (lane == b) mine = m;
    }
    const uint32_t sample = w * 64 + lane;
    if (sample >= 1 && sample < im.num_samples) events[(uint64_t)sample * im.t4_stride + tile] = mine;
  }
}

// explicit-id cohorts (no class rows): the global irregular row, then every carrier record of a slot's node and of its
// out-neighbours sets its sample's bit in the (zeroed) per-sample rows
__global__ void __launch_bounds__(256) k_events_irregular_rows(DevImage im, uint64_t* irr_row) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t tile = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t ntiles = (im.P + 63) >> 6;
  if (tile >= ntiles) return;
  const uint64_t j = tile * 64 + lane;
  const uint64_t irr = __ballot(j < im.P && slot_is_irregular(im, j));
  if (lane == 0) irr_row[tile] = irr;
}
__global__ void __launch_bounds__(256) k_events_explicit(DevImage im, uint64_t* events) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= im.P) return;
  const uint32_t v = im.rp_vid[j];
  const uint64_t c = j >> im.t4_ev_shift;            // (coarse rows: one bit per 2^shift slots)
  const unsigned long long bit = 1ULL << (c & 63);
  const uint32_t e0 = im.row_ptr[v], e1 = im.row_ptr[v + 1];
  for (uint32_t e = e0; e <= e1; ++e) {             // e == e1: the node itself
    const uint32_t u = e < e1 ? im.col[e] : v;
    const uint64_t b = im.v_car_begin[u];
    for (uint32_t i = 0; i < im.v_ncar[u]; ++i) {
      const uint32_t sid = im.car_sid[b + i];
      if (sid >= 1 && sid < im.num_samples) atomicOr((unsigned long long*)&events[(uint64_t)sid * im.t4_stride + (c >> 6)], bit);
    }
  }
}

// Round 6, opt-in (VS_T4_EXACT_ROWS): an explicit-id cohort whose EXACT rows fit the part's HBM gets them (engine.hip: build_t4_rows) -- a bit
// per slot and sample and this hold row, a bit per vertex and sample, set from the carrier records (10,000 samples x 20 M variants: 48 + 73 GB
// of the 288 GB).  The walks then take the class-row cohort's kernels: hold tests are one bit, the cooperative kernels of query types
// 2 / 3 / 5 and the spill-free form of type 4 apply.  The default stays round 4's coarse rows + list look-ups.
__global__ void __launch_bounds__(256) k_hold_explicit(DevImage im, uint64_t* hold) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= im.V) return;
  const uint64_t b = im.v_car_begin[v];
  const uint32_t n = im.v_ncar[v];
  const unsigned long long bit = 1ULL << (v & 63);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t sid = im.car_sid[b + i];
    if (sid >= 1 && sid < im.num_samples) atomicOr((unsigned long long*)&hold[(uint64_t)sid * im.t4_hold_stride + (v >> 6)], bit);
  }
}

// Hold rows (DevImage::t4_hold): one wave per tile of 64 consecutive vertex ids, the same transpose as k_build_events.
__global__ void __launch_bounds__(256) k_build_hold(DevImage im, uint64_t* hold) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t tile = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (tile >= (im.V + 63) >> 6) return;
  const uint64_t v = tile * 64 + lane;
  const bool valid = v < im.V;
  const uint32_t cls = valid ? im.v_class[v] : 0;
  for (uint32_t w = 0; w < im.wpc; ++w) {
    const uint64_t word = valid ? im.class_rows[(uint64_t)cls * im.wpc + w] : 0;
    uint64_t mine = 0;
#pragma unroll 8
    for (uint32_t b = 0; b < 64; ++b) {
      const uint64_t m = __ballot((word >> b) & 1);
      if (lane == b) mine = m;
    }
    const uint32_t sample = w * 64 + lane;
    if (sample >= 1 && sample < im.num_samples) hold[(uint64_t)sample * im.t4_hold_stride + tile] = mine;
  }
}
// Walk records (device_image.hpp): one step of a path walk reads the current vertex in one 32-byte record and each
// neighbour in one 16-byte edge record instead of gathering a dozen 4-byte fields from as many arrays.
struct WalkVertex { uint32_t row_begin, deg, ridx, off, len, cls, ncar; };
__device__ __forceinline__ WalkVertex walk_vertex(const DevImage& im, uint32_t v) {
  const uint4 a = im.w_vertex[2 * (uint64_t)v], b = im.w_vertex[2 * (uint64_t)v + 1];
  return WalkVertex{a.x, a.y, a.z, a.w, b.x, b.y, b.z};
}
struct WalkEdge { uint32_t nbr, ridx, cls; };
__device__ __forceinline__ WalkEdge walk_edge(const DevImage& im, uint32_t e) {
  const uint4 a = im.w_edge[2 * (uint64_t)e];
  return WalkEdge{a.x, a.y, a.z};
}
// the whole edge record: the neighbour and the neighbour's own vertex record (stepping onto it needs no look-up)
__device__ __forceinline__ WalkEdge walk_edge_full(const DevImage& im, uint32_t e, WalkVertex& nv) {
  const uint4 a = im.w_edge[2 * (uint64_t)e], b = im.w_edge[2 * (uint64_t)e + 1];
  nv = WalkVertex{a.w, b.x, a.y, b.y, b.z, a.z, b.w};
  return WalkEdge{a.x, a.y, a.z};
}
// vertex_has_sample on what a record already holds (class rows; explicit-id cohorts fall back to the carrier pool)
__device__ __forceinline__ bool record_has_sample(const DevImage& im, uint32_t v, uint32_t ridx, uint32_t cls, uint32_t sid) {
  if (im.use_bv) return (im.class_rows[(uint64_t)cls * im.wpc + (sid >> 6)] >> (sid & 63)) & 1;
  if (sid == 0) return ridx != 0;
  const uint64_t b = im.v_car_begin[v];
  for (uint32_t i = 0; i < im.v_ncar[v]; ++i)
    if (im.car_sid[b + i] == sid) return true;
  return false;
}

// MODE 0 counts, MODE 1 writes the variant headers at the scanned offsets (a second walk), MODE 2 walks ONCE:
// it records every reported vertex in a scratch list whose per-region capacity is the region's type-6 slot count
// (a sample's variants are branches of the same ref-path range) and flags an overflow instead of writing past it;
// k_emit_from_walk then lays the headers out without walking again.
// Round 5: the walking query types wait for the host ONCE per batch.  The scratch of the recording walk is sized from what the
// handle's previous batch of the kind needed (+ 1/8); the recording walk itself, first thing, compares the capacities' total (the
// scan in front of it left it in device memory) with that allocation and refuses the batch (overflow = 2) when it does not fit --
// or when an id that arrived in device memory is out of range (3: k_walk_setup / k_check_sample_ids left the word) -- and then
// the walk, k_t4_claim and the emitters touch nothing beyond the per-region arrays: the host reads the word together with the
// batch's sizes and redoes a refused batch with an exact allocation (one more wait, once).  (Until the end of round 5 a
// one-thread kernel between scan and walk gave the verdict: one launch more in a string of dependent launches.)
struct WalkAdmit {
  const uint64_t* cap_total;   // the capacities' total, on the device; NULL: the allocation is exact, nothing to admit
  uint64_t cap_alloc;          // what the scratch holds
  const uint64_t* bad_ids;     // non-zero: a sample id of the batch is out of range (NULL: the host checked the ids)
};
__device__ __forceinline__ uint32_t admit_verdict(const WalkAdmit& a) {
  if (!a.cap_total) return 0;
  if (a.bad_ids && *a.bad_ids) return 3;
  return *a.cap_total > a.cap_alloc ? 2u : 0u;
}
struct WalkScratch {
  const uint64_t* cap_begin;   // [Q+1] exclusive scan of the capacities
  uint64_t* pos;
  uint32_t *cur, *ro, *rl, *ao, *al;
  uint64_t* overflow;          // set to 1 when a region outgrew its capacity (the host then takes the two-walk path); 2 / 3: the batch
                               // was refused before the walk started (walk_void)
  unsigned long long* stats;   // tuning builds (VS_TUNING): 16 counters of k_sample_walk (iteration counts, device-clock ticks); else NULL
  WalkAdmit admit;
};
#ifdef VS_TUNING
#define VS_WALK_STAT(i, v) do { if (ws.stats) atomicAdd(&ws.stats[i], (unsigned long long)(v)); } while (0)
#define VS_WALK_STATMAX(i, v) do { if (ws.stats) atomicMax(&ws.stats[i], (unsigned long long)(v)); } while (0)
#define VS_WALK_CLOCK() (ws.stats ? wall_clock64() : 0ULL)
#else
#define VS_WALK_STAT(i, v) do { } while (0)
#define VS_WALK_STATMAX(i, v) do { } while (0)
#define VS_WALK_CLOCK() 0ULL
#endif

// first thing in a recording walk: a refused batch reports no rows (its regions' capacities are what the bounds left in q_nvar;
// every thread of the launch reaches the same verdict, the writers leave it for the host)
__device__ __forceinline__ bool walk_void(const DevResult& r, const WalkScratch& ws, uint64_t q, bool writer) {
  if (!ws.overflow) return false;
  const uint32_t verdict = admit_verdict(ws.admit);
  if (!verdict) return false;
  if (writer) {
    if (q < r.Q) { r.q_nvar[q] = 0; r.q_ncar[q] = 0; r.var_count[q] = 0; }
    *ws.overflow = verdict;
  }
  return true;
}

// What get_sample_var_in_ref reports for a vertex on the sample's path (query.h:680-704), from the walk's state at that
// vertex: kind 0 insertion (ref_pos == next_ref_pos), 1 deletion (the vertex is a ref vertex: ref = sequence of
// find(ref_pos - 1)), 2 substitution (ref = sequence of the previous step's last ref neighbour).  Resolved where it is
// cheap: by the wide k_emit_from_walk for the recording walk, in place for the two-walk fallback.
struct WalkVariant { uint64_t pos; uint32_t ro, rl, ao, al; };
__device__ __forceinline__ WalkVariant resolve_walk_variant(const DevImage& im, uint32_t kind, uint32_t cur, uint64_t ref_pos, uint32_t cur_ref_v) {
  WalkVariant o{0, 0, 0, 0, 0};
  if (kind == 1) {   // (the walk only records a deletion when ref_pos >= 2)
    const uint64_t p = ref_pos - 1;
    const uint64_t rf = (p >= im.ref_length) ? im.R - 1 : (uint64_t)rank1(im, p) - 1;
    const uint32_t fv = im.rp_vid[im.rank_to_slot[rf]];
    o.pos = im.v_ridx[fv]; o.ro = im.v_off[fv]; o.rl = im.v_len[fv];
  } else {
    o.pos = kind == 0 ? ref_pos - 1 : ref_pos;
    o.ao = im.v_off[cur]; o.al = im.v_len[cur];
    if (kind == 2 && cur_ref_v != kNone) { o.ro = im.v_off[cur_ref_v]; o.rl = im.v_len[cur_ref_v]; }
  }
  return o;
}

// One 64-bit-word cache in front of a per-sample bit row (event rows over slots, hold rows over vertex ids): consecutive
// look-ups of a walk fall into the same word more often than not.
// Two compact forms for EXPLICIT-ID cohorts (somatic-like: a handful of carriers per variant, thousands of samples), whose
// full rows are O(samples x slots) -- 125 GB at 10,000 samples x 20 M variants (round 4):
//  * `sh`: one bit of an EVENT row stands for 2^sh consecutive slots ("some slot of the block has an event").  Everything
//    an event row is used for is an acceleration around literal code -- a walk lands on a slot and steps literally, a
//    search tests a candidate literally -- so any SUPERSET of the true events is exact; a coarse bit just makes up to 2^sh
//    cheap literal steps where the full row made one.
//  * `x_sid`: a HOLD row that is not there -- "does vertex v hold the sample" is answered from v's explicit carrier list
//    (v_car_begin / v_ncar / car_sid: most neighbours of a walk are ref vertices without carriers, the rest hold ~8 ids).
struct BitRow {
  const uint64_t* __restrict__ row;
  uint32_t w;          // index of the cached word (kNone: nothing cached)
  uint64_t word;
  uint32_t sh = 0;
  const uint32_t* __restrict__ x_sid = nullptr;
  const uint64_t* __restrict__ x_begin = nullptr;
  const uint32_t* __restrict__ x_ncar = nullptr;
  uint32_t x_id = 0;
  __device__ __forceinline__ uint64_t at(uint32_t wi) {
    if (wi != w) { w = wi; word = row[wi]; }
    return word;
  }
  __device__ __forceinline__ bool bit(uint32_t i) {
    if (x_sid) return holds_explicit(i, x_ncar[i]);
    const uint32_t c = i >> sh;
    return (at(c >> 6) >> (c & 63)) & 1;
  }
  // the same for a vertex whose number of carriers the caller already has in a record (a hold row that is not there then
  // costs nothing for a vertex without carriers -- every ref vertex -- and one look-up of the list's start otherwise)
  __device__ __forceinline__ bool bit_n(uint32_t i, uint32_t ncar) {
    if (x_sid) return holds_explicit(i, ncar);
    return (at(i >> 6) >> (i & 63)) & 1;
  }
  // (eight ids per round trip, two 16-byte loads -- a list holds ~8: an id at a time with an early exit was up to nine
  //  DEPENDENT look-ups per neighbour, most of a walk's chain on the 10,000-sample cohort.  Ids behind the list's end are the
  //  next vertex's or the pool's slack, as in the expansion's group loads: masked by the count.)
  __device__ __forceinline__ bool holds_explicit(uint32_t i, uint32_t n) const {
    if (!n) return false;
    const uint64_t b0 = x_begin[i];
    bool hit = false;
    for (uint32_t k = 0; k < n && !hit; k += 8) {
      uint4 a, b;
      __builtin_memcpy(&a, x_sid + b0 + k, 16);
      __builtin_memcpy(&b, x_sid + b0 + k + 4, 16);
      const uint32_t m = n - k;
      hit = (a.x == x_id) | (a.y == x_id && m > 1) | (a.z == x_id && m > 2) | (a.w == x_id && m > 3) |
            (b.x == x_id && m > 4) | (b.y == x_id && m > 5) | (b.z == x_id && m > 6) | (b.w == x_id && m > 7);
    }
    return hit;
  }
  // first index >= m whose bit is set, or `limit` when there is none below it (m < limit); a coarse row answers with m
  // itself when m's block is set, else with the first index of the next set block
  __device__ __forceinline__ uint32_t next(uint32_t m, uint32_t limit) {
    const uint32_t cm = m >> sh, c_end = ((limit - 1) >> sh) + 1;
    uint32_t wi = cm >> 6;
    const uint32_t w_end = (c_end + 63) >> 6;
    uint64_t x = at(wi) & (~0ULL << (cm & 63));
    while (!x) {
      if (++wi >= w_end) return limit;
      x = at(wi);
    }
    const uint32_t c = (wi << 6) + (uint32_t)__builtin_ctzll(x);
    if (c >= c_end) return limit;
    const uint32_t k = c == cm ? m : c << sh;
    return k < limit ? k : limit;
  }
  // The same for rows that are mostly clear over long ranges -- the break row over the 1024 slots a sequence walk looks
  // ahead, on a cohort whose samples have an event every few thousand slots: eight words per round trip instead of one
  // DEPENDENT load per 64 slots (16 in a row per jump: a third of such a walk's chain).  Costs sixteen registers: not for
  // the kernels that are short of them.
  __device__ __forceinline__ uint32_t next_wide(uint32_t m, uint32_t limit) {
    const uint32_t cm = m >> sh, c_end = ((limit - 1) >> sh) + 1;
    uint32_t wi = cm >> 6;
    const uint32_t w_end = (c_end + 63) >> 6;
    uint64_t x = at(wi) & (~0ULL << (cm & 63));
    while (!x) {
      if (++wi >= w_end) return limit;
      uint64_t q[8];
#pragma unroll
      for (uint32_t t = 0; t < 8; ++t) q[t] = wi + t < w_end ? row[wi + t] : 0ULL;
      uint32_t hit = 8;
#pragma unroll
      for (uint32_t t = 8; t-- > 0;) if (q[t]) hit = t;
      if (hit == 8) { wi += 7; continue; }
#pragma unroll
      for (uint32_t t = 0; t < 8; ++t) if (t == hit) x = q[t];
      wi += hit;
      w = wi; word = x;
    }
    const uint32_t c = (wi << 6) + (uint32_t)__builtin_ctzll(x);
    if (c >= c_end) return limit;
    const uint32_t k = c == cm ? m : c << sh;
    return k < limit ? k : limit;
  }
};
// the rows of one sample (NULL rows: the index has none, or the walk does not use them)
__device__ __forceinline__ BitRow sample_event_row(const DevImage& im, uint32_t sid, bool use) {
  BitRow r{use ? im.t4_events + (uint64_t)sid * im.t4_stride : nullptr, kNone, 0};
  r.sh = im.t4_ev_shift;
  return r;
}
__device__ __forceinline__ BitRow sample_hold_row(const DevImage& im, uint32_t sid, bool use) {
  BitRow r{(use && im.t4_hold) ? im.t4_hold + (uint64_t)sid * im.t4_hold_stride : nullptr, kNone, 0};
  if (use && !im.t4_hold) { r.x_sid = im.car_sid; r.x_begin = im.v_car_begin; r.x_ncar = im.v_ncar; r.x_id = sid; }
  return r;
}
__device__ __forceinline__ BitRow irregular_row(const DevImage& im, bool use) { return BitRow{use ? im.t4_irr : nullptr, kNone, 0}; }
// the global irregular row's word wi, from slot `from` on
__device__ __forceinline__ uint64_t irr_word_from(const uint64_t* __restrict__ irr, uint32_t wi, uint32_t from) {
  if (wi < (from >> 6)) return 0;
  const uint64_t x = irr[wi];
  return wi == (from >> 6) ? x & (~0ULL << (from & 63)) : x;
}
// (Round 5, tried and removed: the cooperative kernels of types 2 / 3 / 5 over the COARSE event rows of explicit-id cohorts -- a coarse
//  bit spread over its eight slots, every slot of a set block a one-step episode.  Text-exact, and slower than the one-lane kernels on
//  the 10,000-sample cohort: 71 / 72 / 84 against 80 / 82 / 98 M regions/s.  A region there has one or two events; its time is the
//  backward search over thousands of ranks, which a group of eight lanes runs redundantly -- eight times the instructions.)
typedef BitRow EventRow;

// ---- the walk of get_sample_var_in_ref as reusable pieces (serial kernel k_sample_walk, cooperative k_sample_walk_coop) ----
// Two data paths, chosen per region: BLOB (the sample has event + hold rows: records from the walk blob, "does v hold
// the sample" from the hold row, jumps over uneventful runs) and plain (sample 0 = "ref", or an index without the rows:
// the round-2 records, class rows, every vertex visited).  WalkVertex::row_begin indexes the blob resp. w_edge.
struct WalkCtx { uint32_t sid; uint64_t x, y; bool use_ev; uint32_t limit; uint32_t irr_from; };   // irr_from: irregular slots count as events from here on (walk_prologue)
// The next slot >= m (m < cx.limit) at which the type-4 walk must step literally: the sample's next event, or an irregular slot
// once the stop slot is within reach (walk_prologue: irr_from); cx.limit when neither comes first.
template <bool WIDE = false>   // WIDE: rows that are clear over long ranges (explicit-id cohorts): BitRow::next_wide
__device__ __forceinline__ uint32_t next_walk_event(BitRow& ev, BitRow& irr, const WalkCtx& cx, uint32_t m) {
  uint32_t k = WIDE ? ev.next_wide(m, cx.limit) : ev.next(m, cx.limit);
  const uint32_t m2 = m > cx.irr_from ? m : cx.irr_from;
  if (m2 < k) k = WIDE ? irr.next_wide(m2, k) : irr.next(m2, k);     // (k itself when no irregular slot lies before it)
  return k;
}
struct WalkSt { uint32_t cur; WalkVertex wc; uint64_t ref_pos; uint32_t cur_ref_v, cur_slot1; };   // cur_slot1: ref-path slot + 1 of cur, 0 = off the path
struct WalkEmit { uint64_t ref_pos; uint32_t cur, kind, cur_ref_v, c; };   // the walk's state at a reported vertex (-> resolve_walk_variant)

constexpr uint32_t kStepEdges = 3;   // out-edges a step reads together (higher degrees -- rare -- one at a time)
struct StepOut {
  uint64_t next_ref_pos;      // ref index of the LAST ref neighbour (unchanged if there is none)
  uint32_t next_ref_v;        // that neighbour (kNone: none)
  uint32_t nxt;               // get_neighbor_vertex: first neighbour holding the sample, else the ref neighbour with the smallest index; 0 = none
  uint32_t nxt_slot1;         // its ref-path slot + 1 (0: not on the path)
  WalkVertex wn;              // its vertex record
};
// does vertex v hold the sample (get_sample_from_vertex_if_exists)?  BLOB: the sample's hold row; else class row / carrier list
template <bool BLOB>
__device__ __forceinline__ bool walk_holds(const DevImage& im, BitRow& hold, uint32_t v, uint32_t ridx, uint32_t cls, uint32_t ncar, uint32_t sid) {
  if (BLOB) return hold.bit_n(v, ncar);
  return record_has_sample(im, v, ridx, cls, sid);
}
// One literal step's view of a vertex's out-edges: all edge records are requested together (one or two lines of the
// blob), the hold bits of the neighbours come from a word that is usually cached already, and the reference's in-order
// decision logic (get_neighbor_vertex, variant_graph.h:1402-1451; "last ref neighbour", query.h:640-667) runs over registers.
template <bool BLOB>
__device__ __forceinline__ StepOut walk_step_edges(const DevImage& im, BitRow& hold, const WalkVertex& wc, uint32_t sid, uint64_t next_ref_pos_default) {
  StepOut o;
  o.next_ref_pos = next_ref_pos_default; o.next_ref_v = kNone; o.nxt = 0; o.nxt_slot1 = 0; o.wn = WalkVertex{};
  uint32_t min_idx = 0xFFFFFFFFu;
  bool nxt_by_sample = false;
  if (BLOB && wc.deg <= kStepEdges) {
    uint4 a[kStepEdges], b[kStepEdges];
#pragma unroll
    for (uint32_t i = 0; i < kStepEdges; ++i) {
      a[i] = uint4{0, 0, 0, 0}; b[i] = uint4{0, 0, 0, 0};
      if (i < wc.deg) { a[i] = im.wblob[2 * (uint64_t)(wc.row_begin + i)]; b[i] = im.wblob[2 * (uint64_t)(wc.row_begin + i) + 1]; }
    }
#pragma unroll
    for (uint32_t i = 0; i < kStepEdges; ++i) {   // (predicated, not `break`: the arrays must stay in registers)
      const bool on = i < wc.deg;
      const uint32_t n = a[i].x, nr = a[i].y;
      if (on && nr) { o.next_ref_pos = nr; o.next_ref_v = n; }  // last ref neighbour wins
      if (on && !nxt_by_sample) {
        const bool holds = hold.bit_n(n, b[i].w);
        if (holds || (nr && min_idx > nr)) {
          o.nxt = n; o.nxt_slot1 = b[i].y;
          o.wn = WalkVertex{a[i].w, b[i].x, a[i].y, 0u, b[i].z, a[i].z, b[i].w};
          if (holds) nxt_by_sample = true; else min_idx = nr;
        }
      }
    }
    return o;
  }
  for (uint32_t e = wc.row_begin; e < wc.row_begin + wc.deg; ++e) {
    uint4 a, b;
    if (BLOB) { a = im.wblob[2 * (uint64_t)e]; b = im.wblob[2 * (uint64_t)e + 1]; }
    else { a = im.w_edge[2 * (uint64_t)e]; b = im.w_edge[2 * (uint64_t)e + 1]; }
    const uint32_t n = a.x, nr = a.y;
    if (nr) { o.next_ref_pos = nr; o.next_ref_v = n; }
    if (!nxt_by_sample) {  // get_neighbor_vertex: first neighbour holding the sample, else smallest ref index
      const bool holds = sid != 0 && walk_holds<BLOB>(im, hold, n, nr, a.z, b.w, sid);
      if (holds || (nr && min_idx > nr)) {
        o.nxt = n;
        o.nxt_slot1 = BLOB ? b.y : 0u;
        o.wn = BLOB ? WalkVertex{a.w, b.x, a.y, 0u, b.z, a.z, b.w} : WalkVertex{a.w, b.x, a.y, b.y, b.z, a.z, b.w};
        if (holds) nxt_by_sample = true; else min_idx = nr;
      }
    }
  }
  return o;
}

// One iteration of the reference's loop body (query.h:640-720) at st.cur: is the vertex reported, then the step to the
// next vertex of the sample's path.  `done`: the path iterator has no next vertex.
template <bool BLOB>
__device__ __forceinline__ bool walk_literal_step(const DevImage& im, const WalkCtx& cx, BitRow& hold, WalkSt& st, WalkEmit& em, bool& done) {
  // does cur hold the sample?  (requested before the edges: it is independent of them)
  const bool cur_holds = st.ref_pos >= cx.x && walk_holds<BLOB>(im, hold, st.cur, st.wc.ridx, st.wc.cls, st.wc.ncar, cx.sid);
  const StepOut so = walk_step_edges<BLOB>(im, hold, st.wc, cx.sid, st.ref_pos + st.wc.len);
  bool emit = false;
  if (cur_holds) {
    const uint32_t kind = st.ref_pos == so.next_ref_pos ? 0u : (st.wc.ridx ? 1u : 2u);
    if (!(kind == 1 && st.ref_pos < 2)) {   // a deletion at ref_pos 1 has no find(ref_pos - 1): skipped
      em = WalkEmit{st.ref_pos, st.cur, kind, st.cur_ref_v, st.wc.ncar};
      emit = true;
    }
  }
  st.cur_ref_v = so.next_ref_v;
  st.ref_pos = so.next_ref_pos;
  done = so.nxt == 0;  // no neighbour: the path iterator is done
  st.cur = so.nxt; st.wc = so.wn; st.cur_slot1 = so.nxt_slot1;
  return emit;
}
// on a ref-path node, in step with it (ref_pos == its index), before the stop slot: where a jump may start / an episode ends
__device__ __forceinline__ bool walk_in_step(const WalkCtx& cx, const WalkSt& st) {
  return cx.use_ev && st.cur_slot1 && st.ref_pos == st.wc.ridx && st.cur_slot1 - 1 < cx.limit;
}
// arrival at event slot k "in step": {k's node, its index}; cur_ref is not read before the step overwrites it (it only
// enters a substitution, and a ref-path node is never reported as one)
__device__ __forceinline__ void walk_arrive_at_slot(const DevImage& im, WalkSt& st, uint32_t k) {
  const uint64_t h = im.blob_of_slot[k];
  const uint4 ra = im.wblob[2 * h], rb = im.wblob[2 * h + 1];   // header record of slot k
  st.cur = rb.w;
  st.wc = WalkVertex{ra.x, ra.y, ra.z, 0u, rb.x, rb.y, rb.z};
  st.ref_pos = st.wc.ridx;
  st.cur_ref_v = st.cur;
  st.cur_slot1 = k + 1;
}
// the blob-mode vertex record of an arbitrary vertex (rare: the walk's start at the head of the path, slow paths)
__device__ __forceinline__ WalkVertex blob_vertex(const DevImage& im, uint32_t v) {
  WalkVertex w = walk_vertex(im, v);
  w.row_begin = im.blob_row[v];
  return w;
}

// get_prev_vertex_with_sample (query.h:57-113) from find(x)'s rank: the start state of the walk
template <bool BLOB>
__device__ __forceinline__ void walk_start_search(const DevImage& im, const WalkCtx& cx, BitRow& ev, BitRow& hold, uint64_t rank0, WalkSt& st,
                                                  uint32_t& st_iters, uint32_t& st_lit) {
  const uint32_t sid = cx.sid;
  uint64_t rank = rank0;
  uint64_t ref_pos = 1;
  uint32_t start_v = 0, start_slot1 = 0;
  WalkVertex wc{};
  bool have_start_rec = false;   // the search found start_v through an edge record that carries its vertex record
  bool jump = BLOB, jumped = false;
  while (true) {
    ++st_iters;
    // Index::previous(rank) is the first ref-path slot of rank - 1; rk_back holds it together with that node's
    // out-degree (what the scan below counts the rank down by): one 8-byte record per iteration of the jumped form
    const uint2 back = im.rk_back[rank == 0 ? 0 : rank - 1];
    const uint32_t pslot = back.x;
    if (rank <= 1) { ref_pos = 1; start_v = im.rp_vid[pslot]; have_start_rec = false; break; }
    if (BLOB && jump && !ev.bit(pslot)) {
      // no out-neighbour of this node holds the sample: the scan below would find nothing and count the rank
      // down once per neighbour
      rank = rank > back.y ? rank - back.y : 0;
      jumped = true;
      continue;
    }
    ++st_lit;
    bool found = false, had_ref = false;
    const uint32_t deg = back.y;
    if (BLOB) {
      const uint32_t rb0 = im.blob_of_slot[pslot] + 1;   // the edge records follow the slot's header
      for (uint32_t e = rb0; e < rb0 + deg; ++e) {
        const uint4 a = im.wblob[2 * (uint64_t)e], b = im.wblob[2 * (uint64_t)e + 1];
        if (a.y) { ref_pos = a.y; had_ref = true; }
        if (hold.bit_n(a.x, b.w)) {
          start_v = a.x; found = true; have_start_rec = true; start_slot1 = b.y;
          wc = WalkVertex{a.w, b.x, a.y, 0u, b.z, a.z, b.w};
        }
      }
      rank = rank > deg ? rank - deg : 0;   // one count per neighbour (the reference's unsigned counter would wrap: clamped)
    } else {
      const uint32_t v = im.rp_vid[pslot];
      const uint32_t rb0 = im.row_ptr[v];
      for (uint32_t e = rb0; e < rb0 + deg; ++e) {
        const WalkEdge ed = walk_edge(im, e);
        if (ed.ridx) { ref_pos = ed.ridx; had_ref = true; }
        if (record_has_sample(im, ed.nbr, ed.ridx, ed.cls, sid)) { start_v = ed.nbr; found = true; }
        rank = rank ? rank - 1 : 0;
      }
    }
    if (found) {
      // ref_pos is the last ref neighbour seen in ANY iteration so far: when this node has none of its own and
      // iterations were jumped over, the value is not known -- search again, literally (a node without a ref
      // neighbour is the end of the path: practically never)
      if (!had_ref && jumped) { jump = false; jumped = false; rank = rank0; ref_pos = 1; start_v = 0; continue; }
      break;
    }
  }
  st.cur = start_v;
  st.ref_pos = ref_pos;
  st.cur_ref_v = kNone;  // cur_ref: the last ref neighbour of the previous vertex (its sequence; none = empty string)
  if (have_start_rec) { st.wc = wc; st.cur_slot1 = start_slot1; }
  else if (BLOB) {       // afterwards the record of a vertex arrives with the edge the walk takes to it
    st.wc = blob_vertex(im, start_v);
    st.cur_slot1 = im.w_vertex[2 * (uint64_t)start_v + 1].w;
  } else { st.wc = walk_vertex(im, start_v); st.cur_slot1 = 0; }
}

// Index::is_empty (index.h:150-166), find(pos, rank) and -- for the event-bitmap walk -- the stop slot, all from two ranks
// requested together.  Returns the region flag (0: walk), fills rank0 and cx.limit.
__device__ __forceinline__ uint8_t walk_prologue(const DevImage& im, WalkCtx& cx, uint64_t& rank0) {
  if (cx.x < 1) return kRegionInvalid;
  const RankLoads lx = rank1_issue(im, cx.x), ly = rank1_issue(im, cx.y ? cx.y - 1 : 0);
  const uint32_t rx = rank1_finish(lx), ry = rank1_finish(ly);
  bool empty = false;
  if (cx.x > im.ref_length) empty = true;
  else if (rx >= im.R) empty = true;
  else if (!((uint64_t)im.idx_pos[rx] - 1 <= cx.y)) empty = true;
  if (empty) return kRegionEmpty;
  rank0 = (cx.x >= im.ref_length) ? im.R - 1 : (uint64_t)rx - 1;  // find(pos, rank)
  // first slot whose node starts at or after y: a walk that reaches it in step with the reference stops there
  cx.limit = (cx.use_ev && cx.y >= 1) ? im.rank_to_slot[ry < im.R ? ry : im.R] : 0;
  // An irregular node's step sets ref_pos to its last ref neighbour's index for ONE step: read only if the next vertex is
  // reported (then the node has an out-neighbour holding the sample: an event of its own) or by the stop test ref_pos >= y
  // (then that neighbour lies at or beyond the stop slot: the node is within irr_reach slots of it).  Everywhere else an
  // irregular-only slot is jumped over like any uneventful one (round 4; round 3 stopped at every ~40th slot).
  cx.irr_from = cx.limit > im.t4_irr_reach ? cx.limit - im.t4_irr_reach : 0;
  return 0;
}

// The serial walk of one region: the reference's loop, with jumps over uneventful runs in BLOB mode.  `sink(em)` takes
// each reported vertex.
template <bool BLOB, bool WIDE = false, typename Sink>
__device__ __forceinline__ void walk_serial(const DevImage& im, const WalkCtx& cx, BitRow& ev, BitRow& irr, BitRow& hold, WalkSt& st, Sink&& sink,
                                            uint32_t& st_jumps, uint32_t& st_steps) {
  bool done = false;
  while (!done) {
    if (st.ref_pos >= cx.y) break;
    if (BLOB && walk_in_step(cx, st)) {
      // On a ref-path node, in step with it (ref_pos == its index): up to the next event slot k the literal loop
      // would take the default step node by node -- no neighbour holds the sample (next = path successor), the
      // node itself does not (nothing emitted), every node is regular (ref_pos and cur_ref follow the path) -- and
      // stop at `limit` if that comes first.
      const uint32_t k = next_walk_event<WIDE>(ev, irr, cx, st.cur_slot1 - 1);
      if (k != st.cur_slot1 - 1) {
        if (k >= cx.limit) break;
        walk_arrive_at_slot(im, st, k);
        ++st_jumps;
      }
    }
    ++st_steps;
    WalkEmit em;
    if (walk_literal_step<BLOB>(im, cx, hold, st, em, done)) sink(em);
  }
}

template <int MODE>
__global__ void __launch_bounds__(64) k_sample_walk(DevImage im, DevResult r, uint32_t sid_all, const uint32_t* sid_per_region,
                                                    WalkScratch ws) {
  constexpr bool EMIT = MODE == 1;
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (MODE == 2 && walk_void(r, ws, q, true)) return;
  if (q >= r.Q) return;
  WalkCtx cx{sid_per_region ? sid_per_region[q] : sid_all, r.regions[2 * q], r.regions[2 * q + 1], false, 0, 0};
  // Event and hold rows of this sample (DevImage::t4_events, t4_hold): clear event bits are ref-path slots where neither
  // the node nor any of its out-neighbours holds the sample and the node is regular -- the reference's loops provably do
  // nothing there but step on, so both the backward search and the walk jump over them.  Everything that happens at a
  // set bit is the literal code.
  cx.use_ev = im.t4_events && cx.sid != 0;
  uint64_t nvar = 0, ncar = 0, ncar_kept = 0, rank0 = 0;
  const uint8_t fl = walk_prologue(im, cx, rank0);
  if (!fl) {
    BitRow ev = sample_event_row(im, cx.sid, cx.use_ev);      // the sample's own events: the search, the walk's jumps
    BitRow irr = irregular_row(im, cx.use_ev);                // | the irregular slots near the stop slot (next_walk_event)
    BitRow hold = sample_hold_row(im, cx.sid, cx.use_ev);
    const uint64_t t_s0 = VS_WALK_CLOCK();
    uint32_t st_iters = 0, st_lit = 0, st_jumps = 0, st_steps = 0;
    WalkSt st;
    if (cx.use_ev) walk_start_search<true>(im, cx, ev, hold, rank0, st, st_iters, st_lit);
    else walk_start_search<false>(im, cx, ev, hold, rank0, st, st_iters, st_lit);
    const uint64_t t_s1 = VS_WALK_CLOCK();
    // ---- walk the sample's path ----
    const uint64_t a0 = EMIT ? r.var_begin[q] : 0;
    const uint64_t cb = EMIT ? r.car_base[q] : 0;
    const uint64_t s0 = MODE == 2 ? ws.cap_begin[q] : 0, scap = MODE == 2 ? ws.cap_begin[q + 1] - s0 : 0;
    auto sink = [&](const WalkEmit& em) {
      if (EMIT) {
        const WalkVariant wv = resolve_walk_variant(im, em.kind, em.cur, em.ref_pos, em.cur_ref_v);
        const uint64_t a = a0 + nvar;
        row_store(r.rows, a, (uint32_t)wv.pos, wv.ro, wv.rl, wv.ao, wv.al, em.c, false, cb + ncar);
        r.r_class[a] = im.v_src[em.cur]; r.r_gt0[a] = im.v_car_begin[em.cur];
      }
      if (MODE == 2) {   // the walk's state at the vertex; k_emit_from_walk turns it into the row
        if (nvar < scap) {
          const uint64_t s = s0 + nvar;
          ws.pos[s] = em.ref_pos; ws.cur[s] = em.cur; ws.ro[s] = em.kind; ws.rl[s] = em.cur_ref_v;
        } else *ws.overflow = 1;
      }
      nvar++; ncar += pad_car(em.c); ncar_kept += em.c;
    };
    if (cx.use_ev) walk_serial<true>(im, cx, ev, irr, hold, st, sink, st_jumps, st_steps);
    else walk_serial<false>(im, cx, ev, irr, hold, st, sink, st_jumps, st_steps);
    const uint64_t t_s2 = VS_WALK_CLOCK();
    VS_WALK_STAT(0, 1); VS_WALK_STAT(1, st_iters); VS_WALK_STAT(2, st_lit); VS_WALK_STAT(3, st_jumps); VS_WALK_STAT(4, st_steps);
    VS_WALK_STAT(5, t_s1 - t_s0); VS_WALK_STAT(6, t_s2 - t_s1); VS_WALK_STAT(7, nvar);
    VS_WALK_STATMAX(8, st_iters); VS_WALK_STATMAX(9, st_steps); VS_WALK_STATMAX(10, t_s1 - t_s0); VS_WALK_STATMAX(11, t_s2 - t_s1);
    (void)st_lit; (void)st_jumps; (void)st_steps; (void)st_iters; (void)t_s0; (void)t_s1; (void)t_s2;
  }
  if (!EMIT) { r.q_flags[q] = fl; r.q_g0[q] = 0; r.q_nvar[q] = nvar; r.q_ncar[q] = ncar; }
  else { r.var_count[q] = nvar; r.q_ncar[q] = ncar_kept; }
}

// ---------------------------------------------------------------------------
// Cooperative form of the recording walk: SUB (16 or 8) lanes per region.
//
// Between two events a walk is in step with the ref path, and what it does from an event slot on depends only on
// {slot's node, its index} (walk_arrive_at_slot) -- so the EPISODES of a region (event slot -> literal steps until the
// walk is in step again, or ends) are independent of each other and run in parallel, one per lane, speculatively from
// every event slot of the region's range.  The group then follows the chain of hand-overs in registers: the head
// episode (from the backward search's start state) ends in step at some slot; the first event at or after it is the
// next episode that really happens; it ends in step at its own slot; and so on until an episode ends the walk or no
// event is left below the stop slot.  Episodes the chain skips (events on ref nodes the sample's path bypasses) are
// discarded.  The accepted episodes' reports are compacted into the region's scratch list in order.
// The prologue, the backward search and the head run redundantly in all 16 lanes (same addresses: one request), so
// a wave diverges four ways instead of sixty-four; an episode that outgrows its registers (more than kEpEmits reports
// or kEpSteps steps -- not seen) sends its region through the serial loop, again redundantly in the 16 lanes.
// ---------------------------------------------------------------------------
constexpr uint32_t kEpEmits = 4, kEpSteps = 12;
constexpr uint32_t kHopAfter = 4;        // windows of the backward search before it goes over to the event row's set bits
constexpr uint32_t kSearchWindow = 32;   // ranks of the backward search's chain read per round trip (a multiple of SUB, at most 32)

template <uint32_t SUB>
__device__ __forceinline__ uint32_t group_inclusive_scan(uint32_t l, uint32_t v) {   // prefix sum inside each group of SUB lanes (l = lane within the group)
  if (SUB == 16) {   // a DPP row
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return v;
  }
#pragma unroll
  for (uint32_t d = 1; d < SUB; d <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, (int)d, 64);
    if (l >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t select_bit(uint64_t word, uint32_t rank) {   // position of the rank-th (0-based) set bit
  for (uint32_t i = 0; i < rank; ++i) word &= word - 1;
  return (uint32_t)__builtin_ctzll(word);
}

// ---- get_prev_vertex_with_sample by a group of SUB lanes, kSearchWindow (32) ranks at a time ----
// The search visits rank, rank - deg(previous(rank)), ... (one count per neighbour of each visited node): a static
// chain.  The group reads the records of the 32 ranks below the current one together, decides which of them the
// chain visits (ancestor labels, rk_anc), tests the visited nodes' event bits, and checks the candidates literally in
// parallel; the first visited node with a neighbour holding the sample is the answer.  Its ref_pos is the node's own
// last ref neighbour; a node without one (the end of the path) would need the history: `no_ref`, the caller walks serially.
// (Shared by the cooperative kernels of query types 4, 2, 3 and 5.)
struct GroupFound {
  uint32_t v, slot1;     // the vertex found (query.h:57-113) and its ref-path slot + 1
  WalkVertex wc;         // its record as the edge that names it carries it (off not included)
  uint64_t ref_pos;      // the found node's last ref neighbour's index; 1 at the head of the path
  bool head;             // the chain ran out: v is the first node of the path, wc / slot1 are not set
  bool no_ref;           // found at a node without a ref neighbour: ref_pos would be an earlier iteration's
};
template <uint32_t SUB>
__device__ __forceinline__ GroupFound group_search_prev(const DevImage& im, BitRow& ev, BitRow& hold, bool searching, uint64_t rank0,
                                                        uint32_t l, uint32_t gbase, uint32_t& n_search) {
  constexpr uint32_t kGroupMask = (1u << SUB) - 1u;
  const uint32_t sh = im.t4_ev_shift;
  GroupFound g{};
  uint64_t rank = rank0;
  const uint32_t tin0 = (searching && rank0 >= 2) ? im.rk_anc[rank0 - 1].x : 0u;   // label of the chain's first rank
  bool hop = false;            // group-uniform: the search has gone over to the event row's set bits
  uint32_t windows = 0, s_top = 0;
  while (__any(searching)) {
    if (searching && rank <= 1) {   // the head of the path (redundant in the group)
      g.v = im.rp_vid[im.rk_back[rank == 0 ? 0 : rank - 1].x];
      g.head = true; g.ref_pos = 1;
      searching = false;
    }
    ++n_search;
    bool found = false, had_ref = false;
    uint32_t f_ref_pos = 0, f_v = 0, f_slot1 = 0;
    WalkVertex f_wc{};
    // the literal test of one visited node (query.h:75-100): its last ref neighbour, its last neighbour holding the sample
    auto check_node = [&](uint32_t slot, uint32_t deg) {
      bool hr = false;
      uint32_t rp = 0;
      const uint32_t rb0 = im.blob_of_slot[slot] + 1;   // the edge records follow the slot's header
      for (uint32_t ed = rb0; ed < rb0 + deg; ++ed) {
        const uint4 a = im.wblob[2 * (uint64_t)ed], b = im.wblob[2 * (uint64_t)ed + 1];
        if (a.y) { rp = a.y; hr = true; }
        if (hold.bit_n(a.x, b.w)) {
          f_v = a.x; found = true; f_slot1 = b.y;
          f_wc = WalkVertex{a.w, b.x, a.y, 0u, b.z, a.z, b.w};
        }
      }
      if (found) { f_ref_pos = rp; had_ref = hr; }
    };
    if (searching && !hop) {
      // kSearchWindow ranks per round trip, E consecutive rank records per lane (lane l: offsets [l * E, l * E + E) below
      // the window's top rank).  Which of them the chain from rank0 visits is an ancestor test on the static forest of
      // chains (DevImage::rk_anc) -- no walking along the chain, every lane decides for its own ranks.
      constexpr uint32_t E = kSearchWindow / SUB;
      uint2 back[E], anc[E];
      bool valid[E];
#pragma unroll
      for (uint32_t e = 0; e < E; ++e) {
        const uint64_t off = (uint64_t)l * E + e;
        valid[e] = rank >= off + 2;
        back[e] = valid[e] ? im.rk_back[rank - off - 1] : uint2{0, 1};
        anc[e] = valid[e] ? im.rk_anc[rank - off - 1] : uint2{1, 0};
      }
      uint64_t ev_word[E];   // the visited nodes' event words, requested together (one memory latency, not one per node)
#pragma unroll
      for (uint32_t e = 0; e < E; ++e) {
        const bool visited = valid[e] && anc[e].x <= tin0 && tin0 - anc[e].x < anc[e].y;
        ev_word[e] = visited ? ev.row[(back[e].x >> sh) >> 6] : 0ULL;
      }
#pragma unroll
      for (uint32_t e = 0; e < E; ++e)   // in chain order within the lane: the first visited node with a holder wins
        if (!found && ((ev_word[e] >> ((back[e].x >> sh) & 63)) & 1)) check_node(back[e].x, back[e].y);
    }
    if (searching && hop) {
      // A long search (a sample with few variants): from here on it goes by the SET BITS of the sample's event row, one
      // 64-slot word per lane and round, highest slot first.  A set bit is a candidate when its slot is the first of its
      // rank (the only node of a rank the chain looks at) and that rank is on the chain from rank0 (ancestor labels).
      // (coarse rows: a bit stands for 2^sh slots, every one of them a candidate -- highest first, none above s_top)
      const uint32_t c_top = s_top >> sh;
      const int64_t wi = (int64_t)(c_top >> 6) - (int64_t)l;
      uint64_t word = wi >= 0 ? ev.row[wi] : 0ULL;
      if (l == 0 && (c_top & 63) != 63) word &= (1ULL << ((c_top & 63) + 1)) - 1;   // nothing above s_top's block
      // is slot k a node the chain looks at (first slot of its rank, rank on the chain)?  Then its literal test.
      auto try_slot = [&](uint32_t k) {
        if (k > s_top || k >= im.P) return;
        const uint32_t r = im.slot_rank[k];                 // chain rank r + 1 looks at the first slot of rank r
        if (r < 1) return;                                  // (the chain stops at rank <= 1 before it would look there)
        const uint2 bk = im.rk_back[r];
        if (bk.x != k) return;
        const uint2 an = im.rk_anc[r];
        if (!(an.x <= tin0 && tin0 - an.x < an.y)) return;
        check_node(k, bk.y);
      };
      if (sh > 0 && (1u << sh) <= SUB) {
        // Coarse rows: the group takes the set bits of its SUB words one at a time, highest block first, and tests the
        // block's 2^sh slots TOGETHER, a slot per lane (lane 0 the highest: the first lane that finds is the answer).
        // (One lane per word working through its blocks slot by slot was up to 2^sh serial tests of ~5 dependent look-ups
        //  per set bit: on the 10,000-sample cohort that was most of the search.)
        while (true) {
          const uint32_t wb = (uint32_t)((__ballot(word != 0) >> gbase) & kGroupMask);
          if (!wb) break;
          const int i0 = (int)gbase + __builtin_ctz(wb);
          const uint64_t w0 = shfl64(word, i0);
          const uint32_t b = 63u - (uint32_t)__builtin_clzll(w0);
          if ((int)(threadIdx.x & 63) == i0) word &= ~(1ULL << b);
          const uint32_t c = (uint32_t)((int64_t)(c_top >> 6) - (int64_t)(i0 - (int)gbase)) * 64u + b;
          if (l < (1u << sh)) try_slot((c << sh) + ((1u << sh) - 1u - l));
          if ((__ballot(found) >> gbase) & kGroupMask) break;
        }
      } else {
        while (word && !found) {
          const uint32_t b = 63u - (uint32_t)__builtin_clzll(word);
          word &= ~(1ULL << b);
          const uint32_t c = (uint32_t)wi * 64u + b;
          for (uint32_t j = 1u << sh; j-- > 0 && !found;) try_slot((c << sh) + j);
        }
      }
    }
    const uint32_t fb = (uint32_t)((__ballot(found) >> gbase) & kGroupMask);
    const int fl_lane = (int)gbase + (fb ? __builtin_ctz(fb) : 0);
    const uint32_t g_v = (uint32_t)__shfl((int)f_v, fl_lane, 64), g_slot1 = (uint32_t)__shfl((int)f_slot1, fl_lane, 64);
    const uint32_t g_ref_pos = (uint32_t)__shfl((int)f_ref_pos, fl_lane, 64);
    const bool g_had_ref = __shfl((int)had_ref, fl_lane, 64) != 0;
    WalkVertex g_wc;
    g_wc.row_begin = (uint32_t)__shfl((int)f_wc.row_begin, fl_lane, 64); g_wc.deg = (uint32_t)__shfl((int)f_wc.deg, fl_lane, 64);
    g_wc.ridx = (uint32_t)__shfl((int)f_wc.ridx, fl_lane, 64); g_wc.off = 0;
    g_wc.len = (uint32_t)__shfl((int)f_wc.len, fl_lane, 64); g_wc.cls = (uint32_t)__shfl((int)f_wc.cls, fl_lane, 64);
    g_wc.ncar = (uint32_t)__shfl((int)f_wc.ncar, fl_lane, 64);
    if (searching) {
      if (fb) {
        g.no_ref = !g_had_ref;
        g.v = g_v; g.wc = g_wc; g.ref_pos = g_ref_pos; g.slot1 = g_slot1;
        searching = false;
      } else if (!hop) {
        rank = rank > kSearchWindow ? rank - kSearchWindow : 0;   // the next window starts right below this one
        if (++windows >= kHopAfter && rank > 1 && im.slot_rank) { hop = true; s_top = im.rk_back[rank - 1].x; }
      } else if (((s_top >> sh) >> 6) >= SUB) s_top = ((((((s_top >> sh) >> 6) - SUB) << 6) | 63u) << sh) | ((1u << sh) - 1u);   // the last slot of the word SUB words down
      else rank = 0;   // nothing left below: the head of the path
    }
  }
  return g;
}

// EXPL: the cohort keeps explicit sample ids (coarse event rows, hold tests from the carrier lists).  The class-row
// instantiation carries none of that code -- at the 128 registers this kernel is held to, the eight ids a hold test reads
// at a time cost it 40 bytes of scratch per lane and 12 % of its time on the chr1 cohort, which never runs them.
// (Registers: 4 waves per SIMD = 128 VGPRs; the explicit-id form spills 23 of them there.  Measured round 5 on the 10,000-sample cohort:
//  3 waves -- 142 VGPRs, nothing spilled -- 143 M regions/s against 158; 5 waves -- 96 VGPRs, 120 spilled -- 118; class-row cohort
//  (chr1-2504) at 5 waves 119 against 124.  The walk is a chain of dependent look-ups: occupancy is worth more than the spills cost.)
template <uint32_t SUB, bool EXPL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) k_sample_walk_coop(DevImage im, DevResult r, uint32_t sid_all, const uint32_t* sid_per_region,
                                                          WalkScratch ws) {
  static_assert(SUB == 8 || SUB == 16, "group width");
  constexpr uint32_t kGroupMask = (1u << SUB) - 1u;
  const uint32_t lane = threadIdx.x & 63, l = lane & (SUB - 1), gbase = lane & (64 - SUB);
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
  if (walk_void(r, ws, q, l == 0)) return;
  const bool live = q < r.Q;
  WalkCtx cx{0, 0, 0, false, 0, 0};
  if (live) { cx.sid = sid_per_region ? sid_per_region[q] : sid_all; cx.x = r.regions[2 * q]; cx.y = r.regions[2 * q + 1]; }
  cx.use_ev = live && im.t4_events && cx.sid != 0;
  // group-uniform state: every lane of a group computes / receives the same values
  uint64_t nvar = 0, ncar = 0, rank0 = 0;
  uint8_t fl = 0;
  bool busy = false;         // the group still has episodes to run
  bool serial = false;       // the group walks its region with the serial loop (no event rows, or a fallback)
  uint32_t cur_slot = 0;     // slot at which the chain is in step
  uint64_t s0 = 0, scap = 0;
  BitRow ev{nullptr, kNone, 0}, hold{nullptr, kNone, 0}, irr{nullptr, kNone, 0};   // the sample's own events; "does v hold the sample"; the global irregular row
  const uint32_t sh = im.t4_ev_shift;   // > 0: coarse event rows (explicit-id cohorts): the search below, then the one-lane walk from where it ends
  WalkSt st{};
  const uint64_t t_c0 = VS_WALK_CLOCK();
  uint64_t t_c1 = t_c0, t_c2 = t_c0;
  uint32_t n_chunks = 0, n_search = 0;
  if (live) {
    fl = walk_prologue(im, cx, rank0);
    s0 = ws.cap_begin[q]; scap = ws.cap_begin[q + 1] - s0;
    if (!fl) {
      ev = sample_event_row(im, cx.sid, cx.use_ev);
      irr = irregular_row(im, cx.use_ev);
      hold = sample_hold_row(im, cx.sid, cx.use_ev);
      if (!EXPL) hold.x_sid = nullptr;     // (class-row cohorts: the hold row answers)
      if (!cx.use_ev) serial = true;
    }
  }
  // ---- get_prev_vertex_with_sample, kSearchWindow (32) ranks at a time (group_search_prev) ----
  {
    const bool searching = live && !fl && cx.use_ev;
    const GroupFound g = group_search_prev<SUB>(im, ev, hold, searching, rank0, l, gbase, n_search);
    if (searching) {
      if (g.head) {
        st.cur = g.v; st.wc = blob_vertex(im, g.v); st.ref_pos = 1; st.cur_ref_v = kNone;
        st.cur_slot1 = im.w_vertex[2 * (uint64_t)g.v + 1].w;
      } else {
        if (g.no_ref) serial = true;   // (ref_pos would be an earlier iteration's: the serial loop knows)
        st.cur = g.v; st.wc = g.wc; st.ref_pos = g.ref_pos; st.cur_ref_v = kNone; st.cur_slot1 = g.slot1;
      }
    }
  }
  t_c1 = VS_WALK_CLOCK();
  bool walked = false;       // coarse event rows: the region was walked by the one-lane loop from the search's end state (below)
  if (sh && live && !fl && cx.use_ev && !serial) {
    // Explicit-id cohorts: a region has one to three events, there is nothing for episodes in parallel to win, and the
    // coarse row's bits do not name slots.  The group-parallel search above did the long part (thousands of ranks on a
    // 10,000-sample cohort); the walk itself runs the serial loop from the state the search left (redundant in the group,
    // lane 0 writes).
    uint32_t jm = 0, sp = 0;
    auto sink = [&](const WalkEmit& e1) {
      if (nvar < scap) {
        if (l == 0) { const uint64_t s = s0 + nvar; ws.pos[s] = e1.ref_pos; ws.cur[s] = e1.cur; ws.ro[s] = e1.kind; ws.rl[s] = e1.cur_ref_v; }
      } else if (l == 0) *ws.overflow = 1;
      nvar++; ncar += pad_car(e1.c);
    };
    walk_serial<true, EXPL>(im, cx, ev, irr, hold, st, sink, jm, sp);
    walked = true;
  }
  if (live && !fl && cx.use_ev && !serial && !walked) {
    // ---- head: literal steps from the start state until the walk is in step (redundant in the group) ----
    bool done = false, term = false;
    uint32_t steps = 0;
    while (true) {
      if (done || st.ref_pos >= cx.y) { term = true; break; }
      if (walk_in_step(cx, st)) { cur_slot = st.cur_slot1 - 1; break; }
      if (++steps > 64) { serial = true; break; }   // (a start state that never falls in step: walk it serially)
      WalkEmit em;
      if (walk_literal_step<true>(im, cx, hold, st, em, done)) {
        if (nvar < scap) {
          if (l == 0) { const uint64_t s = s0 + nvar; ws.pos[s] = em.ref_pos; ws.cur[s] = em.cur; ws.ro[s] = em.kind; ws.rl[s] = em.cur_ref_v; }
        } else if (l == 0) *ws.overflow = 1;
        nvar++; ncar += pad_car(em.c);
      }
    }
    busy = !term && !serial;
  }
  t_c2 = VS_WALK_CLOCK();
  // ---- episodes, 16 events of a group at a time ----
  while (__any(busy)) {
    if (busy) ++n_chunks;
    // the next 16 events at or after cur_slot, one per lane: lane l loads word l of the row from cur_slot's word on
    const uint32_t w0 = cur_slot >> 6, w_end = (cx.limit + 63) >> 6, wi = w0 + l;
    uint64_t word = (busy && wi < w_end) ? (ev.row[wi] | irr_word_from(irr.row, wi, cx.irr_from)) : 0;   // (sh == 0 here)
    if (l == 0) word &= ~0ULL << (cur_slot & 63);
    if (busy && wi == (cx.limit >> 6) && (cx.limit & 63)) word &= (1ULL << (cx.limit & 63)) - 1;
    const uint32_t pc = (uint32_t)__popcll(word), incl = group_inclusive_scan<SUB>(l, pc);
    const uint32_t total = (uint32_t)__shfl((int)incl, (int)gbase + (int)SUB - 1, 64);
    uint32_t j = 0;                                   // the word holding this lane's event: #words whose inclusive count is <= l
#pragma unroll
    for (int t = 0; t < (int)SUB; ++t) j += (uint32_t)__shfl((int)incl, (int)gbase + t, 64) <= l ? 1u : 0u;
    const bool have = busy && l < total;
    const int src = (int)gbase + (int)(j < SUB ? j : SUB - 1);
    const uint64_t wj = shfl64(word, src);
    const uint32_t excl_j = (uint32_t)__shfl((int)(incl - pc), src, 64);
    const bool more = total > SUB || w0 + SUB < w_end;  // events beyond this chunk may exist
    uint32_t slot = 0;
    if (have) slot = ((w0 + j) << 6) + select_bit(wj, l - excl_j);
    // ---- this lane's episode ----
    WalkEmit em[kEpEmits];
    uint32_t n_em = 0, ep_pad = 0, ep_end = 0;
    bool ep_term = false, ep_ovf = false;
    if (have) {
      WalkSt es;
      walk_arrive_at_slot(im, es, slot);
      bool done = false;
      uint32_t steps = 0;
      while (true) {
        WalkEmit e1;
        if (walk_literal_step<true>(im, cx, hold, es, e1, done)) {
          if (n_em < kEpEmits) {
#pragma unroll
            for (uint32_t t = 0; t < kEpEmits; ++t) if (t == n_em) em[t] = e1;
          } else ep_ovf = true;
          ++n_em; ep_pad += pad_car(e1.c);
        }
        if (done || es.ref_pos >= cx.y) { ep_term = true; break; }
        if (walk_in_step(cx, es)) { ep_end = es.cur_slot1 - 1; break; }
        if (++steps >= kEpSteps) { ep_ovf = true; break; }
      }
      if (!ep_term && !ep_ovf && ep_end <= slot) ep_ovf = true;   // (a walk that does not advance: serial loop)
    }
    // ---- the chain of hand-overs (registers only) ----
    bool accepted = false, gdone = !busy, finished = false, fallback = false;
#pragma unroll 1
    for (int t = 0; t < (int)SUB; ++t) {
      const bool cand = !gdone && have && slot >= cur_slot;
      const uint32_t gb = (uint32_t)((__ballot(cand) >> gbase) & kGroupMask);
      const int i = gb ? (int)gbase + __builtin_ctz(gb) : (int)gbase;
      const bool t_i = __shfl((int)ep_term, i, 64), o_i = __shfl((int)ep_ovf, i, 64);
      const uint32_t end_i = (uint32_t)__shfl((int)ep_end, i, 64);
      if (!gdone) {
        if (!gb) gdone = true;                       // no event left in this chunk
        else {
          if ((int)lane == i) accepted = true;
          if (o_i) { fallback = true; gdone = true; }
          else if (t_i) { finished = true; gdone = true; }
          else cur_slot = end_i;
        }
      }
      if (!__any(!gdone)) break;
    }
    if (busy && !fallback && !finished && !more) finished = true;   // nothing below the stop slot any more: the walk runs into it
    // events may remain beyond this chunk: the chain is in step at least up to where the chunk's enumeration ended
    const uint32_t last_slot = (uint32_t)__shfl((int)slot, (int)gbase + (int)SUB - 1, 64);
    const uint32_t chunk_next = total > SUB ? last_slot + 1 : (w0 + SUB) << 6;
    if (busy && !fallback && !finished && chunk_next > cur_slot) cur_slot = chunk_next;
    // ---- the accepted episodes' reports, compacted in order ----
    const uint32_t mine = (accepted && !fallback) ? n_em : 0u;
    const uint32_t inc_e = group_inclusive_scan<SUB>(l, mine), tot_e = (uint32_t)__shfl((int)inc_e, (int)gbase + (int)SUB - 1, 64);
    const uint32_t inc_p = group_inclusive_scan<SUB>(l, (accepted && !fallback) ? ep_pad : 0u), tot_p = (uint32_t)__shfl((int)inc_p, (int)gbase + (int)SUB - 1, 64);
    if (mine) {
      const uint64_t at = nvar + (inc_e - mine);
#pragma unroll
      for (uint32_t t = 0; t < kEpEmits; ++t)
        if (t < mine) {
          if (at + t < scap) { const uint64_t s = s0 + at + t; ws.pos[s] = em[t].ref_pos; ws.cur[s] = em[t].cur; ws.ro[s] = em[t].kind; ws.rl[s] = em[t].cur_ref_v; }
          else *ws.overflow = 1;
        }
    }
    if (busy) { nvar += tot_e; ncar += tot_p; }
    if (fallback) serial = true;
    if (busy && (finished || fallback)) busy = false;
  }
  // ---- regions without event rows, and fallbacks: the serial loop (redundant in the group; lane 0 writes) ----
  if (__any(serial)) {
    if (serial) {
      uint32_t it = 0, lit = 0, jm = 0, sp = 0;
      nvar = 0; ncar = 0;
      ev.w = kNone; hold.w = kNone; irr.w = kNone;
      auto sink = [&](const WalkEmit& e1) {
        if (nvar < scap) {
          if (l == 0) { const uint64_t s = s0 + nvar; ws.pos[s] = e1.ref_pos; ws.cur[s] = e1.cur; ws.ro[s] = e1.kind; ws.rl[s] = e1.cur_ref_v; }
        } else if (l == 0) *ws.overflow = 1;
        nvar++; ncar += pad_car(e1.c);
      };
      if (cx.use_ev) { walk_start_search<true>(im, cx, ev, hold, rank0, st, it, lit); walk_serial<true, EXPL>(im, cx, ev, irr, hold, st, sink, jm, sp); }
      else { walk_start_search<false>(im, cx, ev, hold, rank0, st, it, lit); walk_serial<false>(im, cx, ev, irr, hold, st, sink, jm, sp); }
    }
  }
  if (live && l == 0) { r.q_flags[q] = fl; r.q_g0[q] = 0; r.q_nvar[q] = nvar; r.q_ncar[q] = ncar; }
  if (live && l == 0 && !fl) {
    const uint64_t t_c3 = VS_WALK_CLOCK();
    VS_WALK_STAT(0, 1); VS_WALK_STAT(1, n_search); VS_WALK_STAT(3, n_chunks); VS_WALK_STAT(7, nvar);
    VS_WALK_STAT(5, t_c1 - t_c0); VS_WALK_STAT(6, t_c3 - t_c2); VS_WALK_STAT(12, t_c2 - t_c1);
    VS_WALK_STATMAX(10, t_c1 - t_c0); VS_WALK_STATMAX(11, t_c3 - t_c2); VS_WALK_STATMAX(8, n_search); VS_WALK_STATMAX(9, n_chunks);
    (void)t_c3;
  }
  (void)t_c1; (void)t_c2; (void)n_chunks; (void)n_search;
}

// Headers of a type-4 batch from the scratch list of the single walk (one thread per region; ~10 variants each)
// 16 lanes per region (a sample has ~10 variants in a 10 kb region): coalesced reads of the walk's record and coalesced
// header writes; the arena offsets are a prefix sum inside each 16-lane row (four DPP steps).
// RESOLVE: the scratch holds the type-4 walk's state per reported vertex {ref_pos, vertex, kind, cur_ref} and the row is
// worked out here (resolve_walk_variant); otherwise (type 5) it holds finished rows.
// Shared carrier lists for the walking query types: a vertex that several regions of the batch report -- the same
// common variant on the paths of different samples -- gets ONE list.  The first row to claim the vertex (a 64-bit
// word per vertex, stamped with the batch's generation so that it never needs clearing) owns the list; `own_pad` holds
// the owner rows' padded carrier counts (0 for the others), and its exclusive scan gives the owners their arena places.
struct ListClaims {
  unsigned long long* claim;   // [V] generation << 40 | owner row + 1
  uint64_t gen;
  uint32_t* own_pad;           // [rows] padded carrier count of an owner row, 0 otherwise
  uint64_t* q_own;             // [Q] sum of own_pad over a region's rows; its exclusive scan own_base gives the region's part of the arena
  const uint64_t* own_base;    // [Q + 1]
  uint64_t* own_off;           // [rows] arena offset of an owner row's list
  uint64_t rows_cap;           // entries of own_pad / own_off (the walk's scratch capacity: more rows than that means an overflow, and the batch is redone)
};
constexpr uint64_t kClaimRowMask = (1ULL << 40) - 1;
__global__ void __launch_bounds__(256) k_t4_claim(DevImage im, DevResult r, WalkScratch ws, ListClaims lc) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool live = q < r.Q;
  uint64_t n = live ? r.q_nvar[q] : 0;   // (a refused batch: the walk left zeros)
  const uint64_t a0 = live ? r.var_begin[q] : 0, s0 = live ? ws.cap_begin[q] : 0;
  // a region that outgrew its scratch capacity recorded only the first rows (ws.overflow is set and the host redoes the
  // batch with the two-walk path): nothing beyond the capacity may be read here
  if (live && (n > ws.cap_begin[q + 1] - s0 || a0 + n > lc.rows_cap)) n = 0;
  uint64_t sum = 0;
  for (uint64_t i = threadIdx.x & 15u; i < n; i += 16) {
    const uint32_t v = ws.cur[s0 + i];
    const unsigned long long mine = (lc.gen << 40) | (a0 + i + 1);
    unsigned long long old = lc.claim[v];
    while ((old >> 40) != lc.gen) {
      const unsigned long long prev = atomicCAS(&lc.claim[v], old, mine);
      if (prev == old) { old = mine; break; }
      old = prev;
    }
    const uint32_t pad = old == mine ? pad_car(im.v_ncar[v]) : 0u;
    lc.own_pad[a0 + i] = pad;
    sum += pad;
  }
  for (int d = 8; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 16);
  if (live && (threadIdx.x & 15u) == 0) lc.q_own[q] = sum;
}
// arena offsets of the owner rows: the region's base + the prefix of its own rows' pads (one lane per region: ~10 rows)
__global__ void __launch_bounds__(256) k_t4_offsets(DevResult r, ListClaims lc) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  const uint64_t n = r.q_nvar[q], a0 = r.var_begin[q];
  if (a0 + n > lc.rows_cap) return;   // (overflowed batch: redone by the host)
  uint64_t at = lc.own_base[q];
  for (uint64_t i = 0; i < n; ++i) { lc.own_off[a0 + i] = at; at += lc.own_pad[a0 + i]; }
}

// The scan of q_own and k_t4_offsets as ONE launch for batches of up to kScanSmallMax regions (k_scan_small's scheme: <= 4 blocks, a
// block adds up the tiles in front of its own by itself): a thread has its four regions' bases in registers when the scan is done and
// walks their rows at once.  out_base[Q] = the arena the claimed lists take (the host reads it in the batch's wait).
__global__ void __launch_bounds__(kScanSmallBlock) k_t4_offsets_small(DevResult r, ListClaims lc, uint64_t* __restrict__ out_base) {
  __shared__ uint64_t wsum[kScanSmallBlock / 64], wpre[kScanSmallBlock / 64];
  const uint32_t n = (uint32_t)r.Q, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t tile0 = blockIdx.x * kScanSmallTile, base = tile0 + threadIdx.x * kScanSmallItems;
  const uint64_t* __restrict__ in = lc.q_own;
  uint64_t pre = 0, v[kScanSmallItems];
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallPre; ++k) {
    const uint32_t i = k * kScanSmallBlock + threadIdx.x;
    pre += i < tile0 ? in[i] : 0;
  }
#pragma unroll
  for (uint32_t k = 0; k < kScanSmallItems; ++k) v[k] = base + k < n ? in[base + k] : 0;
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
    const uint32_t q = base + k;
    if (q < n) {
      out_base[q] = ex;
      const uint64_t nr = r.q_nvar[q], a0 = r.var_begin[q];
      if (a0 + nr <= lc.rows_cap) {   // (else: overflowed batch, redone by the host)
        uint64_t at = ex;
        for (uint64_t i = 0; i < nr; ++i) { lc.own_off[a0 + i] = at; at += lc.own_pad[a0 + i]; }
      }
    }
    ex += v[k];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanSmallBlock - 1) out_base[n] = ex;
}

// LISTS 0: private list per row, 1: the list of the vertex's owner row (claims), 2: the index's resident list of the vertex
template <bool RESOLVE, int LISTS>
__global__ void __launch_bounds__(256) k_emit_from_walk(DevImage im, DevResult r, WalkScratch ws, ListClaims lc) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const uint32_t l16 = threadIdx.x & 15u, row_last = (threadIdx.x & 63u) | 15u;
  const bool live = q < r.Q;
  const uint64_t n = live ? r.q_nvar[q] : 0, a0 = live ? r.var_begin[q] : 0, s0 = live ? ws.cap_begin[q] : 0;
  uint64_t cb = live ? r.car_base[q] : 0, kept = 0;
  const uint64_t n_max = __shfl(n, 0, 16) ;   // (uniform per row already; rows of one wave may differ)
  // all four rows of the wave iterate together: the DPP steps need every lane of the wave in the same instruction
  uint64_t rounds = (n_max + 15) / 16;
  for (int d = 16; d < 64; d <<= 1) { const uint64_t o = __shfl_xor(rounds, d, 64); rounds = o > rounds ? o : rounds; }
  for (uint64_t base = 0; base < rounds * 16; base += 16) {
    const uint64_t i = base + l16;
    const bool on = i < n;
    const uint64_t a = a0 + i, s = s0 + i;
    const uint32_t cur = on ? ws.cur[s] : 0u, c = on ? im.v_ncar[cur] : 0u;
    uint32_t incl = pad_car(c);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);   // row_shr:1 .. 8: prefix inside the row
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);
    uint32_t csum = c;
    csum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)csum, 0x111, 0xF, 0xF, true);
    csum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)csum, 0x112, 0xF, 0xF, true);
    csum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)csum, 0x114, 0xF, 0xF, true);
    csum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)csum, 0x118, 0xF, 0xF, true);
    if (on) {
      WalkVariant wv;
      if (RESOLVE) wv = resolve_walk_variant(im, ws.ro[s], cur, ws.pos[s], ws.rl[s]);
      else wv = WalkVariant{ws.pos[s], ws.ro[s], ws.rl[s], ws.ao[s], ws.al[s]};
      uint64_t at = cb + (incl - pad_car(c));
      bool owner = true;
      if (LISTS == 1) {   // the list lives where the vertex's owner row put it
        const uint64_t o = (lc.claim[cur] & kClaimRowMask) - 1;
        at = lc.own_off[o];
        owner = o == a;
      }
      if (LISTS == 2) at = c ? im.v_abegin[cur] : 0;
      row_store(r.rows, a, (uint32_t)wv.pos, wv.ro, wv.rl, wv.ao, wv.al, c, false, at);
      if (LISTS != 2) { r.r_class[a] = owner ? im.v_src[cur] : kNone; r.r_gt0[a] = im.v_car_begin[cur]; }
    }
    cb += (uint32_t)__shfl((int)incl, (int)row_last, 64);
    kept += (uint32_t)__shfl((int)csum, (int)row_last, 64);
  }
  if (live && l16 == 0) { r.var_count[q] = n; r.q_ncar[q] = kept; }
}

// Compact hit lists for a collective: the index (and so the site table) is replicated on every rank,
// therefore a region's variant list is fully described by its site range.  4 x uint64 per region:
//   {region_base + q, first site | region flags << 32 | has-dropped << 40, sites | variants reported << 32, carriers}
__global__ void __launch_bounds__(256) k_pack_regions(DevResult r, uint64_t* dst, uint64_t region_base) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  const uint64_t fl = r.q_flags[q] & ~kRegionSlow;
  const uint64_t dropped = r.var_count[q] != r.q_nvar[q] ? 1ULL : 0ULL;
  dst[4 * q + 0] = region_base + q;
  dst[4 * q + 1] = (uint64_t)r.q_g0[q] | (fl << 32) | (dropped << 40);
  dst[4 * q + 2] = (r.q_nvar[q] & 0xFFFFFFFFULL) | (r.var_count[q] << 32);
  dst[4 * q + 3] = r.q_ncar[q];   // carriers of the reported variants (the arena range car_base[q+1] - car_base[q] is padded)
}

// The receiving side of that collective: region bounds of a batch taken from gathered records instead of from
// (x, y) -- the site range is the answer of Index::find + the walk's stop rule on the rank that produced the record,
// and the replicated site table expands it to the same rows here (k_emit_headers / k_dedup_slow / k_fill_carriers).
// A record whose range does not fit this index's site table marks its region invalid.
__global__ void __launch_bounds__(256) k_bounds_from_records(DevImage im, DevResult r, const uint64_t* recs) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  const uint64_t w1 = recs[4 * q + 1], w2 = recs[4 * q + 2];
  uint32_t g0 = (uint32_t)w1, nsites = (uint32_t)w2;
  uint8_t fl = (uint8_t)((w1 >> 32) & (kRegionEmpty | kRegionInvalid | kRegionNotFound | kRegionEndless));
  if ((uint64_t)g0 + nsites > im.G) { g0 = 0; nsites = 0; fl = kRegionInvalid; }
  if ((w1 >> 40) & 1) fl |= kRegionSlow;   // the producing rank dropped rows: the literal rule runs again here
  r.q_flags[q] = fl;
  r.q_g0[q] = g0;
  r.q_nvar[q] = nsites;
  r.q_ncar[q] = im.s_carpre[g0 + nsites] - im.s_carpre[g0];
}

}  // namespace vsamd
