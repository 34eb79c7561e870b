// k_image.hip.h -- the HBM image (DevImage), the result (DevResult, VariantRow), rank / find, wave helpers.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace vsamd {


constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kSiteAlwaysDrop = 2;  // branch the reference would emit with an uninitialised var_pos
constexpr uint32_t kVarDropped = 1;
constexpr uint8_t kRegionEmpty = 1, kRegionInvalid = 2, kRegionNotFound = 4, kRegionEndless = 8, kRegionSlow = 128;

struct VariantRow;
struct DevImage {
  uint64_t ref_length, nbits;
  uint32_t num_samples, wpc, use_bv, pad_;
  uint64_t V, E, P, R, C, G;
  const uint64_t* bits;
  const uint32_t* blk_rank;
  const uint32_t* idx_pos;
  const uint32_t* rank_to_slot;
  const uint32_t* rp_vid;
  const uint32_t* rp_cand_prefix;
  const uint32_t* row_ptr;
  const uint32_t* col;
  const uint32_t *v_off, *v_len, *v_ridx, *v_class, *v_ncar, *v_nri;
  const uint4* w_vertex;   // walk records, 2 x uint4 per vertex {row_begin, degree, ref index, offset | length, class, #carriers, 0}
  const uint4* w_edge;     // 2 x uint4 per CSR entry {neighbour, its ref index, its class, its row_begin | degree, offset, length, #carriers}
  const uint32_t* v_src;   // per vertex: group index of its class's 16-bit id list (<= list_max carriers) or its class id (row)
  const uint64_t* v_car_begin;
  const uint64_t* class_rows;
  const uint32_t* cls_list_begin;
  const uint32_t* cls_list_ids;
  const uint16_t* cls_list16;   // 16-bit lists, 8-entry aligned and padded, of every class of at most list_max carriers (wpc <= 63)
  const uint8_t* gt_nibbles;    // cohorts above 4032 samples and explicit-id cohorts: 4 bits per carrier record of the pool
  const uint32_t* gt_groups;    // class-row cohorts of at most 4032 samples (their pool is padded: a vertex's records start on a multiple
                                // of 8): one word per 8 carrier records, genotype k at bit 3 (k / 2) + 16 (k & 1)
  const uint32_t* car_sid;
  const uint32_t* car_index;  // sample-coordinate index per carrier record (types 2/3/5); valid when has_car_index
  const uint8_t* seq_codes;
  // site table (one entry per branch of a ref-path node, ref-path order)
  uint32_t *s_pos, *s_ref_off, *s_ref_len, *s_alt_off, *s_alt_len, *s_vid, *s_ncar, *s_flags, *s_dup_prev, *s_class;
  uint64_t* s_carpre;  // [G+1] exclusive prefix of pad_car(s_ncar): arena offsets relative to a region's first site
  uint64_t* s_kpre;    // [G+1] exclusive prefix of s_ncar itself: carriers of the variants a site range reports
  uint64_t* s_gt0;     // [G] carrier-pool index of the branch's first carrier
  const struct VariantRow* s_row;   // [G] the site as a 32-byte row of the variant table with car_begin = s_carpre[g] (k_build_site_rows)
  const uint32_t* sus_g;     // sorted site indexes that can trigger the dedup rule
  const uint32_t* sus_prev;  // nearest earlier equal site, kNone = always dropped
  const uint32_t* rp_sus_prefix;  // [P+1] suspicious sites before each ref-path slot's first site (no bisection per query)
  const uint64_t* rp_carpre;      // [P+1] s_carpre[rp_cand_prefix[slot]]: arena prefix at a slot's first site
  const uint64_t* rp_kpre;        // [P+1] s_kpre likewise
  // The same per-slot / per-rank figures as records (k_slot_records, k_rank_records): a region's bounds touch ONE line per
  // end and level instead of four (round 3: 13 sparse lines per region, 28 us for 100 k regions)
  const uint4* rp_rec;            // [P+1] x 2: {rp_cand_prefix, rp_sus_prefix, rp_carpre lo, hi}, {rp_kpre lo, hi, 0, 0}
  const uint2* rk_rec;            // [R+1]: {idx_pos[r] (0 at r == R), rank_to_slot[r]}
  uint32_t n_sus, has_car_index;
  uint32_t list_max, pad2_;
  // Query type 4: per-sample EVENT bitmaps over the ref-path slots (k_build_events).  Bit j of row s is set when a walk
  // of sample s's path can do anything but step from slot j to slot j + 1 there: the node or one of its out-neighbours
  // holds s, or the node is irregular (its last ref neighbour is not its path successor / it ends the path).  Runs of
  // clear bits are skipped by k_sample_walk.  NULL: not built (over budget, or an index whose slots do not map onto
  // the rank structure one to one) -- the walk then visits every vertex.
  const uint64_t* t4_events;
  uint64_t t4_stride;       // 64-bit words per sample row: ceil(P / 64) + 1
  // Round 4: the sample-INDEPENDENT part of an event -- the node is irregular -- lives in ONE global row (t4_irr, t4_stride
  // words) instead of in every sample's; t4_events holds "the node or an out-neighbour holds the sample" alone.  The walks
  // jump by the OR of the two; the backward searches enumerate the sample's own bits only (every ~40th slot is irregular:
  // on a cohort whose samples have a variant every 15 kb that was 97 % of a long search's candidates).
  const uint64_t* t4_irr;
  uint32_t t4_irr_reach, t4_ev_shift;   // t4_ev_shift: one bit of an event row stands for 2^shift slots (explicit-id cohorts: 3; class-row cohorts: 0);   // HostImage::irr_reach: an irregular slot matters to the type-4 walk only within this many slots of its stop slot
  // Per-sample HOLD rows over the vertex ids (k_build_hold): bit v of row s = vertex v holds sample s (what
  // get_sample_from_vertex_if_exists answers).  Vertex ids grow along the reference, so every test of one walk step --
  // the node, its neighbours, the neighbours' neighbours -- falls into one or two 64-bit words of the sample's row
  // instead of one class-row line per vertex.  Built together with t4_events.
  const uint64_t* t4_hold;
  uint64_t t4_hold_stride;  // 64-bit words per sample row: ceil(V / 64) + 1
  // The walk blob (device_image.hpp): 32-byte records in the order a walk along the reference needs them -- per ref-path
  // slot one header record {first edge record, degree, ref index, sequence offset, length, class, #carriers, vertex id}, the edge
  // records of the slot's node {neighbour, its ref index, its class, ITS first edge record, its degree, its ref-path
  // slot + 1, its length, its #carriers}, then the edge records of its off-path neighbours (and theirs): one or two
  // cache lines hold everything an episode of the type-4 walk reads.
  const uint4* wblob;
  const uint32_t* blob_of_slot;   // [P + 1] header record of each ref-path slot
  const uint32_t* blob_row;       // [V] first edge record of each vertex
  const uint2* rk_back;     // [R] per rank r: {first ref-path slot of r (= Index::previous(r + 1)), out-degree of that node}
  const uint16_t* class_cum;    // [(C + 1) x wpc] ones in words [0, w) of every class row (k_class_cum; indexes with sample coordinates only,
                                //   NULL otherwise): the rank of a sample's bit -- its entry in the carrier pool -- in two loads
  const uint2* rk_anc;      // [R] per chain rank r + 1: {tin, subtree size} in the forest of the backward search's chains (device_image.hpp):
                            //   the chain from r0 visits rank p  <=>  tin[p] <= tin[r0] < tin[p] + size[p]
  const uint32_t* slot_rank;  // [P] rank of every ref-path slot (its node's start index is the rank-th set bit of the rank structure)
  const uint64_t* seq_breaks;   // bit per ref-path slot: the sequence queries must step through it literally (device_image.hpp)
  // RESIDENT carrier lists (option "resident_lists"; engine.hip: build_resident_lists): every list a query can report,
  // expanded once into an arena that stays with the index -- the lists of the sites in site-table order at s_carpre[g]
  // (so a region's lists are ONE arena range, [s_carpre[g0], s_carpre[g1])), then the lists of the vertices only the
  // walking query types report.  A result then holds rows that point into this arena and no arena of its own.
  const uint64_t* v_abegin;   // [V] arena offset of each vertex's list (~0: the vertex has no carriers); NULL: not built
  // The DENSE sites of the index (class-row cohorts: more than list_max carriers, expanded from the class's bit row), ascending:
  // the split form of the shared expansion gives them a launch of their own (k_fill_dense).
  const uint32_t* dense_site;
  uint64_t n_dense;
};

// One row of a result's VARIANT TABLE (what the reference's `Variant` holds, query.h:30-36, with the strings and the
// sample list as references): 32 bytes, written with two 16-byte stores.
struct VariantRow {
  uint32_t pos;            // Variant::var_pos
  uint32_t ref_off, ref_len, alt_off, alt_len;   // Variant::ref / alt = sequence pool [off, off + len)
  uint32_t count_flags;    // carriers | kRowDropped
  uint64_t car_begin;      // first carrier of the row's list in the arena
};
static_assert(sizeof(VariantRow) == 32, "row layout");
constexpr uint32_t kRowDropped = 0x80000000u;   // suppressed by the reference's "already seen" rule (or a branch it never reports)
__device__ __forceinline__ void row_store(VariantRow* rows, uint64_t a, uint32_t pos, uint32_t ro, uint32_t rl, uint32_t ao, uint32_t al,
                                          uint32_t count, bool dropped, uint64_t cb) {
  uint4* p = reinterpret_cast<uint4*>(rows + a);
  p[0] = uint4{pos, ro, rl, ao};
  p[1] = uint4{al, count | (dropped ? kRowDropped : 0u), (uint32_t)cb, (uint32_t)(cb >> 32)};
}
__device__ __forceinline__ VariantRow row_load(const VariantRow* rows, uint64_t a) {
  const uint4* p = reinterpret_cast<const uint4*>(rows + a);
  const uint4 x = p[0], y = p[1];
  return VariantRow{x.x, x.y, x.z, x.w, y.x, y.y, ((uint64_t)y.w << 32) | y.z};
}
__device__ __forceinline__ uint32_t row_count(const VariantRow& v) { return v.count_flags & ~kRowDropped; }
__device__ __forceinline__ bool row_dropped(const VariantRow& v) { return (v.count_flags & kRowDropped) != 0; }

// A result = per-region arrays + the variant table + the carrier arena.  Region q reports rows
// [var_begin[q], var_begin[q] + q_nvar[q]) of the table.  In a sorted batch of overlapping regions the ranges of
// different regions OVERLAP: every site the batch covers has one row and one carrier list, shared by the regions that
// report it (k_share_*); otherwise every region has rows and lists of its own, back to back.
struct DevResult {
  uint64_t Q, A, S;         // regions, rows of the table, arena entries
  const uint64_t* regions;  // [2Q] x,y
  uint8_t* q_flags;         // [Q]
  uint32_t* q_g0;           // [Q] first site of the region
  uint64_t* q_nvar;         // [Q] slots
  uint64_t* q_ncar;         // [Q] arena entries of the region (padded counts) until the offsets are scanned; afterwards the
                            //     header kernels overwrite it with the carriers of the region's REPORTED variants
  uint64_t* var_begin;      // [Q+1] first row of each region ([Q] = A); monotone only when rows are private
  uint64_t* car_base;       // [Q+1] arena offset of each region's first site; with shared carrier lists NOT monotone ([Q] = arena entries used)
  uint64_t* q_car_len;      // [Q] shared carrier lists only: the region's padded arena extent (else NULL: car_base[q + 1] - car_base[q])
  uint64_t* var_count;      // [Q] variants the reference reports (rows minus dropped)
  VariantRow* rows;         // [A]
  // private-row results only: what k_fill_carriers needs per row beside count and arena offset
  uint32_t* r_class;        // DevImage::v_src of the row's vertex: list group index or class id, by the count
  uint64_t* r_gt0;          // carrier-pool index of its first carrier
  void* carriers;           // uint16 (id | gt << 13) for cohorts of at most 4032 samples, else uint32 (id | gt << 29)
  uint32_t car_width, pad3_; // bytes per carrier word in the arena: 2 or 4
  // latency path (k_query_small): the last block posts done_seq | any-slow << 62 | capacities-exceeded << 63 into
  // mapped host memory -- ONE word, one writer -- and the host spins on it instead of waiting for the runtime's
  // completion signal.  host_totals is a debugging aid (VS_LAT_DEBUG: device-clock durations), NULL otherwise.
  unsigned long long* done_counter;
  volatile uint64_t* done_flag;
  uint64_t done_seq;
  volatile uint64_t* host_totals;
};

// Every variant's carrier range in the result arena starts on a multiple of 8 entries (16 bytes of 16-bit carrier
// words) and owns the padding up to the next multiple: k_fill_carriers then writes whole 16-byte groups only.
constexpr uint32_t kCarAlign = 8;
__host__ __device__ __forceinline__ uint32_t pad_car(uint32_t n) { return (n + kCarAlign - 1) & ~(kCarAlign - 1); }
__global__ void __launch_bounds__(256) k_pad_counts(const uint32_t* in, uint32_t* out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pad_car(in[i]);
}

// Rows + parameters of the ONE expansion that builds the resident arena (k_fill_carriers over them): the G sites, then
// the X vertices without a usable site (x_vid, lists at x_begin).
__global__ void __launch_bounds__(256) k_resident_params(DevImage im, const uint32_t* x_vid, const uint64_t* x_begin, uint64_t X,
                                                         VariantRow* rows, uint32_t* r_class, uint64_t* r_gt0) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= im.G + X) return;
  uint32_t cnt, cls; uint64_t gt0, cb;
  if (i < im.G) { cnt = im.s_ncar[i]; cls = im.s_class[i]; gt0 = im.s_gt0[i]; cb = im.s_carpre[i]; }
  else { const uint32_t v = x_vid[i - im.G]; cnt = im.v_ncar[v]; cls = im.v_src[v]; gt0 = im.v_car_begin[v]; cb = x_begin[i - im.G]; }
  row_store(rows, i, 0, 0, 0, 0, 0, cnt, false, cb);
  r_class[i] = cls; r_gt0[i] = gt0;
}

// ones in bit positions [0, p): number of ref-node start indexes <= p
// Branch-free: the whole 512-bit block comes in four independent 16-byte loads issued together with the block's
// cumulative count (ONE memory latency instead of up to nine in a row); words beyond p are masked off.  p == nbits
// (one past the last block) is served from the last block with all eight words counted.
struct RankLoads { uint32_t base; uint4 q[4]; uint32_t full, rem; };
__device__ __forceinline__ RankLoads rank1_issue(const DevImage& im, uint64_t p) {
  if (p > im.nbits) p = im.nbits;
  const uint64_t nblk = im.nbits >> 9;                      // bits holds a whole number of blocks (>= 1)
  const uint64_t blk = (p >> 9) < nblk ? (p >> 9) : nblk - 1;
  RankLoads l;
  l.base = im.blk_rank[blk];
  const uint4* b4 = reinterpret_cast<const uint4*>(im.bits + (blk << 3));
  l.q[0] = b4[0]; l.q[1] = b4[1]; l.q[2] = b4[2]; l.q[3] = b4[3];
  l.full = (uint32_t)((p >> 6) - (blk << 3));               // whole words below p inside the block: 0..8
  l.rem = (uint32_t)(p & 63);
  return l;
}
__device__ __forceinline__ uint32_t rank1_finish(const RankLoads& l) {
  uint32_t r = l.base;
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) {
    const uint4& v = l.q[i >> 1];
    const uint64_t w = (i & 1) ? (((uint64_t)v.w << 32) | v.z) : (((uint64_t)v.y << 32) | v.x);
    const uint64_t m = i < l.full ? ~0ULL : (i == l.full ? ((1ULL << l.rem) - 1) : 0ULL);
    r += __popcll(w & m);
  }
  return r;
}
__device__ __forceinline__ uint32_t rank1(const DevImage& im, uint64_t p) { return rank1_finish(rank1_issue(im, p)); }

// (a & mask) | c in one VOP3 instruction; the mask must sit in an SGPR (no literals in VOP3 on gfx9)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask_sgpr, uint32_t c) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask_sgpr), "v"(c));
  return r;
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves only (no LDS round trips): Hillis-Steele
// inside each row of 16 lanes (row_shr 1, 2, 4, 8; lanes without a source add 0), then lane 15 of each odd row's
// predecessor into rows 1 and 3 (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);
  return v;
}

__device__ __forceinline__ bool seq_equal(const DevImage& im, uint32_t a_off, uint32_t b_off, uint32_t len) {
  for (uint32_t i = 0; i < len; ++i)
    if (im.seq_codes[a_off + i] != im.seq_codes[b_off + i]) return false;
  return true;
}

}  // namespace vsamd
