// kernels.hip.h -- gfx950 device code of the region-query path.
//
// All integer / index work: the roofline that bounds these kernels is HBM
// bandwidth (and memory latency for the gather phases), never MFMA.
// Wavefront width is 64 throughout; one workgroup = 256 threads = 4 waves.
//
// Kernel                replaces (reference file:line)
// --------------------  ------------------------------------------------------
// k_build_sites         next_variant_in_ref's per-node branch classification
//                       (include/query.h:308-393) + the radius-1 neighbour walk
//                       (include/graph.h:394-431), run ONCE over the whole ref
//                       path when an index is opened (the "site table")
// k_mark_dups           the candidates the "already seen" rule can ever hit
//                       (include/query.h:397-414)
// k_region_bounds       Index::is_empty / Index::find (include/index.h:119-166)
//                       + the stop rule ref_index+length >= end (query.h:312)
// k_emit_headers        building std::vector<Variant> (query.h:736-771)
// k_dedup_slow          the literal "already seen" rule for the rare regions
//                       that contain a repeated (pos, alt)
// k_fill_carriers       get_samples -> get_sample_id / get_sample_phasing
//                       (query.h:268-285, variant_graph.h:875-942): expansion of
//                       decoded class id lists / class bit rows + genotype bits into
//                       carrier lists.  This is the dominant kernel.
// k_query_small         a whole get_var_in_ref batch of <= 64 regions in one launch
//                       (bounds, offsets, headers, carriers, dedup): the latency path
// k_query_server        the same, resident: polls requests in mapped host memory
// k_point_bounds        one next_variant_in_ref call from a position (query.h:297-436)
//                       as used by closest_var (query.h:441-483, type 1) and
//                       samples_has_var (query.h:792-823, type 7)
// k_has_var_filter      the (pos, ref, alt) match of samples_has_var (query.h:802-803)
// k_sample_walk_sc      get_sample_var_in_sample (query.h:490-612, type 5)
// k_sample_seq          query_sample_from_ref / query_sample_from_sample
//                       (query.h:118-261, types 2 and 3): the walk emits (offset, length)
//                       pieces of the sequence pool; k_copy_segments decodes them
// k_find                Index::find batched
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace vsamd {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kSiteAlwaysDrop = 2;  // branch the reference would emit with an uninitialised var_pos
constexpr uint32_t kVarDropped = 1;
constexpr uint8_t kRegionEmpty = 1, kRegionInvalid = 2, kRegionNotFound = 4, kRegionEndless = 8, kRegionSlow = 128;

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
  const uint8_t* gt_nibbles;
  const uint32_t* car_sid;
  const uint32_t* car_index;  // sample-coordinate index per carrier record (types 2/3/5); valid when has_car_index
  const uint8_t* seq_codes;
  // site table (one entry per branch of a ref-path node, ref-path order)
  uint32_t *s_pos, *s_ref_off, *s_ref_len, *s_alt_off, *s_alt_len, *s_vid, *s_ncar, *s_flags, *s_dup_prev, *s_class;
  uint64_t* s_carpre;  // [G+1] exclusive prefix of pad_car(s_ncar): arena offsets relative to a region's first site
  uint64_t* s_kpre;    // [G+1] exclusive prefix of s_ncar itself: carriers of the variants a site range reports
  uint64_t* s_gt0;     // [G] carrier-pool index of the branch's first carrier
  const uint32_t* sus_g;     // sorted site indexes that can trigger the dedup rule
  const uint32_t* sus_prev;  // nearest earlier equal site, kNone = always dropped
  const uint32_t* rp_sus_prefix;  // [P+1] suspicious sites before each ref-path slot's first site (no bisection per query)
  const uint64_t* rp_carpre;      // [P+1] s_carpre[rp_cand_prefix[slot]]: arena prefix at a slot's first site
  const uint64_t* rp_kpre;        // [P+1] s_kpre likewise
  uint32_t n_sus, has_car_index;
  uint32_t list_max, pad2_;
  // Query type 4: per-sample EVENT bitmaps over the ref-path slots (k_build_events).  Bit j of row s is set when a walk
  // of sample s's path can do anything but step from slot j to slot j + 1 there: the node or one of its out-neighbours
  // holds s, or the node is irregular (its last ref neighbour is not its path successor / it ends the path).  Runs of
  // clear bits are skipped by k_sample_walk.  NULL: not built (over budget, or an index whose slots do not map onto
  // the rank structure one to one) -- the walk then visits every vertex.
  const uint64_t* t4_events;
  uint64_t t4_stride;       // 64-bit words per sample row: ceil(P / 64) + 1
  // Per-sample HOLD rows over the vertex ids (k_build_hold): bit v of row s = vertex v holds sample s (what
  // get_sample_from_vertex_if_exists answers).  Vertex ids grow along the reference, so every test of one walk step --
  // the node, its neighbours, the neighbours' neighbours -- falls into one or two 64-bit words of the sample's row
  // instead of one class-row line per vertex.  Built together with t4_events.
  const uint64_t* t4_hold;
  uint64_t t4_hold_stride;  // 64-bit words per sample row: ceil(V / 64) + 1
  // The walk blob (device_image.hpp): 32-byte records in the order a walk along the reference needs them -- per ref-path
  // slot one header record {first edge record, degree, ref index, 0, length, class, #carriers, vertex id}, the edge
  // records of the slot's node {neighbour, its ref index, its class, ITS first edge record, its degree, its ref-path
  // slot + 1, its length, its #carriers}, then the edge records of its off-path neighbours (and theirs): one or two
  // cache lines hold everything an episode of the type-4 walk reads.
  const uint4* wblob;
  const uint32_t* blob_of_slot;   // [P + 1] header record of each ref-path slot
  const uint32_t* blob_row;       // [V] first edge record of each vertex
  const uint2* rk_back;     // [R] per rank r: {first ref-path slot of r (= Index::previous(r + 1)), out-degree of that node}
  // RESIDENT carrier lists (option "resident_lists"; engine.hip: build_resident_lists): every list a query can report,
  // expanded once into an arena that stays with the index -- the lists of the sites in site-table order at s_carpre[g]
  // (so a region's lists are ONE arena range, [s_carpre[g0], s_carpre[g1])), then the lists of the vertices only the
  // walking query types report.  A result then holds rows that point into this arena and no arena of its own.
  const uint64_t* v_abegin;   // [V] arena offset of each vertex's list (~0: the vertex has no carriers); NULL: not built
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
  const uint64_t sel = im.idx_pos[rx < R ? rx : R - 1];                  // select(rank(x) + 1)
  // is_empty: x beyond the reference, select past the last one (defined as empty), or no node start in (.., y]
  const bool empty = x > im.ref_length || rx >= R || !(sel - 1 <= y);
  // find(x): rank(x) >= 1 for every x >= 1 because a node starts at index 1
  uint64_t rf = (x >= im.ref_length) ? (uint64_t)R - 1 : (uint64_t)(rx ? rx - 1 : 0);
  if (rf > (uint64_t)R - 1) rf = (uint64_t)R - 1;
  const uint32_t s0 = im.rank_to_slot[rf];
  // first slot whose node ends at or after y stops the walk; node ends tile the reference, so that is the slot before
  // the first start >= y (rank_to_slot[R] == P)
  const uint32_t s1raw = im.rank_to_slot[ry < R ? ry : R];
  uint32_t s1 = s1raw ? s1raw - 1 : 0;
  if (s1 < s0) s1 = s0;
  if (s1 > P) s1 = P;
  uint32_t g0 = im.rp_cand_prefix[s0], g1 = im.rp_cand_prefix[s1];
  // can the "already seen" rule fire inside [g0,g1)?  (g0, g1 are slot boundaries: the list range is tabulated)
  uint32_t lo = im.rp_sus_prefix[s0], hi = im.rp_sus_prefix[s1];
  uint64_t pre0 = im.rp_carpre[s0], npad = im.rp_carpre[s1] - pre0, nkept = im.rp_kpre[s1] - im.rp_kpre[s0];
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

// ---------------------------------------------------------------------------
// Exclusive scan of a uint64 array (three launches; sizes here are <= a few 1e7).
// ---------------------------------------------------------------------------
constexpr int kScanBlock = 256, kScanItems = 8, kScanTile = kScanBlock * kScanItems;

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t* total) {
  __shared__ uint64_t wsum[kScanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint64_t incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  uint64_t woff = 0, tot = 0;
  for (int w = 0; w < kScanBlock / 64; ++w) {
    if (w < wid) woff += wsum[w];
    tot += wsum[w];
  }
  __syncthreads();
  *total = tot;
  return woff + incl - v;
}

template <typename T>
__global__ void __launch_bounds__(kScanBlock) k_scan_tile_sums(const T* in, uint64_t n, uint64_t* tile_sums) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint64_t s = 0;
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += in[base + i];
  uint64_t tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single block: tile_sums -> exclusive prefix in place; writes the grand total to out[n]
__global__ void __launch_bounds__(kScanBlock) k_scan_spine(uint64_t* tile_sums, uint64_t ntiles, uint64_t* grand_total) {
  uint64_t carry = 0;
  for (uint64_t base = 0; base < ntiles; base += kScanBlock) {
    const uint64_t i = base + threadIdx.x;
    const uint64_t v = i < ntiles ? tile_sums[i] : 0;
    uint64_t tot;
    const uint64_t ex = block_exclusive_scan(v, &tot);
    if (i < ntiles) tile_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *grand_total = carry;
}

template <typename T>
__global__ void __launch_bounds__(kScanBlock) k_scan_apply(const T* in, uint64_t n, const uint64_t* tile_sums, uint64_t* out) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint64_t loc[kScanItems];
  uint64_t s = 0;
  for (int i = 0; i < kScanItems; ++i) {
    loc[i] = base + i < n ? (uint64_t)in[base + i] : 0;
    s += loc[i];
  }
  uint64_t tot;
  uint64_t ex = block_exclusive_scan(s, &tot) + tile_sums[blockIdx.x];
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += loc[i];
  }
}

// ---------------------------------------------------------------------------
// Both offset arrays of a batch -- var_begin (slots) and car_base (padded arena entries) -- in ONE pass over the
// regions: three launches instead of six.  The grand totals also go to `totals` (mapped host memory).
// ---------------------------------------------------------------------------
struct Scan2 { uint64_t a, c; };

__device__ __forceinline__ Scan2 block_exclusive_scan2(Scan2 v, Scan2* total) {
  __shared__ Scan2 wsum[kScanBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  Scan2 incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t ta = __shfl_up(incl.a, d, 64), tc = __shfl_up(incl.c, d, 64);
    if (lane >= d) { incl.a += ta; incl.c += tc; }
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  Scan2 woff{0, 0}, tot{0, 0};
  for (int w = 0; w < kScanBlock / 64; ++w) {
    if (w < wid) { woff.a += wsum[w].a; woff.c += wsum[w].c; }
    tot.a += wsum[w].a; tot.c += wsum[w].c;
  }
  __syncthreads();
  *total = tot;
  return Scan2{woff.a + incl.a - v.a, woff.c + incl.c - v.c};
}

__global__ void __launch_bounds__(kScanBlock) k_scan2_tile_sums(const uint64_t* nvar, const uint64_t* ncar, uint64_t n, Scan2* tile_sums) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  Scan2 s{0, 0};
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) { s.a += nvar[base + i]; s.c += ncar[base + i]; }
  Scan2 tot;
  block_exclusive_scan2(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single block: tile sums -> exclusive prefixes in place; grand totals to the two [n] entries and to totals[0..1]
__global__ void __launch_bounds__(kScanBlock) k_scan2_spine(Scan2* tile_sums, uint64_t ntiles, uint64_t* var_end, uint64_t* car_end,
                                                            uint64_t* totals) {
  Scan2 carry{0, 0};
  for (uint64_t base = 0; base < ntiles; base += kScanBlock) {
    const uint64_t i = base + threadIdx.x;
    const Scan2 v = i < ntiles ? tile_sums[i] : Scan2{0, 0};
    Scan2 tot;
    const Scan2 ex = block_exclusive_scan2(v, &tot);
    if (i < ntiles) tile_sums[i] = Scan2{carry.a + ex.a, carry.c + ex.c};
    carry.a += tot.a; carry.c += tot.c;
  }
  if (threadIdx.x == 0) {
    *var_end = carry.a; *car_end = carry.c;
    if (totals) { totals[0] = carry.a; totals[1] = carry.c; }
  }
}

__global__ void __launch_bounds__(kScanBlock) k_scan2_apply(const uint64_t* nvar, const uint64_t* ncar, uint64_t n, const Scan2* tile_sums,
                                                            uint64_t* var_begin, uint64_t* car_base) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  Scan2 loc[kScanItems];
  Scan2 s{0, 0};
  for (int i = 0; i < kScanItems; ++i) {
    loc[i] = base + i < n ? Scan2{nvar[base + i], ncar[base + i]} : Scan2{0, 0};
    s.a += loc[i].a; s.c += loc[i].c;
  }
  Scan2 tot;
  Scan2 ex = block_exclusive_scan2(s, &tot);
  const Scan2 ts = tile_sums[blockIdx.x];
  ex.a += ts.a; ex.c += ts.c;
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) { var_begin[base + i] = ex.a; car_base[base + i] = ex.c; }
    ex.a += loc[i].a; ex.c += loc[i].c;
  }
}

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
// PART: 0 = the whole task; 1 = only its listed variants, 2 = only its denser ones (the two halves of a launch pair that
// runs side by side on two streams: the list half is loads and stores, the row half LDS and vector work).
template <bool WIDE, bool EARLY_NIB, bool TUNE = false, int PART = 0>
__device__ __forceinline__ void expand_task(const DevImage& im, void* arena, uint32_t* lds_wave, uint32_t lane, uint32_t cnt, uint32_t cls,
                                            uint64_t gt0, uint64_t cb, uint32_t ablate_arg, uint32_t gt_words) {
  const uint32_t ablate = (TUNE ? ablate_arg : 0u) | (PART == 2 ? 1u : 0u);   // production instantiations carry no ablation tests
  const uint32_t wpc = im.wpc;
  const uint64_t* __restrict__ class_rows = im.class_rows;
  const uint8_t* __restrict__ gtp = im.gt_nibbles;
  // carrier word in the arena: 16 bits when every sample id fits 13 bits (the non-WIDE instantiation), else 32
  using CT = typename std::conditional<WIDE, uint32_t, uint16_t>::type;
  CT* __restrict__ carriers = reinterpret_cast<CT*>(arena);
  uint32_t m_lo = 0xE000u, m_hi = 0xE0000000u;   // genotype fields of the two 16-bit carrier words in a dword
  asm volatile("" : "+s"(m_lo), "+s"(m_hi));
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
        uint2 nw;
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
  uint64_t dmask = PART == 1 ? 0ULL : __ballot(cnt > list_max && !explicit_ids);
  uint64_t word_cur = 0, word_n1 = 0, word_n2 = 0;   // bit rows of the current dense variant and of the next two
  uint4 nq0 = {0, 0, 0, 0}, nq1 = {0, 0, 0, 0};      // raw genotype nibbles of the current one (then of the next)
  if (dmask) {
    const int t0 = __builtin_ctzll(dmask);
    const uint32_t cls_0 = __builtin_amdgcn_readlane(cls, t0), cnt_0 = __builtin_amdgcn_readlane(cnt, t0);
    const uint64_t gt0_0 = wave_bcast64(gt0, t0);
    if (lane < wpc) word_cur = class_rows[(uint64_t)cls_0 * wpc + lane];
    if (!lists || EARLY_NIB) {
      const uint64_t b0 = (gt0_0 >> 1) & ~15ULL;                        // aligned byte base
      const uint64_t need = ((gt0_0 + cnt_0 + 1) >> 1) - b0;            // bytes that hold this variant's nibbles
      if ((uint64_t)lane * 16 < need) nq0 = *reinterpret_cast<const uint4*>(gtp + b0 + lane * 16);
      if ((uint64_t)lane * 16 + 1024 < need) nq1 = *reinterpret_cast<const uint4*>(gtp + b0 + 1024 + lane * 16);
    }
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
    const bool sp = cnt > 0 && cnt <= list_max && !(ablate & 1);
    const uint32_t c = sp ? (cnt + kCarAlign - 1) / kCarAlign : 0u;
    const uint32_t incl = wave_inclusive_scan(c);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
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
        uint2 nw;
        __builtin_memcpy(&nw, gt32 + (g >> 3), 8);
        const uint32_t n = __builtin_amdgcn_alignbit(nw.y, nw.x, ((uint32_t)g & 7u) * 4);
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
    const bool sp = cnt > 0 && cnt <= list_max && !(ablate & 1);
    const uint32_t c = sp ? (cnt + kCarAlign - 1) / kCarAlign : 0u;
    const uint32_t incl = wave_inclusive_scan(c);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    if (total) {
      uint32_t* s_idb = s_off + 64;
      uint64_t* s_gt0 = reinterpret_cast<uint64_t*>(s_off + 128);
      uint64_t* s_cb = reinterpret_cast<uint64_t*>(s_off + 256);
      s_off[lane] = incl - c;
      s_idb[lane] = cls;      // listed variants: group index of the class's list (DevImage::v_src)
      s_gt0[lane] = gt0;
      s_cb[lane] = cb;
      const uint4* __restrict__ list_groups = reinterpret_cast<const uint4*>(im.cls_list16);
      const uint32_t* __restrict__ gt32 = reinterpret_cast<const uint32_t*>(gtp);
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
        const uint64_t g0 = s_gt0[L0] + (uint64_t)k0 * kCarAlign;   // its first genotype nibble: 32 bits from bit 4g
        const uint64_t g1 = s_gt0[L1] + (uint64_t)k1 * kCarAlign;
        const uint4 iw0 = list_groups[(uint64_t)s_idb[L0] + k0];
        uint2 nw0, nw1 = {0, 0};
        __builtin_memcpy(&nw0, gt32 + (g0 >> 3), 8);
        uint4 iw1 = {0, 0, 0, 0};
        if (two) {
          iw1 = list_groups[(uint64_t)s_idb[L1] + k1];
          __builtin_memcpy(&nw1, gt32 + (g1 >> 3), 8);
        }
        const uint64_t dst0 = (s_cb[L0] >> 3) + k0, dst1 = (s_cb[L1] >> 3) + k1;   // arena ranges start on group boundaries
        {
          const uint32_t n = __builtin_amdgcn_alignbit(nw0.y, nw0.x, ((uint32_t)g0 & 7u) * 4);
          uint4 v;
          v.x = and_or(n << 25, m_hi, and_or(n << 13, m_lo, iw0.x));
          v.y = and_or(n << 17, m_hi, and_or(n << 5, m_lo, iw0.y));
          v.z = and_or(n << 9, m_hi, and_or(n >> 3, m_lo, iw0.z));
          v.w = and_or(n << 1, m_hi, and_or(n >> 11, m_lo, iw0.w));
          store_group_nt(&arena_groups[dst0], v);
        }
        if (two) {
          const uint32_t n = __builtin_amdgcn_alignbit(nw1.y, nw1.x, ((uint32_t)g1 & 7u) * 4);
          uint4 v;
          v.x = and_or(n << 25, m_hi, and_or(n << 13, m_lo, iw1.x));
          v.y = and_or(n << 17, m_hi, and_or(n << 5, m_lo, iw1.y));
          v.z = and_or(n << 9, m_hi, and_or(n >> 3, m_lo, iw1.z));
          v.w = and_or(n << 1, m_hi, and_or(n >> 11, m_lo, iw1.w));
          store_group_nt(&arena_groups[dst1], v);
        }
      }
    }
  }

  if (dmask == 0) return;
  if (lists && !EARLY_NIB) {   // the first dense variant's nibbles (8 registers) are requested after the list phase: its peak register
                 // demand decides how many waves a SIMD holds
    const int t0 = __builtin_ctzll(dmask);
    const uint32_t cnt_0 = __builtin_amdgcn_readlane(cnt, t0);
    const uint64_t gt0_0 = wave_bcast64(gt0, t0);
    const uint64_t b0 = (gt0_0 >> 1) & ~15ULL;
    const uint64_t need = ((gt0_0 + cnt_0 + 1) >> 1) - b0;
    if ((uint64_t)lane * 16 < need) nq0 = *reinterpret_cast<const uint4*>(gtp + b0 + lane * 16);
    if ((uint64_t)lane * 16 + 1024 < need) nq1 = *reinterpret_cast<const uint4*>(gtp + b0 + 1024 + lane * 16);
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
    } else {      // raw nibbles, behind the row staging area
      uint8_t* nib_st = gt_lds + 16;   // 16 bytes (32 nibbles) of bias: see the copy-out
      *reinterpret_cast<uint4*>(nib_st + lane * 16) = nq0;
      if ((uint64_t)lane * 32 + 2048 < (uint64_t)nshift + cnt_t) *reinterpret_cast<uint4*>(nib_st + 1024 + lane * 16) = nq1;
    }
    // request the next variant's nibbles and the row of the one after it before expanding this one (rows are the
    // random 320-byte reads of this kernel: two of them stay in flight per wave)
    if (WIDE) { nq0 = uint4{0, 0, 0, 0}; nq1 = uint4{0, 0, 0, 0}; }   // (the slice path never reads nibbles it did not load)
    word_n2 = 0;
    if (dmask) {
      const int tn = __builtin_ctzll(dmask);
      const uint64_t gt0_n = wave_bcast64(gt0, tn);
      const uint32_t cnt_n = __builtin_amdgcn_readlane(cnt, tn);
      const uint64_t bn = (gt0_n >> 1) & ~15ULL;
      const uint64_t need = ((gt0_n + cnt_n + 1) >> 1) - bn;
      if ((uint64_t)lane * 16 < need) nq0 = *reinterpret_cast<const uint4*>(gtp + bn + lane * 16);
      if ((uint64_t)lane * 16 + 1024 < need) nq1 = *reinterpret_cast<const uint4*>(gtp + bn + 1024 + lane * 16);
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
      const uint8_t* nib_lds = gt_lds;
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
      // nibble index = list index + D; the staging is biased by 32 nibbles
      uint32_t D = nshift + 32 - a1k;
      uint32_t pos = a1k;                                 // list index of the round's first carrier
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
        while (bits) {
          ids16[j++] = (uint16_t)(idb + __builtin_ctz(bits));
          bits &= bits - 1;
        }
        // copy-out: lane q of a pass owns list entries 8q..8q+7 (one 16-byte store); their nibbles are 32 consecutive
        // bits of the stream.  Round 0 writes complete groups only, round 1 everything (the range owns its padding).
        const uint32_t flush = round ? ((end + 7u) & ~7u) : (end & ~7u);
        for (uint32_t q8 = done8 + lane * 8; q8 < flush; q8 += 512) {
          const uint4 iw = *reinterpret_cast<const uint4*>(ids16 + q8);
          const uint32_t n0 = q8 + D;
          const uint32_t* np = reinterpret_cast<const uint32_t*>(nib_lds) + (n0 >> 3);
          const uint32_t n = __builtin_amdgcn_alignbit(np[1], np[0], (n0 & 7u) * 4);
          uint4 v;   // two carriers per word: id | gt << 13 in each half.  The masks live in SGPRs (made opaque once per
                     // kernel) so that every term is a shift plus one v_and_or_b32 -- VOP3 takes no literals on gfx9
          v.x = and_or(n << 25, m_hi, and_or(n << 13, m_lo, iw.x));
          v.y = and_or(n << 17, m_hi, and_or(n << 5, m_lo, iw.y));
          v.z = and_or(n << 9, m_hi, and_or(n >> 3, m_lo, iw.z));
          v.w = and_or(n << 1, m_hi, and_or(n >> 11, m_lo, iw.w));
          store_group_nt(reinterpret_cast<uint4*>(g1k + q8), v);   // a1k is a multiple of 8 and the range owns its padding (pad_car)
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
}

template <bool WIDE, uint32_t CH, bool TUNE, int PART = 0>
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
    expand_task<WIDE, false, TUNE, PART>(im, r.carriers, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], lane, cnt, cls, gt0, cb, ablate, gt_words);
  }
}

// The same expansion over the UNIQUE sites of a batch whose carrier lists are shared: the slot parameters come straight
// from the site table (sequential reads, each site once), the arena offset from k_unique_sites.
template <bool WIDE, uint32_t CH, bool TUNE, int PART = 0>
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
    expand_task<WIDE, false, TUNE, PART>(im, r.carriers, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], lane, cnt, cls, gt0, cb, ablate, gt_words);
  }
}

// ---------------------------------------------------------------------------
// Latency path: a batch of at most 64 regions in ONE launch.  Every wave works the region bounds out for itself (lane
// q takes region q: the same dozen loads in every wave, L2 hits after the first), a wave prefix sum lays the slot,
// arena and task offsets out, and the wave then takes 4-slot tasks straight from the site table: it writes their
// variant headers and expands their carriers (expand_task) -- no header kernel, no kernel-to-kernel dependency, no
// host round trip.  The regions travel in the kernel arguments.  The result buffers were sized on the host from the
// same arithmetic (engine.hip: host_region_size); should the device ever need more it writes nothing and says so.
// The last block to finish applies the literal "already seen" rule to the regions that need it and posts the
// completion mailbox.
// ---------------------------------------------------------------------------
template <int NMAX>
struct SmallRegions { uint64_t xy[2 * NMAX]; };

__device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t v, uint32_t lane) {
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v += t;
  }
  return v;
}

// The result of a small batch lives in ONE slab; host and device lay it out with the same arithmetic (the resident
// server gets only the slab address and the capacities with a request).
__host__ __device__ inline size_t small_result_layout(DevResult& d, uint8_t* slab, uint64_t n, uint64_t capA, uint64_t capS, uint32_t car_width) {
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  const size_t o_flags = take(n), o_g0 = take(n * 4), o_nvar = take(n * 8), o_ncar = take(n * 8), o_vb = take((n + 1) * 8),
               o_cb = take((n + 1) * 8), o_vc = take(n * 8), o_rows = take(capA * sizeof(VariantRow)),
               o_car = take(capS * car_width + 16);
  d.Q = n; d.A = capA; d.S = capS;
  d.regions = nullptr;
  d.q_flags = slab + o_flags; d.q_g0 = (uint32_t*)(slab + o_g0); d.q_nvar = (uint64_t*)(slab + o_nvar);
  d.q_ncar = (uint64_t*)(slab + o_ncar); d.var_begin = (uint64_t*)(slab + o_vb); d.car_base = (uint64_t*)(slab + o_cb);
  d.var_count = (uint64_t*)(slab + o_vc);
  d.rows = (VariantRow*)(slab + o_rows); d.r_class = nullptr; d.r_gt0 = nullptr; d.q_car_len = nullptr; d.carriers = slab + o_car;
  d.car_width = car_width; d.pad3_ = 0;
  return off;
}

// Bounds, offsets, per-region arrays and tasks of a small batch for ONE wave (see k_query_small).  `wave`/`nwaves`:
// this wave's place among the waves sharing the batch.  Returns any-slow | over << 1.
template <bool WIDE>
__device__ __forceinline__ uint32_t small_batch_wave(const DevImage& im, const DevResult& r, uint64_t x, uint64_t y, uint32_t n, uint32_t wave,
                                                     uint32_t nwaves, uint32_t* lds_wave, uint32_t gt_words, uint64_t cap_slots,
                                                     uint64_t cap_carriers, bool stamps) {
  constexpr uint32_t CH = kFillChunkSmall;
  const uint32_t lane = threadIdx.x & 63;
  // ---- bounds of region `lane`, offsets of all regions ----
  RegionBounds b{0, 0, 0, 0, 0, 0};
  if (lane < n) b = region_bounds_of(im, x, y);
  const uint64_t pre0 = b.pre0, npad = b.npad, nkept = b.nkept;
  const uint32_t nv = b.g1 - b.g0, ntask = (nv + CH - 1) / CH;
  const uint64_t vend = wave_inclusive_scan64(nv, lane), cend = wave_inclusive_scan64(npad, lane);
  const uint32_t tend = wave_inclusive_scan(ntask);
  const uint64_t A = wave_bcast64(vend, 63), S = wave_bcast64(cend, 63);
  const uint32_t T = __builtin_amdgcn_readlane(tend, 63);
  const bool any_slow = __ballot(b.flags & kRegionSlow) != 0;
  const bool over = A > cap_slots || S > cap_carriers;
  if (stamps && wave == 0 && lane == 0) __hip_atomic_store(&r.done_counter[2], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bounds and offsets done
  if (wave == 0 && !over) {   // the per-region arrays of the result
    if (lane < n) {
      r.q_flags[lane] = b.flags; r.q_g0[lane] = b.g0; r.q_nvar[lane] = nv;
      r.var_begin[lane] = vend - nv; r.car_base[lane] = cend - npad;
      if (!(b.flags & kRegionSlow)) { r.var_count[lane] = nv; r.q_ncar[lane] = nkept; }
    }
    if (lane == 0) { r.var_begin[n] = A; r.car_base[n] = S; }
  }
  // ---- tasks: 8 consecutive sites of one region ----
  if (!over) {
    for (uint32_t c = wave; c < T; c += nwaves) {
      // the region of task c: the last one whose first task is <= c (regions without tasks share their successor's offset)
      const uint32_t q = (uint32_t)__popcll(__ballot(lane < n && tend - ntask <= c)) - 1u;
      const uint32_t g0_q = __builtin_amdgcn_readlane(b.g0, q), nv_q = __builtin_amdgcn_readlane(nv, q);
      const uint32_t t0_q = __builtin_amdgcn_readlane(tend - ntask, q);
      const uint64_t a0_q = wave_bcast64(vend - nv, q), cb_q = wave_bcast64(cend - npad, q) - wave_bcast64(pre0, q);
      const uint32_t j = (c - t0_q) * CH + lane;
      uint32_t cnt = 0, cls = 0;
      uint64_t gt0 = 0, cb = 0;
      if (lane < CH && j < nv_q) {
        const uint32_t g = g0_q + j;
        const uint64_t a = a0_q + j;
        const uint32_t fl = im.s_flags[g];
        cnt = im.s_ncar[g];
        cls = im.s_class[g];
        gt0 = im.s_gt0[g];
        cb = cb_q + im.s_carpre[g];
        // the variant row (building std::vector<Variant>, query.h:736-771)
        row_store(r.rows, a, im.s_pos[g], im.s_ref_off[g], im.s_ref_len[g], im.s_alt_off[g], im.s_alt_len[g], cnt,
                  (fl & kSiteAlwaysDrop) != 0, cb);
      }
      expand_task<WIDE, true>(im, r.carriers, lds_wave, lane, cnt, cls, gt0, cb, 0u, gt_words);
    }
  }
  return (any_slow ? 1u : 0u) | (over ? 2u : 0u);
}

template <bool WIDE, int NMAX>
__global__ void __launch_bounds__(256) k_query_small(DevImage im, DevResult r, SmallRegions<NMAX> regs, uint32_t gt_words,
                                                     uint64_t cap_slots, uint64_t cap_carriers) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t n = (uint32_t)r.Q;
  const bool stamps = r.host_totals != nullptr;   // VS_LAT_DEBUG: device clock, 100 MHz
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&r.done_counter[1], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  const uint32_t lds_words_per_wave = WIDE ? gt_words + kRingWords : slice_lds_words(im.num_samples);
  uint64_t x = 0, y = 0;
  if (lane < n) { x = regs.xy[2 * lane]; y = regs.xy[2 * lane + 1]; }
  const uint32_t st = small_batch_wave<WIDE>(im, r, x, y, n, wave, nwaves, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], gt_words,
                                             cap_slots, cap_carriers, stamps);
  const bool any_slow = st & 1u, over = st & 2u;
  // ---- completion: the last block applies the literal dedup rule where needed, then posts the mailbox.  The
  //      flag word carries everything the host does not know yet: sequence number | any-slow << 62 | over << 63 (the
  //      sizes are the host's own).  Without dedup work nobody reads another block's data inside this launch, so a
  //      block only waits until its own stores are acknowledged (no L2 write-back) before it counts itself done. ----
  __shared__ uint32_t s_last;
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&r.done_counter[3], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // block 0's first wave is through
  if (any_slow) __threadfence();
  else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(r.done_counter, 1ULL) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last) {
    if (any_slow && !over) {
      __threadfence();   // the headers other blocks wrote
      for (uint32_t q = threadIdx.x; q < n; q += blockDim.x)
        if (r.q_flags[q] & kRegionSlow) dedup_region(im, r, q);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      *r.done_counter = 0;   // re-armed for the next launch on this stream
      if (stamps) {   // device-clock durations in 10 ns ticks {kernel, bounds + offsets, block 0's tasks}
        const uint64_t t_end = wall_clock64();   // (the stamps come from another block, possibly another XCD: agent-scope loads)
        const uint64_t t1 = __hip_atomic_load(&r.done_counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t2 = __hip_atomic_load(&r.done_counter[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t3 = __hip_atomic_load(&r.done_counter[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.host_totals[0] = t_end - t1;
        r.host_totals[1] = t2 - t1;
        r.host_totals[2] = t3 - t2;
        __threadfence_system();
      }
      __hip_atomic_store(const_cast<uint64_t*>(r.done_flag), r.done_seq | (any_slow ? 1ULL << 62 : 0ULL) | (over ? 1ULL << 63 : 0ULL),
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------------------
// Resident query server: the same work as k_query_small without a launch per query.  A small grid stays on the GPU for
// a bounded time and polls a request line in mapped host memory; the host posts {sequence number, slab, capacities,
// regions}, every block picks the request up by itself (no device-side broadcast), the waves share the tasks, each
// block pushes its results out with a system-scope release and counts itself done, the last one applies the dedup
// rule where needed and posts the sequence number back.  A request costs one PCIe round trip instead of a kernel
// dispatch (2.5 us against 6 us + the launch call, tools/microbench/pingpong.hip / latency_floor.hip).
// Every loop is bounded by the device clock: a block leaves `life_ticks` after its start or `idle_ticks` after the last
// request whatever the host does (so a device-wide synchronisation elsewhere in the process waits a millisecond at most), no
// block ever waits for another one, and a request caught by a block's exit simply is not answered -- the host then
// falls back to the launch path (engine.hip).
// ---------------------------------------------------------------------------
struct ServerRequest {        // mapped host memory, 64-byte aligned; the host writes the body first, then tail, then head
  uint64_t head;              // sequence number; ~0 = leave
  uint64_t slab;              // device address of the result slab (small_result_layout)
  uint64_t cap_slots, cap_carriers;
  uint64_t n_and_width;       // regions | carrier width << 32
  uint64_t x0, y0;            // the first region (a single-region request is this one line)
  uint64_t tail;              // server_request_tail(head, body): seals the six words above
  uint64_t xy[128];           // all regions
};
static_assert(sizeof(ServerRequest) == 64 + 1024, "request layout");
// The tail word seals the line: sequence number mixed with a checksum of the six body words.  The device accepts a
// line only when head == the expected sequence number AND tail matches the body it read, so the hand-off does not
// depend on the eight 8-byte loads of the poll being served as one 64-byte transaction (a torn read -- new head and
// tail, old body -- fails the checksum and is simply polled again).
__host__ __device__ inline uint64_t server_request_tail(uint64_t seq, const uint64_t body[6]) {
  uint64_t h = seq * 0x9E3779B97F4A7C15ULL;
  for (int i = 0; i < 6; ++i) { h = (h ^ body[i]) * 0xff51afd7ed558ccdULL; h ^= h >> 29; }
  return h;
}

template <bool WIDE>
__global__ void __launch_bounds__(256) k_query_server(DevImage im, const ServerRequest* req, unsigned long long* done_counter,
                                                      volatile uint64_t* done_flag, uint64_t first_seq, uint32_t gt_words,
                                                      uint64_t life_ticks, uint64_t idle_ticks) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  const uint32_t lds_words_per_wave = WIDE ? gt_words + kRingWords : slice_lds_words(im.num_samples);
  __shared__ uint64_t s_req[8];
  __shared__ uint64_t s_xy[128];
  __shared__ uint32_t s_state;   // 0 idle, 1 request in s_req, 2 leave
  __shared__ uint32_t s_last;
  const uint64_t t_start = wall_clock64();
  uint64_t t_last = t_start;       // (every block sees the same requests: the idle clocks agree to within microseconds)
  uint64_t expect = first_seq;
  const uint64_t* reqw = reinterpret_cast<const uint64_t*>(req);
  while (true) {
    // ---- wave 0 polls the request line: lanes 0..7 read its eight words in one access ----
    if (threadIdx.x < 64) {
      uint32_t state = 0;
      while (state == 0) {
        uint64_t w = 0;
        if (lane < 8) w = __hip_atomic_load(reqw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t head = wave_bcast64(w, 0), tail = wave_bcast64(w, 7);
        const uint64_t now = wall_clock64();
        bool sealed = false;
        if (head == expect) {
          const uint64_t body[6] = {wave_bcast64(w, 1), wave_bcast64(w, 2), wave_bcast64(w, 3), wave_bcast64(w, 4), wave_bcast64(w, 5), wave_bcast64(w, 6)};
          sealed = tail == server_request_tail(expect, body);
        }
        if (head == ~0ULL || now - t_start > life_ticks || now - t_last > idle_ticks) state = 2;
        else if (sealed) {
          if (lane < 8) s_req[lane] = w;
          const uint32_t n = (uint32_t)wave_bcast64(w, 4) & 0xFFFFu;
          if (n > 1) {   // the other regions: one more round trip
            s_xy[2 * lane] = lane < n ? __hip_atomic_load(reqw + 8 + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
            s_xy[2 * lane + 1] = lane < n ? __hip_atomic_load(reqw + 9 + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
          }
          state = 1;
        } else __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) s_state = state;
    }
    __syncthreads();
    if (s_state == 2) break;
    // ---- the request ----
    const uint64_t seq = s_req[0];
    const uint32_t n = (uint32_t)s_req[4] & 0xFFFFu, car_width = (uint32_t)(s_req[4] >> 32);
    const bool stamps = (s_req[4] >> 16) & 1;   // VS_LAT_DEBUG: device-clock stamps into done_flag[1..4] (block 0, last block)
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) done_flag[1] = wall_clock64();
    DevResult r{};
    small_result_layout(r, reinterpret_cast<uint8_t*>(s_req[1]), n, s_req[2], s_req[3], car_width);
    r.done_counter = done_counter;
    uint64_t x = 0, y = 0;
    if (lane < n) { x = n > 1 ? s_xy[2 * lane] : s_req[5]; y = n > 1 ? s_xy[2 * lane + 1] : s_req[6]; }
    const uint32_t st = small_batch_wave<WIDE>(im, r, x, y, n, wave, nwaves, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], gt_words,
                                               s_req[2], s_req[3], false);
    const bool any_slow = st & 1u, over = st & 2u;
    // ---- completion: results must be out of this XCD's L2 before anybody is told (the kernel does not end here) ----
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) done_flag[2] = wall_clock64();
    __threadfence_system();
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) done_flag[3] = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done_counter, 1ULL) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last) {
      if (any_slow && !over) {
        __threadfence();
        for (uint32_t q = threadIdx.x; q < n; q += blockDim.x)
          if (r.q_flags[q] & kRegionSlow) dedup_region(im, r, q);
        __threadfence_system();
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        *done_counter = 0;
        __threadfence();
        if (stamps) { done_flag[4] = wall_clock64(); __threadfence_system(); }
        __hip_atomic_store(const_cast<uint64_t*>(done_flag), seq | (any_slow ? 1ULL << 62 : 0ULL) | (over ? 1ULL << 63 : 0ULL),
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    expect = seq + 1;
    t_last = wall_clock64();
    __syncthreads();   // s_req / s_xy are rewritten by the next poll
  }
}

// Hit-list records for a collective: 4 x uint64 per reported row of every region, regions back to back
//   {pos | dropped << 63, ref_off | ref_len << 32, alt_off | alt_len << 32, region | car_count << 32}
// (one wave per region; slot_begin = exclusive scan of the regions' row counts)
__global__ void __launch_bounds__(256) k_pack_headers(DevResult r, uint64_t* dst, const uint64_t* slot_begin, uint64_t region_base) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= r.Q) return;
  const uint64_t n = r.q_nvar[q], a0 = r.var_begin[q], o0 = slot_begin[q];
  for (uint64_t j = threadIdx.x & 63; j < n; j += 64) {
    const VariantRow v = row_load(r.rows, a0 + j);
    uint64_t* d = dst + 4 * (o0 + j);
    d[0] = (uint64_t)v.pos | (row_dropped(v) ? (1ULL << 63) : 0ULL);
    d[1] = (uint64_t)v.ref_off | ((uint64_t)v.ref_len << 32);
    d[2] = (uint64_t)v.alt_off | ((uint64_t)v.alt_len << 32);
    d[3] = (region_base + q) | ((uint64_t)row_count(v) << 32);
  }
}

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
// Event bitmaps of query type 4 (DevImage::t4_events), built once when an index is opened.
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

__global__ void __launch_bounds__(256) k_build_events(DevImage im, uint64_t* events) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t tile = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t ntiles = (im.P + 63) >> 6;
  if (tile >= ntiles) return;
  const uint64_t j = tile * 64 + lane;
  const bool valid = j < im.P;
  const uint64_t irr = __ballot(valid && slot_is_irregular(im, j));
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
    if (sample >= 1 && sample < im.num_samples) events[(uint64_t)sample * im.t4_stride + tile] = mine | irr;
  }
}

// explicit-id cohorts (no class rows): the rows start as the irregular mask, then every carrier record of a slot's
// node and of its out-neighbours sets its sample's bit
__global__ void __launch_bounds__(256) k_events_irregular_rows(DevImage im, uint64_t* events) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t tile = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t ntiles = (im.P + 63) >> 6;
  if (tile >= ntiles) return;
  const uint64_t j = tile * 64 + lane;
  const uint64_t irr = __ballot(j < im.P && slot_is_irregular(im, j));
  for (uint32_t s = 1 + lane; s < im.num_samples; s += 64) events[(uint64_t)s * im.t4_stride + tile] = irr;
}
__global__ void __launch_bounds__(256) k_events_explicit(DevImage im, uint64_t* events) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= im.P) return;
  const uint32_t v = im.rp_vid[j];
  const unsigned long long bit = 1ULL << (j & 63);
  const uint32_t e0 = im.row_ptr[v], e1 = im.row_ptr[v + 1];
  for (uint32_t e = e0; e <= e1; ++e) {             // e == e1: the node itself
    const uint32_t u = e < e1 ? im.col[e] : v;
    const uint64_t b = im.v_car_begin[u];
    for (uint32_t i = 0; i < im.v_ncar[u]; ++i) {
      const uint32_t sid = im.car_sid[b + i];
      if (sid >= 1 && sid < im.num_samples) atomicOr((unsigned long long*)&events[(uint64_t)sid * im.t4_stride + (j >> 6)], bit);
    }
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
__global__ void __launch_bounds__(256) k_hold_explicit(DevImage im, uint64_t* hold) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= im.V) return;
  const uint64_t b = im.v_car_begin[v];
  for (uint32_t i = 0; i < im.v_ncar[v]; ++i) {
    const uint32_t sid = im.car_sid[b + i];
    if (sid >= 1 && sid < im.num_samples) atomicOr((unsigned long long*)&hold[(uint64_t)sid * im.t4_hold_stride + (v >> 6)], 1ULL << (v & 63));
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
struct WalkScratch {
  const uint64_t* cap_begin;   // [Q+1] exclusive scan of the capacities
  uint64_t* pos;
  uint32_t *cur, *ro, *rl, *ao, *al;
  uint64_t* overflow;          // set to 1 when a region outgrew its capacity (the host then takes the two-walk path)
  unsigned long long* stats;   // tuning builds (VS_TUNING): 16 counters of k_sample_walk (iteration counts, device-clock ticks); else NULL
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
struct BitRow {
  const uint64_t* __restrict__ row;
  uint32_t w;          // index of the cached word (kNone: nothing cached)
  uint64_t word;
  __device__ __forceinline__ uint64_t at(uint32_t wi) {
    if (wi != w) { w = wi; word = row[wi]; }
    return word;
  }
  __device__ __forceinline__ bool bit(uint32_t i) { return (at(i >> 6) >> (i & 63)) & 1; }
  // first index >= m whose bit is set, or `limit` when there is none below it (m < limit)
  __device__ __forceinline__ uint32_t next(uint32_t m, uint32_t limit) {
    uint32_t wi = m >> 6;
    const uint32_t w_end = (limit + 63) >> 6;
    uint64_t x = at(wi) & (~0ULL << (m & 63));
    while (!x) {
      if (++wi >= w_end) return limit;
      x = at(wi);
    }
    const uint32_t k = (wi << 6) + (uint32_t)__builtin_ctzll(x);
    return k < limit ? k : limit;
  }
};
typedef BitRow EventRow;

// ---- the walk of get_sample_var_in_ref as reusable pieces (serial kernel k_sample_walk, cooperative k_sample_walk_coop) ----
// Two data paths, chosen per region: BLOB (the sample has event + hold rows: records from the walk blob, "does v hold
// the sample" from the hold row, jumps over uneventful runs) and plain (sample 0 = "ref", or an index without the rows:
// the round-2 records, class rows, every vertex visited).  WalkVertex::row_begin indexes the blob resp. w_edge.
struct WalkCtx { uint32_t sid; uint64_t x, y; bool use_ev; uint32_t limit; };
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
__device__ __forceinline__ bool walk_holds(const DevImage& im, BitRow& hold, uint32_t v, uint32_t ridx, uint32_t cls, uint32_t sid) {
  if (BLOB) return hold.bit(v);
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
        const bool holds = hold.bit(n);
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
      const bool holds = sid != 0 && walk_holds<BLOB>(im, hold, n, nr, a.z, sid);
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
  const bool cur_holds = st.ref_pos >= cx.x && walk_holds<BLOB>(im, hold, st.cur, st.wc.ridx, st.wc.cls, cx.sid);
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
        const uint4 a = im.wblob[2 * (uint64_t)e];
        if (a.y) { ref_pos = a.y; had_ref = true; }
        if (hold.bit(a.x)) {
          const uint4 b = im.wblob[2 * (uint64_t)e + 1];
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
  return 0;
}

// The serial walk of one region: the reference's loop, with jumps over uneventful runs in BLOB mode.  `sink(em)` takes
// each reported vertex.
template <bool BLOB, typename Sink>
__device__ __forceinline__ void walk_serial(const DevImage& im, const WalkCtx& cx, BitRow& ev, BitRow& hold, WalkSt& st, Sink&& sink,
                                            uint32_t& st_jumps, uint32_t& st_steps) {
  bool done = false;
  while (!done) {
    if (st.ref_pos >= cx.y) break;
    if (BLOB && walk_in_step(cx, st)) {
      // On a ref-path node, in step with it (ref_pos == its index): up to the next event slot k the literal loop
      // would take the default step node by node -- no neighbour holds the sample (next = path successor), the
      // node itself does not (nothing emitted), every node is regular (ref_pos and cur_ref follow the path) -- and
      // stop at `limit` if that comes first.
      const uint32_t k = ev.next(st.cur_slot1 - 1, cx.limit);
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
  if (q >= r.Q) return;
  WalkCtx cx{sid_per_region ? sid_per_region[q] : sid_all, r.regions[2 * q], r.regions[2 * q + 1], false, 0};
  // Event and hold rows of this sample (DevImage::t4_events, t4_hold): clear event bits are ref-path slots where neither
  // the node nor any of its out-neighbours holds the sample and the node is regular -- the reference's loops provably do
  // nothing there but step on, so both the backward search and the walk jump over them.  Everything that happens at a
  // set bit is the literal code.
  cx.use_ev = im.t4_events && cx.sid != 0;
  uint64_t nvar = 0, ncar = 0, ncar_kept = 0, rank0 = 0;
  const uint8_t fl = walk_prologue(im, cx, rank0);
  if (!fl) {
    BitRow ev{cx.use_ev ? im.t4_events + (uint64_t)cx.sid * im.t4_stride : nullptr, kNone, 0};
    BitRow hold{cx.use_ev ? im.t4_hold + (uint64_t)cx.sid * im.t4_hold_stride : nullptr, kNone, 0};
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
    if (cx.use_ev) walk_serial<true>(im, cx, ev, hold, st, sink, st_jumps, st_steps);
    else walk_serial<false>(im, cx, ev, hold, st, sink, st_jumps, st_steps);
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

template <uint32_t SUB>
__global__ void __launch_bounds__(256) k_sample_walk_coop(DevImage im, DevResult r, uint32_t sid_all, const uint32_t* sid_per_region,
                                                          WalkScratch ws) {
  static_assert(SUB == 8 || SUB == 16, "group width");
  constexpr uint32_t kGroupMask = (1u << SUB) - 1u;
  const uint32_t lane = threadIdx.x & 63, l = lane & (SUB - 1), gbase = lane & (64 - SUB);
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
  const bool live = q < r.Q;
  WalkCtx cx{0, 0, 0, false, 0};
  if (live) { cx.sid = sid_per_region ? sid_per_region[q] : sid_all; cx.x = r.regions[2 * q]; cx.y = r.regions[2 * q + 1]; }
  cx.use_ev = live && im.t4_events && cx.sid != 0;
  // group-uniform state: every lane of a group computes / receives the same values
  uint64_t nvar = 0, ncar = 0, rank0 = 0;
  uint8_t fl = 0;
  bool busy = false;         // the group still has episodes to run
  bool serial = false;       // the group walks its region with the serial loop (no event rows, or a fallback)
  uint32_t cur_slot = 0;     // slot at which the chain is in step
  uint64_t s0 = 0, scap = 0;
  BitRow ev{nullptr, kNone, 0}, hold{nullptr, kNone, 0};
  WalkSt st{};
  const uint64_t t_c0 = VS_WALK_CLOCK();
  uint64_t t_c1 = t_c0, t_c2 = t_c0;
  uint32_t n_chunks = 0, n_search = 0;
  if (live) {
    fl = walk_prologue(im, cx, rank0);
    s0 = ws.cap_begin[q]; scap = ws.cap_begin[q + 1] - s0;
    if (!fl) {
      ev.row = cx.use_ev ? im.t4_events + (uint64_t)cx.sid * im.t4_stride : nullptr;
      hold.row = cx.use_ev ? im.t4_hold + (uint64_t)cx.sid * im.t4_hold_stride : nullptr;
      if (!cx.use_ev) serial = true;
    }
  }
  // ---- get_prev_vertex_with_sample, 16 ranks of the chain at a time ----
  // The search visits rank, rank - deg(previous(rank)), ... (one count per neighbour of each visited node).  The group
  // reads the records of the 16 ranks below the current one together, follows the chain through them in registers
  // (which of the 16 are visited), tests the visited nodes' event bits, and checks the candidates literally in
  // parallel; the first visited node with a neighbour holding the sample is the answer.  Its ref_pos is the node's own
  // last ref neighbour; a node without one (the end of the path) would need the history: serial loop.
  bool searching = live && !fl && cx.use_ev;
  uint64_t rank = rank0;
  while (__any(searching)) {
    if (searching && rank <= 1) {   // the head of the path (redundant in the group)
      const uint32_t v = im.rp_vid[im.rk_back[rank == 0 ? 0 : rank - 1].x];
      st.cur = v; st.wc = blob_vertex(im, v); st.ref_pos = 1; st.cur_ref_v = kNone;
      st.cur_slot1 = im.w_vertex[2 * (uint64_t)v + 1].w;
      searching = false;
    }
    ++n_search;
    const bool valid = searching && rank >= (uint64_t)l + 2;
    uint2 back{0, 1};
    if (valid) back = im.rk_back[rank - l - 1];
    uint32_t vis = 0, pos = 0;   // group-uniform: chain positions visited among the 16, next position
#pragma unroll 1
    for (int t = 0; t < (int)SUB; ++t) {
      const int src = (int)gbase + (int)(pos < SUB ? pos : SUB - 1);
      const uint32_t deg_c = (uint32_t)__shfl((int)back.y, src, 64);
      const bool val_c = __shfl((int)valid, src, 64) != 0;
      if (pos < SUB && val_c) { vis |= 1u << pos; pos += deg_c ? deg_c : 1u; }
    }
    const bool cand = valid && ((vis >> l) & 1) && ev.bit(back.x);
    bool found = false, had_ref = false;
    uint32_t f_ref_pos = 0, f_v = 0, f_slot1 = 0;
    WalkVertex f_wc{};
    if (cand) {
      const uint32_t rb0 = im.blob_of_slot[back.x] + 1;   // the edge records follow the slot's header
      for (uint32_t e = rb0; e < rb0 + back.y; ++e) {
        const uint4 a = im.wblob[2 * (uint64_t)e];
        if (a.y) { f_ref_pos = a.y; had_ref = true; }
        if (hold.bit(a.x)) {
          const uint4 b = im.wblob[2 * (uint64_t)e + 1];
          f_v = a.x; found = true; f_slot1 = b.y;
          f_wc = WalkVertex{a.w, b.x, a.y, 0u, b.z, a.z, b.w};
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
        if (!g_had_ref) serial = true;   // (ref_pos would be an earlier iteration's: the serial loop knows)
        st.cur = g_v; st.wc = g_wc; st.ref_pos = g_ref_pos; st.cur_ref_v = kNone; st.cur_slot1 = g_slot1;
        searching = false;
      } else rank = rank > pos ? rank - pos : 0;
    }
  }
  t_c1 = VS_WALK_CLOCK();
  if (live && !fl && cx.use_ev && !serial) {
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
    uint64_t word = (busy && wi < w_end) ? ev.row[wi] : 0;
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
      ev.w = kNone; hold.w = kNone;
      auto sink = [&](const WalkEmit& e1) {
        if (nvar < scap) {
          if (l == 0) { const uint64_t s = s0 + nvar; ws.pos[s] = e1.ref_pos; ws.cur[s] = e1.cur; ws.ro[s] = e1.kind; ws.rl[s] = e1.cur_ref_v; }
        } else if (l == 0) *ws.overflow = 1;
        nvar++; ncar += pad_car(e1.c);
      };
      if (cx.use_ev) { walk_start_search<true>(im, cx, ev, hold, rank0, st, it, lit); walk_serial<true>(im, cx, ev, hold, st, sink, jm, sp); }
      else { walk_start_search<false>(im, cx, ev, hold, rank0, st, it, lit); walk_serial<false>(im, cx, ev, hold, st, sink, jm, sp); }
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
  uint64_t n = live ? r.q_nvar[q] : 0;
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

// ---------------------------------------------------------------------------
// Point queries (types 1 and 7).  A single next_variant_in_ref(pos) call with an empty `vars`
// walks the ref path from find(pos) and stops at the first node with a reportable branch, so its
// answer is the branch list of ONE ref-path slot: the first slot >= slot(find(pos)) whose sites
// carry anybody (always-dropped sites have s_ncar == 0, reportable ones >= 1).  s_carpre over
// rp_cand_prefix is monotone in the slot, so that slot is found by bisection.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t slot_of_find(const DevImage& im, uint64_t pos) {  // Index::find, index.h:119-133
  uint64_t rf;
  if (pos >= im.ref_length) rf = im.R - 1;
  else { const uint32_t k = rank1(im, pos); rf = k == 0 ? 0 : k - 1; }
  return im.rank_to_slot[rf];
}

__device__ __forceinline__ uint32_t next_valid_slot(const DevImage& im, uint32_t s0) {
  const uint32_t P = (uint32_t)im.P;
  if (s0 >= P) return P;
  const uint64_t base = im.s_carpre[im.rp_cand_prefix[s0]];
  if (im.s_carpre[im.G] == base) return P;
  uint32_t lo = s0, hi = P - 1;
  while (lo < hi) {
    const uint32_t m = lo + ((hi - lo) >> 1);
    if (im.s_carpre[im.rp_cand_prefix[m + 1]] > base) hi = m; else lo = m + 1;
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
    for (uint32_t i = 0; i < w; ++i) rank += __popcll(row[i]);
    rank -= (uint32_t)(row[0] & 1);  // the ref entry is not part of the pool
    index = im.car_index[im.v_car_begin[v] + rank];
    return true;
  }
  const uint64_t b = im.v_car_begin[v];
  for (uint32_t i = 0; i < im.v_ncar[v]; ++i)
    if (im.car_sid[b + i] == sid) { index = im.car_index[b + i]; return true; }
  return false;
}
__device__ __forceinline__ bool sample_entry(const DevImage& im, uint32_t v, uint32_t sid, uint32_t& index) {
  return sample_entry_rec(im, v, im.v_ridx[v], im.use_bv ? im.v_class[v] : 0u, sid, index);
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

// Capacities of the single recording walk of type 5: branch sites of the reference range [x, y) widened by the
// region's own length (the sample's coordinates are shifted against the reference's by its net indel length).
__global__ void __launch_bounds__(256) k_walk_caps_sc(DevImage im, DevResult r) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  const uint64_t x = r.regions[2 * q], y = r.regions[2 * q + 1];
  const uint64_t margin = (y > x ? y - x : 0) + 256;
  const uint64_t lo = x > margin + 1 ? x - margin : 1, hi = (y > x ? y : x) + margin;
  const uint32_t s0 = slot_of_find(im, lo), s1 = slot_of_find(im, hi);
  r.q_nvar[q] = (s1 >= s0 ? (uint64_t)(im.rp_cand_prefix[s1 + 1] - im.rp_cand_prefix[s0]) : 0) + 8;
}

// Query type 5.  One thread per region; MODE as in k_sample_walk (0 count, 1 emit, 2 record once).
template <int MODE>
__global__ void __launch_bounds__(64) k_sample_walk_sc(DevImage im, DevResult r, const uint32_t* sid_per_region, WalkScratch ws) {
  constexpr bool EMIT = MODE == 1;
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  const uint32_t sid = sid_per_region[q];
  const uint64_t x = r.regions[2 * q], y = r.regions[2 * q + 1];
  uint8_t fl = 0;
  uint64_t nvar = 0, ncar = 0, ncar_kept = 0;
  uint64_t ref_pos = 0, sample_pos = 0;
  uint32_t closest_v = 0;
  if (!rewind_to_sample_pos(im, x, sid, closest_v, ref_pos, sample_pos)) fl = kRegionEndless;
  else {
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
        if (EMIT) {
          const uint64_t a = a0 + nvar;
          row_store(r.rows, a, (uint32_t)pos, ro, rl, ao, al, c, false, cb + ncar);
          r.r_class[a] = im.v_src[cur]; r.r_gt0[a] = im.v_car_begin[cur];
        }
        if (MODE == 2) {
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
  uint64_t* overflow;               // single-walk mode: set when a region outgrew its piece capacity
  uint32_t relative, pad_;          // single-walk mode: pieces sit at seg_begin[q] .. + q_nseg[q], seg_begin = capacities' scan
};

struct SeqSink {
  uint64_t nseg, nbytes;
};

// PASS 0 counts, PASS 1 writes the pieces at their scanned places (second walk), PASS 2 is the single walk: pieces go
// to the region's slice of a capacity-sized list with byte offsets relative to the region's first byte.
template <int PASS>
__device__ __forceinline__ void seq_append(const DevSeqResult& r, SeqSink& s, uint64_t seg0, uint64_t byte0, uint32_t off,
                                           uint64_t len, uint64_t cap) {
  if (len == 0) return;
  if (PASS == 1 || (PASS == 2 && s.nseg < cap)) {
    r.seg_src[seg0 + s.nseg] = off; r.seg_len[seg0 + s.nseg] = (uint32_t)len; r.seg_dst[seg0 + s.nseg] = byte0 + s.nbytes;
  } else if (PASS == 2) *r.overflow = 1;
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

template <int MODE, int PASS>
__global__ void __launch_bounds__(64) k_sample_seq(DevImage im, DevSeqResult r) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  if (PASS == 1 && r.q_flags[q]) return;
  const uint32_t sid = r.sids[q];
  const uint64_t x = r.regions[2 * q], y = r.regions[2 * q + 1];
  const uint64_t seg0 = PASS ? r.seg_begin[q] : 0, byte0 = PASS == 1 ? r.byte_begin[q] : 0;
  const uint64_t cap = PASS == 2 ? r.seg_begin[q + 1] - seg0 : 0;
  SeqSink s{0, 0};
  uint8_t fl = 0;
  uint64_t ref_pos = 0, sample_pos = 0;
  uint32_t cur = 0;
  bool ok = true;
  if (MODE == 2) cur = prev_vertex_with_sample(im, x, sid, ref_pos, sample_pos);
  else ok = rewind_to_sample_pos(im, x, sid, cur, ref_pos, sample_pos);
  if (!ok) fl = kRegionEndless;
  else {
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
  if (PASS != 1) {
    r.q_flags[q] = fl;
    r.q_nseg[q] = fl ? 0 : s.nseg;
    r.q_nbytes[q] = fl ? 0 : s.nbytes;
  }
}

// Piece capacity of a region for the single walk: twice the ref-path slots plus branch sites of the (for sample
// coordinates: generously widened) reference range, plus slack.  Too small a guess only costs the fallback.
__global__ void __launch_bounds__(256) k_seq_caps(DevImage im, DevSeqResult r, uint32_t sample_coordinates) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= r.Q) return;
  const uint64_t x = r.regions[2 * q], y = r.regions[2 * q + 1];
  const uint64_t margin = sample_coordinates ? (y > x ? y - x : 0) + 256 : 0;
  const uint64_t lo = x > margin + 1 ? x - margin : 1, hi = (y > x ? y : x) + margin;
  const uint32_t s0 = slot_of_find(im, lo), s1 = slot_of_find(im, hi);
  const uint64_t slots = s1 >= s0 ? (uint64_t)(s1 - s0) + 1 : 1;
  const uint64_t sites = s1 >= s0 ? (uint64_t)(im.rp_cand_prefix[s1 + 1] - im.rp_cand_prefix[s0]) : 0;
  r.q_nseg[q] = 2 * (slots + sites) + 8;
}

// Decode the pieces into characters: one wave per region, 64 piece descriptors at a time.
__global__ void __launch_bounds__(256) k_copy_segments(DevImage im, DevSeqResult r) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= r.Q) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t s0 = r.seg_begin[q], s1 = r.relative ? s0 + r.q_nseg[q] : r.seg_begin[q + 1];
  const uint64_t dst0 = r.relative ? r.byte_begin[q] : 0;
  for (uint64_t base = s0; base < s1; base += 64) {
    const uint64_t mine = base + lane;
    uint32_t src = 0, len = 0;
    uint64_t dst = 0;
    if (mine < s1) { src = r.seg_src[mine]; len = r.seg_len[mine]; dst = dst0 + r.seg_dst[mine]; }
    const uint32_t cnt = (uint32_t)((s1 - base) < 64 ? (s1 - base) : 64);
    for (uint32_t k = 0; k < cnt; ++k) {
      const uint32_t ksrc = __builtin_amdgcn_readlane(src, k), klen = __builtin_amdgcn_readlane(len, k);
      const uint64_t kdst = wave_bcast64(dst, k);
      for (uint32_t i = lane; i < klen; i += 64) {
        const uint32_t c = im.seq_codes[ksrc + i] & 7;
        r.chars[kdst + i] = (uint8_t)(0x0505054E47544341ULL >> (8 * c));  // "ACTGN" then char 5 (map_int, util.cc:32-41)
      }
    }
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
