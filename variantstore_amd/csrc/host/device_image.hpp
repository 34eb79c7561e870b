// device_image.hpp -- HostGraph -> structure-of-arrays image that is copied to HBM.
//
// The reference answers a region query by pointer chasing:
// Index::find (rank on an RRR vector, include/index.h:119-133) -> walk the ref
// path with VariantGraphPathIterator (include/variant_graph.h:1999-2049) -> per
// node a radius-1 BFS over CQF neighbour sets (include/graph.h:265-280,394-431)
// -> per branch get_neighbor_vertex / get_sample_from_vertex_if_exists
// (variant_graph.h:1296-1451).  Everything those calls look up is static once an
// index is loaded, so it is flattened here, once, into arrays whose order is the
// order the reference visits things in:
//
//   rank structure   plain bit-vector of ref-node starts + per-512-bit counts
//                    (replaces rrr_vector<127> rank/select)
//   ref path         every node on the "ref" path in path order, zero-length
//                    dummy nodes included (a superset of Index::node_list)
//   CSR adjacency    out-neighbours of every vertex in the reference's hash-set
//                    iteration order (RefOrderSet), replaces CQF + aux lists
//   vertex table     offset / length / ref index / class / #carriers / nri
//   class rows       one word-aligned bit row per sample class (class 0 = {ref})
//   genotype pool    4 bits per carrier (phase, gt_1, gt_2), s_info order
//
// Host-only code, no GPU calls.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <stdexcept>
#include "host_graph.hpp"

namespace vsamd {

constexpr uint32_t VS_NONE = 0xFFFFFFFFu;
constexpr uint32_t kListGroup = 8;        // must equal kCarAlign in kernels.hip.h: carriers per 16-byte arena group
constexpr uint32_t kListMaxDefault = 640; // classes with at most this many carriers also get a decoded 16-bit id list

struct HostImage {
  // scalars
  uint64_t ref_length = 0;
  uint32_t num_samples = 0, wpc = 0;
  uint32_t use_bit_vector = 0;
  uint64_t V = 0, E = 0, P = 0, R = 0, C = 0;

  // rank structure over start indexes: bit (idx-1) set <=> a ref node starts at 1-based idx
  std::vector<uint64_t> bits;        // ceil(ref_length/64)+1 words
  std::vector<uint32_t> blk_rank;    // ones before each 512-bit block, [nblk+1]
  std::vector<uint32_t> idx_pos;     // [R] ascending start indexes (select)
  std::vector<uint32_t> rank_to_slot;  // [R+1] first ref-path slot with the r-th start; [R] = P

  // ref path, slot order
  std::vector<uint32_t> rp_vid;      // [P+1]; [P] = 0 (the path iterator wraps to vertex 0)
  std::vector<uint32_t> rp_cand_prefix;  // [P+1] branches (out-neighbours != ref successor) before slot

  // CSR
  std::vector<uint32_t> row_ptr;     // [V+1]
  std::vector<uint32_t> col;         // [E]

  // vertex table
  std::vector<uint32_t> v_off, v_len, v_ridx, v_class, v_ncar, v_nri;
  std::vector<uint64_t> v_car_begin;  // carrier-pool index of the first non-ref s_info entry

  // classes / genotypes / explicit ids / sequence
  std::vector<uint64_t> class_rows;  // (C+1) rows of wpc words; row 0 = {bit 0}
  // sparse classes (<= kSparseClassMax carriers) also decoded once into explicit id lists, so the
  // expansion of a rare variant reads a few ids instead of scanning a whole bit row
  std::vector<uint32_t> cls_list_begin;  // [C+2]
  // Every class of at most list_max carriers is also decoded into an ascending id list (ref excluded); every list
  // starts on a multiple of 8 entries (one arena group) and is zero-padded to the next one.  k_fill_carriers expands
  // those variants lane-per-GROUP straight from the list, only denser classes go through their bit row.  Entries are
  // 16-bit for cohorts of at most 4032 samples (16-bit carrier words: one 16-byte load per group), else 32-bit.
  std::vector<uint32_t> cls_list_ids;    // cohorts above 4032 samples
  std::vector<uint16_t> cls_list16;      // cohorts of at most 4032 samples
  uint32_t list_max = kListMaxDefault;
  // What k_fill_carriers needs to find a vertex's carrier ids, resolved once here so that the kernel has no dependent
  // look-up in front of its first id load: for a listed vertex (<= list_max carriers, 16-bit lists in use) the GROUP
  // index of its class's list in cls_list16 (entry offset / 8), otherwise its class id (row index).
  std::vector<uint32_t> v_src;
  // Walk records (query types 4, 5, 2, 3 step along a sample's path vertex by vertex): everything one step reads about
  // the current vertex in ONE 32-byte record {row_begin, degree, ref index, offset, length, class, #carriers, 0}, and
  // everything it reads about a neighbour in ONE 32-byte edge record in CSR order {neighbour, its ref index, its class,
  // its row_begin | its degree, offset, length, #carriers}: the first half answers the tests of a step, the second
  // half IS the neighbour's vertex record, so stepping onto it needs no further look-up -- two or three memory
  // accesses per step instead of a dozen scattered 4-byte reads.
  std::vector<uint32_t> w_vertex;   // 8 words per vertex; word 7 = ref-path slot + 1 (0 for vertices off the path)
  // every ref-path slot's node starts at the index of its rank (slots of rank r: [rank_to_slot[r], rank_to_slot[r+1])):
  // what the type-4 event bitmaps rely on to find where a walk stops; false for an index that breaks it (never seen)
  bool slots_follow_ranks = false;
  std::vector<uint32_t> w_edge;     // 8 words per CSR entry
  // The walk blob of query type 4: 8-word records laid out in the order a walk along the reference meets them.  Per
  // ref-path slot, in slot order: a header {first edge record, degree, ref index, 0, length, class, #carriers, vertex id},
  // the edge records of the slot's node {neighbour, its ref index, its class, its first edge record, its degree, its
  // ref-path slot + 1, its length, its #carriers}, then -- breadth first -- the edge records of the off-path vertices
  // reachable from it that have not been placed yet (alt alleles hang off the node in front of them): the node, its
  // branches and where they rejoin sit in one or two cache lines.
  std::vector<uint32_t> wblob;
  std::vector<uint32_t> blob_of_slot;   // [P + 1] header record of each slot
  std::vector<uint32_t> blob_row;       // [V] first edge record of each vertex
  // Sequence queries (types 2 / 3) merge a run of ref-path slots into ONE piece; bit k set = slot k must be stepped
  // through literally whatever the sample: its first ref neighbour (edge order) or its smallest-index ref neighbour is
  // not its path successor, the successor's sequence does not follow its own in the pool or in ref coordinates, or it
  // ends the path.
  std::vector<uint64_t> seq_breaks;   // ceil(P / 64) + 1 words
  // How far, in ref-path slots, the last ref neighbour of an IRREGULAR node (one whose last ref neighbour is not its path
  // successor: a deletion edge listed after the successor) lies ahead of it, at most.  Such a node sets the walk's ref_pos
  // to that neighbour's index for one step: observable to query type 4 only through its stop test ref_pos >= y, i.e. only
  // for irregular nodes within this many slots of the stop slot.
  uint32_t irr_reach = 1;
  // The backward search of get_prev_vertex_with_sample visits rank, rank - deg(previous(rank)), ...: a STATIC chain, so
  // the ranks form a forest (parent = the next rank of the chain) and "does the chain from r0 visit rank p" is "p is an
  // ancestor of r0": DFS interval labels {tin, subtree size} per chain rank, 2 words at index rank - 1 like rk_back.
  std::vector<uint32_t> rk_anc;
  std::vector<uint32_t> slot_rank;   // [P] rank of every ref-path slot (slots of rank r: [rank_to_slot[r], rank_to_slot[r + 1]))
  std::vector<uint32_t> rk_back;    // 2 words per rank: {first ref-path slot of the rank, out-degree of that slot's node}
  // The carrier pool of a CLASS-ROW cohort's image is padded: every vertex's run starts on a multiple of 8 records (v_car_begin), so
  // that a group of 8 carriers of the result -- one 16-byte arena group -- is also one group of the pool (round 5).
  std::vector<uint8_t> gt_nibbles;   // cohorts above 4032 samples and explicit-id cohorts: 2 carriers per byte, low nibble first
  // class-row cohorts of at most 4032 samples (16-bit carrier words): ONE 32-bit word per group of 8 carriers, genotype k of the group
  // (3 bits: phase, gt_1, gt_2) at bit 3 (k / 2) + 16 (k & 1) -- the word shifted left by 13 - 3 j and masked with 0xE000E000
  // is the genotype part of the group's j-th pair of carrier words (id | gt << 13 in each half): one shift and one
  // v_and_or_b32 per pair where a nibble stream cost two of each and an unaligned 8-byte load
  std::vector<uint32_t> gt_groups;
  std::vector<uint32_t> car_sid;     // explicit mode only
  std::vector<uint32_t> car_index;   // sample-coordinate index per carrier record (query types 2/3/5); may be empty
  std::vector<uint8_t> seq_codes;    // DNA_MAP codes, one per base
};

// first non-ref carrier id of vertex b (sample_ids[0] in query.h:356-357), VS_NONE if none
inline uint32_t first_carrier(const HostGraph& g, uint32_t b) {
  if (g.car_begin[b + 1] == g.car_begin[b]) return VS_NONE;
  if (!g.use_bit_vector) return g.car_sid[g.car_begin[b]];
  uint32_t cls = g.class_id[b];
  if (cls == 0) return VS_NONE;
  const uint64_t* row = &g.class_bits[(uint64_t)(cls - 1) * g.words_per_class()];
  for (uint32_t w = 0; w < g.words_per_class(); ++w) {
    uint64_t x = row[w];
    if (w == 0) x &= ~1ULL;
    if (x) return w * 64 + __builtin_ctzll(x);
  }
  return VS_NONE;
}

inline bool vertex_has_sample(const HostGraph& g, uint32_t v, uint32_t sid) {
  if (sid == 0) return g.ref_index[v] != 0;
  if (g.use_bit_vector) return g.class_id[v] != 0 && g.class_has(g.class_id[v], sid);
  for (uint64_t c = g.car_begin[v]; c < g.car_begin[v + 1]; ++c)
    if (g.car_sid[c] == sid) return true;
  return false;
}

inline void build_host_image(const HostGraph& g, HostImage& im) {
  const uint64_t V = g.num_vertices();
  im.ref_length = g.ref_length;
  im.num_samples = g.num_samples;
  im.wpc = g.words_per_class();
  im.use_bit_vector = g.use_bit_vector;
  im.V = V;
  im.C = g.num_classes;

  // ---- CSR in query-time neighbour order (graph.h:149-172, 265-280) ----
  im.row_ptr.assign(V + 1, 0);
  std::vector<std::vector<uint32_t>> aux_order(g.aux_lists.size());
  for (size_t i = 0; i < g.aux_lists.size(); ++i) {
    RefOrderSet s;
    for (uint32_t n : g.aux_lists[i]) s.insert(n);
    aux_order[i] = s.order();
  }
  for (uint64_t v = 0; v < V; ++v) {
    uint32_t deg = 0;
    if (g.topo_val[v] != 0) deg = g.topo_inplace[v] ? 1 : (uint32_t)aux_order[g.topo_val[v] - 1].size();
    im.row_ptr[v + 1] = im.row_ptr[v] + deg;
  }
  im.E = im.row_ptr[V];
  im.col.resize(im.E);
  for (uint64_t v = 0; v < V; ++v) {
    if (g.topo_val[v] == 0) continue;
    uint32_t* dst = &im.col[im.row_ptr[v]];
    if (g.topo_inplace[v]) dst[0] = g.topo_val[v];
    else {
      const auto& l = aux_order[g.topo_val[v] - 1];
      for (size_t i = 0; i < l.size(); ++i) dst[i] = l[i];
    }
  }
  for (uint64_t e = 0; e < im.E; ++e)
    if (im.col[e] >= V) throw std::runtime_error("adjacency refers to a vertex that does not exist");

  // ---- vertex table ----
  im.v_off = g.off; im.v_len = g.len; im.v_ridx = g.ref_index; im.v_class = g.class_id;
  im.v_ncar.resize(V); im.v_car_begin.resize(V);
  // the image's pool: for class-row cohorts a vertex's records start on a multiple of 8 (explicit-id cohorts -- a handful of carriers per
  // vertex, ids in the pool itself -- keep the dense pool: padding it made the 10,000-sample cohort's pool 44 % larger and its
  // expansion 6 % slower)
  const bool pad_pool = g.use_bit_vector != 0;
  uint64_t pool_padded = 0;
  for (uint64_t v = 0; v < V; ++v) {
    im.v_ncar[v] = g.num_carriers((uint32_t)v);
    im.v_car_begin[v] = pool_padded;
    pool_padded += pad_pool ? (((uint64_t)im.v_ncar[v] + 7) & ~7ULL) : im.v_ncar[v];
  }

  // The device expands a class row into exactly v_ncar carriers without bounds checks:
  // a vertex whose s_info count disagrees with its class row is refused here
  // (the reference only logs this condition, variant_graph.h:1303-1308).
  if (g.use_bit_vector) {
    std::vector<uint32_t> class_pop(g.num_classes + 1, 0);
    for (uint64_t c = 1; c <= g.num_classes; ++c) {
      const uint64_t* row = &g.class_bits[(c - 1) * im.wpc];
      uint32_t pc = 0;
      for (uint32_t w = 0; w < im.wpc; ++w) pc += __builtin_popcountll(w == 0 ? (row[w] & ~1ULL) : row[w]);
      class_pop[c] = pc;
    }
    for (uint64_t v = 0; v < V; ++v) {
      if (g.class_id[v] > g.num_classes) throw std::runtime_error("vertex refers to a sample class that does not exist");
      if (class_pop[g.class_id[v]] != im.v_ncar[v]) throw std::runtime_error("s_info count of a vertex differs from its sample class");
    }
  } else {
    for (uint32_t sid : g.car_sid)
      if (sid >= g.num_samples) throw std::runtime_error("explicit sample id out of range");
  }

  // ---- ref path: vertex 0, then get_neighbor_vertex(., ref) (variant_graph.h:1402-1451, 2025-2032) ----
  auto ref_successor = [&](uint32_t v) -> uint32_t {
    uint32_t best = 0, min_idx = UINT32_MAX;
    for (uint32_t e = im.row_ptr[v]; e < im.row_ptr[v + 1]; ++e) {
      uint32_t n = im.col[e];
      if (g.ref_index[n] != 0 && min_idx > g.ref_index[n]) { best = n; min_idx = g.ref_index[n]; }
    }
    return best;  // 0 == none (no edge ever points at vertex 0, graph.h:214-215)
  };
  im.rp_vid.clear();
  {
    std::vector<uint8_t> on_path(V, 0);
    uint32_t cur = 0;
    while (true) {
      if (on_path[cur]) throw std::runtime_error("ref path revisits a vertex");
      on_path[cur] = 1;
      im.rp_vid.push_back(cur);
      uint32_t nxt = ref_successor(cur);
      if (nxt == 0) break;
      cur = nxt;
    }
  }
  im.P = im.rp_vid.size();
  im.rp_vid.push_back(0);  // sentinel: *next_it of the last node is vertex 0
  // the range form relies on contiguous coverage: ridx[i+1] == ridx[i] + len[i]
  for (uint64_t i = 0; i + 1 < im.P; ++i) {
    uint32_t a = im.rp_vid[i], b = im.rp_vid[i + 1];
    if (g.ref_index[a] == 0 || (uint64_t)g.ref_index[a] + g.len[a] != g.ref_index[b])
      throw std::runtime_error("ref path is not a contiguous tiling of the reference");
  }
  im.rp_cand_prefix.assign(im.P + 1, 0);
  for (uint64_t i = 0; i < im.P; ++i) {
    uint32_t v = im.rp_vid[i], succ = im.rp_vid[i + 1], c = 0;
    for (uint32_t e = im.row_ptr[v]; e < im.row_ptr[v + 1]; ++e) c += im.col[e] != succ;
    im.rp_cand_prefix[i + 1] = im.rp_cand_prefix[i] + c;
  }

  // ---- rank structure + rank -> first slot ----
  im.R = g.idx_pos.size();
  im.idx_pos = g.idx_pos;
  const uint64_t nwords = (g.ref_length + 63) / 64 + 1;
  im.bits.assign((nwords + 7) / 8 * 8, 0);
  for (uint32_t p : g.idx_pos) {
    if (p < 1 || p > g.ref_length + 1) throw std::runtime_error("index position out of range");
    im.bits[(p - 1) >> 6] |= 1ULL << ((p - 1) & 63);
  }
  const uint64_t nblk = im.bits.size() / 8;
  im.blk_rank.assign(nblk + 1, 0);
  for (uint64_t b = 0; b < nblk; ++b) {
    uint32_t c = 0;
    for (int w = 0; w < 8; ++w) c += __builtin_popcountll(im.bits[b * 8 + w]);
    im.blk_rank[b + 1] = im.blk_rank[b] + c;
  }
  {
    // slot of each node_list entry: the FIRST path node at that start index (index.h:86-90)
    std::vector<uint32_t> slot_of(V, VS_NONE);
    for (uint64_t i = 0; i < im.P; ++i) slot_of[im.rp_vid[i]] = (uint32_t)i;
    im.rank_to_slot.assign(im.R + 1, (uint32_t)im.P);
    for (uint64_t r = 0; r < im.R; ++r) {
      uint32_t v = g.node_list[r];
      if (v >= V || slot_of[v] == VS_NONE) throw std::runtime_error("node_list entry is not on the ref path");
      if (g.ref_index[v] != g.idx_pos[r]) throw std::runtime_error("node_list entry does not start at its index bit");
      im.rank_to_slot[r] = slot_of[v];
      if (slot_of[v] > 0 && g.ref_index[im.rp_vid[slot_of[v] - 1]] == g.idx_pos[r])
        throw std::runtime_error("node_list entry is not the first path node at its index");
    }
  }

  im.slots_follow_ranks = im.R > 0 && im.rank_to_slot[0] == 0;
  for (uint64_t r = 0; r < im.R && im.slots_follow_ranks; ++r) {
    if (im.rank_to_slot[r] >= im.rank_to_slot[r + 1]) { im.slots_follow_ranks = false; break; }
    for (uint32_t i = im.rank_to_slot[r]; i < im.rank_to_slot[r + 1]; ++i)
      if (g.ref_index[im.rp_vid[i]] != g.idx_pos[r]) { im.slots_follow_ranks = false; break; }
  }

  // ---- nri: ref index reached by one path step from a branch along its first carrier
  //      (query.h:353-365).  VS_NONE = "consecutive mutation": the reference then
  //      keeps the ref entry of the node the branch hangs off. ----
  im.v_nri.assign(V, 0);
  for (uint64_t b = 0; b < V; ++b) {
    if (g.ref_index[b] != 0) continue;  // only alt vertices are classified through nri
    uint32_t fc = first_carrier(g, (uint32_t)b);
    if (fc == VS_NONE) continue;
    uint32_t d = 0, min_idx = UINT32_MAX;
    bool hit = false;
    for (uint32_t e = im.row_ptr[b]; e < im.row_ptr[b + 1] && !hit; ++e) {
      uint32_t n = im.col[e];
      if (g.ref_index[n] != 0 && min_idx > g.ref_index[n]) { d = n; min_idx = g.ref_index[n]; }
      if (vertex_has_sample(g, n, fc)) { d = n; hit = true; }
    }
    // d == 0: iterator done, it now points at vertex 0 whose ref index is 1
    im.v_nri[b] = g.ref_index[d] != 0 ? g.ref_index[d] : VS_NONE;
  }

  // ---- class rows, genotype nibbles, explicit ids, sequence ----
  im.class_rows.assign((im.C + 1) * (uint64_t)im.wpc, 0);
  if (im.wpc) im.class_rows[0] = 1;  // class 0: only "ref"
  if (g.use_bit_vector && !g.class_bits.empty())
    std::copy(g.class_bits.begin(), g.class_bits.end(), im.class_rows.begin() + im.wpc);
  im.cls_list_begin.assign(im.C + 2, 0);
  im.cls_list_ids.clear();
  im.cls_list16.clear();
  const bool narrow = im.wpc <= 63;   // 16-bit carrier words (kernels.hip.h: the non-WIDE instantiation)
  if (g.use_bit_vector) {
    for (uint64_t c = 1; c <= im.C; ++c) {
      const uint64_t* row = &im.class_rows[c * im.wpc];
      uint32_t pc = 0;
      for (uint32_t w = 0; w < im.wpc; ++w) pc += __builtin_popcountll(w == 0 ? (row[w] & ~1ULL) : row[w]);
      if (narrow) {
        if (im.cls_list16.size() > 0xFFFFFFF0ull - im.num_samples) throw std::runtime_error("decoded class lists exceed 2^32 entries");
        im.cls_list_begin[c] = (uint32_t)im.cls_list16.size();
        if (pc <= im.list_max) {
          for (uint32_t w = 0; w < im.wpc; ++w) {
            uint64_t x = w == 0 ? (row[w] & ~1ULL) : row[w];
            while (x) { im.cls_list16.push_back((uint16_t)(w * 64 + __builtin_ctzll(x))); x &= x - 1; }
          }
          im.cls_list16.resize((im.cls_list16.size() + kListGroup - 1) / kListGroup * kListGroup, 0);
        }
      } else {   // cohorts above 4032 samples: the same lists with 32-bit entries (a group of 8 = two 16-byte loads)
        if (im.cls_list_ids.size() > 0xFFFFFFF0ull - im.num_samples) throw std::runtime_error("decoded class lists exceed 2^32 entries");
        im.cls_list_begin[c] = (uint32_t)im.cls_list_ids.size();
        if (pc <= im.list_max) {
          for (uint32_t w = 0; w < im.wpc; ++w) {
            uint64_t x = w == 0 ? (row[w] & ~1ULL) : row[w];
            while (x) { im.cls_list_ids.push_back(w * 64 + __builtin_ctzll(x)); x &= x - 1; }
          }
          im.cls_list_ids.resize((im.cls_list_ids.size() + kListGroup - 1) / kListGroup * kListGroup, 0);
        }
      }
    }
    im.cls_list_begin[im.C + 1] = (uint32_t)(narrow ? im.cls_list16.size() : im.cls_list_ids.size());
  }
  im.cls_list_ids.resize(im.cls_list_ids.size() + 8, 0);  // 16-byte reads may run past the last list
  im.cls_list16.resize(im.cls_list16.size() + 8, 0);
  im.v_src = im.v_class;
  if (g.use_bit_vector)
    for (uint64_t v = 0; v < V; ++v)
      if (im.v_ncar[v] <= im.list_max) im.v_src[v] = im.cls_list_begin[im.v_class[v]] / kListGroup;
  // genotypes, explicit ids and sample-coordinate indexes in the padded pool's order
  const bool groups = narrow && pad_pool;                      // genotype group words need group-aligned runs
  if (groups) im.gt_groups.assign(pool_padded / 8 + 16, 0);   // (16-byte reads of the staging run up to 3 words past a run)
  else im.gt_nibbles.assign(pool_padded / 2 + 32, 0);         // windowed 64-bit reads run up to 24 bytes past a list
  if (!g.car_sid.empty()) im.car_sid.assign(pool_padded, 0);
  if (!g.car_index.empty()) im.car_index.assign(pool_padded, 0);
  for (uint64_t v = 0; v < V; ++v) {
    const uint64_t src = g.car_begin[v], dst = im.v_car_begin[v];
    for (uint32_t j = 0; j < im.v_ncar[v]; ++j) {
      const uint64_t c = dst + j;
      const uint32_t gt = g.car_flags[src + j] & 7u;
      if (groups) im.gt_groups[c >> 3] |= gt << (3 * ((c & 7) >> 1) + 16 * (c & 1));
      else im.gt_nibbles[c >> 1] |= (uint8_t)(gt << ((c & 1) * 4));
      if (!g.car_sid.empty()) im.car_sid[c] = g.car_sid[src + j];
      if (!g.car_index.empty()) im.car_index[c] = g.car_index[src + j];
    }
  }
  im.car_sid.resize(im.car_sid.size() + 8, 0);   // a group of 8 ids is read whole: the last variant's may run past its records
  im.seq_codes = g.seq;

  im.w_vertex.assign(V * 8, 0);
  for (uint64_t v = 0; v < V; ++v) {
    uint32_t* w = &im.w_vertex[v * 8];
    w[0] = im.row_ptr[v]; w[1] = im.row_ptr[v + 1] - im.row_ptr[v]; w[2] = im.v_ridx[v]; w[3] = im.v_off[v];
    w[4] = im.v_len[v]; w[5] = im.v_class[v]; w[6] = im.v_ncar[v];
  }
  for (uint64_t i = 0; i < im.P; ++i) im.w_vertex[(uint64_t)im.rp_vid[i] * 8 + 7] = (uint32_t)i + 1;   // ref-path slot + 1 (0: off the path)
  im.w_edge.assign(im.E * 8, 0);
  for (uint64_t e = 0; e < im.E; ++e) {
    const uint32_t n = im.col[e];
    uint32_t* w = &im.w_edge[e * 8];
    w[0] = n; w[1] = im.v_ridx[n]; w[2] = im.v_class[n]; w[3] = im.row_ptr[n];
    w[4] = im.row_ptr[n + 1] - im.row_ptr[n]; w[5] = im.v_off[n]; w[6] = im.v_len[n]; w[7] = im.v_ncar[n];
  }
  {  // ---- walk blob ----
    im.blob_row.assign(V + 1, VS_NONE);
    im.blob_of_slot.assign(im.P + 1, 0);
    std::vector<uint32_t> placed;   // vertices in the order their edge rows are laid out
    placed.reserve(V);
    uint64_t at = 0;
    std::vector<uint32_t> queue;
    auto deg_of = [&](uint32_t v) { return im.row_ptr[v + 1] - im.row_ptr[v]; };
    for (uint64_t k = 0; k < im.P; ++k) {
      const uint32_t R = im.rp_vid[k];
      im.blob_of_slot[k] = (uint32_t)at++;           // the header
      im.blob_row[R] = (uint32_t)at; at += deg_of(R); placed.push_back(R);
      queue.assign(1, R);
      for (size_t qi = 0; qi < queue.size(); ++qi) {
        const uint32_t u = queue[qi];
        for (uint32_t e = im.row_ptr[u]; e < im.row_ptr[u + 1]; ++e) {
          const uint32_t n = im.col[e];
          if (im.w_vertex[(uint64_t)n * 8 + 7] != 0 || im.blob_row[n] != VS_NONE) continue;   // on the path, or placed already
          im.blob_row[n] = (uint32_t)at; at += deg_of(n); placed.push_back(n);
          queue.push_back(n);
        }
      }
    }
    im.blob_of_slot[im.P] = (uint32_t)at;
    for (uint64_t v = 0; v < V; ++v)
      if (im.blob_row[v] == VS_NONE) { im.blob_row[v] = (uint32_t)at; at += deg_of((uint32_t)v); placed.push_back((uint32_t)v); }
    if (at > 0xFFFFFFF0ull) throw std::runtime_error("walk blob exceeds 2^32 records");
    im.wblob.assign((at + 1) * 8, 0);
    for (uint32_t u : placed)
      for (uint32_t i = 0; i < deg_of(u); ++i) {
        const uint32_t n = im.col[im.row_ptr[u] + i];
        uint32_t* w = &im.wblob[((uint64_t)im.blob_row[u] + i) * 8];
        w[0] = n; w[1] = im.v_ridx[n]; w[2] = im.v_class[n]; w[3] = im.blob_row[n];
        w[4] = deg_of(n); w[5] = im.w_vertex[(uint64_t)n * 8 + 7]; w[6] = im.v_len[n]; w[7] = im.v_ncar[n];
      }
    for (uint64_t k = 0; k < im.P; ++k) {
      const uint32_t R = im.rp_vid[k];
      uint32_t* w = &im.wblob[(uint64_t)im.blob_of_slot[k] * 8];
      w[0] = im.blob_row[R]; w[1] = deg_of(R); w[2] = im.v_ridx[R]; w[3] = im.v_off[R]; w[4] = im.v_len[R]; w[5] = im.v_class[R];
      w[6] = im.v_ncar[R]; w[7] = R;
    }
  }
  im.seq_breaks.assign((im.P + 63) / 64 + 1, 0);
  im.irr_reach = 1;
  for (uint64_t k = 0; k < im.P; ++k) {
    const uint32_t R = im.rp_vid[k];
    bool brk = k + 1 >= im.P;
    if (!brk) {
      const uint32_t succ = im.rp_vid[k + 1];
      uint32_t first_ref = VS_NONE, min_ref = VS_NONE, min_idx = 0xFFFFFFFFu, last_ref = VS_NONE;
      for (uint32_t e = im.row_ptr[R]; e < im.row_ptr[R + 1]; ++e) {
        const uint32_t n = im.col[e], nr = im.v_ridx[n];
        if (!nr) continue;
        if (first_ref == VS_NONE) first_ref = n;
        if (nr < min_idx) { min_idx = nr; min_ref = n; }
        last_ref = n;
      }
      if (last_ref != VS_NONE && last_ref != succ) {   // irregular: how far ahead its last ref neighbour lies
        const uint32_t s1 = im.w_vertex[(uint64_t)last_ref * 8 + 7];
        // (a ref neighbour off the ref path or behind the node cannot be bounded: every irregular slot then counts everywhere)
        if (s1 == 0 || s1 - 1 <= k) im.irr_reach = 0xFFFFFFFFu;
        else if (im.irr_reach != 0xFFFFFFFFu && s1 - 1 - k > im.irr_reach) im.irr_reach = (uint32_t)(s1 - 1 - k);
      }
      brk = first_ref != succ || min_ref != succ || im.v_off[succ] != im.v_off[R] + im.v_len[R] || im.v_ridx[succ] != im.v_ridx[R] + im.v_len[R];
    }
    if (brk) im.seq_breaks[k >> 6] |= 1ull << (k & 63);
  }
  im.rk_back.assign((im.R + 1) * 2, 0);
  for (uint64_t r = 0; r < im.R; ++r) {
    const uint32_t slot = im.rank_to_slot[r];
    im.rk_back[2 * r] = slot;
    if (slot < im.P) { const uint32_t v = im.rp_vid[slot]; im.rk_back[2 * r + 1] = im.row_ptr[v + 1] - im.row_ptr[v]; }
  }
  im.slot_rank.assign(im.P, 0);
  for (uint64_t r = 0; r < im.R; ++r)
    for (uint64_t k = im.rank_to_slot[r]; k < im.rank_to_slot[r + 1] && k < im.P; ++k) im.slot_rank[k] = (uint32_t)r;
  {  // chain ranks 0 .. R; rank p >= 2 steps to p - max(deg, 1) with deg = out-degree of previous(p) = rk_back[p - 1]
    const uint64_t n = im.R + 1;
    auto parent = [&](uint64_t p) -> uint64_t {
      const uint32_t deg = im.rk_back[2 * (p - 1) + 1];
      const uint64_t step = deg ? deg : 1;
      return p > step ? p - step : 0;
    };
    std::vector<uint32_t> size(n, 1), off(n, 1), tin(n, 0);
    for (uint64_t p = n; p-- > 2;) size[parent(p)] += size[p];
    if (n > 1) tin[1] = size[0];
    for (uint64_t p = 2; p < n; ++p) { const uint64_t q = parent(p); tin[p] = tin[q] + off[q]; off[q] += size[p]; }
    im.rk_anc.assign(n * 2, 0);
    for (uint64_t p = 1; p < n; ++p) { im.rk_anc[2 * (p - 1)] = tin[p]; im.rk_anc[2 * (p - 1) + 1] = size[p]; }
  }
}

}  // namespace vsamd
