// host_graph.hpp -- the logical content of a VariantStore index directory.
//
// One HostGraph holds exactly what the reference persists for a chromosome
// (reference include/variant_graph.h:501-557 serialize, include/index.h:174-179,
// include/graph.h:179-208), decoded into flat arrays:
//
//   vertex_list_<k>.proto   -> off/len/class_id/ref_index + carrier pool
//   seq_buffer.sdsl         -> seq (one DNA_MAP code per base, util.h:44)
//   sample_vector.sdsl      -> class_bits (one word-aligned row per class >= 1)
//   sampleid_map.lst        -> chr, ref_length, num_samples, sample_names
//   adj_list.cqf            -> topo_inplace / topo_val  (key -> value bit, count)
//   aux_vertex_list*.sdsl   -> aux_lists (ON-DISK order; the query-time order is
//                              what a fresh hash set yields after inserting them,
//                              graph.h:162-171 -- see out_neighbors())
//   index.sdsl              -> idx_pos (sorted 1-based start indexes with a set bit)
//   ref_node_id.sdsl        -> node_list
//
// It is produced either by the constructor (builder.hpp) or by the on-disk
// loader (formats/), and consumed by the device-image builder.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <stdexcept>
#include "ref_order_set.hpp"

namespace vsamd {

// s_info flag bits (variantgraphvertex.proto:14-18)
enum : uint8_t { GT_PHASE = 1, GT_1 = 2, GT_2 = 4 };

struct HostGraph {
  std::string chr;
  uint64_t ref_length = 0;
  uint32_t num_samples = 0;      // includes "ref" (variant_graph.h:632)
  bool use_bit_vector = false;   // class bit-vector mode vs explicit sample ids
  uint64_t num_classes = 0;      // classes 1..num_classes are stored
  uint64_t num_edges = 0;        // construct-time counter only (graph.h:135)
  uint64_t num_keys = 0;         // distinct CQF keys == get_num_vertices()

  // ---- vertices, id == array index (creation order) ----
  std::vector<uint32_t> off, len;
  std::vector<uint32_t> class_id;   // sampleclass_id(0); 0 = "ref only"
  std::vector<uint32_t> ref_index;  // index of the "ref" s_info entry, 0 = not a ref vertex
  std::vector<uint64_t> car_begin;  // [V+1] into the carrier pool (non-ref s_info entries, s_info order)
  std::vector<uint8_t> car_flags;   // GT_PHASE | GT_1 | GT_2 per carrier
  std::vector<uint32_t> car_index;  // sample-coordinate index per carrier (types 2/3/5); may be empty
  std::vector<uint32_t> car_sid;    // explicit mode only: sample id per carrier

  // ---- sequence pool ----
  std::vector<uint8_t> seq;         // DNA_MAP codes A0 C1 T2 G3 N4 (5 = unknown char)

  // ---- sample classes (bit-vector mode) ----
  uint32_t words_per_class() const { return (num_samples + 63) / 64; }
  std::vector<uint64_t> class_bits;  // row c-1 = class c, bit j = sample id j

  // ---- samples ----
  std::vector<std::string> sample_names;  // id -> name; [0] == "ref"
  std::vector<std::string> sample_file_order;  // names in sampleid_map.lst order (optional)

  // ---- topology ----
  std::vector<uint8_t> topo_inplace;  // CQF value bit
  std::vector<uint32_t> topo_val;     // CQF count: neighbour id (inplace) or 1-based aux list index; 0 = no key
  std::vector<std::vector<uint32_t>> aux_lists;  // on-disk element order

  // ---- position index ----
  std::vector<uint32_t> idx_pos;     // ascending 1-based indexes i with bit i-1 set
  std::vector<uint32_t> node_list;   // first ref vertex at each of those indexes

  uint64_t num_vertices() const { return off.size(); }
  uint32_t num_carriers(uint32_t v) const { return (uint32_t)(car_begin[v + 1] - car_begin[v]); }

  // Neighbours of v in the order the reference iterates them after loading the
  // index from disk (graph.h:149-172 then :265-280).
  void out_neighbors(uint32_t v, std::vector<uint32_t>& out) const {
    out.clear();
    if (v >= topo_val.size() || topo_val[v] == 0) return;
    if (topo_inplace[v]) {
      out.push_back(topo_val[v]);
      return;
    }
    RefOrderSet s;
    for (uint32_t n : aux_lists[topo_val[v] - 1]) s.insert(n);
    out.assign(s.begin(), s.end());
  }

  bool class_has(uint32_t cls, uint32_t sid) const {
    if (cls == 0) return sid == 0;
    const uint64_t* row = &class_bits[(uint64_t)(cls - 1) * words_per_class()];
    return (row[sid >> 6] >> (sid & 63)) & 1;
  }

  // ------------------------------------------------------------------------
  // Plain dump: the flat little-endian file the CPU oracle (oracle/) reads.
  // Layout: magic "VSPLAIN1", then sections of {u64 count, payload}.
  // ------------------------------------------------------------------------
  void write_plain(const std::string& path) const {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open " + path);
    auto w64 = [&](uint64_t x) { fwrite(&x, 8, 1, f); };
    auto wstr = [&](const std::string& s) { w64(s.size()); fwrite(s.data(), 1, s.size(), f); };
    auto wv32 = [&](const std::vector<uint32_t>& v) { w64(v.size()); if (!v.empty()) fwrite(v.data(), 4, v.size(), f); };
    auto wv64 = [&](const std::vector<uint64_t>& v) { w64(v.size()); if (!v.empty()) fwrite(v.data(), 8, v.size(), f); };
    auto wv8 = [&](const std::vector<uint8_t>& v) { w64(v.size()); if (!v.empty()) fwrite(v.data(), 1, v.size(), f); };
    fwrite("VSPLAIN1", 1, 8, f);
    wstr(chr);
    w64(ref_length); w64(num_samples); w64(use_bit_vector ? 1 : 0); w64(num_classes);
    wv32(off); wv32(len); wv32(class_id); wv32(ref_index);
    wv64(car_begin); wv8(car_flags); wv32(car_index); wv32(car_sid);
    wv8(seq);
    wv64(class_bits);
    w64(sample_names.size());
    for (auto& s : sample_names) wstr(s);
    wv8(topo_inplace); wv32(topo_val);
    w64(aux_lists.size());
    for (auto& l : aux_lists) wv32(l);
    wv32(idx_pos); wv32(node_list);
    fclose(f);
  }
};

inline char map_int(uint8_t code) {  // reference src/util.cc:32-41
  switch (code) {
    case 0: return 'A';
    case 1: return 'C';
    case 2: return 'T';
    case 3: return 'G';
    case 4: return 'N';
    default: return (char)5;
  }
}
inline uint8_t map_base(char c) {  // reference src/util.cc:44-53
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'T': return 2;
    case 'G': return 3;
    case 'N': return 4;
    default: return 5;
  }
}

}  // namespace vsamd
