// builder.hpp -- variation-graph construction (`variantstore construct`).
//
// Restates, over flat arrays, the reference's write side so that the graphs the
// query path sees are the ones the reference would have produced from the same
// FASTA + VCF: VariantGraph ctor (reference include/variant_graph.h:323-364),
// add_mutation (:1509-1881), split_vertex (:1115-1167), add_vertex (:735-784),
// find_sample_vector_or_add (:803-832), update_vertex_sample_class (:834-873),
// get_neighbor_vertex (:1402-1451), Graph::add_edge / remove_edge
// (include/graph.h:210-263) and the Index build (include/index.h:53-106).
// Host-only code: it prepares an index, it is not the query hot path.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>
#include <stdexcept>
#include "host_graph.hpp"
#include "ref_order_set.hpp"

namespace vsamd {

struct SampleGT {  // sample_struct, variant_graph.h:78-83
  uint32_t sample_id;
  bool phase, gt1, gt2;
};

// MurmurHash64A (Austin Appleby, public domain) -- the reference keys sample
// classes by this hash of the class bit-vector (variant_graph.h:811-813).
inline uint64_t murmur_hash_64a(const void* key, int len, uint64_t seed) {
  const uint64_t m = 0xc6a4a7935bd1e995ULL;
  const int r = 47;
  uint64_t h = seed ^ (len * m);
  const uint64_t* data = (const uint64_t*)key;
  const uint64_t* end = data + (len / 8);
  while (data != end) {
    uint64_t k;
    memcpy(&k, data++, 8);
    k *= m; k ^= k >> r; k *= m;
    h ^= k; h *= m;
  }
  const unsigned char* data2 = (const unsigned char*)data;
  switch (len & 7) {
    case 7: h ^= (uint64_t)data2[6] << 48; [[fallthrough]];
    case 6: h ^= (uint64_t)data2[5] << 40; [[fallthrough]];
    case 5: h ^= (uint64_t)data2[4] << 32; [[fallthrough]];
    case 4: h ^= (uint64_t)data2[3] << 24; [[fallthrough]];
    case 3: h ^= (uint64_t)data2[2] << 16; [[fallthrough]];
    case 2: h ^= (uint64_t)data2[1] << 8; [[fallthrough]];
    case 1: h ^= (uint64_t)data2[0]; h *= m;
  }
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}

class GraphBuilder {
 public:
  // sample_names: VCF column order (ids 1..n); "ref" is id 0.
  GraphBuilder(const std::string& chr, const std::string& ref, const std::vector<std::string>& sample_names,
               bool use_bit_vector)
      : chr_(chr), use_bv_(use_bit_vector) {
    ref_length_ = ref.size();
    names_.push_back("ref");
    for (auto& s : sample_names) names_.push_back(s);
    num_samples_ = (uint32_t)names_.size();
    wpc_ = (num_samples_ + 63) / 64;
    // whole reference as vertex 0, index 1 (variant_graph.h:352-356)
    uint32_t v = add_vertex_seq(ref, 0);
    ref_index_[v] = 1;
    idx_vertex_id_[1] = v;
  }

  uint64_t num_vertices() const { return off_.size(); }
  uint64_t num_edges() const { return num_edges_; }
  uint64_t num_keys() const { return num_keys_; }
  uint64_t seq_length() const { return seq_.size(); }
  uint64_t num_classes() const { return sampleclass_map_.size(); }
  uint64_t ref_length() const { return ref_length_; }
  uint32_t num_samples() const { return num_samples_; }
  bool use_bit_vector() const { return use_bv_; }
  std::string get_sequence(uint64_t start, uint32_t length) const {
    std::string s;
    for (uint64_t i = start; i < start + length && i < seq_.size(); ++i) s += map_int(seq_[i]);
    return s;
  }

  // variant_graph.h:1509-1881.  `pos` is the 1-based VCF POS.
  void add_mutation(std::string ref, std::string alt, uint64_t pos, std::vector<SampleGT>& sample_list) {
    enum { INSERTION, DELETION, SUBSTITUTION } mutation;
    if (ref.size() == alt.size()) mutation = SUBSTITUTION;
    else if (ref.size() > alt.size()) mutation = DELETION;
    else mutation = INSERTION;

    if (mutation == INSERTION) {
      pos = pos + ref.size();
      alt = alt.substr(ref.size());
    } else if (mutation == DELETION) {
      pos = pos + alt.size();
      ref = ref.substr(alt.size());
    }

    // the ref vertex at or before @pos (:1534-1545).  The reference dereferences
    // end() when pos lies beyond the last start index; libstdc++ then reads the
    // node count as the key -- defined here as "not equal", i.e. step back.
    auto ref_idx_itr = idx_vertex_id_.lower_bound(pos);
    if (ref_idx_itr == idx_vertex_id_.end() || ref_idx_itr->first != pos) {
      if (ref_idx_itr == idx_vertex_id_.begin()) throw std::runtime_error("no ref vertex before pos");
      --ref_idx_itr;
    }
    const uint64_t ref_vertex_idx = ref_idx_itr->first;
    uint32_t ref_vertex_id = (uint32_t)ref_idx_itr->second;
    const uint64_t rv_len = len_[ref_vertex_id];  // copies taken before any split (:1547)
    const uint64_t rv_off = off_[ref_vertex_id];
    const uint64_t rsz = ref.size();

    uint32_t prev_ref_vertex_id = 0, next_ref_vertex_id = 0;

    // advance through the start-index map until the vertex covering pos+|ref| (:1600-1605 and twins)
    auto seek_end = [&](std::map<uint64_t, uint64_t>::iterator& it, uint32_t& vid, uint64_t& vlen) {
      it = idx_vertex_id_.lower_bound(ref_vertex_idx);
      do {
        ++it;
        if (it == idx_vertex_id_.end()) throw std::runtime_error("mutation runs past the reference end");
        vid = (uint32_t)it->second;
        vlen = len_[vid];
      } while (it->first + vlen < pos + rsz);
    };

    if (mutation == SUBSTITUTION) {
      if (ref_vertex_idx == pos && rv_len == rsz) {                      // :1553-1573
        split_vertex(ref_vertex_id, 1, &next_ref_vertex_id);
        prev_ref_vertex_id = ref_vertex_id;
        ref_vertex_id = next_ref_vertex_id;
        get_neighbor_vertex(ref_vertex_id, 0, &next_ref_vertex_id);
      } else if (ref_vertex_idx == pos && rv_len > rsz) {                 // :1574-1593
        split_vertex(ref_vertex_id, 1, &next_ref_vertex_id);
        prev_ref_vertex_id = ref_vertex_id;
        ref_vertex_id = next_ref_vertex_id;
        split_vertex(ref_vertex_id, rsz + 1, &next_ref_vertex_id);
      } else if (ref_vertex_idx == pos && rv_len < rsz) {                 // :1594-1632
        if (rv_len > 1) split_vertex(ref_vertex_id, 1, &next_ref_vertex_id);
        prev_ref_vertex_id = ref_vertex_id;
        ref_vertex_id = next_ref_vertex_id;
        std::map<uint64_t, uint64_t>::iterator t; uint32_t nv; uint64_t nl;
        seek_end(t, nv, nl);
        if (t->first + nl == pos + rsz) get_neighbor_vertex((uint32_t)t->second, 0, &next_ref_vertex_id);
        else next_ref_vertex_id = nv;
      } else if (ref_vertex_idx < pos && ref_vertex_idx + rv_len > pos + rsz) {  // :1633-1641
        uint64_t split_pos = pos - rv_off;
        split_vertex2(ref_vertex_id, split_pos, split_pos + rsz, &prev_ref_vertex_id, &next_ref_vertex_id);
        std::swap(ref_vertex_id, prev_ref_vertex_id);
      } else if (ref_vertex_idx < pos && ref_vertex_idx + rv_len < pos + rsz) {  // :1642-1669
        prev_ref_vertex_id = ref_vertex_id;
        if (rv_len > 1) split_vertex(prev_ref_vertex_id, pos - ref_vertex_idx + 1, &ref_vertex_id);
        std::map<uint64_t, uint64_t>::iterator t; uint32_t nv; uint64_t nl;
        seek_end(t, nv, nl);
        if (t->first + nl == pos + rsz) get_neighbor_vertex((uint32_t)t->second, 0, &next_ref_vertex_id);
        else next_ref_vertex_id = nv;
      } else if (ref_vertex_idx < pos && ref_vertex_idx + rv_len == pos + rsz) {  // :1670-1678
        prev_ref_vertex_id = ref_vertex_id;
        split_vertex(prev_ref_vertex_id, pos - ref_vertex_idx + 1, &ref_vertex_id);
        get_neighbor_vertex(ref_vertex_id, 0, &next_ref_vertex_id);
      }
      uint32_t sv = add_allele_vertex(alt, sample_list);                  // :1679-1707
      add_edge(prev_ref_vertex_id, sv);
      add_edge(sv, next_ref_vertex_id);
    } else if (mutation == INSERTION) {
      if (ref_vertex_idx == pos) {                                        // :1712-1724
        auto t = idx_vertex_id_.lower_bound(ref_vertex_idx);
        if (t == idx_vertex_id_.begin()) throw std::runtime_error("insertion before the first ref vertex");
        --t;
        prev_ref_vertex_id = (uint32_t)t->second;
        next_ref_vertex_id = ref_vertex_id;
      } else if (ref_vertex_idx < pos && ref_vertex_idx + rv_len > pos) {  // :1725-1730
        split_vertex(ref_vertex_id, pos - ref_vertex_idx + 1, &next_ref_vertex_id);
        prev_ref_vertex_id = ref_vertex_id;
      } else if (ref_vertex_idx + rv_len == pos) {                        // :1731-1735
        prev_ref_vertex_id = ref_vertex_id;
        get_neighbor_vertex(ref_vertex_id, 0, &next_ref_vertex_id);
      } else {                                                            // :1736-1738
        prev_ref_vertex_id = ref_vertex_id;
      }
      uint32_t sv = add_allele_vertex(alt, sample_list);                  // :1739-1766
      add_edge(prev_ref_vertex_id, sv);
      if (next_ref_vertex_id != 0) add_edge(sv, next_ref_vertex_id);
    } else {  // DELETION
      auto prev_by_index = [&]() {
        auto t = idx_vertex_id_.lower_bound(ref_vertex_idx);
        if (t == idx_vertex_id_.begin()) throw std::runtime_error("deletion at the first ref vertex");
        --t;
        prev_ref_vertex_id = (uint32_t)t->second;
      };
      if (ref_vertex_idx == pos && rv_len == rsz) {                       // :1771-1784
        prev_by_index();
        get_neighbor_vertex(ref_vertex_id, 0, &next_ref_vertex_id);
      } else if (ref_vertex_idx == pos && rv_len > rsz) {                 // :1785-1798
        prev_by_index();
        split_vertex(ref_vertex_id, rsz + 1, &next_ref_vertex_id);
      } else if (ref_vertex_idx == pos && rv_len < rsz) {                 // :1799-1823
        std::map<uint64_t, uint64_t>::iterator t; uint32_t nv; uint64_t nl;
        seek_end(t, nv, nl);
        if (t->first + nl == pos + rsz) get_neighbor_vertex((uint32_t)t->second, 0, &next_ref_vertex_id);
        else split_vertex((uint32_t)t->second, pos + rsz - t->first + 1, &next_ref_vertex_id);
        prev_by_index();
      } else if (ref_vertex_idx < pos && ref_vertex_idx + rv_len > pos + rsz) {  // :1824-1830
        uint64_t split_pos = pos - rv_off;
        split_vertex2(ref_vertex_id, split_pos, split_pos + rsz, &prev_ref_vertex_id, &next_ref_vertex_id);
        std::swap(ref_vertex_id, prev_ref_vertex_id);
      } else if (ref_vertex_idx < pos && ref_vertex_idx + rv_len < pos + rsz) {  // :1831-1855
        prev_ref_vertex_id = ref_vertex_id;
        if (rv_len > 1) split_vertex(prev_ref_vertex_id, pos - ref_vertex_idx + 1, &ref_vertex_id);
        std::map<uint64_t, uint64_t>::iterator t; uint32_t nv; uint64_t nl;
        seek_end(t, nv, nl);
        if (t->first + nl == pos + rsz) get_neighbor_vertex((uint32_t)t->second, 0, &next_ref_vertex_id);
        else next_ref_vertex_id = nv;
      } else if (ref_vertex_idx < pos && ref_vertex_idx + rv_len == pos + rsz) {  // :1856-1863
        prev_ref_vertex_id = ref_vertex_id;
        split_vertex(prev_ref_vertex_id, pos - ref_vertex_idx + 1, &ref_vertex_id);
        get_neighbor_vertex(ref_vertex_id, 0, &next_ref_vertex_id);
      }
      if (use_bv_) update_vertex_sample_class(next_ref_vertex_id, sample_list);   // :1866-1873
      else for (const auto& s : sample_list) append_carrier(next_ref_vertex_id, s);  // :1874-1877
      add_edge(prev_ref_vertex_id, next_ref_vertex_id);                   // :1879-1880
    }
  }

  // Flatten into the persisted form and build the position index.
  // with_sample_indexes: run fix_sample_indexes (per-carrier sample-coordinate indexes, only read by
  // query types 2/3/5; 4 bytes per carrier record)
  void finish(HostGraph& g, bool with_sample_indexes = true) {
    g.chr = chr_;
    g.ref_length = ref_length_;
    g.num_samples = num_samples_;
    g.use_bit_vector = use_bv_;
    g.num_classes = sampleclass_map_.size();
    g.num_edges = num_edges_;
    g.num_keys = num_keys_;
    const uint64_t V = off_.size();
    g.off = off_; g.len = len_; g.class_id = class_; g.ref_index = ref_index_;
    g.car_begin.assign(V + 1, 0);
    for (uint64_t v = 0; v < V; ++v) g.car_begin[v + 1] = g.car_begin[v] + car_cnt_[v];
    g.car_flags.resize(g.car_begin[V]);
    if (!use_bv_) g.car_sid.resize(g.car_begin[V]);
    for (uint64_t v = 0; v < V; ++v) {
      for (uint32_t i = 0; i < car_cnt_[v]; ++i) {
        g.car_flags[g.car_begin[v] + i] = pool_flags_[car_at_[v] + i];
        if (!use_bv_) g.car_sid[g.car_begin[v] + i] = pool_sid_[car_at_[v] + i];
      }
    }
    g.seq = seq_;
    g.class_bits = sample_vector_;
    g.sample_names = names_;
    g.topo_inplace = topo_inplace_; g.topo_val = topo_val_;
    g.topo_inplace.resize(V, 0); g.topo_val.resize(V, 0);
    g.aux_lists.clear();
    for (auto& s : aux_) g.aux_lists.emplace_back(s.order());  // serialize iterates the sets (graph.h:199-203)
    if (with_sample_indexes) fix_sample_indexes(g);
    build_index(g);
  }

 private:
  // ---- vertices ----
  uint32_t new_vertex(uint64_t offset, uint64_t length, uint32_t cls) {  // create_vertex :1068-1113
    off_.push_back((uint32_t)offset); len_.push_back((uint32_t)length); class_.push_back(cls);
    ref_index_.push_back(0); car_at_.push_back(pool_flags_.size()); car_cnt_.push_back(0);
    topo_val_.push_back(0); topo_inplace_.push_back(0);
    return (uint32_t)(off_.size() - 1);
  }
  uint32_t add_vertex_seq(const std::string& s, uint32_t cls) {  // add_vertex(seq,...) :758-784
    uint64_t start = seq_.size();
    for (char c : s) seq_.push_back(map_base(c));
    return new_vertex(start, s.size(), cls);
  }
  static uint8_t flags_of(const SampleGT& s) {
    return (s.phase ? GT_PHASE : 0) | (s.gt1 ? GT_1 : 0) | (s.gt2 ? GT_2 : 0);
  }
  // carriers of a freshly created vertex are appended contiguously; a vertex
  // whose list is rewritten or extended gets a fresh region of the pool.
  void append_carrier(uint32_t v, const SampleGT& s) {  // add_sample_to_vertex :1453-1463 (index 0)
    if (car_at_[v] + car_cnt_[v] != pool_flags_.size()) {
      uint64_t at = pool_flags_.size();
      for (uint32_t i = 0; i < car_cnt_[v]; ++i) {
        pool_flags_.push_back(pool_flags_[car_at_[v] + i]);
        if (!use_bv_) pool_sid_.push_back(pool_sid_[car_at_[v] + i]);
      }
      car_at_[v] = at;
    }
    pool_flags_.push_back(flags_of(s));
    if (!use_bv_) pool_sid_.push_back(s.sample_id);
    car_cnt_[v]++;
  }
  uint32_t add_allele_vertex(const std::string& alt, const std::vector<SampleGT>& sample_list) {
    uint32_t cls = use_bv_ ? find_sample_vector_or_add(sample_list) : 0;
    uint32_t v = add_vertex_seq(alt, cls);
    for (const auto& s : sample_list) append_carrier(v, s);
    return v;
  }

  // ---- sample classes ----
  uint32_t find_sample_vector_or_add(const std::vector<SampleGT>& list) {  // :803-832
    scratch_.assign(wpc_, 0);
    for (const auto& s : list) scratch_[s.sample_id >> 6] |= 1ULL << (s.sample_id & 63);
    uint64_t h = murmur_hash_64a(scratch_.data(), (int)(wpc_ * 8), 2038074743);
    auto it = sampleclass_map_.find(h);
    if (it != sampleclass_map_.end()) return it->second;
    uint32_t id = (uint32_t)sampleclass_map_.size() + 1;
    sampleclass_map_.emplace(h, id);
    sample_vector_.insert(sample_vector_.end(), scratch_.begin(), scratch_.end());
    return id;
  }
  // ids of v's s_info entries excluding ref, in s_info order
  void carrier_ids(uint32_t v, std::vector<uint32_t>& ids) const {
    ids.clear();
    if (!use_bv_) {
      for (uint32_t i = 0; i < car_cnt_[v]; ++i) ids.push_back(pool_sid_[car_at_[v] + i]);
      return;
    }
    if (class_[v] == 0) return;
    const uint64_t* row = &sample_vector_[(uint64_t)(class_[v] - 1) * wpc_];
    for (uint32_t w = 0; w < wpc_; ++w) {
      uint64_t x = row[w];
      while (x) {
        uint32_t id = w * 64 + __builtin_ctzll(x);
        if (id != 0) ids.push_back(id);
        x &= x - 1;
      }
    }
  }
  void update_vertex_sample_class(uint32_t v, const std::vector<SampleGT>& sample_list) {  // :834-873
    std::map<uint32_t, SampleGT> by_id;
    for (const auto& s : sample_list) by_id.insert({s.sample_id, s});
    // existing entries re-enter with cleared genotype bits (`{id, 0, 0}`) and never override
    std::vector<uint32_t> ids;
    carrier_ids(v, ids);
    for (uint32_t id : ids) by_id.insert({id, SampleGT{id, false, false, false}});
    const bool is_ref = ref_index_[v] != 0;
    if (is_ref) by_id.insert({0, SampleGT{0, false, false, false}});
    std::vector<SampleGT> list;
    for (auto& kv : by_id) list.push_back(kv.second);
    class_[v] = find_sample_vector_or_add(list);
    car_at_[v] = pool_flags_.size();
    car_cnt_[v] = 0;
    for (const auto& s : list)
      if (s.sample_id != 0) append_carrier(v, s);
  }

  // ---- topology (graph.h:210-280) ----
  void out_neighbors(uint32_t v, std::vector<uint32_t>& out) const {
    out.clear();
    if (topo_val_[v] == 0) return;
    if (topo_inplace_[v]) out.push_back(topo_val_[v]);
    else out.assign(aux_[topo_val_[v] - 1].begin(), aux_[topo_val_[v] - 1].end());
  }
  void add_edge(uint32_t s, uint32_t d) {
    uint32_t val = topo_val_[s];
    if (d == 0) return;
    if (val != 0 && topo_inplace_[s] && val == d) return;
    if (val == 0) {
      num_edges_++; num_keys_++;
      topo_val_[s] = d; topo_inplace_[s] = 1;
    } else if (topo_inplace_[s]) {
      RefOrderSet nb;
      nb.insert(val);
      nb.insert(d);
      aux_.push_back(nb);
      num_edges_++;
      topo_val_[s] = (uint32_t)aux_.size(); topo_inplace_[s] = 0;
    } else {
      if (aux_[val - 1].insert(d)) num_edges_++;
    }
  }
  void remove_edge(uint32_t s, uint32_t d) {
    uint32_t val = topo_val_[s];
    if (val == 0) return;
    if (topo_inplace_[s]) {  // delete_key(KeyObject(s,1,d)) drops the key whatever d is
      topo_val_[s] = 0; topo_inplace_[s] = 0; num_keys_--;
    } else {
      RefOrderSet& l = aux_[val - 1];
      if (l.contains(d)) l.erase_begin();  // erases begin(), not d (graph.h:252-257)
    }
  }
  // first out-neighbour carrying sample_id, else the ref out-neighbour with the
  // smallest index; *v keeps its incoming value when nothing qualifies (:1402-1451)
  bool get_neighbor_vertex(uint32_t id, uint32_t sample_id, uint32_t* v) const {
    uint32_t min_idx = UINT32_MAX;
    std::vector<uint32_t> nb, ids;
    out_neighbors(id, nb);
    for (uint32_t n : nb) {
      if (ref_index_[n] != 0 && min_idx > ref_index_[n]) { *v = n; min_idx = ref_index_[n]; }
      if (sample_id != 0) {
        carrier_ids(n, ids);
        for (uint32_t c : ids) if (c == sample_id) { *v = n; return true; }
      }
    }
    return *v != 0;
  }
  void split_vertex(uint32_t vertex_id, uint64_t pos, uint32_t* new_vertex_out) {  // :1115-1156
    const uint64_t cur_off = off_[vertex_id], cur_len = len_[vertex_id], cur_idx = ref_index_[vertex_id];
    if (pos > cur_len) throw std::runtime_error("split position is greater than vertex length");
    uint64_t offset = cur_off + pos - 1;
    uint64_t length = cur_len - pos + 1;
    uint32_t nv = new_vertex(offset, length, 0);
    ref_index_[nv] = (uint32_t)(cur_idx + pos - 1);
    *new_vertex_out = nv;
    if (cur_idx != ref_index_[nv]) idx_vertex_id_[ref_index_[nv]] = nv;
    len_[vertex_id] = (uint32_t)(cur_len - length);
    std::vector<uint32_t> nb;
    out_neighbors(vertex_id, nb);
    for (uint32_t n : nb) {
      add_edge(nv, n);
      remove_edge(vertex_id, n);
    }
    add_edge(vertex_id, nv);
  }
  void split_vertex2(uint32_t vertex_id, uint64_t pos1, uint64_t pos2, uint32_t* n1, uint32_t* n2) {  // :1158-1167
    split_vertex(vertex_id, pos1, n1);
    split_vertex(*n1, pos2 - pos1 + 1, n2);
  }

  // ---- fix_sample_indexes, variant_graph.h:1883-1997 (the "optimized solution") ----
  // One breadth-first pass over the whole graph in Graph::GraphIterator order (graph.h:394-459, with
  // its "shared neighbour first" reordering) carrying a per-sample offset between sample and ref
  // coordinates; uses the construct-time neighbour order, as the reference does.
  void fix_sample_indexes(HostGraph& g) const {
    const uint64_t V = off_.size();
    g.car_index.assign(g.car_flags.size(), 0);
    std::vector<int32_t> delta(num_samples_, 0);
    std::vector<uint8_t> visited(V, 0);
    std::vector<std::pair<uint32_t, uint64_t>> q;  // FIFO via head index
    size_t head = 0;
    std::vector<uint32_t> nb, nb2, ids, a, b;
    auto process = [&](uint32_t cur) {
      out_neighbors(cur, nb);
      if (ref_index_[cur] != 0) {
        const uint32_t ref_index = ref_index_[cur];
        for (uint32_t n : nb) {
          carrier_ids(n, ids);
          for (size_t i = 0; i < ids.size(); ++i) {
            uint32_t& idx = g.car_index[g.car_begin[n] + i];
            if (idx == 0) {
              const int32_t si = (int32_t)(ref_index + len_[cur] + (uint32_t)delta[ids[i]]);
              if (si < 0) throw std::runtime_error("Sample index is less than 0");
              idx = (uint32_t)si;
            } else {
              delta[ids[i]] = (int32_t)(idx - (ref_index + len_[cur]));
            }
          }
        }
      } else {
        if (nb.size() > 1) throw std::runtime_error("Sample vertex has more than 1 neighbor");
        for (uint32_t n : nb) {
          if (ref_index_[n] == 0) throw std::runtime_error("Ref vertex not found as a neighbor from sample vertex");
          carrier_ids(cur, ids);
          for (size_t i = 0; i < ids.size(); ++i)
            delta[ids[i]] = (int32_t)(g.car_index[g.car_begin[cur] + i] + len_[cur] - ref_index_[n]);
        }
      }
    };
    visited[0] = 1;
    out_neighbors(0, nb);
    for (uint32_t n : nb) q.emplace_back(n, 1);
    process(0);
    while (true) {
      uint32_t cur = 0;
      bool found = false;
      while (head < q.size()) {
        cur = q[head].first;
        if (!visited[cur]) { visited[cur] = 1; found = true; break; }
        ++head;
      }
      if (!found) break;
      const uint64_t hop = q[head].second;
      ++head;
      std::vector<uint32_t> ordered;
      out_neighbors(cur, nb2);
      a = nb2;
      std::sort(a.begin(), a.end());
      for (uint32_t v : nb2) {
        out_neighbors(v, b);
        std::sort(b.begin(), b.end());
        bool inter = false;
        for (size_t i = 0, j = 0; i < a.size() && j < b.size();) {
          if (a[i] == b[j]) { inter = true; break; }
          if (a[i] < b[j]) ++i; else ++j;
        }
        if (inter) ordered.insert(ordered.begin(), v);
        else ordered.push_back(v);
      }
      for (uint32_t v : ordered) q.emplace_back(v, hop + 1);
      process(cur);
      if (head > (1u << 20) && head * 2 > q.size()) {  // drop the consumed prefix
        q.erase(q.begin(), q.begin() + head);
        head = 0;
      }
    }
  }

  // ---- Index(const VariantGraph*) index.h:53-106 ----
  void build_index(HostGraph& g) const {
    std::vector<uint8_t> seen(ref_length_ + 2, 0);
    std::vector<std::pair<uint32_t, uint32_t>> ent;  // (index, first node)
    uint32_t cur = 0;
    while (true) {
      uint32_t idx = ref_index_[cur];
      if (idx == 0) throw std::runtime_error("Ref sample not found in the vertex");
      if (idx - 1 < seen.size() && !seen[idx - 1]) { seen[idx - 1] = 1; ent.emplace_back(idx, cur); }
      uint32_t nxt = 0;
      if (!get_neighbor_vertex(cur, 0, &nxt) && nxt == 0) break;
      cur = nxt;
    }
    std::sort(ent.begin(), ent.end());
    // node_list is filled in path order, idx_pos is the bit-vector: both ascending
    // because ref indexes never decrease along the ref path.
    g.idx_pos.clear(); g.node_list.clear();
    for (auto& e : ent) { g.idx_pos.push_back(e.first); g.node_list.push_back(e.second); }
  }

  std::string chr_;
  bool use_bv_;
  uint64_t ref_length_ = 0;
  uint32_t num_samples_ = 0, wpc_ = 0;
  uint64_t num_edges_ = 0, num_keys_ = 0;
  std::vector<std::string> names_;
  std::vector<uint32_t> off_, len_, class_, ref_index_, car_cnt_;
  std::vector<uint64_t> car_at_;
  std::vector<uint8_t> pool_flags_;
  std::vector<uint32_t> pool_sid_;
  std::vector<uint8_t> seq_;
  std::vector<uint64_t> sample_vector_, scratch_;
  std::unordered_map<uint64_t, uint32_t> sampleclass_map_;
  std::vector<uint32_t> topo_val_;
  std::vector<uint8_t> topo_inplace_;
  std::vector<RefOrderSet> aux_;
  std::map<uint64_t, uint64_t> idx_vertex_id_;
};

}  // namespace vsamd
