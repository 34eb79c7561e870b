// index_files.hpp -- index directory <-> HostGraph: the on-disk drop-in surface.
//
// Reads and writes the files `variantstore construct` leaves in <prefix>/ (SURVEY.md §8c):
//   index.sdsl, ref_node_id.sdsl            Index::serialize            include/index.h:174-179
//   adj_list.cqf, aux_vertex_list*.sdsl     Graph::serialize            include/graph.h:179-208
//   vertex_list_<k>.proto                   serialize_vertex_list       include/variant_graph.h:453-477
//   seq_buffer.sdsl, sample_vector.sdsl,
//   sampleid_map.lst                        VariantGraph::serialize     include/variant_graph.h:501-557
// and the matching loaders (index.h:108-117, graph.h:149-172, variant_graph.h:366-446).
#pragma once
#include <algorithm>
#include <dirent.h>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <unordered_map>
#include "host_graph.hpp"
#include "formats/cqf_io.hpp"
#include "formats/proto_io.hpp"
#include "formats/sdsl_io.hpp"

namespace vsamd {

constexpr uint32_t kVertexesInBlock = 200000;  // NUM_VERTEXES_IN_BLOCK, variant_graph.h:44

inline void save_index_dir(const HostGraph& g, const std::string& prefix) {
  mkdir(prefix.c_str(), 0755);
  const uint64_t V = g.num_vertices();
  auto path = [&](const std::string& f) { return prefix + "/" + f; };
  auto open_out = [&](const std::string& f) {
    std::ofstream o(path(f), std::ios::binary | std::ios::trunc);
    if (!o) throw std::runtime_error("cannot write " + path(f));
    return o;
  };

  // ---- vertex blocks ----
  const uint32_t wpc = g.words_per_class();
  for (uint64_t b0 = 0, k = 0; b0 < V; b0 += kVertexesInBlock, ++k) {
    std::string msg;
    const uint64_t b1 = std::min<uint64_t>(V, b0 + kVertexesInBlock);
    proto::Vertex pv;
    for (uint64_t v = b0; v < b1; ++v) {
      pv.vertex_id = (uint32_t)v; pv.offset = g.off[v]; pv.length = g.len[v];
      pv.has_class = g.use_bit_vector; pv.class_id = g.class_id[v];
      pv.s_info.clear();
      if (g.ref_index[v]) {
        proto::SInfo s;
        s.index = g.ref_index[v];
        if (!g.use_bit_vector) { s.has_sid = true; s.sid = 0; }
        pv.s_info.push_back(s);
      }
      for (uint64_t c = g.car_begin[v]; c < g.car_begin[v + 1]; ++c) {
        proto::SInfo s;
        s.index = g.car_index.empty() ? 0 : g.car_index[c];
        s.flags = g.car_flags[c] & 7;
        if (!g.use_bit_vector) { s.has_sid = true; s.sid = g.car_sid[c]; }
        pv.s_info.push_back(s);
      }
      proto::encode_list_entry(msg, pv);
    }
    proto::write_framed_gzip(path("vertex_list_" + std::to_string(k) + ".proto"), msg);
  }

  // ---- seq_buffer.sdsl: int_vector<>, bit-compressed (variant_graph.h:519-525) ----
  {
    uint8_t mx = 0;
    for (uint8_t c : g.seq) mx = c > mx ? c : mx;
    sdsl::IntVector iv;
    iv.init(g.seq.size(), (uint8_t)(sdsl::hi(mx) + 1));
    iv.words.push_back(0);
    for (uint64_t i = 0; i < g.seq.size(); ++i) iv.set(i, g.seq[i]);
    auto o = open_out("seq_buffer.sdsl");
    sdsl::write_int_vector(o, iv, 0);
  }

  // ---- topology ----
  {
    std::vector<cqf::Entry> entries;
    for (uint64_t v = 0; v < V; ++v)
      if (g.topo_val[v]) entries.push_back(cqf::Entry{v, g.topo_inplace[v] ? 1ULL : 0ULL, g.topo_val[v]});
    cqf::Filter f;
    cqf::build(entries, f);
    f.save(path("adj_list.cqf"));
    std::vector<uint32_t> flat, lens;
    for (const auto& l : g.aux_lists) {
      flat.insert(flat.end(), l.begin(), l.end());
      lens.push_back((uint32_t)l.size());
    }
    auto o1 = open_out("aux_vertex_list.sdsl");
    sdsl::write_int_vector(o1, sdsl::pack_u32(flat, 32), 32);
    auto o2 = open_out("aux_vertex_list_lengths.sdsl");
    sdsl::write_int_vector(o2, sdsl::pack_u32(lens, 32), 32);
  }

  // ---- sample_vector.sdsl: rrr over num_classes x num_samples bits, row stride = num_samples ----
  {
    sdsl::PlainBits bv;
    const uint64_t N = g.num_samples;
    bv.init(g.use_bit_vector ? g.num_classes * N : 0);
    if (g.use_bit_vector)
      for (uint64_t c = 0; c < g.num_classes; ++c)
        for (uint64_t j = 0; j < N; ++j)
          if ((g.class_bits[c * wpc + (j >> 6)] >> (j & 63)) & 1) bv.set(c * N + j);
    auto o = open_out("sample_vector.sdsl");
    sdsl::write_rrr127(o, bv);
  }

  // ---- sampleid_map.lst (variant_graph.h:542-556): unordered_map iteration order ----
  {
    std::unordered_map<std::string, uint32_t> m;
    for (uint32_t i = 0; i < g.sample_names.size(); ++i) m.insert(std::make_pair(g.sample_names[i], i));
    std::ofstream o(path("sampleid_map.lst"), std::ios::trunc);
    if (!o) throw std::runtime_error("cannot write sampleid_map.lst");
    o << g.chr << " " << g.ref_length << "\n" << g.chr << " " << g.num_samples << "\n";
    for (const auto& kv : m) o << kv.first << " " << kv.second << "\n";
  }

  // ---- index ----
  {
    sdsl::PlainBits bv;
    bv.init(g.ref_length);
    for (uint32_t p : g.idx_pos)
      if (p >= 1 && p <= g.ref_length) bv.set(p - 1);
    auto o = open_out("index.sdsl");
    sdsl::write_rrr127(o, bv);
    auto o2 = open_out("ref_node_id.sdsl");
    sdsl::write_int_vector(o2, sdsl::pack_u32(g.node_list, sdsl::bit_compress_width(g.node_list)), 0);
  }
}

inline void load_index_dir(const std::string& prefix, HostGraph& g) {
  auto path = [&](const std::string& f) { return prefix + "/" + f; };
  auto open_in = [&](const std::string& f) {
    std::ifstream in(path(f), std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path(f));
    return in;
  };
  // ---- sampleid_map.lst ----
  {
    std::ifstream in(path("sampleid_map.lst"));
    if (!in.good()) throw std::runtime_error("Failed to open sampleid map file " + path("sampleid_map.lst"));
    std::string s;
    uint64_t n = 0;
    in >> g.chr >> g.ref_length;
    in >> s >> n;
    if (n == 0) throw std::runtime_error("Num samples is less or equal to 0.");
    g.num_samples = (uint32_t)n;
    g.sample_names.assign(n, "");
    uint32_t id;
    uint64_t cnt = 0;
    while (in >> s >> id) {
      if (id >= n) throw std::runtime_error("sample id out of range in sampleid_map.lst");
      g.sample_names[id] = s;
      g.sample_file_order.push_back(s);
      ++cnt;
    }
    if (cnt != n) throw std::runtime_error("Num samples is not equal to num entries in samples file.");
  }
  // ---- vertex blocks, sorted by their number (variant_graph.h:370-397) ----
  std::map<long, std::string> files;
  {
    DIR* d = opendir(prefix.c_str());
    if (!d) throw std::runtime_error("cannot open directory " + prefix);
    while (dirent* e = readdir(d)) {
      std::string f = e->d_name;
      if (f.size() > 6 && f.substr(f.size() - 6) == ".proto" && f.rfind("vertex_list_", 0) == 0)
        files[atol(f.substr(12, f.size() - 18).c_str())] = f;
    }
    closedir(d);
  }
  if (files.empty()) throw std::runtime_error("no vertex_list_*.proto in " + prefix);
  bool any_sid = false, any_class = false;
  std::vector<uint32_t> pending_first_;  // class-vector mode: index of every vertex's first s_info entry
  std::vector<uint8_t> pending_flags_;
  g.car_begin.assign(1, 0);
  for (auto& kv : files) {
    proto::read_framed_gzip(path(kv.second), [&](proto::Reader list) {
      while (!list.done()) {
        uint64_t tag = list.varint();
        if ((tag >> 3) != 1 || (tag & 7) != 2) { list.skip((uint32_t)(tag & 7)); continue; }
        proto::Vertex v;
        proto::decode_vertex(list.sub(), v);
        if (v.vertex_id != g.off.size()) throw std::runtime_error("vertex ids are not consecutive in " + kv.second);
        if (v.s_info.empty()) throw std::runtime_error("vertex without s_info in " + kv.second);
        g.off.push_back(v.offset); g.len.push_back(v.length); g.class_id.push_back(v.class_id);
        any_class |= v.has_class;
        any_sid |= v.s_info[0].has_sid;
        // is the first entry "ref"?  explicit ids say so directly; with class vectors it is
        // decided below once the class rows are known (bit 0 of the row); remember the raw list
        uint32_t ridx = 0;
        size_t first = 0;
        if (v.s_info[0].has_sid) {
          if (v.s_info[0].sid == 0) { ridx = v.s_info[0].index; first = 1; }
        } else {
          pending_first_.push_back(v.s_info[0].index);  // resolved later
          pending_flags_.push_back(v.s_info[0].flags);
          first = 1;
        }
        g.ref_index.push_back(ridx);
        for (size_t i = first; i < v.s_info.size(); ++i) {
          g.car_flags.push_back(v.s_info[i].flags);
          g.car_index.push_back(v.s_info[i].index);
          if (v.s_info[i].has_sid) g.car_sid.push_back(v.s_info[i].sid);
        }
        g.car_begin.push_back(g.car_flags.size());
      }
    });
  }
  g.use_bit_vector = !any_sid;
  if (any_sid && !pending_first_.empty()) throw std::runtime_error("index mixes explicit sample ids and class vectors");
  (void)any_class;
  const uint64_t V = g.off.size();

  // ---- sample_vector.sdsl ----
  {
    auto in = open_in("sample_vector.sdsl");
    sdsl::PlainBits bv;
    sdsl::read_rrr127(in, bv);
    const uint64_t N = g.num_samples;
    const uint32_t wpc = g.words_per_class();
    g.num_classes = N ? bv.size / N : 0;
    g.class_bits.assign(g.num_classes * wpc, 0);
    for (uint64_t c = 0; c < g.num_classes; ++c)
      for (uint64_t j = 0; j < N; ++j)
        if (bv.get(c * N + j)) g.class_bits[c * wpc + (j >> 6)] |= 1ULL << (j & 63);
  }
  // resolve "is the first s_info entry the ref sample" for class-vector vertices
  if (g.use_bit_vector) {
    if (pending_first_.size() != V) throw std::runtime_error("internal: pending s_info mismatch");
    // the carrier pool was filled assuming entry 0 is ref; vertices whose class lacks bit 0 get
    // entry 0 back as their first carrier
    std::vector<uint8_t> nf; std::vector<uint32_t> ni; std::vector<uint64_t> nb(1, 0);
    nf.reserve(g.car_flags.size() + V); ni.reserve(g.car_flags.size() + V);
    for (uint64_t v = 0; v < V; ++v) {
      const uint32_t cls = g.class_id[v];
      if (cls > g.num_classes) throw std::runtime_error("vertex refers to a sample class that does not exist");
      const bool has_ref = cls == 0 || (g.class_bits[(uint64_t)(cls - 1) * g.words_per_class()] & 1);
      if (has_ref) g.ref_index[v] = pending_first_[v];
      else { nf.push_back(pending_flags_[v]); ni.push_back(pending_first_[v]); }
      for (uint64_t c = g.car_begin[v]; c < g.car_begin[v + 1]; ++c) { nf.push_back(g.car_flags[c]); ni.push_back(g.car_index[c]); }
      nb.push_back(nf.size());
    }
    g.car_flags.swap(nf); g.car_index.swap(ni); g.car_begin.swap(nb);
    pending_first_.clear(); pending_flags_.clear();
  }
  bool any_index = false;
  for (uint32_t x : g.car_index) if (x) { any_index = true; break; }
  if (!any_index) g.car_index.clear();

  // ---- seq_buffer.sdsl ----
  {
    auto in = open_in("seq_buffer.sdsl");
    sdsl::IntVector iv;
    sdsl::read_int_vector(in, iv, 0);
    g.seq.resize(iv.n);
    for (uint64_t i = 0; i < iv.n; ++i) g.seq[i] = (uint8_t)iv.get(i);
  }
  // ---- topology ----
  {
    cqf::Filter f;
    f.load(path("adj_list.cqf"));
    g.topo_val.assign(V, 0); g.topo_inplace.assign(V, 0);
    g.num_keys = 0;
    for (uint64_t v = 0; v < V; ++v) {
      uint64_t val = 0;
      uint64_t cnt = f.query(v, &val);
      if (cnt) { g.topo_val[v] = (uint32_t)cnt; g.topo_inplace[v] = (uint8_t)(val & 1); g.num_keys++; }
    }
    auto in1 = open_in("aux_vertex_list.sdsl");
    auto in2 = open_in("aux_vertex_list_lengths.sdsl");
    sdsl::IntVector flat, lens;
    sdsl::read_int_vector(in1, flat, 32);
    sdsl::read_int_vector(in2, lens, 32);
    g.aux_lists.clear();
    uint64_t at = 0;
    for (uint64_t i = 0; i < lens.n; ++i) {
      uint64_t l = lens.get(i);
      if (at + l > flat.n) throw std::runtime_error("aux_vertex_list lengths exceed the list");
      std::vector<uint32_t> lst(l);
      for (uint64_t j = 0; j < l; ++j) lst[j] = (uint32_t)flat.get(at + j);
      g.aux_lists.push_back(std::move(lst));
      at += l;
    }
    for (uint64_t v = 0; v < V; ++v)
      if (g.topo_val[v] && !g.topo_inplace[v] && g.topo_val[v] > g.aux_lists.size())
        throw std::runtime_error("adjacency points past the aux vertex lists");
  }
  // ---- index ----
  {
    auto in = open_in("index.sdsl");
    sdsl::PlainBits bv;
    sdsl::read_rrr127(in, bv);
    g.idx_pos.clear();
    for (uint64_t i = 0; i < bv.size; ++i) if (bv.get(i)) g.idx_pos.push_back((uint32_t)(i + 1));
    auto in2 = open_in("ref_node_id.sdsl");
    sdsl::IntVector nl;
    sdsl::read_int_vector(in2, nl, 0);
    g.node_list.resize(nl.n);
    for (uint64_t i = 0; i < nl.n; ++i) g.node_list[i] = (uint32_t)nl.get(i);
    if (g.node_list.size() != g.idx_pos.size()) throw std::runtime_error("index.sdsl and ref_node_id.sdsl disagree");
  }
}

}  // namespace vsamd
