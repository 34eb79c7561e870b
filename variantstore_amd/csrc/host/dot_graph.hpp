// dot_graph.hpp -- `variantstore draw`: the radius-limited neighbourhood of a vertex as a Graphviz file.
//
// Host-only (a debugging visualiser: a few hundred vertices at most; nothing here is worth a GPU).
// Restates, over HostGraph:
//   draw_subgraph                      reference include/query.h:825-842
//   createDotGraph / get_samples / is_ref_node   include/dot_graph.h:42-132
//   Graph::GraphIterator with a radius (the common-neighbour ordering of the expansion)   include/graph.h:394-459
//   get_prev_vertex_with_sample        include/query.h:57-113 (start vertex for a non-ref sample)
#pragma once
#include <algorithm>
#include <deque>
#include <string>
#include <unordered_set>
#include <vector>
#include "host_graph.hpp"

namespace vsamd {

// s_info of a vertex as (sample id, index) pairs in s_info order: the ref entry first, then the carrier pool.
// In bit-vector mode the id attached to entry i is the i-th set bit of the class row (variant_graph.h:875-880).
inline void vertex_entries(const HostGraph& g, uint32_t v, std::vector<std::pair<uint32_t, uint32_t>>& out) {
  out.clear();
  std::vector<uint32_t> idx;
  if (g.ref_index[v]) idx.push_back(g.ref_index[v]);
  for (uint64_t c = g.car_begin[v]; c < g.car_begin[v + 1]; ++c) idx.push_back(g.car_index.empty() ? 0u : g.car_index[c]);
  if (g.use_bit_vector) {
    std::vector<uint32_t> ids;
    if (g.class_id[v] == 0) ids.push_back(0);
    else {
      const uint64_t* row = &g.class_bits[(uint64_t)(g.class_id[v] - 1) * g.words_per_class()];
      for (uint32_t j = 0; j < g.num_samples; ++j)
        if ((row[j >> 6] >> (j & 63)) & 1) ids.push_back(j);
    }
    for (size_t i = 0; i < idx.size(); ++i) out.emplace_back(i < ids.size() ? ids[i] : 0xFFFFFFFFu, idx[i]);
  } else {
    size_t i = 0;
    if (g.ref_index[v]) out.emplace_back(0u, idx[i++]);
    for (uint64_t c = g.car_begin[v]; c < g.car_begin[v + 1]; ++c) out.emplace_back(g.car_sid[c], idx[i++]);
  }
}

// visiting order of Graph::GraphIterator(v, radius)
inline std::vector<uint32_t> bfs_order(const HostGraph& g, uint32_t v, uint64_t radius) {
  std::vector<uint32_t> order{v}, nb, nb2;
  std::unordered_set<uint32_t> visited{v};
  std::deque<std::pair<uint32_t, uint64_t>> q;
  if (radius > 0) {
    g.out_neighbors(v, nb);
    for (uint32_t n : nb) q.emplace_back(n, 1);
  }
  while (true) {
    uint32_t cur = 0;
    uint64_t hop = 0;
    bool got = false;
    while (!q.empty()) {
      cur = q.front().first; hop = q.front().second;
      q.pop_front();
      if (visited.insert(cur).second) { got = true; break; }
    }
    if (!got) break;
    order.push_back(cur);
    if (hop < radius) {
      // neighbours that share a neighbour with `cur` go to the front of the expansion, the others to the back
      g.out_neighbors(cur, nb);
      std::vector<uint32_t> sorted_cur(nb);
      std::sort(sorted_cur.begin(), sorted_cur.end());
      std::vector<uint32_t> ordered;
      for (uint32_t n : nb) {
        g.out_neighbors(n, nb2);
        std::sort(nb2.begin(), nb2.end());
        std::vector<uint32_t> common;
        std::set_intersection(sorted_cur.begin(), sorted_cur.end(), nb2.begin(), nb2.end(), std::back_inserter(common));
        if (!common.empty()) ordered.insert(ordered.begin(), n);
        else ordered.push_back(n);
      }
      for (uint32_t n : ordered) q.emplace_back(n, hop + 1);
    }
  }
  return order;
}

inline std::string dot_text(const HostGraph& g, uint32_t v, uint64_t radius) {
  const std::vector<uint32_t> order = bfs_order(g, v, radius);
  std::string labels, ref, sample;
  std::vector<std::pair<uint32_t, uint32_t>> ent;
  for (uint32_t n : order) {
    vertex_entries(g, n, ent);
    labels += std::to_string(n) + "[ label=\"" + std::to_string(n) + " l:" + std::to_string((int)g.len[n]) + "\n(";
    for (size_t i = 0; i < ent.size(); ++i) {
      labels += (ent[i].first < g.sample_names.size() ? g.sample_names[ent[i].first] : std::string("?")) + " i:" +
                std::to_string((int)ent[i].second);
      if (i + 1 < ent.size()) labels += "\n";
    }
    labels += ")\"]\n";
  }
  ref += "\tsubgraph cluster_0 {\n\t\tlabel=\"reference\";\n";
  std::vector<uint32_t> nb;
  for (uint32_t n : order) {
    g.out_neighbors(n, nb);
    for (uint32_t m : nb) {
      if (g.ref_index[m] != 0 && g.ref_index[n] != 0) ref += "\t\t" + std::to_string(n) + " -> " + std::to_string(m) + "\n";
      else sample += "\t" + std::to_string(n) + " -> " + std::to_string(m) + "\n";
    }
  }
  ref += "\t}\n";
  return "digraph {\n" + labels + ref + sample + "}";
}

// Index::find(pos) and find(pos, rank) on the host (index.h:119-148)
inline uint32_t host_find(const HostGraph& g, uint64_t pos, uint64_t* rank = nullptr) {
  if (pos >= g.ref_length) {
    if (rank) *rank = g.node_list.size() - 1;
    return g.node_list.back();
  }
  const uint64_t k = std::upper_bound(g.idx_pos.begin(), g.idx_pos.end(), (uint32_t)pos) - g.idx_pos.begin();
  if (k == 0) { if (rank) *rank = 0; return g.node_list[0]; }
  if (rank) *rank = k - 1;
  return g.node_list[k - 1];
}

inline bool host_has_sample(const HostGraph& g, uint32_t v, uint32_t sid) {
  if (sid == 0) return g.ref_index[v] != 0;
  if (g.use_bit_vector) return g.class_has(g.class_id[v], sid);
  for (uint64_t c = g.car_begin[v]; c < g.car_begin[v + 1]; ++c)
    if (g.car_sid[c] == sid) return true;
  return false;
}

// start vertex of draw_subgraph: find(pos) for "ref", else get_prev_vertex_with_sample (query.h:57-113)
inline uint32_t draw_start_vertex(const HostGraph& g, uint64_t pos, uint32_t sid) {
  if (sid == 0) return host_find(g, pos);
  uint64_t rank = 0;
  uint32_t v_find = host_find(g, pos, &rank);
  std::vector<uint32_t> nb;
  while (true) {
    const uint32_t v = g.node_list[rank == 0 ? 0 : rank - 1];
    if (rank <= 1) return v;
    bool found = false;
    g.out_neighbors(v, nb);
    for (uint32_t n : nb) {
      if (host_has_sample(g, n, sid)) { v_find = n; found = true; }
      rank = rank ? rank - 1 : 0;
    }
    if (found) return v_find;
  }
}

}  // namespace vsamd
