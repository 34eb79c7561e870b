// proto_io.hpp -- vertex_list_<k>.proto files: gzip( varint64 count, { varint32 size, bytes }* ) of
// protobuf-encoded VariantGraphVertexList messages (reference include/stream.hpp:25-52,70-112;
// schema include/variantgraphvertex.proto:6-26; written by variant_graph.h:453-477).
//
// A from-scratch proto3 wire codec for exactly these three messages (no protoc / libprotobuf in
// this image):
//   VariantGraphVertexList { repeated VariantGraphVertex vertex = 1; }
//   VariantGraphVertex     { uint32 vertex_id = 1; uint32 offset = 2; uint32 length = 3;
//                            repeated uint32 sampleclass_id = 4 [packed];
//                            repeated sample_info s_info = 5; }
//   sample_info            { uint32 index = 1; repeated uint32 sample_id = 2 [packed];
//                            bool phase = 3; bool gt_1 = 4; bool gt_2 = 5; }
// Known answer (produced with python google.protobuf): a list holding one vertex
// {vertex_id 3, offset 80, length 1, sampleclass_id [1], s_info [{index 9, phase, gt_1}]} is
// 0a11 0803 1050 1801 220101 2a06 0809 1801 2001.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>

namespace vsamd {
namespace proto {

inline void put_varint(std::string& o, uint64_t v) {
  while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; }
  o.push_back((char)v);
}
inline size_t varint_size(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; ++n; } return n; }

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    int shift = 0;
    while (true) {
      if (p >= end) throw std::runtime_error("protobuf: truncated varint");
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
      if (shift > 63) throw std::runtime_error("protobuf: varint too long");
    }
    return v;
  }
  Reader sub() {
    uint64_t n = varint();
    if ((uint64_t)(end - p) < n) throw std::runtime_error("protobuf: truncated field");
    Reader r{p, p + n};
    p += n;
    return r;
  }
  void skip(uint32_t wire) {
    switch (wire) {
      case 0: varint(); break;
      case 1: if (end - p < 8) throw std::runtime_error("protobuf: truncated"); p += 8; break;
      case 2: sub(); break;
      case 5: if (end - p < 4) throw std::runtime_error("protobuf: truncated"); p += 4; break;
      default: throw std::runtime_error("protobuf: unsupported wire type");
    }
  }
};

struct SInfo {
  uint32_t index = 0;
  bool has_sid = false;
  uint32_t sid = 0;
  uint8_t flags = 0;  // bit0 phase, bit1 gt_1, bit2 gt_2
};
struct Vertex {
  uint32_t vertex_id = 0, offset = 0, length = 0;
  bool has_class = false;
  uint32_t class_id = 0;
  std::vector<SInfo> s_info;
};

inline void encode_sinfo(std::string& o, const SInfo& s) {
  if (s.index) { o.push_back(0x08); put_varint(o, s.index); }
  if (s.has_sid) { o.push_back(0x12); put_varint(o, varint_size(s.sid)); put_varint(o, s.sid); }
  if (s.flags & 1) { o.push_back(0x18); o.push_back(1); }
  if (s.flags & 2) { o.push_back(0x20); o.push_back(1); }
  if (s.flags & 4) { o.push_back(0x28); o.push_back(1); }
}
inline size_t sinfo_size(const SInfo& s) {
  size_t n = 0;
  if (s.index) n += 1 + varint_size(s.index);
  if (s.has_sid) n += 2 + varint_size(s.sid);
  n += 2 * ((s.flags & 1) + ((s.flags >> 1) & 1) + ((s.flags >> 2) & 1));
  return n;
}
inline void encode_vertex_body(std::string& o, const Vertex& v) {
  if (v.vertex_id) { o.push_back(0x08); put_varint(o, v.vertex_id); }
  if (v.offset) { o.push_back(0x10); put_varint(o, v.offset); }
  if (v.length) { o.push_back(0x18); put_varint(o, v.length); }
  if (v.has_class) { o.push_back(0x22); put_varint(o, varint_size(v.class_id)); put_varint(o, v.class_id); }
  for (const auto& s : v.s_info) {
    o.push_back(0x2A);
    put_varint(o, sinfo_size(s));
    encode_sinfo(o, s);
  }
}
inline size_t vertex_body_size(const Vertex& v) {
  size_t n = 0;
  if (v.vertex_id) n += 1 + varint_size(v.vertex_id);
  if (v.offset) n += 1 + varint_size(v.offset);
  if (v.length) n += 1 + varint_size(v.length);
  if (v.has_class) n += 2 + varint_size(v.class_id);
  for (const auto& s : v.s_info) { size_t k = sinfo_size(s); n += 1 + varint_size(k) + k; }
  return n;
}
// appends `vertex = 1` of a VariantGraphVertexList
inline void encode_list_entry(std::string& o, const Vertex& v) {
  o.push_back(0x0A);
  put_varint(o, vertex_body_size(v));
  encode_vertex_body(o, v);
}

inline void decode_sinfo(Reader r, SInfo& s) {
  while (!r.done()) {
    uint64_t tag = r.varint();
    uint32_t f = (uint32_t)(tag >> 3), w = (uint32_t)(tag & 7);
    if (f == 1 && w == 0) s.index = (uint32_t)r.varint();
    else if (f == 2 && w == 2) { Reader q = r.sub(); while (!q.done()) { s.sid = (uint32_t)q.varint(); s.has_sid = true; break; } }
    else if (f == 2 && w == 0) { s.sid = (uint32_t)r.varint(); s.has_sid = true; }
    else if (f == 3 && w == 0) { if (r.varint()) s.flags |= 1; }
    else if (f == 4 && w == 0) { if (r.varint()) s.flags |= 2; }
    else if (f == 5 && w == 0) { if (r.varint()) s.flags |= 4; }
    else r.skip(w);
  }
}
inline void decode_vertex(Reader r, Vertex& v) {
  while (!r.done()) {
    uint64_t tag = r.varint();
    uint32_t f = (uint32_t)(tag >> 3), w = (uint32_t)(tag & 7);
    if (f == 1 && w == 0) v.vertex_id = (uint32_t)r.varint();
    else if (f == 2 && w == 0) v.offset = (uint32_t)r.varint();
    else if (f == 3 && w == 0) v.length = (uint32_t)r.varint();
    else if (f == 4 && w == 2) { Reader q = r.sub(); bool first = true; while (!q.done()) { uint32_t x = (uint32_t)q.varint(); if (first) { v.class_id = x; v.has_class = true; first = false; } } }
    else if (f == 4 && w == 0) { uint32_t x = (uint32_t)r.varint(); if (!v.has_class) { v.class_id = x; v.has_class = true; } }
    else if (f == 5 && w == 2) { v.s_info.emplace_back(); decode_sinfo(r.sub(), v.s_info.back()); }
    else r.skip(w);
  }
}

// ---- gzip framing (stream.hpp) ----
inline void write_framed_gzip(const std::string& path, const std::string& message) {
  gzFile f = gzopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  std::string head;
  put_varint(head, 1);                 // count
  put_varint(head, message.size());    // WriteVarint32(s.size())
  bool ok = gzwrite(f, head.data(), (unsigned)head.size()) == (int)head.size();
  size_t off = 0;
  while (ok && off < message.size()) {
    unsigned n = (unsigned)std::min<size_t>(message.size() - off, 1u << 30);
    ok = gzwrite(f, message.data() + off, n) == (int)n;
    off += n;
  }
  if (gzclose(f) != Z_OK || !ok) throw std::runtime_error("write failed: " + path);
}
// calls fn(Reader over one VariantGraphVertexList message) for every message in the file
template <typename Fn>
inline void read_framed_gzip(const std::string& path, Fn fn) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot read " + path);
  std::vector<uint8_t> buf;
  {
    std::vector<uint8_t> chunk(1 << 22);
    int n;
    while ((n = gzread(f, chunk.data(), (unsigned)chunk.size())) > 0) buf.insert(buf.end(), chunk.begin(), chunk.begin() + n);
    gzclose(f);
    if (n < 0) throw std::runtime_error("gzip error in " + path);
  }
  Reader r{buf.data(), buf.data() + buf.size()};
  while (!r.done()) {
    uint64_t count = r.varint();
    if (!count) break;
    for (uint64_t i = 0; i < count; ++i) {
      uint64_t sz = r.varint();
      if ((uint64_t)(r.end - r.p) < sz) throw std::runtime_error("truncated message in " + path);
      if (sz) fn(Reader{r.p, r.p + sz});
      r.p += sz;
    }
  }
}

}  // namespace proto
}  // namespace vsamd
