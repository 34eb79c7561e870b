// sdsl_io.hpp -- readers/writers for the sdsl-lite containers the reference persists
// (reference include/index.h:174-179, include/graph.h:191-207,
//  include/variant_graph.h:519-539): int_vector<0>, int_vector<32>, bit_vector and
// rrr_vector<127, int_vector<>, 32>.
//
// sdsl-lite is neither vendored in the reference nor installed here, so these are
// written to the published serialisation of sdsl-lite 2.x (the reference pins no
// version, README:29):
//   int_vector<w>::serialize   u64 bit-length, [u8 width when w == 0], then
//                              ceil(bits/64) little-endian words, elements packed LSB-first
//   rrr_vector::serialize      u64 size, bt (int_vector<>, width 7), btnr (bit_vector),
//                              btnrp (int_vector<>), rank samples (int_vector<>), invert
//                              (bit_vector); rank/select supports are NOT stored
// No file produced by a real sdsl build exists in this image: byte-level parity
// with genuine files is unpinned; these codecs round-trip with
// each other and are exercised by tests/test_formats.py.
#pragma once
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace vsamd {
namespace sdsl {

inline uint32_t hi(uint64_t x) { return x ? 63 - __builtin_clzll(x) : 0; }  // bits::hi (hi(0) == 0)

struct IntVector {  // int_vector<>: packed, runtime width
  uint8_t width = 64;
  uint64_t n = 0;
  std::vector<uint64_t> words;

  void init(uint64_t count, uint8_t w) {
    width = w; n = count;
    words.assign((count * w + 63) / 64, 0);
  }
  uint64_t get(uint64_t i) const { return get_bits(i * width, width); }
  void set(uint64_t i, uint64_t v) { set_bits(i * width, v, width); }
  uint64_t get_bits(uint64_t pos, uint32_t len) const {
    if (len == 0) return 0;
    const uint64_t w = pos >> 6, o = pos & 63;
    uint64_t v = words[w] >> o;
    if (o + len > 64) v |= words[w + 1] << (64 - o);
    return len == 64 ? v : v & ((1ULL << len) - 1);
  }
  void set_bits(uint64_t pos, uint64_t v, uint32_t len) {
    if (len == 0) return;
    const uint64_t w = pos >> 6, o = pos & 63;
    const uint64_t m = len == 64 ? ~0ULL : ((1ULL << len) - 1);
    v &= m;
    words[w] = (words[w] & ~(m << o)) | (v << o);
    if (o + len > 64) {
      const uint32_t r = (uint32_t)(o + len - 64);
      const uint64_t m2 = (1ULL << r) - 1;
      words[w + 1] = (words[w + 1] & ~m2) | (v >> (64 - o));
    }
  }
};

inline void write_u64(std::ostream& o, uint64_t v) { o.write((const char*)&v, 8); }
inline uint64_t read_u64(std::istream& in) {
  uint64_t v = 0;
  in.read((char*)&v, 8);
  if (!in) throw std::runtime_error("truncated sdsl file");
  return v;
}

// fixed_width == 0: dynamic (width byte in the header)
inline void write_int_vector(std::ostream& o, const IntVector& v, uint8_t fixed_width) {
  write_u64(o, v.n * v.width);
  if (fixed_width == 0) o.write((const char*)&v.width, 1);
  const uint64_t nw = (v.n * v.width + 63) / 64;
  if (nw) o.write((const char*)v.words.data(), nw * 8);
}
inline void read_int_vector(std::istream& in, IntVector& v, uint8_t fixed_width) {
  const uint64_t bits = read_u64(in);
  uint8_t w = fixed_width;
  if (fixed_width == 0) {
    in.read((char*)&w, 1);
    if (!in) throw std::runtime_error("truncated sdsl int_vector header");
  }
  if (w == 0 || w > 64) throw std::runtime_error("bad int_vector width");
  v.width = w;
  v.n = bits / w;
  const uint64_t nw = (bits + 63) / 64;
  v.words.assign(nw + 1, 0);  // +1: get_bits may touch the next word
  if (nw) in.read((char*)v.words.data(), nw * 8);
  if (!in) throw std::runtime_error("truncated sdsl int_vector data");
}

inline IntVector pack_u32(const std::vector<uint32_t>& src, uint8_t width) {
  IntVector v;
  v.init(src.size(), width);
  v.words.push_back(0);
  for (uint64_t i = 0; i < src.size(); ++i) v.set(i, src[i]);
  return v;
}
inline uint8_t bit_compress_width(const std::vector<uint32_t>& v) {  // util::bit_compress
  uint32_t mx = 0;
  for (uint32_t x : v) mx = x > mx ? x : mx;
  return (uint8_t)(hi(mx) + 1);
}

// ------------------------------------------------------------------ rrr_vector<127>
struct Binomial127 {
  static constexpr int N = 127;
  unsigned __int128 c[N + 1][N + 1];
  uint16_t space[N + 1];
  Binomial127() {
    for (int n = 0; n <= N; ++n)
      for (int k = 0; k <= N; ++k) c[n][k] = 0;
    for (int n = 0; n <= N; ++n) {
      c[n][0] = 1;
      for (int k = 1; k <= n; ++k) c[n][k] = c[n - 1][k - 1] + (k <= n - 1 ? c[n - 1][k] : 0);
    }
    for (int k = 0; k <= N; ++k) {
      unsigned __int128 x = c[N][k];
      if (x == 1) { space[k] = 0; continue; }
      uint16_t h = 0;
      while (x >>= 1) ++h;
      space[k] = h + 1;
    }
  }
  static const Binomial127& get() { static Binomial127 b; return b; }
};

typedef unsigned __int128 u128;

inline u128 bin_to_nr(u128 bin) {  // rrr_helper::bin_to_nr (combinatorial number system)
  if (bin == 0) return 0;
  const Binomial127& B = Binomial127::get();
  u128 nr = 0;
  int k = __builtin_popcountll((uint64_t)bin) + __builtin_popcountll((uint64_t)(bin >> 64));
  int pos = 0;
  while (bin != 0) {
    if (bin & 1) { nr += B.c[127 - pos - 1][k]; --k; }
    ++pos;
    bin >>= 1;
  }
  return nr;
}
inline u128 nr_to_bin(int k, u128 nr) {  // inverse of bin_to_nr
  const Binomial127& B = Binomial127::get();
  if (k == 0) return 0;
  u128 bin = 0;
  for (int pos = 0; pos < 127 && k > 0; ++pos) {
    const u128 c = B.c[127 - pos - 1][k];
    if (nr >= c) { nr -= c; bin |= (u128)1 << pos; --k; }
  }
  return bin;
}

struct PlainBits {
  uint64_t size = 0;
  std::vector<uint64_t> words;
  void init(uint64_t n) { size = n; words.assign((n + 63) / 64 + 2, 0); }
  bool get(uint64_t i) const { return (words[i >> 6] >> (i & 63)) & 1; }
  void set(uint64_t i) { words[i >> 6] |= 1ULL << (i & 63); }
  u128 get128(uint64_t pos, uint32_t len) const {  // up to 127 bits starting at pos, bit 0 = bit pos
    u128 v = 0;
    uint32_t got = 0;
    while (got < len) {
      const uint64_t w = (pos + got) >> 6, o = (pos + got) & 63;
      const uint32_t take = std::min<uint32_t>(64 - (uint32_t)o, len - got);
      uint64_t part = words[w] >> o;
      if (take < 64) part &= (1ULL << take) - 1;
      v |= (u128)part << got;
      got += take;
    }
    return v;
  }
};

// rrr_vector<127, int_vector<>, 32>(const bit_vector&) + serialize
inline void write_rrr127(std::ostream& o, const PlainBits& bv) {
  const Binomial127& B = Binomial127::get();
  const uint64_t t_bs = 127, t_k = 32;
  const uint64_t m_size = bv.size;
  const uint64_t nblocks = (m_size + t_bs) / t_bs;
  std::vector<uint16_t> bt(nblocks, 0);
  uint64_t pos = 0, i = 0, btnr_pos = 0, sum_rank = 0;
  auto popc = [&](uint64_t p, uint32_t len) {
    u128 v = bv.get128(p, len);
    return (uint16_t)(__builtin_popcountll((uint64_t)v) + __builtin_popcountll((uint64_t)(v >> 64)));
  };
  while (pos + t_bs <= m_size) {
    uint16_t x = popc(pos, 127);
    bt[i++] = x; sum_rank += x; btnr_pos += B.space[x]; pos += t_bs;
  }
  if (pos < m_size) {
    uint16_t x = popc(pos, (uint32_t)(m_size - pos));
    bt[i++] = x; sum_rank += x; btnr_pos += B.space[x];
  }
  IntVector btnr;  // bit_vector
  btnr.init(std::max<uint64_t>(btnr_pos, 64), 1);
  btnr.words.push_back(0); btnr.words.push_back(0);
  const uint64_t nsb = (nblocks + t_k - 1) / t_k;
  IntVector btnrp, rank, invert;
  btnrp.init(nsb, (uint8_t)(hi(btnr_pos) + 1)); btnrp.words.push_back(0);
  rank.init(nsb + ((m_size % (t_k * t_bs)) > 0 ? 1 : 0), (uint8_t)(hi(sum_rank) + 1)); rank.words.push_back(0);
  invert.init(nsb, 1); invert.words.push_back(0);

  pos = 0; i = 0; btnr_pos = 0; sum_rank = 0;
  bool inv = false;
  auto put_nr = [&](u128 nr, uint16_t len) {
    if (len <= 64) btnr.set_bits(btnr_pos, (uint64_t)nr, len);
    else { btnr.set_bits(btnr_pos, (uint64_t)nr, 64); btnr.set_bits(btnr_pos + 64, (uint64_t)(nr >> 64), len - 64); }
  };
  while (pos + t_bs <= m_size) {
    if (i % t_k == 0) {
      btnrp.set(i / t_k, btnr_pos);
      rank.set(i / t_k, sum_rank);
      if (i + t_k <= nblocks) {
        uint64_t gt_half = 0;
        for (uint64_t j = i; j < i + t_k; ++j) if (bt[j] > t_bs / 2) ++gt_half;
        if (gt_half > t_k / 2) {
          invert.set(i / t_k, 1);
          for (uint64_t j = i; j < i + t_k; ++j) bt[j] = (uint16_t)(t_bs - bt[j]);
          inv = true;
        } else inv = false;
      } else inv = false;
    }
    const uint16_t x = bt[i++];
    const uint16_t sp = B.space[x];
    sum_rank += inv ? (t_bs - x) : x;
    if (sp) put_nr(bin_to_nr(bv.get128(pos, 127)), sp);
    btnr_pos += sp;
    pos += t_bs;
  }
  if (pos < m_size) {
    if (i % t_k == 0) {
      btnrp.set(i / t_k, btnr_pos);
      rank.set(i / t_k, sum_rank);
      invert.set(i / t_k, 0);
      inv = false;
    }
    const uint16_t x = bt[i++];
    const uint16_t sp = B.space[x];
    sum_rank += inv ? (t_bs - x) : x;
    if (sp) put_nr(bin_to_nr(bv.get128(pos, (uint32_t)(m_size - pos))), sp);
    btnr_pos += sp;
  }
  rank.set(rank.n - 1, sum_rank);
  IntVector btv;
  btv.init(nblocks, 7); btv.words.push_back(0);
  for (uint64_t j = 0; j < nblocks; ++j) btv.set(j, bt[j]);
  write_u64(o, m_size);
  write_int_vector(o, btv, 0);
  write_int_vector(o, btnr, 1);
  write_int_vector(o, btnrp, 0);
  write_int_vector(o, rank, 0);
  write_int_vector(o, invert, 1);
}

// rrr_vector::load + full decode into plain bits
inline void read_rrr127(std::istream& in, PlainBits& bv) {
  const Binomial127& B = Binomial127::get();
  const uint64_t t_bs = 127, t_k = 32;
  const uint64_t m_size = read_u64(in);
  IntVector bt, btnr, btnrp, rank, invert;
  read_int_vector(in, bt, 0);
  read_int_vector(in, btnr, 1);
  read_int_vector(in, btnrp, 0);
  read_int_vector(in, rank, 0);
  read_int_vector(in, invert, 1);
  btnr.words.push_back(0); btnr.words.push_back(0);
  bv.init(m_size);
  const uint64_t nblocks = (m_size + t_bs) / t_bs;
  if (bt.n < nblocks && m_size) throw std::runtime_error("rrr_vector: block-type array too short");
  uint64_t p = 0;
  for (uint64_t b = 0; b < nblocks; ++b) {
    const uint64_t sb = b / t_k;
    if (b % t_k == 0) p = sb < btnrp.n ? btnrp.get(sb) : p;
    uint16_t stored = (uint16_t)bt.get(b);
    uint16_t k = (sb < invert.n && invert.get(sb)) ? (uint16_t)(t_bs - stored) : stored;
    const uint16_t sp = B.space[stored];
    const uint64_t base = b * t_bs;
    if (base >= m_size) break;
    u128 bin;
    if (k == 0) bin = 0;
    else if (k == 127) bin = (((u128)1) << 127) - 1;
    else {
      u128 nr;
      if (sp <= 64) nr = btnr.get_bits(p, sp);
      else nr = (u128)btnr.get_bits(p, 64) | ((u128)btnr.get_bits(p + 64, sp - 64) << 64);
      bin = nr_to_bin(k, nr);
    }
    p += sp;
    const uint32_t len = (uint32_t)std::min<uint64_t>(t_bs, m_size - base);
    for (uint32_t j = 0; j < len; ++j)
      if ((bin >> j) & 1) bv.set(base + j);
  }
}

}  // namespace sdsl
}  // namespace vsamd
