// Test-only harness around the product's file codecs (variantstore_amd/csrc/host/formats/):
//   formats_check proto      -> hex of the protobuf known answer
//   formats_check rrr <seed> -> random bit-vectors through write_rrr127 / read_rrr127
//   formats_check intvec     -> int_vector widths 1..64 round trip
//   formats_check cqf <seed> -> cqf::build + Filter::query on random (key, value, count) sets
#include <cstdio>
#include <cstring>
#include <random>
#include <sstream>
#include "../../variantstore_amd/csrc/host/formats/sdsl_io.hpp"
#include "../../variantstore_amd/csrc/host/formats/proto_io.hpp"
#include "../../variantstore_amd/csrc/host/formats/cqf_io.hpp"
using namespace vsamd;

int main(int argc, char** argv) {
  std::string mode = argc > 1 ? argv[1] : "";
  uint64_t seed = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
  std::mt19937_64 rng(seed);
  if (mode == "proto") {
    proto::Vertex v;
    v.vertex_id = 3; v.offset = 80; v.length = 1; v.has_class = true; v.class_id = 1;
    proto::SInfo s; s.index = 9; s.flags = 1 | 2;
    v.s_info.push_back(s);
    std::string o;
    proto::encode_list_entry(o, v);
    for (unsigned char c : o) printf("%02x", c);
    printf("\n");
    proto::Vertex back;
    proto::Reader r{(const uint8_t*)o.data() + 2, (const uint8_t*)o.data() + o.size()};
    proto::decode_vertex(r, back);
    if (back.vertex_id != 3 || back.offset != 80 || back.length != 1 || back.class_id != 1 || back.s_info.size() != 1 ||
        back.s_info[0].index != 9 || back.s_info[0].flags != 3) { printf("FAIL decode\n"); return 1; }
    return 0;
  }
  if (mode == "rrr") {
    for (int round = 0; round < 60; ++round) {
      uint64_t n = round < 8 ? (uint64_t[]){0, 1, 63, 126, 127, 128, 4064, 4065}[round] : 1 + rng() % 40000;
      double dens = (round % 5 == 0) ? 0.9 : (round % 5 == 1 ? 0.5 : (round % 5 == 2 ? 0.001 : 0.05));
      sdsl::PlainBits bv; bv.init(n);
      for (uint64_t i = 0; i < n; ++i) if ((rng() >> 11) * (1.0 / 9007199254740992.0) < dens) bv.set(i);
      if (round % 7 == 3) for (uint64_t i = 0; i < n; ++i) bv.set(i);  // all ones
      std::stringstream ss;
      sdsl::write_rrr127(ss, bv);
      sdsl::PlainBits back;
      sdsl::read_rrr127(ss, back);
      if (back.size != n) { printf("FAIL size %lu\n", n); return 1; }
      for (uint64_t i = 0; i < n; ++i) if (back.get(i) != bv.get(i)) { printf("FAIL bit %lu of %lu (round %d)\n", i, n, round); return 1; }
    }
    // combinatorial number system is a bijection on 127-bit blocks
    for (int t = 0; t < 2000; ++t) {
      sdsl::u128 b = ((sdsl::u128)rng() << 64 | rng()) & ((((sdsl::u128)1) << 127) - 1);
      if (t % 3 == 0) b &= ((sdsl::u128)rng() << 64 | rng());
      int k = __builtin_popcountll((uint64_t)b) + __builtin_popcountll((uint64_t)(b >> 64));
      if (sdsl::nr_to_bin(k, sdsl::bin_to_nr(b)) != b) { printf("FAIL nr\n"); return 1; }
      if (sdsl::bin_to_nr(b) >= sdsl::Binomial127::get().c[127][k]) { printf("FAIL range\n"); return 1; }
    }
    printf("OK\n");
    return 0;
  }
  if (mode == "intvec") {
    for (int w = 1; w <= 64; ++w) {
      sdsl::IntVector v; v.init(1000, (uint8_t)w); v.words.push_back(0);
      std::vector<uint64_t> ref(1000);
      for (int i = 0; i < 1000; ++i) { ref[i] = rng() & (w == 64 ? ~0ULL : ((1ULL << w) - 1)); v.set(i, ref[i]); }
      std::stringstream ss;
      sdsl::write_int_vector(ss, v, 0);
      sdsl::IntVector b; sdsl::read_int_vector(ss, b, 0);
      if (b.n != 1000 || b.width != w) { printf("FAIL hdr %d\n", w); return 1; }
      for (int i = 0; i < 1000; ++i) if (b.get(i) != ref[i]) { printf("FAIL w=%d i=%d\n", w, i); return 1; }
    }
    // header bytes of an int_vector<32> with 3 elements: bit length 96 then 2 words
    { sdsl::IntVector v = sdsl::pack_u32({1, 2, 3}, 32); std::stringstream ss; sdsl::write_int_vector(ss, v, 32);
      std::string s = ss.str(); uint64_t bits; memcpy(&bits, s.data(), 8);
      if (bits != 96 || s.size() != 8 + 16) { printf("FAIL int_vector<32> layout\n"); return 1; } }
    printf("OK\n");
    return 0;
  }
  if (mode == "cqf") {
    std::vector<cqf::Entry> es;
    uint64_t n = 50000;
    for (uint64_t k = 0; k < n; ++k) {
      if (rng() % 10 == 0) continue;
      uint64_t cnt = (rng() % 4 == 0) ? 1 + rng() % 5 : 1 + rng() % 3000000;
      es.push_back(cqf::Entry{k, rng() & 1, cnt});
    }
    cqf::Filter f; cqf::build(es, f);
    size_t at = 0;
    for (uint64_t k = 0; k < n; ++k) {
      uint64_t val = 9, cnt = f.query(k, &val);
      if (at < es.size() && es[at].key == k) {
        if (cnt != es[at].count || val != es[at].value) { printf("FAIL key %lu: %lu/%lu vs %lu/%lu\n", k, cnt, val, es[at].count, es[at].value); return 1; }
        ++at;
      } else if (cnt != 0) { printf("FAIL absent key %lu\n", k); return 1; }
    }
    if (argc > 3) f.save(argv[3]);
    printf("OK %lu %lu %lu\n", (unsigned long)f.md.nslots, (unsigned long)f.md.noccupied_slots, (unsigned long)f.md.ndistinct_elts);
    return 0;
  }
  return 2;
}
