// cqf_io.hpp -- adj_list.cqf: the counting quotient filter that holds the graph topology
// (reference include/graph.h:139-147,179-208; file = raw qfmetadata + blocks,
// src/gqf/gqf_file.c:259-332, structs include/gqf/gqf_int.h:37-101).
//
// key   = vertex id, hashed with the invertible 40-bit hash (src/gqf/hashutil.c:132-142)
// value = 1 bit: "count is the single out-neighbour" vs "count is a 1-based aux-list index"
// count = stored in the filter's variable-length counter encoding (src/gqf/gqf.c:1052-1182)
//
// Reader: qf_query restated (gqf.c:2081-2118 with is_occupied / run_end / decode_counter).
// Writer: lays the filter out directly from the sorted (bucket, remainder) entries -- a
// quotient filter's slot array is canonical for a given multiset, so this reproduces what
// the reference's sequence of inserts leaves behind.  tests/test_formats.py checks both
// directions against the reference's own gqf code (oracle/_ref/libgqf_ref.so): byte-equal
// files, and qf_query on files written here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace vsamd {
namespace cqf {

constexpr uint64_t kMagic = 1018874902021329732ULL;
constexpr uint32_t kSeed = 2038074761u;   // GQF_SEED, gqf_cpp.h:27
constexpr uint64_t kKeyBits = 40;          // KEYBITS, graph.h:30
constexpr uint64_t kValueBits = 1;
constexpr uint64_t kDefaultSlots = 1ULL << 25;  // DEFAULT_SIZE, graph.h:29

#pragma pack(push, 1)
struct Metadata {  // quotient_filter_metadata, 128 bytes
  uint64_t magic_endian_number;
  uint32_t hash_mode;
  uint32_t auto_resize;
  uint64_t total_size_in_bytes;
  uint32_t seed;
  uint32_t pad_;
  uint64_t nslots, xnslots, key_bits, value_bits, key_remainder_bits, bits_per_slot;
  unsigned __int128 range;
  uint64_t nblocks, nelts, ndistinct_elts, noccupied_slots;
};
#pragma pack(pop)
static_assert(sizeof(Metadata) == 128, "qfmetadata layout");

inline uint64_t bitmask(uint64_t n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1); }

inline uint64_t hash_64(uint64_t key, uint64_t mask) {  // Thomas Wang's invertible integer hash
  key = (~key + (key << 21)) & mask;
  key = key ^ key >> 24;
  key = ((key + (key << 3)) + (key << 8)) & mask;
  key = key ^ key >> 14;
  key = ((key + (key << 2)) + (key << 4)) & mask;
  key = key ^ key >> 28;
  key = (key + (key << 31)) & mask;
  return key;
}

struct Filter {
  Metadata md{};
  std::vector<uint8_t> blocks;  // nblocks * block_bytes (+8 bytes slack for the unaligned slot loads)
  uint64_t block_bytes = 0;

  void init(uint64_t nslots) {
    memset(&md, 0, sizeof(md));
    md.magic_endian_number = kMagic;
    md.hash_mode = 1;  // QF_HASH_INVERTIBLE
    md.auto_resize = 1;
    md.seed = kSeed;
    md.nslots = nslots;
    md.xnslots = nslots + (uint64_t)(10 * std::sqrt((double)nslots));
    md.key_bits = kKeyBits;
    md.value_bits = kValueBits;
    uint64_t krb = kKeyBits;
    for (uint64_t n = nslots; n > 1; n >>= 1) --krb;
    md.key_remainder_bits = krb;
    md.bits_per_slot = krb + kValueBits;
    md.range = (unsigned __int128)nslots << krb;
    md.nblocks = (md.xnslots + 63) / 64;
    block_bytes = 18 + 8 * md.bits_per_slot;
    md.total_size_in_bytes = md.nblocks * block_bytes;
    blocks.assign(md.total_size_in_bytes + 8, 0);
  }
  uint8_t* blk(uint64_t b) { return &blocks[b * block_bytes]; }
  const uint8_t* blk(uint64_t b) const { return &blocks[b * block_bytes]; }
  uint16_t offset(uint64_t b) const { uint16_t v; memcpy(&v, blk(b), 2); return v; }
  uint64_t occupieds(uint64_t b) const { uint64_t v; memcpy(&v, blk(b) + 2, 8); return v; }
  uint64_t runends(uint64_t b) const { uint64_t v; memcpy(&v, blk(b) + 10, 8); return v; }
  void set_offset(uint64_t b, uint16_t v) { memcpy(blk(b), &v, 2); }
  void or_occupied(uint64_t i) { uint64_t v = occupieds(i / 64) | (1ULL << (i % 64)); memcpy(blk(i / 64) + 2, &v, 8); }
  void or_runend(uint64_t i) { uint64_t v = runends(i / 64) | (1ULL << (i % 64)); memcpy(blk(i / 64) + 10, &v, 8); }
  bool is_occupied(uint64_t i) const { return (occupieds(i / 64) >> (i % 64)) & 1; }
  bool is_runend(uint64_t i) const { return (runends(i / 64) >> (i % 64)) & 1; }
  uint64_t get_slot(uint64_t i) const {  // gqf.c:532-544
    const uint64_t bit = (i % 64) * md.bits_per_slot;
    uint64_t w;
    memcpy(&w, blk(i / 64) + 18 + bit / 8, 8);
    return (w >> (bit % 8)) & bitmask(md.bits_per_slot);
  }
  void set_slot(uint64_t i, uint64_t v) {  // gqf.c:546-568
    const uint64_t bit = (i % 64) * md.bits_per_slot;
    uint8_t* p = blk(i / 64) + 18 + bit / 8;
    uint64_t w;
    memcpy(&w, p, 8);
    const uint64_t sh = bit % 8, m = bitmask(md.bits_per_slot) << sh;
    w = (w & ~m) | ((v << sh) & m);
    memcpy(p, &w, 8);
  }

  // ---- query (gqf.c:2081-2118) ----
  static uint64_t bitrank(uint64_t val, uint64_t pos) { return __builtin_popcountll(val & ((2ULL << pos) - 1)); }
  static uint64_t bitselect(uint64_t val, int rank) {
    for (int i = 0; i < rank; ++i) val &= val - 1;
    return val ? (uint64_t)__builtin_ctzll(val) : 64;
  }
  static uint64_t bitselectv(uint64_t val, uint64_t ignore, int rank) { return bitselect(val & ~bitmask(ignore % 64), rank); }
  static uint64_t popcntv(uint64_t val, uint64_t ignore) {
    return (ignore % 64) ? (uint64_t)__builtin_popcountll(val & ~bitmask(ignore % 64)) : (uint64_t)__builtin_popcountll(val);
  }
  uint64_t run_end(uint64_t bucket) const {  // gqf.c:583-632
    const uint64_t bb = bucket / 64, bo = bucket % 64, boff = offset(bb);
    const uint64_t rank = bitrank(occupieds(bb), bo);
    if (rank == 0) return boff <= bo ? bucket : 64 * bb + boff - 1;
    uint64_t rb = bb + boff / 64, ignore = boff % 64, rrank = rank - 1;
    uint64_t ro = bitselectv(runends(rb), ignore, (int)rrank);
    if (ro == 64) {
      if (boff == 0 && rank == 0) return bucket;
      do {
        rrank -= popcntv(runends(rb), ignore);
        rb++;
        ignore = 0;
        if (rb >= md.nblocks) throw std::runtime_error("cqf: run end not found");
        ro = bitselectv(runends(rb), ignore, (int)rrank);
      } while (ro == 64);
    }
    const uint64_t idx = 64 * rb + ro;
    return idx < bucket ? bucket : idx;
  }
  uint64_t decode_counter(uint64_t index, uint64_t* remainder, uint64_t* count) const {  // gqf.c:1112-1182
    const uint64_t rem = get_slot(index);
    *remainder = rem;
    if (is_runend(index)) { *count = 1; return index; }
    uint64_t digit = get_slot(index + 1);
    if (is_runend(index + 1)) { *count = digit == rem ? 2 : 1; return index + (digit == rem ? 1 : 0); }
    if (rem > 0 && digit >= rem) { *count = digit == rem ? 2 : 1; return index + (digit == rem ? 1 : 0); }
    if (rem > 0 && digit == 0 && get_slot(index + 2) == rem) { *count = 3; return index + 2; }
    if (rem == 0 && digit == 0) {
      if (get_slot(index + 2) == 0) { *count = 3; return index + 2; }
      *count = 2;
      return index + 1;
    }
    uint64_t cnt = 0;
    const uint64_t base = (1ULL << md.bits_per_slot) - (rem ? 2 : 1);
    uint64_t end = index + 1;
    while (digit != rem && !is_runend(end)) {
      if (digit > rem) digit--;
      if (digit && rem) digit--;
      cnt = cnt * base + digit;
      end++;
      digit = get_slot(end);
    }
    if (rem) { *count = cnt + 3; return end; }
    if (is_runend(end) || get_slot(end + 1) != 0) { *count = 1; return index; }
    *count = cnt + 4;
    return end + 1;
  }
  // returns the count of `key` (0 = absent) and its value bits
  uint64_t query(uint64_t key, uint64_t* value) const {
    const uint64_t hash = hash_64(key, bitmask(md.key_bits));
    const uint64_t hrem = hash & bitmask(md.key_remainder_bits);
    const uint64_t bucket = hash >> md.key_remainder_bits;
    if (!is_occupied(bucket)) return 0;
    uint64_t start = bucket == 0 ? 0 : run_end(bucket - 1) + 1;
    if (start < bucket) start = bucket;
    uint64_t cur_rem, cur_cnt, cur_end;
    do {
      cur_end = decode_counter(start, &cur_rem, &cur_cnt);
      *value = cur_rem & bitmask(md.value_bits);
      if ((cur_rem >> md.value_bits) == hrem) return cur_cnt;
      start = cur_end + 1;
    } while (!is_runend(cur_end));
    return 0;
  }

  void load(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    in.read((char*)&md, sizeof(md));
    if (!in || md.magic_endian_number != kMagic) throw std::runtime_error("not a CQF file: " + path);
    block_bytes = 18 + 8 * md.bits_per_slot;
    if (md.total_size_in_bytes != md.nblocks * block_bytes) throw std::runtime_error("CQF header is inconsistent: " + path);
    blocks.assign(md.total_size_in_bytes + 8, 0);
    in.read((char*)blocks.data(), md.total_size_in_bytes);
    if (!in) throw std::runtime_error("truncated CQF file: " + path);
  }
  void save(const std::string& path) const {
    std::ofstream out(path, std::ios::binary | std::ios::trunc);
    if (!out) throw std::runtime_error("cannot write " + path);
    out.write((const char*)&md, sizeof(md));
    out.write((const char*)blocks.data(), md.total_size_in_bytes);
    if (!out) throw std::runtime_error("write failed: " + path);
  }
};

// slots of one (remainder, count) entry, in memory order (encode_counter, gqf.c:1052-1108)
inline void encode_counter(uint64_t bits_per_slot, uint64_t remainder, uint64_t counter, std::vector<uint64_t>& out) {
  out.clear();
  if (counter == 0) return;
  std::vector<uint64_t> rev;  // built back to front like the reference (*--p = ...)
  rev.push_back(remainder);
  if (counter == 1) { out = rev; return; }
  if (counter == 2) { out = {remainder, remainder}; return; }
  if (counter == 3 && remainder == 0) { out = {0, 0, 0}; return; }
  if (counter == 3) { out = {remainder, 0, remainder}; return; }
  uint64_t base = (1ULL << bits_per_slot) - 1;
  if (remainder == 0) rev.push_back(0);
  else base--;
  counter -= remainder ? 3 : 4;
  uint64_t digit;
  do {
    digit = counter % base;
    digit++;
    if (remainder && digit >= remainder) digit++;
    rev.push_back(digit);
    counter /= base;
  } while (counter);
  if (remainder && digit >= remainder) rev.push_back(0);
  rev.push_back(remainder);
  out.assign(rev.rbegin(), rev.rend());
}

struct Entry { uint64_t key, value, count; };

// Build the filter for a set of distinct keys.  Starts at the reference's default size and doubles
// while the reference's auto-resize rule (gqf.c:1916-1926: 75 % occupied slots) would have fired.
inline void build(const std::vector<Entry>& entries, Filter& f) {
  uint64_t nslots = kDefaultSlots;
  while (true) {
    f.init(nslots);
    const uint64_t bps = f.md.bits_per_slot;
    struct Item { uint64_t bucket, rem, count; };
    std::vector<Item> items;
    items.reserve(entries.size());
    for (const auto& e : entries) {
      if (e.count == 0) continue;
      const uint64_t h = (hash_64(e.key, bitmask(kKeyBits)) << kValueBits) | (e.value & bitmask(kValueBits));
      items.push_back(Item{h >> bps, h & bitmask(bps), e.count});
    }
    std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) {
      return a.bucket != b.bucket ? a.bucket < b.bucket : a.rem < b.rem;
    });
    uint64_t cursor = 0, nelts = 0;
    bool overflow = false;
    std::vector<uint64_t> enc;
    std::vector<std::pair<uint64_t, uint64_t>> run_ends;  // (bucket, end exclusive)
    size_t i = 0;
    while (i < items.size() && !overflow) {
      const uint64_t bucket = items[i].bucket;
      uint64_t pos = std::max(cursor, bucket);
      size_t j = i;
      while (j < items.size() && items[j].bucket == bucket) {
        encode_counter(bps, items[j].rem, items[j].count, enc);
        if (pos + enc.size() > f.md.xnslots) { overflow = true; break; }
        for (uint64_t s : enc) f.set_slot(pos++, s);
        nelts += items[j].count;
        ++j;
      }
      if (overflow) break;
      f.or_occupied(bucket);
      f.or_runend(pos - 1);
      run_ends.emplace_back(bucket, pos);
      cursor = pos;
      i = j;
    }
    uint64_t used = 0;
    {
      uint64_t prev_end = 0;
      for (auto& re : run_ends) { used += re.second - std::max(prev_end, re.first); prev_end = re.second; }
    }
    if (overflow || (double)used >= (double)nslots * 0.75) { nslots *= 2; continue; }
    // block offsets: slots of block b taken by runs of buckets before the block
    size_t g = 0;
    uint64_t end_before = 0;
    for (uint64_t b = 0; b < f.md.nblocks; ++b) {
      while (g < run_ends.size() && run_ends[g].first < 64 * b) { end_before = run_ends[g].second; ++g; }
      const uint64_t off = end_before > 64 * b ? end_before - 64 * b : 0;
      if (off > 0xFFFF) throw std::runtime_error("cqf: block offset overflows 16 bits");
      f.set_offset(b, (uint16_t)off);
    }
    f.md.nelts = nelts;
    f.md.ndistinct_elts = items.size();
    f.md.noccupied_slots = used;
    return;
  }
}

}  // namespace cqf
}  // namespace vsamd
