// Drives vsamd::RefOrderSet against the toolchain's real std::unordered_set on
// random insert / erase(begin) / copy sequences and compares iteration order.
// Also checks the embedded prime-table prefix against libstdc++'s __prime_list.
#include <unordered_set>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../variantstore_amd/csrc/host/ref_order_set.hpp"

namespace std { namespace __detail { extern const unsigned long __prime_list[]; } }

static bool same(const std::unordered_set<uint32_t>& a, const vsamd::RefOrderSet& b) {
  if (a.size() != b.size()) return false;
  auto it = b.begin();
  for (uint32_t v : a) { if (*it != v) return false; ++it; }
  return a.bucket_count() == b.bucket_count() || a.size() == 0;
}

int main(int argc, char** argv) {
  uint64_t seed = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  int rounds = argc > 2 ? atoi(argv[2]) : 20000;
  std::mt19937_64 rng(seed);
  // prime table prefix
  {
    vsamd::RefOrderSet s; // grow a set through many rehashes and compare bucket counts
    std::unordered_set<uint32_t> r;
    for (uint32_t i = 0; i < 70000; ++i) {
      uint32_t v = (uint32_t)rng();
      r.insert(v);
      // RefOrderSet is O(n) per insert; only follow bucket counts cheaply for big sizes
      if (i < 3000) { s.insert(v); if (!same(r, s)) { printf("FAIL grow at %u\n", i); return 1; } }
    }
  }
  for (int round = 0; round < rounds; ++round) {
    std::unordered_set<uint32_t> r; vsamd::RefOrderSet s;
    int nops = 1 + rng() % 40;
    uint32_t range = (round % 3 == 0) ? 64 : (round % 3 == 1 ? 100000 : 0xffffffffu);
    for (int k = 0; k < nops; ++k) {
      int op = rng() % 10;
      if (op < 7) {
        uint32_t v = (uint32_t)(rng() % range);
        bool a = r.insert(v).second, b = s.insert(v);
        if (a != b) { printf("FAIL insert ret\n"); return 1; }
      } else if (op < 8) {
        if (!r.empty()) { r.erase(r.begin()); s.erase_begin(); }
      } else if (op < 9) {
        std::unordered_set<uint32_t> c(r); vsamd::RefOrderSet d(s);  // copy-construct
        r = c; s = d;                                                // copy-assign back
      } else {
        std::unordered_set<uint32_t> c; c = r; r = c;               // assign into empty
      }
      if (!same(r, s)) {
        printf("FAIL round %d op %d: real:", round, k);
        for (auto v : r) printf(" %u", v);
        printf(" (nb %zu) model:", r.bucket_count());
        for (auto v : s) printf(" %u", v);
        printf(" (nb %lu)\n", (unsigned long)s.bucket_count());
        return 1;
      }
    }
  }
  // survey probes
  {
    vsamd::RefOrderSet s; for (uint32_t v : {1u, 14u, 27u, 2u}) s.insert(v);
    const uint32_t exp[] = {2, 27, 14, 1}; int i = 0;
    for (auto v : s) if (v != exp[i++]) { printf("FAIL probe\n"); return 1; }
  }
  printf("OK\n");
  return 0;
}
