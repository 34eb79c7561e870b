// ref_order_set.hpp -- iteration-order model of the reference's neighbour sets.
//
// The reference keeps every multi-neighbour adjacency list in a
// std::unordered_set<uint32_t> (reference include/graph.h:44,134) and all of
// its traversals iterate those sets (graph.h:265-280, 402-405), so the order in
// which alleles are reported -- and which duplicate wins the de-duplication in
// query.h:397-414 -- is the iteration order of a libstdc++ hash set.  The
// product must not depend on whichever libstdc++ happens to be installed on the
// box it runs on, so the rule is restated here (GCC 11 libstdc++, identity hash
// for uint32_t, max_load_factor 1.0):
//
//   * bucket = value % bucket_count; bucket_count starts at 1
//   * before inserting, if size+1 > next_resize: the table grows to
//     next_bkt(max(size+1 (11 on the very first insert) + 1, 2*bucket_count))
//     and every node is re-linked, walking the old list front to back
//   * a node is linked at the FRONT OF ITS BUCKET'S RUN when the bucket is
//     non-empty, otherwise at the FRONT OF THE WHOLE LIST
//   * copies (copy-construct / copy-assign) preserve order and bucket count
//
// tests/test_ref_order_set.py drives this model against the real
// std::unordered_set of the build toolchain on random operation sequences.
#pragma once
#include <cstdint>
#include <vector>
#include <algorithm>

namespace vsamd {

class RefOrderSet {
 public:
  using const_iterator = std::vector<uint32_t>::const_iterator;

  size_t size() const { return ord_.size(); }
  bool empty() const { return ord_.empty(); }
  const_iterator begin() const { return ord_.begin(); }
  const_iterator end() const { return ord_.end(); }
  const std::vector<uint32_t>& order() const { return ord_; }
  uint64_t bucket_count() const { return nb_; }

  bool contains(uint32_t x) const {
    return std::find(ord_.begin(), ord_.end(), x) != ord_.end();
  }

  // returns true when x was not present (std::unordered_set::insert().second)
  bool insert(uint32_t x) {
    if (contains(x)) return false;
    const uint64_t n_elt = ord_.size();
    if (n_elt + 1 > next_resize_) {
      uint64_t min_bkts = std::max<uint64_t>(n_elt + 1, next_resize_ ? 0 : 11);
      if (min_bkts >= nb_) {
        rehash(next_bkt(std::max<uint64_t>(min_bkts + 1, nb_ * 2)));
      } else {
        next_resize_ = nb_;
      }
    }
    link(ord_, x, nb_);
    return true;
  }

  // std::unordered_set::erase(begin())
  void erase_begin() {
    if (!ord_.empty()) ord_.erase(ord_.begin());
  }

  // std::unordered_set::erase(key): unlinks the node, order of the others is
  // unchanged.
  bool erase(uint32_t x) {
    auto it = std::find(ord_.begin(), ord_.end(), x);
    if (it == ord_.end()) return false;
    ord_.erase(it);
    return true;
  }

 private:
  static void link(std::vector<uint32_t>& ord, uint32_t x, uint64_t nb) {
    const uint64_t b = x % nb;
    for (size_t i = 0; i < ord.size(); ++i) {
      if (ord[i] % nb == b) {  // bucket non-empty: front of the bucket's run
        ord.insert(ord.begin() + i, x);
        return;
      }
    }
    ord.insert(ord.begin(), x);  // empty bucket: front of the list
  }

  void rehash(uint64_t n) {
    std::vector<uint32_t> fresh;
    fresh.reserve(ord_.size() + 1);
    for (uint32_t v : ord_) link(fresh, v, n);
    ord_.swap(fresh);
    nb_ = n;
  }

  // std::__detail::_Prime_rehash_policy::_M_next_bkt (GCC 11): a 14-entry fast
  // table, then the first prime >= n of libstdc++'s __prime_list.
  uint64_t next_bkt(uint64_t n) {
    static const unsigned char fast_bkt[] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
    if (n < sizeof(fast_bkt)) {
      if (n == 0) return 1;
      next_resize_ = fast_bkt[n];
      return fast_bkt[n];
    }
    // Prefix of libstdc++'s __prime_list (checked against the toolchain's own
    // table by tests/test_ref_order_set.py).  65521 neighbours of one vertex is
    // far beyond anything a variation graph produces.
    static const uint32_t primes[] = {
        2,     3,     5,     7,     11,    13,    17,    19,    23,    29,    31,    37,   41,
        43,    47,    53,    59,    61,    67,    71,    73,    79,    83,    89,    97,   103,
        109,   113,   127,   137,   139,   149,   157,   167,   179,   193,   199,   211,  227,
        241,   257,   277,   293,   313,   337,   359,   383,   409,   439,   467,   503,  541,
        577,   619,   661,   709,   761,   823,   887,   953,   1031,  1109,  1193,  1289, 1381,
        1493,  1613,  1741,  1879,  2029,  2179,  2357,  2549,  2753,  2971,  3209,  3469, 3739,
        4027,  4349,  4703,  5087,  5503,  5953,  6427,  6949,  7517,  8123,  8783,  9497, 10273,
        11113, 12011, 12983, 14033, 15173, 16411, 17749, 19183, 20753, 22447, 24281, 26267,
        28411, 30727, 33223, 35933, 38873, 42043, 45481, 49201, 53201, 57557, 62233, 67307};
    const size_t np = sizeof(primes) / sizeof(primes[0]);
    const uint32_t* p = std::lower_bound(primes, primes + np, (uint32_t)std::min<uint64_t>(n, primes[np - 1]));
    next_resize_ = *p;
    return *p;
  }

  std::vector<uint32_t> ord_;
  uint64_t nb_ = 1;
  uint64_t next_resize_ = 0;
};

}  // namespace vsamd
