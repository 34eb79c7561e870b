// Test-only: the static structures device_image.hpp derives for the walking query types, against brute force on a
// synthetic cohort with indels: the ancestor labels of the backward search's chains (rk_anc), the slot -> rank table,
// and the break bits of the sequence queries.  usage: image_labels_check [<fasta> <vcf>]  -> "ok <ranks> <slots> <tested> <breaks>"
// (with arguments: the graph the constructor builds from the files, whose edges lie in the reference's hash-set order)
#include <cstdio>
#include <set>
#include <vector>
#include "../../variantstore_amd/csrc/host/host_graph.hpp"
#include "../../variantstore_amd/csrc/host/builder.hpp"
#include "../../variantstore_amd/csrc/host/vcf.hpp"
#include "../../variantstore_amd/csrc/host/synth.hpp"
#include "../../variantstore_amd/csrc/host/device_image.hpp"
using namespace vsamd;

int main(int argc, char** argv) {
  HostGraph g;
  if (argc >= 3) {
    uint64_t nk = 0, ne = 0, sl = 0;
    construct_from_files(argv[1], argv[2], g, &nk, &ne, &sl);
  } else {
    SynthParams p;
    p.ref_length = 200000; p.num_variants = 6000; p.num_samples = 60; p.frac_ins = 0.1; p.frac_del = 0.1; p.frac_multi = 0.05;
    p.max_indel = 6; p.sample_coordinates = true;
    construct_synthetic(p, g);
  }
  HostImage im;
  build_host_image(g, im);
  const uint64_t R = im.R, P = im.P;
  if (im.rk_anc.size() != (R + 1) * 2 || im.slot_rank.size() != P || im.seq_breaks.size() != (P + 63) / 64 + 1) { printf("FAIL sizes\n"); return 1; }
  // slot -> rank
  for (uint64_t r = 0; r < R; ++r)
    for (uint64_t k = im.rank_to_slot[r]; k < im.rank_to_slot[r + 1] && k < P; ++k)
      if (im.slot_rank[k] != r) { printf("FAIL slot_rank %llu\n", (unsigned long long)k); return 1; }
  // chains: the reference's scan visits rank, rank - deg(previous(rank)), ... while rank >= 2
  auto deg_of_rank = [&](uint64_t rank) { const uint32_t d = im.rk_back[2 * (rank - 1) + 1]; return (uint64_t)(d ? d : 1); };
  auto visits = [&](uint64_t pr, uint64_t r0) {
    const uint32_t tin = im.rk_anc[2 * (pr - 1)], size = im.rk_anc[2 * (pr - 1) + 1], tin0 = im.rk_anc[2 * (r0 - 1)];
    return tin <= tin0 && tin0 - tin < size;
  };
  uint64_t tested = 0;
  for (uint64_t r0 = 2; r0 <= R; r0 += 37) {
    std::set<uint64_t> chain;
    for (uint64_t pr = r0; pr >= 2;) { chain.insert(pr); const uint64_t d = deg_of_rank(pr); pr = pr > d ? pr - d : 0; }
    const uint64_t lo = r0 > 400 ? r0 - 400 : 2;
    for (uint64_t pr = lo; pr <= r0; ++pr, ++tested)
      if (visits(pr, r0) != (chain.count(pr) != 0)) { printf("FAIL labels r0 %llu rank %llu\n", (unsigned long long)r0, (unsigned long long)pr); return 1; }
    if (r0 + 1 <= R && visits(r0 + 1, r0)) { printf("FAIL labels: a rank above the start\n"); return 1; }
  }
  // break bits: a clear bit promises that the slot's successor continues it in every respect a merged run relies on
  uint64_t breaks = 0;
  for (uint64_t k = 0; k < P; ++k) {
    const bool brk = (im.seq_breaks[k >> 6] >> (k & 63)) & 1;
    breaks += brk;
    if (brk) continue;
    if (k + 1 >= P) { printf("FAIL last slot without a break\n"); return 1; }
    const uint32_t v = im.rp_vid[k], succ = im.rp_vid[k + 1];
    uint32_t first_ref = VS_NONE, min_ref = VS_NONE, min_idx = 0xFFFFFFFFu;
    for (uint32_t e = im.row_ptr[v]; e < im.row_ptr[v + 1]; ++e) {
      const uint32_t n = im.col[e], nr = im.v_ridx[n];
      if (!nr) continue;
      if (first_ref == VS_NONE) first_ref = n;
      if (nr < min_idx) { min_idx = nr; min_ref = n; }
    }
    if (first_ref != succ || min_ref != succ || im.v_off[succ] != im.v_off[v] + im.v_len[v] || im.v_ridx[succ] != im.v_ridx[v] + im.v_len[v]) {
      printf("FAIL break bit missing at slot %llu\n", (unsigned long long)k);
      return 1;
    }
  }
  printf("ok %llu %llu %llu %llu\n", (unsigned long long)R, (unsigned long long)P, (unsigned long long)tested, (unsigned long long)breaks);
  return 0;
}
