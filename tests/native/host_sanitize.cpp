// Test-only: the host side of the product (constructor, index-directory codecs, image builder, synthetic
// generator, dot graph) under AddressSanitizer + UBSan.  usage: host_sanitize <fasta> <vcf> <scratch dir>
#include <cstdio>
#include <string>
#include <sys/stat.h>
#include "../../variantstore_amd/csrc/host/host_graph.hpp"
#include "../../variantstore_amd/csrc/host/builder.hpp"
#include "../../variantstore_amd/csrc/host/vcf.hpp"
#include "../../variantstore_amd/csrc/host/synth.hpp"
#include "../../variantstore_amd/csrc/host/device_image.hpp"
#include "../../variantstore_amd/csrc/host/index_files.hpp"
#include "../../variantstore_amd/csrc/host/dot_graph.hpp"
using namespace vsamd;

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  HostGraph g;
  uint64_t nk = 0, ne = 0, sl = 0;
  construct_from_files(argv[1], argv[2], g, &nk, &ne, &sl);
  const std::string dir = argv[3];
  mkdir(dir.c_str(), 0755);
  save_index_dir(g, dir);
  HostGraph h;
  load_index_dir(dir, h);
  HostImage im;
  build_host_image(h, im);
  if (h.off.size() != g.off.size() || h.car_flags != g.car_flags || h.seq != g.seq || h.idx_pos != g.idx_pos) { printf("FAIL round trip\n"); return 1; }
  SynthParams p;
  p.ref_length = 300000; p.num_variants = 3000; p.num_samples = 300; p.frac_ins = 0.05; p.frac_del = 0.05; p.frac_multi = 0.02;
  p.sample_coordinates = true;
  HostGraph s;
  construct_synthetic(p, s);
  HostImage im2;
  build_host_image(s, im2);
  const std::string dot = dot_text(s, host_find(s, 5000), 3);
  printf("ok %zu %llu %zu %llu %zu\n", h.off.size(), (unsigned long long)im.P, s.off.size(), (unsigned long long)im2.P, dot.size());
  return 0;
}
