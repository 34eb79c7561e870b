// Test infrastructure: WINDOWS of a synthetic cohort as FASTA + VCF files.
//
// The full-size indexes of BASELINE.json (5 M sites x 2504 samples, 20 M sites x 10,000 samples) are too big for the CPU
// oracle, so the oracle sees them through windows: the generator (variantstore_amd/csrc/host/synth.hpp -- the records a
// VCF would hold) is run once over the whole cohort, and the records whose reference span lies inside a window
// [lo, hi] are written as a VCF relative to the window (POS - lo + 1) next to the window's slice of the reference.  A test
// builds the small index from those FILES through the product's VCF path (`from_vcf`), runs the oracle on it and compares
// the rows, shifted back by lo - 1, with what the GPU answers on the FULL index (tests/test_gpu_full_size.py,
// tests/test_zz_gpu_full_size_tcga.py).
//
//   synth_windows <ref_length> <num_variants> <num_samples> <seed> <first_pos> <frac_ins> <frac_del> <frac_multi> <max_indel>
//                 <af_exponent> <max_af> <outdir>   < windows ("lo hi" per line, 1-based, inclusive)
// writes <outdir>/w<k>.fa and <outdir>/w<k>.vcf for the k-th line.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../variantstore_amd/csrc/host/synth.hpp"

int main(int argc, char** argv) {
  if (argc != 13) { fprintf(stderr, "usage: see the header of synth_windows.cpp\n"); return 2; }
  vsamd::SynthParams p;
  p.ref_length = strtoull(argv[1], 0, 10); p.num_variants = strtoull(argv[2], 0, 10); p.num_samples = (uint32_t)strtoul(argv[3], 0, 10);
  p.seed = strtoull(argv[4], 0, 10); p.first_pos = strtoull(argv[5], 0, 10);
  p.frac_ins = atof(argv[6]); p.frac_del = atof(argv[7]); p.frac_multi = atof(argv[8]); p.max_indel = (uint32_t)strtoul(argv[9], 0, 10);
  p.af_exponent = atof(argv[10]); p.max_af = atof(argv[11]);
  const std::string outdir = argv[12];
  struct Win { uint64_t lo, hi; FILE* vf; uint64_t n; };
  std::vector<Win> wins;
  uint64_t lo, hi;
  while (scanf("%" SCNu64 " %" SCNu64, &lo, &hi) == 2) {
    if (lo < 1 || hi < lo || hi > p.ref_length) { fprintf(stderr, "window %" PRIu64 " %" PRIu64 " outside the reference\n", lo, hi); return 2; }
    wins.push_back(Win{lo, hi, nullptr, 0});
  }
  vsamd::SynthSource src(p);
  const std::string& ref = src.reference();
  for (size_t k = 0; k < wins.size(); ++k) {
    const std::string base = outdir + "/w" + std::to_string(k);
    FILE* ff = fopen((base + ".fa").c_str(), "w");
    if (!ff) { perror("fasta"); return 1; }
    fprintf(ff, ">syn\n");
    for (uint64_t i = wins[k].lo - 1; i < wins[k].hi; i += 80)
      fprintf(ff, "%.*s\n", (int)std::min<uint64_t>(80, wins[k].hi - i), ref.data() + i);
    fclose(ff);
    FILE* vf = fopen((base + ".vcf").c_str(), "w");
    if (!vf) { perror("vcf"); return 1; }
    fprintf(vf, "##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
    fprintf(vf, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");
    for (auto& n : src.sample_names()) fprintf(vf, "\t%s", n.c_str());
    fprintf(vf, "\n");
    wins[k].vf = vf;
  }
  vsamd::SynthRecord r;
  std::string line;
  while (src.next(r)) {
    const uint64_t end = r.pos + r.ref.size() - 1;   // the record's reference span [pos, end]
    bool any = false;
    for (auto& w : wins) if (r.pos > w.lo && end <= w.hi) { any = true; break; }   // (POS 1 of a window is left alone: a VCF record needs its anchor base)
    if (!any) continue;
    // the genotype columns once per record: "a|b" with 1 for a carried haplotype (any non-zero allele number makes a
    // carrier of every ALT of the record: variant_graph.h:666-691, as construct_synthetic feeds the builder)
    line.clear();
    size_t ci = 0;
    for (uint32_t s = 1; s <= p.num_samples; ++s) {
      if (ci < r.carriers.size() && r.carriers[ci].sample_id == s) {
        line += r.carriers[ci].gt1 ? "\t1|" : "\t0|";
        line += r.carriers[ci].gt2 ? "1" : "0";
        ++ci;
      } else line += "\t0|0";
    }
    std::string alts = r.alts[0];
    for (size_t i = 1; i < r.alts.size(); ++i) alts += "," + r.alts[i];
    for (auto& w : wins) {
      if (!(r.pos > w.lo && end <= w.hi)) continue;
      fprintf(w.vf, "syn\t%" PRIu64 "\t.\t%s\t%s\t99\t.\t.\tGT", r.pos - w.lo + 1, r.ref.c_str(), alts.c_str());
      fwrite(line.data(), 1, line.size(), w.vf);
      fputc('\n', w.vf);
      w.n++;
    }
  }
  for (size_t k = 0; k < wins.size(); ++k) {
    fclose(wins[k].vf);
    printf("%zu %" PRIu64 " %" PRIu64 " %" PRIu64 "\n", k, wins[k].lo, wins[k].hi, wins[k].n);
  }
  return 0;
}
