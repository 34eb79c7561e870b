// synth.hpp -- deterministic synthetic cohorts (BASELINE.json configs #2-#4).
//
// Generates the records a VCF would hold -- position, REF, ALT list and one
// genotype per sample -- and feeds them through the same GraphBuilder that the
// FASTA+VCF path uses (the reference's add_vcfs -> add_mutation,
// include/variant_graph.h:619-733,1509-1881), so a synthetic index is exactly
// what `variantstore construct` would have produced from the equivalent files.
// Shapes follow SURVEY.md §8(d): random ACGT reference, stratified-uniform
// variant positions, AF = min(0.5, 10^(-a*U)) per variant, every haplotype an
// independent Bernoulli(AF) draw, at least one carrier, all genotypes phased.
// Sample names are zero padded (S00001...) so name order equals column order.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "builder.hpp"

namespace vsamd {

struct SynthParams {
  uint64_t ref_length = 1000000;
  uint64_t num_variants = 10000;
  uint32_t num_samples = 100;
  uint64_t seed = 1;
  uint64_t first_pos = 1000;
  double frac_ins = 0.0, frac_del = 0.0, frac_multi = 0.0;
  uint32_t max_indel = 6;
  double af_exponent = 3.0;
  bool sample_coordinates = false;
  double max_af = 0.5;
};

struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
  uint64_t below(uint64_t n) { return (uint64_t)(uniform() * (double)n); }
};

struct SynthRecord {
  uint64_t pos;
  std::string ref;
  std::vector<std::string> alts;
  std::vector<SampleGT> carriers;  // ascending sample id == name order
};

class SynthSource {
 public:
  explicit SynthSource(const SynthParams& p) : p_(p), rng_(p.seed * 0x2545F4914F6CDD1DULL + 12345) {
    static const char B[4] = {'A', 'C', 'G', 'T'};
    ref_.resize(p.ref_length);
    SplitMix64 r(p.seed ^ 0xA5A5A5A55A5A5A5AULL);
    for (uint64_t i = 0; i < p.ref_length; i += 32) {
      uint64_t x = r.next();
      for (uint64_t j = 0; j < 32 && i + j < p.ref_length; ++j, x >>= 2) ref_[i + j] = B[x & 3];
    }
    uint64_t lo = p.first_pos < 2 ? 2 : p.first_pos;
    uint64_t hi = p.ref_length > (uint64_t)p.max_indel + 8 ? p.ref_length - p.max_indel - 8 : lo + 1;
    span_ = hi > lo ? hi - lo : 1;
    lo_ = lo;
    for (uint32_t i = 0; i < p.num_samples; ++i) {
      char buf[32];
      snprintf(buf, sizeof(buf), "S%05u", i + 1);
      names_.push_back(buf);
    }
  }
  const std::string& reference() const { return ref_; }
  const std::vector<std::string>& sample_names() const { return names_; }

  // record i of num_variants; positions are stratified so they are strictly increasing
  bool next(SynthRecord& rec) {
    if (i_ >= p_.num_variants) return false;
    const double gap = (double)span_ / (double)p_.num_variants;
    uint64_t a = lo_ + (uint64_t)(i_ * gap), b = lo_ + (uint64_t)((i_ + 1) * gap);
    if (b <= a) b = a + 1;
    uint64_t pos = a + rng_.below(b - a);
    if (pos <= last_pos_) pos = last_pos_ + 1;
    last_pos_ = pos;
    ++i_;
    rec.pos = pos;
    rec.alts.clear();
    const char r0 = ref_[pos - 1];
    const double t = rng_.uniform();
    auto rand_base = [&]() { return "ACGT"[rng_.next() & 3]; };
    if (t < p_.frac_ins) {
      uint32_t k = 1 + (uint32_t)rng_.below(p_.max_indel);
      std::string alt(1, r0);
      for (uint32_t j = 0; j < k; ++j) alt += rand_base();
      rec.ref.assign(1, r0);
      rec.alts.push_back(alt);
    } else if (t < p_.frac_ins + p_.frac_del) {
      uint32_t k = 1 + (uint32_t)rng_.below(p_.max_indel);
      rec.ref = ref_.substr(pos - 1, k + 1);
      rec.alts.push_back(std::string(1, r0));
    } else {
      rec.ref.assign(1, r0);
      char a1;
      do a1 = rand_base(); while (a1 == r0);
      rec.alts.push_back(std::string(1, a1));
      if (rng_.uniform() < p_.frac_multi) {
        char a2;
        do a2 = rand_base(); while (a2 == r0 || a2 == a1);
        rec.alts.push_back(std::string(1, a2));
      }
    }
    // genotypes: geometric skipping over the 2N haplotypes
    double af = std::pow(10.0, -p_.af_exponent * rng_.uniform());
    if (af > p_.max_af) af = p_.max_af;
    rec.carriers.clear();
    const uint64_t H = 2ull * p_.num_samples;
    const double lq = std::log1p(-af);
    uint64_t h = 0;
    while (true) {
      double u = rng_.uniform();
      if (u <= 0) u = 1e-300;
      uint64_t skip = (uint64_t)std::floor(std::log(u) / lq);
      h += skip;
      if (h >= H) break;
      add_hap(rec, h);
      ++h;
    }
    if (rec.carriers.empty()) add_hap(rec, rng_.below(H));
    return true;
  }

 private:
  static void add_hap(SynthRecord& rec, uint64_t h) {
    uint32_t sid = (uint32_t)(h >> 1) + 1;
    if (rec.carriers.empty() || rec.carriers.back().sample_id != sid)
      rec.carriers.push_back(SampleGT{sid, true, false, false});
    if (h & 1) rec.carriers.back().gt2 = true;
    else rec.carriers.back().gt1 = true;
  }
  SynthParams p_;
  SplitMix64 rng_;
  std::string ref_;
  std::vector<std::string> names_;
  uint64_t span_ = 1, lo_ = 2, i_ = 0, last_pos_ = 0;
};

struct SynthStats {
  uint64_t num_vars = 0, num_mutations = 0, num_mutations_samples = 0;
  bool use_bit_vector = false;
};

// Construct the index of a synthetic cohort.  When vcf_out / fasta_out are given
// the same records are also written as files (small configs, parity tests).
inline SynthStats construct_synthetic(const SynthParams& p, HostGraph& out, uint64_t* n_keys = nullptr,
                                      uint64_t* n_edges = nullptr, uint64_t* seq_len = nullptr,
                                      const char* fasta_out = nullptr, const char* vcf_out = nullptr) {
  SynthStats st;
  // use_bit_vector_encoding(): carrier density over the first 99 records (variant_graph.h:568-617)
  {
    SynthSource probe(p);
    SynthRecord r;
    float density = 0;
    for (uint32_t cnt = 1; cnt < 100 && probe.next(r); ++cnt) {
      float cur = r.carriers.size() / (float)p.num_samples;
      density = density > cur ? density : cur;
    }
    st.use_bit_vector = density > 0.05f;
  }
  SynthSource src(p);
  FILE* vf = nullptr;
  if (fasta_out) {
    FILE* ff = fopen(fasta_out, "w");
    if (!ff) throw std::runtime_error("cannot write fasta");
    fprintf(ff, ">syn\n");
    const std::string& ref = src.reference();
    for (uint64_t i = 0; i < ref.size(); i += 80) fprintf(ff, "%.*s\n", (int)std::min<uint64_t>(80, ref.size() - i), ref.data() + i);
    fclose(ff);
  }
  if (vcf_out) {
    vf = fopen(vcf_out, "w");
    if (!vf) throw std::runtime_error("cannot write vcf");
    fprintf(vf, "##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
    fprintf(vf, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");
    for (auto& n : src.sample_names()) fprintf(vf, "\t%s", n.c_str());
    fprintf(vf, "\n");
  }
  GraphBuilder b("syn", src.reference(), src.sample_names(), st.use_bit_vector);
  SynthRecord r;
  SplitMix64 allele_rng(p.seed + 99);
  std::vector<char> row;
  while (src.next(r)) {
    st.num_vars++;
    if (vf) {
      std::string alts = r.alts[0];
      for (size_t i = 1; i < r.alts.size(); ++i) alts += "," + r.alts[i];
      fprintf(vf, "syn\t%lu\t.\t%s\t%s\t99\t.\t.\tGT", (unsigned long)r.pos, r.ref.c_str(), alts.c_str());
      size_t ci = 0;
      for (uint32_t s = 1; s <= p.num_samples; ++s) {
        if (ci < r.carriers.size() && r.carriers[ci].sample_id == s) {
          // allele numbers: any non-zero number makes a carrier (variant_graph.h:666-691)
          int a1 = r.carriers[ci].gt1 ? 1 + (int)(r.alts.size() > 1 ? allele_rng.below(r.alts.size()) : 0) : 0;
          int a2 = r.carriers[ci].gt2 ? 1 + (int)(r.alts.size() > 1 ? allele_rng.below(r.alts.size()) : 0) : 0;
          fprintf(vf, "\t%d|%d", a1, a2);
          ++ci;
        } else fprintf(vf, "\t0|0");
      }
      fprintf(vf, "\n");
    }
    for (const auto& alt : r.alts) {
      std::vector<SampleGT> list = r.carriers;
      st.num_mutations++;
      st.num_mutations_samples += list.size();
      b.add_mutation(r.ref, alt, r.pos, list);
    }
  }
  if (vf) fclose(vf);
  if (n_keys) *n_keys = b.num_keys();
  if (n_edges) *n_edges = b.num_edges();
  if (seq_len) *seq_len = b.seq_length();
  // per-carrier sample-coordinate indexes cost 4 bytes per carrier record and are read only by query
  // types 2, 3 and 5; off unless asked for
  b.finish(out, p.sample_coordinates || getenv("VS_SYNTH_SAMPLE_INDEXES") != nullptr);
  return st;
}

}  // namespace vsamd
