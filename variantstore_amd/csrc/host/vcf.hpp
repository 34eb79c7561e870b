// vcf.hpp -- FASTA / VCF ingest for `variantstore construct`.
//
// Mirrors what the reference extracts from vcflib records in
// VariantGraph::add_vcfs (reference include/variant_graph.h:619-733),
// use_bit_vector_encoding (:568-617) and read_fasta (src/util.cc:82-105).
// vcflib keeps the per-sample fields of a record in a std::map keyed by sample
// NAME (vcflib/Variant.h:242), so carriers are gathered in name order while
// sample ids follow VCF column order; that asymmetry is part of the reference's
// observable output and is reproduced here.
#pragma once
#include <algorithm>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <zlib.h>
#include "builder.hpp"

namespace vsamd {

inline void read_fasta(const std::string& path, std::string& chr, std::string& ref) {
  std::ifstream in(path);
  if (!in.good()) throw std::runtime_error("Failed to open input fasta file: " + path);
  bool found = false;
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    if (line[0] == '>') {
      if (found) throw std::runtime_error("Found multiple references in the fasta file");
      std::stringstream ls(line);
      std::getline(ls, chr, ' ');
      chr = chr.substr(1);
      found = true;
    } else {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      ref.append(line);
    }
  }
}

struct VcfRecord {
  std::string chrom, ref;
  uint64_t pos = 0;
  std::vector<std::string> alts;
  std::vector<std::string> gt;  // GT string per sample column ("" = absent)
};

class VcfReader {
 public:
  explicit VcfReader(const std::string& path) {
    gz_ = gzopen(path.c_str(), "rb");  // transparently reads plain text too
    if (!gz_) throw std::runtime_error("cannot open VCF " + path);
    std::string line;
    while (getline(line)) {
      if (line.rfind("##", 0) == 0) continue;
      if (line.rfind("#CHROM", 0) == 0) {
        auto f = split(line, '\t');
        for (size_t i = 9; i < f.size(); ++i) sample_names.push_back(f[i]);
        break;
      }
    }
  }
  ~VcfReader() { if (gz_) gzclose(gz_); }
  VcfReader(const VcfReader&) = delete;

  std::vector<std::string> sample_names;

  bool next(VcfRecord& r) {
    std::string line;
    while (getline(line)) {
      if (line.empty() || line[0] == '#') continue;
      auto f = split(line, '\t');
      if (f.size() < 5) continue;
      r.chrom = f[0];
      r.pos = strtoull(f[1].c_str(), nullptr, 10);
      r.ref = f[3];
      r.alts = split(f[4], ',');
      r.gt.assign(sample_names.size(), "");
      if (f.size() > 9) {
        auto fmt = split(f[8], ':');
        size_t gi = std::find(fmt.begin(), fmt.end(), "GT") - fmt.begin();
        for (size_t s = 0; s < sample_names.size() && 9 + s < f.size(); ++s) {
          auto sf = split(f[9 + s], ':');
          if (gi < sf.size()) {
            auto vals = split(sf[gi], ',');
            r.gt[s] = vals.empty() ? "" : vals[0];
          }
        }
      }
      return true;
    }
    return false;
  }

  static std::vector<std::string> split(const std::string& s, char d) {
    std::vector<std::string> out;
    size_t a = 0;
    while (true) {
      size_t b = s.find(d, a);
      out.push_back(s.substr(a, b == std::string::npos ? b : b - a));
      if (b == std::string::npos) break;
      a = b + 1;
    }
    return out;
  }

 private:
  bool getline(std::string& line) {
    line.clear();
    char buf[1 << 16];
    while (gzgets(gz_, buf, sizeof(buf))) {
      line.append(buf);
      if (!line.empty() && line.back() == '\n') {
        line.pop_back();
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
      }
    }
    return !line.empty();
  }
  gzFile gz_ = nullptr;
};

// GT string -> carrier?  (variant_graph.h:666-691; the allele NUMBER is never
// compared with the ALT index: any non-zero allele makes the sample a carrier)
inline bool parse_gt(const std::string& gt, SampleGT& s) {
  if (gt.size() == 3) {
    int first = gt[0] - '0', second = gt[2] - '0';
    if (first > 0 || second > 0) {
      s.gt1 = first > 0;
      s.gt2 = second > 0;
      if (gt[1] == '|') s.phase = true;
      else if (gt[1] == '/') s.phase = false;
      else throw std::runtime_error(std::string("Unknown phase: ") + gt[1]);
      return true;
    }
  } else if (gt.size() == 1) {
    int present = 0;
    if (gt[0] >= '0' && gt[0] <= '9') present = gt[0] - '0';
    if (present) { s.phase = false; s.gt1 = true; s.gt2 = false; return true; }
  }
  return false;
}

inline bool is_acgt(const std::string& s) {  // std::regex("^[ACTG]+$"), variant_graph.h:660
  if (s.empty()) return false;
  for (char c : s) if (c != 'A' && c != 'C' && c != 'T' && c != 'G') return false;
  return true;
}

struct ConstructStats {
  uint64_t num_vars = 0, num_mutations = 0, num_mutations_samples = 0, num_unsupported = 0;
  bool use_bit_vector = false;
};

// VariantGraph(ref_file, vcf_file, ...) : variant_graph.h:323-364
inline ConstructStats construct_from_files(const std::string& fasta, const std::string& vcf, HostGraph& out,
                                           uint64_t* n_vertices = nullptr, uint64_t* n_edges = nullptr,
                                           uint64_t* seq_len = nullptr) {
  ConstructStats st;
  std::string chr, ref;
  read_fasta(fasta, chr, ref);

  // use_bit_vector_encoding(): density over the first 99 records (:568-617)
  {
    VcfReader rd(vcf);
    VcfRecord r;
    uint32_t cnt = 1;
    float density = 0;
    while (cnt < 100 && rd.next(r)) {
      uint32_t n = 0;
      SampleGT s{};
      for (auto& g : r.gt) if (parse_gt(g, s)) n++;
      float cur = rd.sample_names.empty() ? 0.f : n / (float)rd.sample_names.size();
      density = density > cur ? density : cur;
      cnt++;
    }
    st.use_bit_vector = density > 0.05f;
  }

  VcfReader rd(vcf);
  GraphBuilder b(chr, ref, rd.sample_names, st.use_bit_vector);
  // name-sorted view of the sample columns (vcflib's std::map<string, ...>)
  std::vector<uint32_t> by_name(rd.sample_names.size());
  for (uint32_t i = 0; i < by_name.size(); ++i) by_name[i] = i;
  std::stable_sort(by_name.begin(), by_name.end(),
                   [&](uint32_t a, uint32_t c) { return rd.sample_names[a] < rd.sample_names[c]; });

  VcfRecord r;
  while (rd.next(r)) {
    st.num_vars++;
    bool chr_ok = r.chrom == chr || (r.chrom.size() >= 3 && r.chrom.substr(3) == chr);
    if (!chr_ok || r.pos < 1 || r.pos > b.ref_length() || r.ref != b.get_sequence(r.pos - 1, (uint32_t)r.ref.size())) {
      st.num_unsupported++;
      continue;
    }
    for (const auto& alt : r.alts) {
      std::vector<SampleGT> list;
      if (is_acgt(alt)) {
        for (uint32_t col : by_name) {
          SampleGT s{};
          if (parse_gt(r.gt[col], s)) {
            s.sample_id = col + 1;
            list.push_back(s);
          }
        }
      }
      if (!list.empty()) {
        st.num_mutations++;
        st.num_mutations_samples += list.size();
        b.add_mutation(r.ref, alt, r.pos, list);
      }
    }
  }
  if (n_vertices) *n_vertices = b.num_keys();
  if (n_edges) *n_edges = b.num_edges();
  if (seq_len) *seq_len = b.seq_length();
  b.finish(out);
  return st;
}

}  // namespace vsamd
