// variantstore -- drop-in command line for the query path, on top of the C ABI.
//
// Same sub-commands, flags, log lines and output files as the reference driver
// (reference src/variantstore.cc:81-156 clipp grammar, src/commands.cc:33-60 construct_main,
// :64-93 read_regions, :114-215 query_main, src/util.cc:67-80 print_time_elapsed; log lines in
// spdlog's default pattern "[Y-m-d H:M:S.ms] [level] msg").  All seven query types run on the GPU
// through include/variantstore_hip.h.
//
// Extensions for batches that do not fit a command line:
//   -r @FILE            one "<start>:<end>" (or "<start>") per line instead of a comma list
//   --batch-out FILE    rows of ALL regions ("#region <i> <x>:<y>" before each), since the
//                       reference's -o file only ever holds the last region (query.h:774-781)
//   --device N          GPU ordinal (default 0)
//   --resident-lists    expand every carrier list of the index once, when it is opened, into an arena that stays in
//                       HBM (vs_index_set_option "resident_lists"): batches then copy rows only across PCIe
//   --nprocs N          any query type (1 - 7: the dispatch of src/commands.cc:150-193) as N PROCESSES, one per GPU (device ..
//                       device + N - 1): the parent forks before anything touches a GPU, every rank answers its contiguous shard
//                       of the sorted region list, the ranks all-gather the per-region records (counts and flags: what the
//                       reference prints per region) through the C ABI's collective (vs_comm_*: RCCL over xGMI; the unique id
//                       travels through a file) and rank 0 prints the log lines of ALL regions from the gathered records;
//                       --batch-out text is written per rank and concatenated in rank order, -o holds the last region's
//                       text (the last FOUND position's for types 1 and 7).  A rank that fails ends the others.  --ngpus
//                       (threads in one process, no collective) remains the fallback.
//   --nprocs-same-device  TEST-ONLY: every rank of --nprocs opens GPU `--device` (real RCCL refuses two ranks on one device:
//                       tests load tests/native/fake_rccl.cpp through VS_RCCL_LIB to run the multi-rank code on a one-GPU box)
//   --ngpus N           shard the sorted region list over GPUs device .. device + N - 1 (query types 4, 5, 6): one handle
//                       and one host thread per GPU, contiguous shards (the reference's serial loop, commands.cc:145,
//                       carries no state between regions), results printed in region order
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <signal.h>
#include <unistd.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <tuple>
#include <vector>
#include "variantstore_hip.h"

namespace {

void log_line(const char* level, const std::string& msg) {
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  struct tm tm;
  localtime_r(&tv.tv_sec, &tm);
  char ts[64];
  strftime(ts, sizeof(ts), "%Y-%m-%d %H:%M:%S", &tm);
  printf("[%s.%03d] [%s] %s\n", ts, (int)(tv.tv_usec / 1000), level, msg.c_str());
  fflush(stdout);
}
void info(const std::string& m) { log_line("info", m); }
void error(const std::string& m) { log_line("error", m); }

[[noreturn]] void die(int rc, const char* what) {
  error(std::string(what) + ": " + vs_strerror(rc) + ": " + vs_last_error());
  abort();
}

// std::stoi semantics of read_regions (commands.cc:64-93): throws std::invalid_argument on garbage
uint64_t stoi_like(const std::string& s) { return (uint64_t)std::stoi(s); }

std::vector<std::tuple<uint64_t, uint64_t>> read_regions(std::string region) {
  std::vector<std::tuple<uint64_t, uint64_t>> regions;
  if (!region.empty() && region[0] == '@') {
    std::ifstream in(region.substr(1));
    if (!in) throw std::runtime_error("cannot open region file " + region.substr(1));
    std::string line;
    while (std::getline(in, line)) {
      if (line.empty()) continue;
      auto p = line.find(':');
      if (p == std::string::npos) regions.emplace_back(stoi_like(line), 0);
      else regions.emplace_back(stoi_like(line.substr(0, p)), stoi_like(line.substr(p + 1)));
    }
  } else {
    auto pos = region.find(',');
    while (true) {
      std::string token = region.substr(0, pos);
      auto pos2 = token.find(':');
      uint64_t beg = 0, end = 0;
      if (pos2 == std::string::npos) beg = stoi_like(token);
      else { end = stoi_like(token.substr(pos2 + 1)); beg = stoi_like(token.substr(0, pos2)); }
      regions.emplace_back(beg, end);
      if (pos == std::string::npos) break;
      region = region.substr(pos + 1);
      pos = region.find(',');
    }
  }
  std::sort(regions.begin(), regions.end());
  return regions;
}

void print_time_elapsed(const std::string& desc, const timeval& start, timeval end) {  // util.cc:67-80
  if (start.tv_usec > end.tv_usec) { end.tv_usec += 1000000; end.tv_sec--; }
  float t = ((end.tv_sec - start.tv_sec) * 1000000 + (end.tv_usec - start.tv_usec)) / 1000000.f;
  std::cout << desc << "Total Time Elapsed: " << std::to_string(t) << "seconds" << std::endl;
}

bool dir_exists(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

struct Args {
  std::string cmd, ref, vcf, prefix, region, outfile, sample, alt, refseq, batch_out;
  uint32_t type = 0, mode = 0;
  uint64_t hops = 0;
  bool have_hops = false;
  bool have_type = false, have_mode = false, verbose = false;
  int device = 0, ngpus = 1, nprocs = 0;
  bool resident_lists = false;
  bool nprocs_same_device = false;   // test-only (--nprocs-same-device): every rank of --nprocs opens GPU `device`
};

int usage() {
  std::cout << "SYNOPSIS\n"
               "        variantstore construct -r <reference-file> -v <vcf-file> -p <output-prefix>\n"
               "        variantstore query -p <output-prefix> -t <query-type> -r <region> -m <mode> [-o <outfile>]\n"
               "                     [-s <sample-name>] [-a <alt-seq>] [-b <ref-seq>] [-v]\n"
               "                     [--batch-out <file>] [--device <n>] [--ngpus <n>] [--nprocs <n>] [--resident-lists]\n"
               "        variantstore draw -p <output-prefix> -r <region> -h <hops> [-s <sample-name>]\n"
               "        variantstore help\n\n"
               "OPTIONS\n"
               "        <query-type>  1  Return closest variant in reference coordinates.\n"
               "                      2  Get sample's sequence in reference coordinates.\n"
               "                      3  Get sample's sequence in sample coordinates.\n"
               "                      4  Get sample's variants in reference coordinates.\n"
               "                      5  Get sample's variants in sample coordinates.\n"
               "                      6  Get variants in reference coordinates.\n"
               "                      7  Return samples with a given mutation.\n"
               "        <region>      <start>:<end>[,<start>:<end>...] or @file with one region per line\n"
               "        <mode>        READ_INDEX_ONLY: 0, READ_COMPLETE_GRAPH: 1 (the index is always resident here)\n";
  return EXIT_FAILURE;
}

int construct_main(const Args& a) {
  info("Creating variant graph");
  vs_construct_stats st;
  vs_index* idx = nullptr;
  // stdout lines of the VariantGraph constructor (variant_graph.h:342-344, 625-626, 729-731, 1919)
  int rc = vs_index_from_vcf(a.ref.c_str(), a.vcf.c_str(), -1, &st, &idx);
  if (rc != VS_OK) die(rc, "construct");
  if (st.use_bit_vector) info("Building sample vector based variant graph.");
  vs_index_info inf;
  vs_index_get_info(idx, &inf);
  info("Adding mutations from: " + a.vcf + " #Samples: " + std::to_string(inf.num_samples - 1));
  info("Num mutations: " + std::to_string(st.num_mutations) + " num mutations-sample: " + std::to_string(st.num_mutations_samples));
  info("Num vars: " + std::to_string(st.num_vars));
  info("Fixing sample indexes in the graph");
  info("Graph stats:");
  info(std::string("Chromosome: ") + vs_index_chr(idx) + " #Vertices: " + std::to_string(st.num_vertices) +
       " #Edges: " + std::to_string(st.num_edges) + " Seq length: " + std::to_string(st.seq_length));
  info("Serializing variant graph to disk");
  info("Number of sample vector classes: " + std::to_string(st.num_classes));
  info("Creating Index");
  info("Serializing index to disk");
  rc = vs_index_save(idx, a.prefix.c_str());
  if (rc != VS_OK) die(rc, "serialize");
  vs_index_close(idx);
  return EXIT_SUCCESS;
}

// read_sequences, commands.cc:96-111
std::vector<std::string> read_sequences(std::string s) {
  std::vector<std::string> seqs;
  auto pos = s.find(',');
  while (true) {
    seqs.push_back(s.substr(0, pos));
    if (pos == std::string::npos) break;
    s = s.substr(pos + 1);
    pos = s.find(',');
  }
  return seqs;
}

// Query types 2 (query_sample_from_ref) and 3 (query_sample_from_sample): commands.cc:156-165.
int sequence_query_main(const Args& a, vs_index* idx, const std::vector<vs_region>& batch) {
  struct timeval start, end;
  gettimeofday(&start, nullptr);
  uint32_t sid = 0;
  if (vs_index_sample_id(idx, a.sample.c_str(), &sid) != VS_OK) { error("Sample not found"); abort(); }  // variant_graph.h:2010-2013
  std::vector<uint32_t> sids(batch.size(), sid);
  vs_result* res = nullptr;
  int rc = vs_query_sample_seq(idx, batch.data(), batch.size(), sids.data(), a.type == 3 ? 1 : 0, &res);
  if (rc != VS_OK) die(rc, "query");
  uint64_t nq = 0;
  const uint8_t* flags = nullptr;
  rc = vs_result_get_sequences(res, &nq, &flags, nullptr, nullptr);
  if (rc != VS_OK) die(rc, "result");
  gettimeofday(&end, nullptr);
  std::ofstream batch_out;
  if (!a.batch_out.empty()) batch_out.open(a.batch_out);
  uint32_t query_num = 0;
  for (uint64_t i = 0; i < nq; ++i) {
    if (a.type == 2) info("2. Get sample's sequence in ref coordinate. " + std::to_string(i));
    else info("3. Get sample's sequence in sample's coordinate. " + std::to_string(i));
    if (flags[i] & VS_REGION_INVALID) {  // an uncaught exception of std::string::substr in the reference
      std::cerr << "terminate called after throwing an instance of 'std::out_of_range'\n";
      abort();
    }
    if (flags[i] & VS_REGION_ENDLESS) {
      error("the reference's backward search does not terminate on region " + std::to_string(batch[i].x) + ":" + std::to_string(batch[i].y));
      return EXIT_FAILURE;
    }
    const char* text = nullptr; uint64_t len = 0;
    if (batch_out.is_open() || a.verbose) {
      rc = vs_result_format_region(res, i, &text, &len);
      if (rc != VS_OK) die(rc, "result");
    }
    if (batch_out.is_open()) {
      batch_out << "#region " << i << " " << batch[i].x << ":" << batch[i].y << "\n";
      batch_out.write(text, len);
    }
    if (a.verbose && i + 1 == nq) {
      std::ofstream out;
      out.open(a.outfile);
      out.write(text, len);
    }
    query_num += 1;
    if (query_num == 10 || query_num == 100 || query_num == 1000) {
      gettimeofday(&end, nullptr);   // cumulative up to this line, as in the reference's loop (commands.cc:196-211)
      print_time_elapsed("Query" + std::to_string(query_num) + ": ", start, end);
    }
  }
  gettimeofday(&end, nullptr);
  print_time_elapsed("Query" + std::to_string(query_num) + ": ", start, end);
  vs_result_free(res);
  vs_index_close(idx);
  return EXIT_SUCCESS;
}

// Query types 1 (closest_var) and 7 (samples_has_var): commands.cc:151-155, 181-189.
int point_query_main(const Args& a, vs_index* idx, const std::vector<vs_region>& batch) {
  struct timeval start, end;
  gettimeofday(&start, nullptr);
  std::vector<uint64_t> positions;
  for (auto& b : batch) positions.push_back(b.x);
  std::vector<std::string> refs, alts;
  std::vector<const char*> refp, altp;
  vs_result* res = nullptr;
  int rc;
  if (a.type == 7) {
    alts = read_sequences(a.alt);
    refs = read_sequences(a.refseq);
    // the reference indexes refs[i]/alts[i] with the position of the region in the SORTED list and does not
    // check the lengths (commands.cc:185); a shorter list is out of bounds there and refused here
    if (refs.size() < batch.size() || alts.size() < batch.size()) {
      error("-a/-b must list one sequence per region");
      vs_index_close(idx);
      return EXIT_FAILURE;
    }
    for (size_t i = 0; i < batch.size(); ++i) { refp.push_back(refs[i].c_str()); altp.push_back(alts[i].c_str()); }
    rc = vs_query_samples_has_var(idx, positions.data(), refp.data(), altp.data(), positions.size(), &res);
  } else {
    rc = vs_query_closest_var(idx, positions.data(), positions.size(), &res);
  }
  if (rc != VS_OK) die(rc, "query");
  vs_result_view v;
  rc = vs_result_get_view(res, 0, &v);
  if (rc != VS_OK) die(rc, "result");
  gettimeofday(&end, nullptr);
  std::ofstream batch_out;
  if (!a.batch_out.empty()) batch_out.open(a.batch_out);
  uint32_t query_num = 0;
  for (uint64_t i = 0; i < v.n_regions; ++i) {
    const bool found = !(v.region_flags[i] & VS_REGION_NOT_FOUND);
    if (a.type == 1) info("1. return closest mutation in ref coordinate. " + std::to_string(i));
    else {
      info("7. Get samples have given variant. " + std::to_string(i));
      info("Looking for variant POS: " + std::to_string(positions[i]) + ", REF: " + refs[i] + ", ALT: " + alts[i]);
      if (!found) error("There is no such variant!");
    }
    const char* text = nullptr; uint64_t len = 0;
    if (found && (batch_out.is_open() || a.verbose)) {
      rc = vs_result_format_region(res, i, &text, &len);
      if (rc != VS_OK) die(rc, "result");
    }
    if (batch_out.is_open()) {
      batch_out << "#region " << i << " " << positions[i] << (found ? "" : " not-found") << "\n";
      if (found) batch_out.write(text, len);
    }
    if (a.verbose && found) {  // rewritten by every call that finds something: the last one stays
      std::ofstream out;
      out.open(a.outfile);
      out.write(text, len);
    }
    query_num += 1;
    if (query_num == 10 || query_num == 100 || query_num == 1000) {
      gettimeofday(&end, nullptr);   // cumulative up to this line, as in the reference's loop (commands.cc:196-211)
      print_time_elapsed("Query" + std::to_string(query_num) + ": ", start, end);
    }
  }
  gettimeofday(&end, nullptr);
  print_time_elapsed("Query" + std::to_string(query_num) + ": ", start, end);
  vs_result_free(res);
  vs_index_close(idx);
  return EXIT_SUCCESS;
}

// `variantstore draw` (commands.cc:217-242): <prefix>/graph.dot for the neighbourhood of the vertex at the region's
// start.  The reference's option grammar stores -s into the QUERY options (variantstore.cc:146-150), so its draw
// always starts from the reference path; the same here.
int draw_main(const Args& a) {
  info("Loading Index ...");
  info("Loading variant graph ...");
  info("Read complete graph ..");
  vs_index* idx = nullptr;
  int rc = vs_index_open(a.prefix.c_str(), -1, &idx);   // host-only: nothing here runs on the GPU
  if (rc != VS_OK) die(rc, "load");
  vs_index_info inf;
  vs_index_get_info(idx, &inf);
  info("Graph stats:");
  info(std::string("Chromosome: ") + vs_index_chr(idx) + " #Vertices: " + std::to_string(inf.num_topology_keys) +
       " #Edges: 0 Seq length: " + std::to_string(inf.seq_length));
  auto regions = read_regions(a.region);
  info("Looking up vertex corresponding to the queried region");
  rc = vs_index_draw_subgraph(idx, std::get<0>(regions[0]), a.hops, "ref", (a.prefix + "/graph.dot").c_str());
  if (rc != VS_OK) die(rc, "draw");
  vs_index_close(idx);
  return EXIT_SUCCESS;
}

// `--nprocs N`: one process per GPU and the collective of the C ABI (see the header comment), for every query type the
// reference's loop dispatches (src/commands.cc:150-193).  The sorted region list is cut into contiguous shards, every rank
// answers its shard with the type's own entry point, the per-region records (vs_result_pack_regions: counts and flags --
// all the reference prints per region) are all-gathered through vs_comm_*, rank 0 prints the log lines of EVERY region from
// the gathered records, and the ranks' shards of the --batch-out text are put together by the parent.
int query_multiproc_main(const Args& a) {
  auto regions = read_regions(a.region);
  if (a.type < 1 || a.type > 7) {
    for (size_t i = 0; i < regions.size(); ++i) error("Unsupported query type");  // commands.cc:191
    return EXIT_SUCCESS;
  }
  std::vector<vs_region> batch;
  for (auto& r : regions) batch.push_back(vs_region{std::get<0>(r), std::get<1>(r)});
  std::vector<std::string> refs, alts;
  if (a.type == 7) {
    alts = read_sequences(a.alt);
    refs = read_sequences(a.refseq);
    if (refs.size() < batch.size() || alts.size() < batch.size()) { error("-a/-b must list one sequence per region"); return EXIT_FAILURE; }
  }
  const int world = (int)std::min<size_t>((size_t)a.nprocs, std::max<size_t>(batch.size(), 1));
  char tmpl[] = "/tmp/vs_nprocs_XXXXXX";
  if (!mkdtemp(tmpl)) { error("cannot create a temporary directory"); return EXIT_FAILURE; }
  const std::string dir = tmpl, uid_file = dir + "/uid";
  auto shard = [&](int k, size_t* lo, size_t* hi) {
    const size_t base = batch.size() / world, rem = batch.size() % world;
    *lo = k * base + std::min<size_t>(k, rem);
    *hi = *lo + base + ((size_t)k < rem ? 1 : 0);
  };
  size_t max_count = 0;
  for (int k = 0; k < world; ++k) { size_t lo, hi; shard(k, &lo, &hi); max_count = std::max(max_count, hi - lo); }
  std::vector<pid_t> kids;
  fflush(stdout);
  for (int rank = 0; rank < world; ++rank) {
    const pid_t pid = fork();   // (nothing in this process has touched a GPU)
    if (pid < 0) { error("fork failed"); return EXIT_FAILURE; }
    if (pid > 0) { kids.push_back(pid); continue; }
    // ---- rank `rank` ----
    struct timeval start, end;
    if (rank == 0) { info("Loading Index ..."); info("Loading variant graph ..."); info(a.mode == 0 ? "Read index only .." : "Read complete graph .."); }
    vs_index* idx = nullptr;
    int rc = vs_index_open(a.prefix.c_str(), a.nprocs_same_device ? a.device : a.device + rank, &idx);
    if (rc != VS_OK) die(rc, "load");
    if (const char* k = getenv("VS_NPROCS_TEST_KILL_RANK"))   // TEST-ONLY: this rank dies the hard way before it joins the communicator
      if (*k && atoi(k) == rank) raise(SIGKILL);
    if (rank == 0) {
      vs_index_info inf;
      vs_index_get_info(idx, &inf);
      info("Graph stats:");
      info(std::string("Chromosome: ") + vs_index_chr(idx) + " #Vertices: " + std::to_string(inf.num_topology_keys) +
           " #Edges: 0 Seq length: " + std::to_string(inf.seq_length));
    }
    if (a.resident_lists && vs_index_set_option(idx, "resident_lists", 1) != VS_OK)
      log_line("warning", std::string("--resident-lists: ") + vs_last_error() + " (lists are expanded per batch)");
    uint32_t sid = 0;
    if (a.type >= 2 && a.type <= 5 && vs_index_sample_id(idx, a.sample.c_str(), &sid) != VS_OK) {   // variant_graph.h:2010-2013
      if (rank == 0) error("Sample not found");
      _exit(EXIT_FAILURE);
    }
    // the communicator: rank 0 makes the unique id, the others wait for its file
    unsigned char uid[VS_COMM_ID_BYTES];
    if (rank == 0) {
      rc = vs_comm_unique_id(uid);
      if (rc != VS_OK) die(rc, "vs_comm_unique_id");
      std::ofstream f(uid_file + ".tmp", std::ios::binary);
      f.write((const char*)uid, sizeof(uid));
      f.close();
      rename((uid_file + ".tmp").c_str(), uid_file.c_str());
    } else {
      for (int tries = 0;; ++tries) {
        std::ifstream f(uid_file, std::ios::binary);
        if (f && f.read((char*)uid, sizeof(uid))) break;
        if (tries > 60000) { error("no unique id from rank 0"); _exit(EXIT_FAILURE); }
        usleep(2000);
      }
    }
    vs_comm* comm = nullptr;
    {   // RCCL prints a version banner through C stdio when a communicator is made: stdout is the reference's log, so file
        // descriptor 1 points at stderr for the duration of the call
      fflush(stdout);
      const int saved = dup(1);
      dup2(2, 1);
      rc = vs_comm_init(idx, rank, world, uid, &comm);
      fflush(stdout);
      dup2(saved, 1);
      close(saved);
    }
    if (rc != VS_OK) die(rc, "vs_comm_init");
    size_t lo, hi;
    shard(rank, &lo, &hi);
    const uint64_t n = hi - lo;
    gettimeofday(&start, nullptr);
    vs_result* res = nullptr;
    if (a.type == 6) rc = vs_query_var_in_ref(idx, batch.data() + lo, n, &res);
    else if (a.type == 4) rc = vs_query_sample_var_in_ref(idx, batch.data() + lo, n, sid, &res);
    else if (a.type == 5) { std::vector<uint32_t> sids(n, sid); rc = vs_query_sample_var_in_sample(idx, batch.data() + lo, n, sids.data(), &res); }
    else if (a.type == 2 || a.type == 3) { std::vector<uint32_t> sids(n, sid); rc = vs_query_sample_seq(idx, batch.data() + lo, n, sids.data(), a.type == 3 ? 1 : 0, &res); }
    else {
      std::vector<uint64_t> positions;
      for (size_t i = lo; i < hi; ++i) positions.push_back(batch[i].x);
      if (a.type == 7) {
        std::vector<const char*> refp, altp;
        for (size_t i = lo; i < hi; ++i) { refp.push_back(refs[i].c_str()); altp.push_back(alts[i].c_str()); }
        rc = vs_query_samples_has_var(idx, positions.data(), refp.data(), altp.data(), n, &res);
      } else rc = vs_query_closest_var(idx, positions.data(), n, &res);
    }
    if (rc != VS_OK) die(rc, "query");
    std::vector<uint64_t> recs((size_t)world * max_count * 4);
    rc = vs_comm_allgather_regions_host(comm, res, lo, max_count, recs.data());
    if (rc != VS_OK) die(rc, "vs_comm_allgather_regions_host");
    const bool point = a.type == 1 || a.type == 7;
    auto own_flags = [&](size_t k) { return (recs[((size_t)rank * max_count + k) * 4 + 1] >> 32) & 0xFF; };
    if (!a.batch_out.empty()) {   // this rank's shard of the text
      std::ofstream out(dir + "/out." + std::to_string(rank));
      for (size_t k = 0; k < n; ++k) {
        const bool found = !(own_flags(k) & VS_REGION_NOT_FOUND);
        if (point) out << "#region " << (lo + k) << " " << batch[lo + k].x << (found ? "" : " not-found") << "\n";
        else out << "#region " << (lo + k) << " " << batch[lo + k].x << ":" << batch[lo + k].y << "\n";
        const char* text; uint64_t len;
        if ((!point || found) && vs_result_format_region(res, k, &text, &len) == VS_OK) out.write(text, len);
      }
    }
    if (a.verbose) {   // the -o file: the last region's text (query.h:774-781) -- the last FOUND one for the point queries; a rank that
                       // holds a candidate leaves it for the parent, which keeps the highest rank's
      size_t last = n;
      for (size_t k = n; k-- > 0;)
        if (!point || !(own_flags(k) & VS_REGION_NOT_FOUND)) { last = k; break; }
      const char* text; uint64_t len;
      if (last < n && (point || hi == batch.size()) && vs_result_format_region(res, last, &text, &len) == VS_OK) {
        std::ofstream out(dir + "/last." + std::to_string(rank));
        out.write(text, len);
      }
    }
    int status = EXIT_SUCCESS;
    if (rank == 0) {   // the log lines of every region, from the gathered records
      uint32_t query_num = 0;
      for (int k = 0; k < world && status == EXIT_SUCCESS; ++k) {
        size_t klo, khi;
        shard(k, &klo, &khi);
        for (size_t j = 0; j < khi - klo && status == EXIT_SUCCESS; ++j) {
          const uint64_t* rec = &recs[((size_t)k * max_count + j) * 4];
          const uint64_t i = rec[0], flags = (rec[1] >> 32) & 0xFF, nvar = rec[2] >> 32;
          switch (a.type) {
            case 1: info("1. return closest mutation in ref coordinate. " + std::to_string(i)); break;
            case 2: info("2. Get sample's sequence in ref coordinate. " + std::to_string(i)); break;
            case 3: info("3. Get sample's sequence in sample's coordinate. " + std::to_string(i)); break;
            case 4: info("4. Get sample's variants in ref coordinate. " + std::to_string(i)); break;
            case 5: info("5. Get sample's variants in sample coordinate. " + std::to_string(i)); break;
            case 6: info("6. Get variants in ref coordinate. " + std::to_string(i)); break;
            default:
              info("7. Get samples have given variant. " + std::to_string(i));
              info("Looking for variant POS: " + std::to_string(batch[i].x) + ", REF: " + refs[i] + ", ALT: " + alts[i]);
              if (flags & VS_REGION_NOT_FOUND) error("There is no such variant!");
          }
          if (!point && (flags & VS_REGION_ENDLESS)) {
            error("the reference's backward search does not terminate on region " + std::to_string(batch[i].x) + ":" + std::to_string(batch[i].y));
            status = EXIT_FAILURE; break;
          }
          if (!point && (flags & VS_REGION_INVALID)) {
            if (a.type == 2 || a.type == 3) std::cerr << "terminate called after throwing an instance of 'std::out_of_range'\n";
            else error("Can't find node corresponding to pos " + std::to_string(batch[i].x));   // index.h:151-154
            status = EXIT_FAILURE; break;
          }
          if (a.type >= 4 && a.type <= 6) {
            const char* label = (a.type == 6 && !(flags & VS_REGION_EMPTY)) ? "get_var_in_ref" : "get_sample_var_in_ref";   // query.h:746
            if (a.type == 5) label = "get_sample_var_in_sample";  // query.h:599
            std::cout << "Number of variants " << label << ": " << nvar << '\n';
          }
          query_num += 1;
          if (query_num == 10 || query_num == 100 || query_num == 1000) {
            gettimeofday(&end, nullptr);
            print_time_elapsed("Query" + std::to_string(query_num) + ": ", start, end);
          }
        }
      }
      gettimeofday(&end, nullptr);
      print_time_elapsed("Query" + std::to_string(query_num) + ": " + (a.type == 6 ? "(query_var_in_ref) " : ""), start, end);
    }
    fflush(stdout);
    vs_result_free(res);
    vs_comm_destroy(comm);
    vs_index_close(idx);
    _exit(status);
  }
  // Whichever rank ends first is reaped first; a rank that fails (no such device, an RCCL error) leaves the others inside
  // ncclCommInitRank or the all-gather for good, so the first abnormal exit ends them all (fresh forks: nothing to save).
  int failed = 0;
  for (size_t left = kids.size(); left > 0; --left) {
    int st = 0;
    const pid_t pid = waitpid(-1, &st, 0);
    if (pid < 0) { failed = 1; break; }
    auto it = std::find(kids.begin(), kids.end(), pid);
    if (it == kids.end()) { ++left; continue; }   // (not one of the ranks)
    *it = -1;
    if (!WIFEXITED(st) || WEXITSTATUS(st) != EXIT_SUCCESS) {
      if (!failed) {
        error("a rank of --nprocs failed; stopping the others");
        for (pid_t other : kids) if (other > 0) kill(other, SIGKILL);
      }
      failed = 1;
    }
  }
  if (!failed && !a.batch_out.empty()) {
    std::ofstream out(a.batch_out, std::ios::binary);
    for (int k = 0; k < world; ++k) {
      std::ifstream in(dir + "/out." + std::to_string(k), std::ios::binary);
      out << in.rdbuf();
    }
  }
  if (!failed && a.verbose)
    for (int k = world; k-- > 0;) {
      std::ifstream in(dir + "/last." + std::to_string(k), std::ios::binary);
      if (!in) continue;
      std::ofstream out(a.outfile, std::ios::binary);
      out << in.rdbuf();
      break;
    }
  for (int k = 0; k < world; ++k) { remove((dir + "/out." + std::to_string(k)).c_str()); remove((dir + "/last." + std::to_string(k)).c_str()); }
  remove(uid_file.c_str());
  rmdir(dir.c_str());
  return failed ? EXIT_FAILURE : EXIT_SUCCESS;
}

int query_main(const Args& a) {
  if (a.nprocs >= 1) return query_multiproc_main(a);
  info("Loading Index ...");
  info("Loading variant graph ...");
  info(a.mode == 0 ? "Read index only .." : "Read complete graph ..");
  vs_index* idx = nullptr;
  int rc = vs_index_open(a.prefix.c_str(), a.device, &idx);
  if (rc != VS_OK) die(rc, "load");
  vs_index_info inf;
  vs_index_get_info(idx, &inf);
  info("Graph stats:");
  info(std::string("Chromosome: ") + vs_index_chr(idx) + " #Vertices: " + std::to_string(inf.num_topology_keys) +
       " #Edges: 0 Seq length: " + std::to_string(inf.seq_length));

  auto regions = read_regions(a.region);
  struct timeval start, end;
  gettimeofday(&start, nullptr);
  if (a.type < 1 || a.type > 7) {
    for (size_t i = 0; i < regions.size(); ++i) error("Unsupported query type");  // commands.cc:191
    vs_index_close(idx);
    return EXIT_SUCCESS;
  }
  std::vector<vs_region> batch;
  for (auto& r : regions) batch.push_back(vs_region{std::get<0>(r), std::get<1>(r)});
  if (a.type == 1 || a.type == 7) return point_query_main(a, idx, batch);
  if (a.type == 2 || a.type == 3) return sequence_query_main(a, idx, batch);
  uint32_t sid = 0;
  if (a.type != 6 && vs_index_sample_id(idx, a.sample.c_str(), &sid) != VS_OK) { error("Sample not found"); abort(); }  // variant_graph.h:2010-2013
  // ---- shards: one per GPU (--ngpus), contiguous pieces of the sorted list ----
  const int ng = (int)std::min<size_t>((size_t)a.ngpus, std::max<size_t>(batch.size(), 1));
  struct Shard { vs_index* idx = nullptr; size_t lo = 0, hi = 0; vs_result* res = nullptr; vs_result_raw v{}; int rc = VS_OK; std::string err; };
  std::vector<Shard> shards(ng);
  shards[0].idx = idx;
  for (int g = 0; g < ng; ++g) {
    const size_t base = batch.size() / ng, rem = batch.size() % ng;
    shards[g].lo = g * base + std::min<size_t>(g, rem);
    shards[g].hi = shards[g].lo + base + ((size_t)g < rem ? 1 : 0);
  }
  auto run_shard = [&](Shard& sh, int g) {
    if (!sh.idx) {
      sh.rc = vs_index_open(a.prefix.c_str(), a.device + g, &sh.idx);
      if (sh.rc != VS_OK) { sh.err = vs_last_error(); return; }
    }
    if (a.resident_lists && vs_index_set_option(sh.idx, "resident_lists", 1) != VS_OK)
      log_line("warning", std::string("--resident-lists: ") + vs_last_error() + " (lists are expanded per batch)");
    const vs_region* rg = batch.data() + sh.lo;
    const uint64_t n = sh.hi - sh.lo;
    if (a.type == 6) sh.rc = vs_query_var_in_ref(sh.idx, rg, n, &sh.res);
    else if (a.type == 4) sh.rc = vs_query_sample_var_in_ref(sh.idx, rg, n, sid, &sh.res);
    else {
      std::vector<uint32_t> sids(n, sid);
      sh.rc = vs_query_sample_var_in_sample(sh.idx, rg, n, sids.data(), &sh.res);
    }
    // the per-region counts and flags are all the log lines need (the raw form: no per-region expansion on the host); when
    // the text of every region is wanted, rows and carriers come over in ONE raw copy (not one copy per region)
    if (sh.rc == VS_OK) sh.rc = vs_result_get_raw(sh.res, a.batch_out.empty() ? 0 : 1, &sh.v);
    if (sh.rc != VS_OK) sh.err = vs_last_error();
  };
  if (ng == 1) run_shard(shards[0], 0);
  else {
    std::vector<std::thread> th;
    for (int g = 0; g < ng; ++g) th.emplace_back(run_shard, std::ref(shards[g]), g);
    for (auto& t : th) t.join();
  }
  for (auto& sh : shards)
    if (sh.rc != VS_OK) { error(std::string("query: ") + vs_strerror(sh.rc) + ": " + sh.err); abort(); }
  gettimeofday(&end, nullptr);

  std::ofstream batch_out;
  if (!a.batch_out.empty()) batch_out.open(a.batch_out);
  uint32_t query_num = 0;
  for (auto& sh : shards) {
    const vs_result_raw& v = sh.v;
    for (uint64_t k = 0; k < v.n_regions; ++k) {
      const uint64_t i = sh.lo + k;
      if (a.type == 6) info("6. Get variants in ref coordinate. " + std::to_string(i));
      else if (a.type == 5) info("5. Get sample's variants in sample coordinate. " + std::to_string(i));
      else info("4. Get sample's variants in ref coordinate. " + std::to_string(i));
      if (v.region_flags[k] & VS_REGION_ENDLESS) {
        error("the reference's backward search does not terminate on region " + std::to_string(batch[i].x) + ":" + std::to_string(batch[i].y));
        return EXIT_FAILURE;
      }
      if (v.region_flags[k] & VS_REGION_INVALID) {  // index.h:151-154
        error("Can't find node corresponding to pos " + std::to_string(batch[i].x));
        abort();
      }
      // query.h:746 prints the type-4 label on the early-out of type 6 as well
      const char* label = (a.type == 6 && !(v.region_flags[k] & VS_REGION_EMPTY)) ? "get_var_in_ref" : "get_sample_var_in_ref";
      if (a.type == 5) label = "get_sample_var_in_sample";  // query.h:599
      std::cout << "Number of variants " << label << ": " << v.var_count[k] << '\n';
      if (batch_out.is_open()) {
        const char* text; uint64_t len;
        if (vs_result_format_region(sh.res, k, &text, &len) == VS_OK) {
          batch_out << "#region " << i << " " << batch[i].x << ":" << batch[i].y << "\n";
          batch_out.write(text, len);
        }
      }
      if (a.verbose && i + 1 == batch.size()) {  // the -o file is reopened (truncated) per region: the last one stays
        const char* text; uint64_t len;
        if (vs_result_format_region(sh.res, k, &text, &len) == VS_OK) {
          std::ofstream out;
          out.open(a.outfile);
          out.write(text, len);
        }
      }
      query_num += 1;
      if (query_num == 10 || query_num == 100 || query_num == 1000) {
        gettimeofday(&end, nullptr);   // cumulative up to this line, as in the reference's loop (commands.cc:196-211)
        print_time_elapsed("Query" + std::to_string(query_num) + ": ", start, end);
      }
    }
  }
  std::string dsc = "Query" + std::to_string(query_num) + ": ";
  if (a.type == 6) dsc.append("(query_var_in_ref) ");
  gettimeofday(&end, nullptr);   // the whole loop, output included: what the reference's last line measures
  print_time_elapsed(dsc, start, end);
  for (auto& sh : shards) {
    vs_result_free(sh.res);
    vs_index_close(sh.idx);
  }
  return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  if (argc < 2) return usage();
  a.cmd = argv[1];
  auto need = [&](int& i) -> std::string {
    if (i + 1 >= argc) { std::cerr << "missing value for " << argv[i] << "\n"; exit(EXIT_FAILURE); }
    return argv[++i];
  };
  for (int i = 2; i < argc; ++i) {
    std::string f = argv[i];
    if (a.cmd == "construct") {
      if (f == "-r" || f == "--reference") a.ref = need(i);
      else if (f == "-v" || f == "--vcf") a.vcf = need(i);
      else if (f == "-p" || f == "--output-prefix") a.prefix = need(i);
      else { std::cerr << "unknown option " << f << "\n"; return EXIT_FAILURE; }
    } else if (a.cmd == "draw") {
      if (f == "-p" || f == "--output-prefix") a.prefix = need(i);
      else if (f == "-r" || f == "--region") a.region = need(i);
      else if (f == "-h" || f == "--hops") { a.hops = (uint64_t)atoll(need(i).c_str()); a.have_hops = true; }
      else if (f == "-s" || f == "--sample-name") a.sample = need(i);
      else { std::cerr << "unknown option " << f << "\n"; return EXIT_FAILURE; }
    } else if (a.cmd == "query") {
      if (f == "-p" || f == "--output-prefix") a.prefix = need(i);
      else if (f == "-t" || f == "--type") { a.type = (uint32_t)atoi(need(i).c_str()); a.have_type = true; }
      else if (f == "-r" || f == "--region") a.region = need(i);
      else if (f == "-m" || f == "--mode") { a.mode = (uint32_t)atoi(need(i).c_str()); a.have_mode = true; }
      else if (f == "-o" || f == "--output_file") a.outfile = need(i);
      else if (f == "-s" || f == "--sample-name") a.sample = need(i);
      else if (f == "-a" || f == "--alt-seq") a.alt = need(i);
      else if (f == "-b" || f == "--ref-seq") a.refseq = need(i);
      else if (f == "-v" || f == "--verbose") a.verbose = true;
      else if (f == "--batch-out") a.batch_out = need(i);
      else if (f == "--device") a.device = atoi(need(i).c_str());
      else if (f == "--ngpus") a.ngpus = std::max(1, atoi(need(i).c_str()));
      else if (f == "--nprocs") a.nprocs = std::max(1, atoi(need(i).c_str()));
      else if (f == "--nprocs-same-device") a.nprocs_same_device = true;
      else if (f == "--resident-lists") a.resident_lists = true;
      else { std::cerr << "unknown option " << f << "\n"; return EXIT_FAILURE; }
    }
  }
  if (a.cmd == "construct") {
    if (a.ref.empty() || a.vcf.empty() || a.prefix.empty()) return usage();
    if (!dir_exists(a.prefix)) {
      std::cerr << "The required input directory " << a.prefix << " does not seem to exist.\n";
      return EXIT_FAILURE;
    }
    return construct_main(a);
  }
  if (a.cmd == "query") {
    if (a.prefix.empty() || !a.have_type || a.region.empty() || !a.have_mode) return usage();
    return query_main(a);
  }
  if (a.cmd == "draw") {
    if (a.prefix.empty() || a.region.empty() || !a.have_hops) return usage();
    return draw_main(a);
  }
  return usage();
}
