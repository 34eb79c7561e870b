// engine.hip -- host side of the C ABI (include/variantstore_hip.h).
//
// Owns the decoded index (HostGraph), its flattened image (HostImage), the HBM
// copy (DevImage), one HIP stream per handle and a small device-buffer pool so
// that steady-state batches do no hipMalloc.  Mirrors the reference's seam
// between query_main and query.h (reference src/commands.cc:114-215).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../../include/variantstore_hip.h"
#include "../host/host_graph.hpp"
#include "../host/builder.hpp"
#include "../host/vcf.hpp"
#include "../host/synth.hpp"
#include "../host/device_image.hpp"
#include "../host/index_files.hpp"
#include "../host/dot_graph.hpp"
#include "kernels.hip.h"

using namespace vsamd;

static thread_local std::string g_last_error;
static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(VS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define VS_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != VS_OK) return rc_; \
  } while (0)

struct DevBuf { void* p; size_t cap; };

// Switches of one handle.  The environment is read ONCE, when the handle is opened (read_env_opts); afterwards only
// vs_index_set_option changes them -- no getenv on any query path.
struct EngineOpts {
  // ---- the production switches (vs_index_set_option; eight keys) ----
  int server = 1;               // "latency_server": 0 never use the resident server, 1 start it for a back-to-back streak of small queries
                                // only (default), 2 start it with the first small query
  unsigned srv_blocks = 16;     // "server_blocks": grid of the resident server (>= 1)
  bool share_lists = true;      // "share_lists": type-6 batches of more than 64 regions hold one row and one carrier list per covered site, shared
                                // by the regions that report it; type-4 / 5 batches one list per reported vertex
  bool resident_lists = false;  // "resident_lists": carrier lists expanded once into an arena that stays with the index (build_resident_lists)
  bool async_submit = true;     // "async_submit": a type-6 batch of more than 64 regions returns once it is ENQUEUED (its sizes are known -- or speculated: t6_speculate --, its
                                // buffers allocated, its last kernel launched); everything that reads the result is ordered behind it on the
                                // handle's stream, so callers see no difference except that the host is free while the GPU works
  bool async_fill = false;      // "async_fill": the carrier expansion of a type-6 batch runs on a second stream and the call returns while it
                                // is in flight (every accessor of the result waits for it): the next batch's plan and rows run beside it
  int t4_walk = 2;              // "t4_walk": the walk of query type 4 -- 2 cooperative (8 lanes per region, episodes in parallel; default),
                                // 1 one lane per region jumping over uneventful ref-path runs, 0 literal (every vertex of the sample's path)
  bool t6_speculate = true;     // "t6_speculate": a type-6 batch that returns when it is enqueued does not wait for its plan's totals either -- table and
                                // arena are sized from the handle's previous batch (+ 1/8), the kernels read the totals on the device, and a batch
                                // that does not fit is refused there and redone with the exact sizes when its result is first asked for anything
  bool force_fallbacks = false; // "force_fallbacks": the count-then-emit pairs of walks that query types 2 - 5 fall back to when a region
                                // outgrows the capacity of its recording walk (tests of those paths)
  // ---- read from the environment when the handle is opened ----
  bool phase_events = false;    // walking batches record all five phase events (vs_index_last_timing's phases); default: first and last only
  bool no_t4_events = false;    // VS_T4_NO_EVENTS: do not build the event bitmaps at all
  int plan_stream_priority = 0; // VS_PLAN_STREAM_PRIORITY: high (+1) / low (-1): the priority of the stream the plan of a type-6 batch and the part of a walking
                                // batch in front of its host wait run on (0: default priority)
  uint32_t plan_items = 1;      // VS_PLAN_ITEMS: regions per thread of the plan's kernels for batches of 64 k regions and more (fewer, longer waves beside
                                // the previous batch's expansion: what the plan costs the expansion is the wave slots its waves hold)
  bool t4_exact_rows = false;   // VS_T4_EXACT_ROWS: an explicit-id cohort gets exact per-sample rows (a bit per slot and a hold row) instead of round 4's
                                // coarse event rows (a bit per 8 slots, hold tests from the carrier lists) when they fit the budget
  // ---- tuning builds only (VS_TUNING: VS_BUILD_TUNING=1 python -m variantstore_amd.build --force) ----
  bool lat_debug = false;       // device-clock stamps of the latency kernels on stderr
  bool fill_fused = true;       // shared batches: the expansion writes the shared rows as well (k_fill_sites2); false: k_share_rows2 + k_fill_sites
  uint32_t fill_chunk = 0;      // rows per task of the expansion: 0 = by the batch's shape, else 8 / 16 / 32 / 64
  int fill_mode = 0;            // shared expansion: 0 one launch, 2 split (lists + rows, then the dense sites: what a profiler wants to see apart)
  uint32_t fill_dense_k = 16;   // dense sites per wave of k_fill_dense: 8 / 16 / 32 / 64
  bool fill_stats = false;      // device-clock ticks per phase of the expansion's tasks (k_fill_sites2)
  int sc_group = 1;             // lanes per region of the walks of query types 2 / 3 / 5: 1, or 8 lanes running the same chain
  bool walk_stats = false;      // iteration counts and device-clock ticks of k_sample_walk
  uint32_t fill_ablate = 0;     // skip a regime of the expansion
  size_t fill_lds_pad = 0;      // pad the fill kernel's LDS block (occupancy experiments)
};


struct vs_index {
  HostGraph g;
  HostImage im;
  DevImage d{};
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<void*> image_allocs;
  uint64_t device_bytes = 0;
  uint64_t t4_rows_bytes = 0;   // of which: the per-sample event and hold rows of query type 4
  uint64_t t4_alloc_bytes = 0;   // what build_t4_rows added to device_bytes (drop_t4_rows takes it off again)
  uint64_t pool_mallocs = 0, pool_frees = 0;   // hipMalloc / hipFree calls of the batch-buffer pool (vs_index_info)
  std::string seq_chars;
  std::unordered_map<std::string, uint32_t> sample_ids;
  std::vector<DevBuf> pool;
  // [V] list claims of the walking query types (kernels.hip.h: k_t4_claim), generation-stamped.  TWO tables, taken in turn: the claims of
  // a batch are made in front of its host wait on the second stream while the emitter of the batch before may still be reading its own
  // on the handle's stream; a table's next user (two batches on) waits for the completion event of its last (t4_claim_done).
  unsigned long long* t4_claim[2] = {nullptr, nullptr};
  hipEvent_t t4_claim_done[2] = {nullptr, nullptr};
  uint64_t t4_gen = 0;
  hipStream_t plan_stream = nullptr;        // the plan of an async_submit batch runs here, beside the previous batch's expansion on `stream`
  hipStream_t fill_stream = nullptr;        // second stream: the expansion of an async_fill batch
  hipEvent_t fill_ev[2] = {nullptr, nullptr};
  bool sort_hint = false;                   // the last shared batch arrived unsorted and was sorted on the device
  uint32_t sort_probe_in = 0;
  uint64_t seq_cap_hint[2] = {0, 0};        // the same for the piece lists of query types 2 / 3
  uint64_t walk_cap_hint[2] = {0, 0};       // scratch entries the last type-4 / type-5 batch's recording walk needed (+ 1/8): the next batch's allocation
  // A hint shrinks only after kHintShrinkAfter CONSECUTIVE batches that needed less than a quarter of it, and then to the largest of
  // those: a handle that alternates large and small batches keeps the large scratch (one host wait per batch) instead of having every
  // large batch refused on the device and redone.  [0 / 1]: walk hints, [2 / 3]: sequence hints.
  uint32_t hint_small_runs[4] = {0, 0, 0, 0};
  uint64_t hint_small_max[4] = {0, 0, 0, 0};
  // speculative type-6 batches (run_type6_shared, result_sizes): what the last shared batch needed (+ 1/8) and how many regions it had;
  // a ring of totals mailboxes in mapped host memory (kPinPlanRing), one per batch in flight, and who is still to read which
  uint64_t t6_hint_rows = 0, t6_hint_arena = 0, t6_hint_n = 0;
  uint32_t t6_small_runs = 0;
  uint64_t t6_small_rows = 0, t6_small_arena = 0;
  static constexpr int kPlanSlots = 8;
  vs_result* plan_slot_owner[kPlanSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t plan_slot_next = 0;
  uint64_t t6_speculated = 0, t6_refused = 0;   // batches submitted without a host wait / of those, refused on the device and redone (vs_index_info)
  uint64_t share_seq = 0;                   // sequence number of the plan's totals mailbox
  uint64_t done_seq = 0;                    // sequence number of the batch completion word
  uint64_t* walk_words = nullptr;           // the flag words of the walking batches (batch_words): zero between batches
  bool walk_words_dirty = true;
  hipStream_t pre = nullptr;                // the stream the part of a walking batch IN FRONT of its host wait is on (PreStream below); NULL: the handle's
  bool batch_in_flight = false;             // a batch returned when it was enqueued (async_submit) and nothing has synchronised the stream since
  hipEvent_t tev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // the events collect_timing reads: the handle's own (ev[]) or, for a
                                            // lean batch, the result's pair around its expansion (timing_owner)
  vs_result* timing_owner = nullptr;
  bool timing_total_only = false;           // ... and recorded only its first and last event (query types 2 / 3)
  bool timing_pending = false;              // the last batch returned before its kernels had finished: vs_index_last_timing reads the events then
  uint64_t timing_fill_launches = 0;
  std::vector<hipEvent_t> ev_pool;          // timing events of results (two per shared batch), reused between results
  // resident carrier lists (DevImage::v_abegin): the arena, its length in entries, and -- once a caller has asked for
  // carriers on the host -- its page-locked host mirror, shared by every result of the handle
  void* res_arena = nullptr;
  uint64_t res_entries = 0, res_site_entries = 0;
  void* res_mirror = nullptr;
  std::vector<DevBuf> pin_pool;        // page-locked host buffers of raw result copies (vs_result_get_raw), reused between results
  hipStream_t copy_stream = nullptr;   // device-to-host copies of the streamed form run here, beside the next chunk's kernels
  vs_timing timing{};
  vs_construct_stats cstats{};
  uint64_t live_results = 0;
  uint64_t live_comms = 0;     // communicators (vs_comm) made on this handle: they keep its device and stream in use
  bool close_pending = false;  // vs_index_close was called while results or communicators were alive
  unsigned long long* done_counter = nullptr;   // device word of the latency path's completion mailbox
  uint64_t lat_seq = 0;
  // resident query server of the latency path (kernels.hip.h: k_query_server)
  hipStream_t srv_stream = nullptr;
  bool srv_alive = false;
  std::chrono::steady_clock::time_point srv_started{}, srv_last{};
  uint64_t srv_seq = 1;                          // sequence number of the next request
  unsigned long long* srv_counter = nullptr;     // device word: blocks done with the current request
  std::vector<uint64_t> h_carpre;   // host copy of DevImage::s_carpre (arena prefix): sizes of the latency path's results
  EngineOpts opts;
  // back-to-back detection for opts.server == 1: small queries that followed the previous one within kSrvStreakGap
  uint32_t small_streak = 0;
  std::chrono::steady_clock::time_point last_small_done{};
  uint64_t* pinned = nullptr;  // 16 KiB of mapped host memory (uint64 words): latency-path mailboxes, server request, batch totals
  static constexpr size_t kPinTotals = 0;     // [0..2] latency path, lat_debug only: device-clock durations (kernel, bounds, tasks)
  static constexpr size_t kPinFlag = 6;       // latency path: completion sequence number
  static constexpr size_t kPinSrvResp = 16;   // [16..20] resident server: sequence number of the last request answered (+ debug stamps)
  static constexpr size_t kPinSrvReq = 256;   // [256 .. 256 + 136) resident server: the request (ServerRequest, 64-byte aligned)
  static constexpr size_t kPinBatch = 1040;   // [1040..1041] throughput path, private rows: slots and arena entries of the batch
  static constexpr size_t kPinPlan = 1056;    // [1056..1063] throughput path, shared rows: PlanTotals of the batch (k_t6_apply), sequence word last
  static constexpr size_t kPinDone = 1072;    // completion word of a batch (k_post_done)
  static constexpr size_t kPinWords = 1080;   // [1080..1083] three device words + sequence (k_post_words: read_device_words)
  static constexpr size_t kPinPlanRing = 1104;   // [1104 .. 1104 + 8 x 8) PlanTotals of the speculative batches in flight (kPlanSlots mailboxes)
};

struct vs_result {
  vs_index* idx = nullptr;
  DevResult d{};
  std::vector<DevBuf> bufs;
  // host copies
  bool have_headers = false, have_carriers = false;
  std::vector<uint8_t> h_flags;
  std::vector<uint64_t> h_var_begin, h_nvar, h_var_count, h_car_base;   // per region: first table row, rows, variants reported, arena offset
  std::vector<VariantRow> h_rows;           // the variant table as it lies in HBM (rows may be shared between regions)
  // the host VIEW: every region's rows expanded back to back (slot = region-major row), carrier lists back to back
  std::vector<uint64_t> h_view_begin, h_pos, h_car_begin, h_car_begin_view;   // h_car_begin: arena offsets per slot
  std::vector<uint32_t> h_ref_off, h_ref_len, h_alt_off, h_alt_len, h_vflags, h_car_count, h_carriers;
  uint64_t n_variants = 0, n_carriers_kept = 0, n_bases = 0, n_view_carriers = 0;
  bool have_totals = false;
  std::string text;
  std::vector<uint32_t> slice_carriers;
  // per-region arrays alone (flags, slot and arena offsets, counts): all that totals and single-region formatting need
  bool have_meta = false;
  bool shared_lists = false;          // rows and carrier lists shared between the regions of the batch (DevResult::q_car_len valid)
  uint64_t n_unique_sites = 0;        // lists actually expanded: unique covered sites when shared, else the rows
  uint64_t n_rows_reported = 0;       // rows over all regions (shared rows counted once per region that reports them)
  // async_fill: the expansion of this result is (or was) in flight on the handle's second stream
  bool pending = false;
  hipEvent_t ev_fill[2] = {nullptr, nullptr};   // around the expansion, on the stream it runs on
  hipEvent_t ev_done = nullptr;       // behind the LAST kernel of a batch that returned when it was enqueued (async_submit): nothing of the result
                                      // -- its buffers, the call's temporaries it keeps -- goes back to the pool before this has happened
  float fill_ms = -1.0f;
  // a SPECULATIVE type-6 batch: d.A / d.S hold what was ALLOCATED until somebody asks for the result's sizes (result_sizes)
  bool sizes_pending = false, totals_captured = false;
  int plan_slot = -1;
  uint64_t plan_seq = 0, cap_rows = 0, cap_arena = 0;
  PlanTotals plan_copy{};
  bool resident = false;              // the carrier arena is the index's (vs_index::res_arena), not this result's
  bool scattered_lists = false;       // lists shared per vertex (walking query types): a region's carriers are not one arena range
  std::vector<uint64_t> h_car_len;
  // the rows of ONE region, fetched when the whole table is not on the host (vs_result_format_region)
  std::vector<VariantRow> sl_rows;
  // the RAW host copy (vs_result_get_raw): variant table and arena exactly as they lie in HBM, in page-locked memory
  DevBuf raw_pin{nullptr, 0};
  std::vector<DevBuf> old_pins;   // earlier raw copies of this result (rows only, then rows + carriers): pointers handed out stay valid until it is freed
  const VariantRow* raw_rows = nullptr;
  const uint8_t* raw_arena = nullptr;   // NULL: carriers not copied
  int kind = 0;  // 7: samples_has_var result (vs_result_format_region writes the sample line); 2 / 3: sequences
  // sequence results (query types 2 and 3)
  DevSeqResult sq{};
  uint64_t seq_bytes = 0;
  bool have_seq = false;
  std::vector<uint64_t> h_byte_begin;
  std::vector<uint8_t> h_chars;
};

#define VS_NOT_SEQ(r) \
  if ((r)->kind == 2 || (r)->kind == 3) return fail(VS_ERR_ARG, "a sequence result (query types 2/3) has no variant table; use vs_result_get_sequences")

// ------------------------------------------------------------------ helpers
static int server_stop(vs_index* idx);   // the resident latency server must be gone before anything that synchronises the
                                         // whole device (hipFree does): it would otherwise wait for the server's idle clock

static int dev_alloc(vs_index* idx, size_t bytes, void** out, std::vector<DevBuf>* owner) {
  if (bytes == 0) bytes = 8;
  bytes = (bytes + 255) & ~(size_t)255;
  // best fit from the pool
  int best = -1;
  for (size_t i = 0; i < idx->pool.size(); ++i)
    if (idx->pool[i].cap >= bytes && (best < 0 || idx->pool[i].cap < idx->pool[best].cap)) best = (int)i;
  if (best >= 0 && idx->pool[best].cap <= 2 * bytes + (256 << 10)) {   // never more than twice (+256 KiB) what was asked for
    DevBuf b = idx->pool[best];
    idx->pool.erase(idx->pool.begin() + best);
    *out = b.p;
    if (owner) owner->push_back(b);
    return VS_OK;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  idx->pool_mallocs++;
  if (e != hipSuccess) {
    // release the pool and retry once
    (void)server_stop(idx);
    for (auto& b : idx->pool) { (void)hipFree(b.p); idx->pool_frees++; }
    idx->pool.clear();
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(VS_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  *out = p;
  if (owner) owner->push_back(DevBuf{p, bytes});
  return VS_OK;
}

template <typename T>
static int upload_image(vs_index* idx, const std::vector<T>& v, const T** dptr) {
  size_t bytes = v.size() * sizeof(T);
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes + 64));   // (slack: k_copy_segments reads 16 bytes at a time, up to 15 beyond a piece)
  idx->image_allocs.push_back(p);
  idx->device_bytes += bytes;
  if (bytes) HIP_TRY(hipMemcpyAsync(p, v.data(), bytes, hipMemcpyHostToDevice, idx->stream));
  *dptr = (const T*)p;
  return VS_OK;
}
template <typename T>
static int alloc_image(vs_index* idx, size_t n, T** dptr) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, n ? n * sizeof(T) : 8));
  idx->image_allocs.push_back(p);
  idx->device_bytes += n * sizeof(T);
  *dptr = (T*)p;
  return VS_OK;
}

// Walking batches (query types 4, 5, 2, 3) under async_submit run everything in front of their one host wait -- copies, capacities,
// scans, the recording walk, the claims -- on the handle's SECOND stream (vs_index::plan_stream), beside the rows and the carrier
// expansion of the batch before, which are on the handle's stream; the host has seen that part's last kernel post its words before
// it enqueues the rest, so no event links the two.  What the part reads is the caller's input and the index, what it writes is the
// batch's own (a batch that returned when it was enqueued keeps its temporaries until its completion event) -- but for the claim
// tables of the type-4 lists, which belong to the handle: two, taken in turn (vs_index::t4_claim).
// The helpers below launch on work_stream(idx): the second stream while a PreStream guard is alive, else the handle's.
static inline hipStream_t work_stream(const vs_index* idx) { return idx->pre ? idx->pre : idx->stream; }
struct PreStream {
  vs_index* idx;
  PreStream(vs_index* i, hipStream_t s) : idx(i) { i->pre = s; }
  // (left without done(): an error on the way -- whatever was launched has to be over before the caller releases the result's buffers,
  //  which nothing on the handle's own stream is ordered behind)
  ~PreStream() { if (idx->pre) (void)hipStreamSynchronize(idx->pre); idx->pre = nullptr; }
  void done() { idx->pre = nullptr; }
};

// out[0..n) = exclusive prefix of in, out[n] = total
template <typename T>
static int exclusive_scan(vs_index* idx, const T* in, uint64_t n, uint64_t* out, std::vector<DevBuf>* scratch_owner) {
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(out, 0, 8, work_stream(idx)));
    return VS_OK;
  }
  if (n <= kScanSmallMax && (const void*)in != (const void*)out) {   // one launch instead of three
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_small<T>), dim3((unsigned)((n + kScanSmallTile - 1) / kScanSmallTile)), dim3(kScanSmallBlock), 0, work_stream(idx), in, (uint32_t)n, out);
    HIP_TRY(hipGetLastError());
    return VS_OK;
  }
  const uint64_t ntiles = (n + kScanTile - 1) / kScanTile;
  void* ts = nullptr;
  VS_TRY(dev_alloc(idx, ntiles * 8, &ts, scratch_owner));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_tile_sums<T>), dim3((unsigned)ntiles), dim3(kScanBlock), 0, work_stream(idx), in, n, (uint64_t*)ts);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(kScanBlock), 0, work_stream(idx), (uint64_t*)ts, ntiles, out + n);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<T>), dim3((unsigned)ntiles), dim3(kScanBlock), 0, work_stream(idx), in, n, (const uint64_t*)ts, out);
  HIP_TRY(hipGetLastError());
  return VS_OK;
}

// var_begin / car_base of a batch in one pass (k_scan2_*): out[0..n) = exclusive prefixes, out[n] = totals, which the
// spine kernel also writes to `host_totals` (mapped host memory; NULL = not wanted) -- no staged device-to-host copy.
static int scan_offsets(vs_index* idx, const uint64_t* nvar, const uint64_t* ncar, uint64_t n, uint64_t* var_begin, uint64_t* car_base,
                        uint64_t* host_totals, std::vector<DevBuf>* scratch_owner) {
  if (n <= kScanSmallMax) {   // one launch instead of three
    hipLaunchKernelGGL(k_scan2_small, dim3((unsigned)std::max<uint64_t>(1, (n + kScanSmallTile - 1) / kScanSmallTile)), dim3(kScanSmallBlock), 0, work_stream(idx), nvar, ncar,
                       (uint32_t)n, var_begin, car_base, host_totals);
    HIP_TRY(hipGetLastError());
    return VS_OK;
  }
  const uint64_t ntiles = n ? (n + kScanTile - 1) / kScanTile : 0;
  void* ts = nullptr;
  VS_TRY(dev_alloc(idx, (ntiles + 1) * sizeof(Scan2), &ts, scratch_owner));
  if (ntiles) hipLaunchKernelGGL(k_scan2_tile_sums, dim3((unsigned)ntiles), dim3(kScanBlock), 0, work_stream(idx), nvar, ncar, n, (Scan2*)ts);
  hipLaunchKernelGGL(k_scan2_spine, dim3(1), dim3(kScanBlock), 0, work_stream(idx), (Scan2*)ts, ntiles, var_begin + n, car_base + n, host_totals);
  if (ntiles) hipLaunchKernelGGL(k_scan2_apply, dim3((unsigned)ntiles), dim3(kScanBlock), 0, work_stream(idx), nvar, ncar, n, (const Scan2*)ts, var_begin, car_base);
  HIP_TRY(hipGetLastError());
  return VS_OK;
}

static void release_bufs(vs_index* idx, std::vector<DevBuf>& bufs) {
  for (auto& b : bufs) idx->pool.push_back(b);
  bufs.clear();
  // keep the pool bounded: drop the smallest buffers beyond kPoolEntries.  (hipFree waits for the whole device: the bound
  // has to sit well above what a loop of batches in flight cycles through -- a type-6 batch that returned when it was
  // enqueued holds ~20 buffers until it is freed, two of them are alive at a time, and buffers of other batch shapes stay
  // around -- or every step frees and re-allocates its small buffers and stalls on the expansion in flight: 64 entries
  // made the bench loop take 2.1 ms per step instead of 0.61 on some runs.)
  constexpr size_t kPoolEntries = 320;
  if (idx->pool.size() > kPoolEntries) (void)server_stop(idx);
  while (idx->pool.size() > kPoolEntries) {
    size_t k = 0;
    for (size_t i = 1; i < idx->pool.size(); ++i)
      if (idx->pool[i].cap < idx->pool[k].cap) k = i;
    (void)hipFree(idx->pool[k].p);
    idx->pool_frees++;
    idx->pool.erase(idx->pool.begin() + k);
  }
}

// Page-locked host memory for raw result copies: allocating it costs as much as copying into it, so buffers go back to a
// small per-handle pool when their result is freed.
static int pin_alloc(vs_index* idx, size_t bytes, DevBuf* out) {
  bytes = (bytes + 4095) & ~(size_t)4095;
  int best = -1;
  for (size_t i = 0; i < idx->pin_pool.size(); ++i)
    if (idx->pin_pool[i].cap >= bytes && (best < 0 || idx->pin_pool[i].cap < idx->pin_pool[best].cap)) best = (int)i;
  if (best >= 0) { *out = idx->pin_pool[best]; idx->pin_pool.erase(idx->pin_pool.begin() + best); return VS_OK; }
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  *out = DevBuf{p, bytes};
  return VS_OK;
}
static void pin_release(vs_index* idx, DevBuf b) {
  if (!b.p) return;
  idx->pin_pool.push_back(b);
  while (idx->pin_pool.size() > 4) {   // keep the largest few
    size_t k = 0;
    for (size_t i = 1; i < idx->pin_pool.size(); ++i)
      if (idx->pin_pool[i].cap < idx->pin_pool[k].cap) k = i;
    (void)hipHostFree(idx->pin_pool[k].p);
    idx->pin_pool.erase(idx->pin_pool.begin() + k);
  }
}

// Temporaries of one call.  Every early return (VS_TRY / HIP_TRY) runs the destructor: the stream is drained first
// -- queued kernels may still be using the buffers -- and then they go back to the pool, so a failing call on a
// long-lived handle neither leaks HBM nor recycles memory that is still in flight.  The success path calls
// release() itself after its own synchronisation.
struct ScratchBufs {
  vs_index* idx;
  std::vector<DevBuf> bufs;
  explicit ScratchBufs(vs_index* i) : idx(i) {}
  ScratchBufs(const ScratchBufs&) = delete;
  ScratchBufs& operator=(const ScratchBufs&) = delete;
  void release() { release_bufs(idx, bufs); }
  ~ScratchBufs() {
    if (bufs.empty()) return;
    if (idx->plan_stream) (void)hipStreamSynchronize(idx->plan_stream);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    release_bufs(idx, bufs);
  }
};

// ----------------------------------------------------------- image on device
// ---- event bitmaps of query type 4: one row of P bits per sample (3 GB for 2504 samples x 9.6 M ref-path slots; HBM is
//      what this part has plenty of).  Skipped when they would take more than half of the free memory or `cap` bytes; dropped
//      and rebuilt on an open handle by option t4_rows_max_mb (several handles on one GPU: the caller decides who gets them --
//      without the rows type 4 walks one lane per region). ----
static void free_image_alloc(vs_index* idx, const void* p) {
  if (!p) return;
  auto it = std::find(idx->image_allocs.begin(), idx->image_allocs.end(), const_cast<void*>(p));
  if (it != idx->image_allocs.end()) idx->image_allocs.erase(it);
  (void)hipFree(const_cast<void*>(p));
}
constexpr uint32_t kHintShrinkAfter = 8;
static void hint_after_batch(vs_index* idx, int which, uint64_t& hint, uint64_t cap_seen) {
  if (cap_seen >= hint / 4) { idx->hint_small_runs[which] = 0; idx->hint_small_max[which] = 0; return; }
  idx->hint_small_max[which] = std::max(idx->hint_small_max[which], cap_seen);
  if (++idx->hint_small_runs[which] >= kHintShrinkAfter) {
    const uint64_t m = idx->hint_small_max[which];
    hint = m + m / 8 + 1024;
    idx->hint_small_runs[which] = 0; idx->hint_small_max[which] = 0;
  }
}
static int drop_t4_rows(vs_index* idx) {
  DevImage& d = idx->d;
  if (!d.t4_events) return VS_OK;
  HIP_TRY(hipDeviceSynchronize());   // (batches that returned when they were enqueued may still be walking over the rows)
  free_image_alloc(idx, d.t4_events); free_image_alloc(idx, d.t4_hold); free_image_alloc(idx, d.t4_irr);
  idx->device_bytes -= idx->t4_alloc_bytes;
  idx->t4_alloc_bytes = 0; idx->t4_rows_bytes = 0;
  d.t4_events = nullptr; d.t4_stride = 0; d.t4_hold = nullptr; d.t4_hold_stride = 0; d.t4_irr = nullptr; d.t4_ev_shift = 0; d.t4_irr_reach = 1;
  return VS_OK;
}
static int build_t4_rows(vs_index* idx, uint64_t cap) {
  const HostImage& im = idx->im;
  DevImage& d = idx->d;
  if (d.t4_events) return VS_OK;
  if (im.slots_follow_ranks && im.P && d.num_samples > 1) {
    // Class-row cohorts: a bit per slot and sample, a bit per vertex and sample (3.0 + 4.6 GB at 2504 samples).  Explicit-id
    // cohorts (thousands of samples, a handful of carriers per variant): one bit per EIGHT slots -- any superset of the
    // events is exact, a coarse bit costs a few literal steps where the sample does have an event, and those are rare -- and
    // no hold rows at all: "does v hold the sample" is read from v's carrier list (k_walk.hip.h: BitRow).  10,000 samples x
    // 20 M variants: 6.3 GB instead of round 3's 125 GB.
    const uint64_t istride = (im.P + 63) / 64 + 1;                                   // the global irregular row: a bit per slot
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    // Round 6, opt-in (VS_T4_EXACT_ROWS=1 when the handle is opened): an explicit-id cohort takes the EXACT rows too when they fit the
    // budget (10,000 samples x 20 M variants: 48 + 73 GB of the part's 288): hold tests become one bit, the spill-free type-4 walk and the
    // cooperative kernels of types 2 / 3 / 5 apply.  Measured on that cohort: type 4 160 -> 172 M regions/s, types 2 / 3 / 5 unchanged
    // (82 / 83 / 101 -> 84 / 84 / 109) for 116 GB more image -- not the default.
    const uint64_t exact_bytes = (uint64_t)d.num_samples * (istride + (im.V + 63) / 64 + 1) * 8;
    const bool exact = d.use_bv || (idx->opts.t4_exact_rows && exact_bytes <= cap && exact_bytes <= free_b / 2);
    const uint32_t shift = exact ? 0u : 3u;
    const uint64_t stride = exact ? istride : ((im.P >> shift) + 64) / 64 + 1, hstride = exact ? (im.V + 63) / 64 + 1 : 0;
    const uint64_t bytes = (uint64_t)d.num_samples * stride * 8, hbytes = (uint64_t)d.num_samples * hstride * 8;
    if (bytes + hbytes <= cap && bytes + hbytes <= free_b / 2) {
      const uint64_t bytes_before = idx->device_bytes;
      uint64_t *events = nullptr, *hold = nullptr, *irr = nullptr;
      // every allocation and launch first, the handle's fields last: a failure half way (the check against the free memory is not atomic with
      // the allocations) gives back what it took and leaves the handle without rows, as if none had been asked for
      auto build = [&]() -> int {
        VS_TRY(alloc_image(idx, (size_t)istride, &irr));
        HIP_TRY(hipMemsetAsync(irr, 0, istride * 8, idx->stream));
        VS_TRY(alloc_image(idx, (size_t)d.num_samples * stride, &events));
        HIP_TRY(hipMemsetAsync(events, 0, bytes, idx->stream));
        if (hstride) {
          VS_TRY(alloc_image(idx, (size_t)d.num_samples * hstride, &hold));
          HIP_TRY(hipMemsetAsync(hold, 0, hbytes, idx->stream));
        }
        DevImage dd = d;   // (the build kernels read the strides from their copy of the image)
        dd.t4_stride = stride; dd.t4_hold_stride = hstride; dd.t4_ev_shift = shift;
        const unsigned tiles = (unsigned)((im.P + 63) / 64), vtiles = (unsigned)((im.V + 63) / 64);
        if (d.use_bv) {
          hipLaunchKernelGGL(k_build_events, dim3((tiles + 3) / 4), dim3(256), 0, idx->stream, dd, events, irr);
          hipLaunchKernelGGL(k_build_hold, dim3((vtiles + 3) / 4), dim3(256), 0, idx->stream, dd, hold);
        } else {
          hipLaunchKernelGGL(k_events_irregular_rows, dim3((tiles + 3) / 4), dim3(256), 0, idx->stream, dd, irr);
          hipLaunchKernelGGL(k_events_explicit, dim3((unsigned)((im.P + 255) / 256)), dim3(256), 0, idx->stream, dd, events);
          if (hstride) hipLaunchKernelGGL(k_hold_explicit, dim3((unsigned)((im.V + 255) / 256)), dim3(256), 0, idx->stream, dd, hold);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(idx->stream));
        return VS_OK;
      };
      const int rc = build();
      if (rc != VS_OK) {
        (void)hipStreamSynchronize(idx->stream);
        free_image_alloc(idx, events); free_image_alloc(idx, hold); free_image_alloc(idx, irr);
        idx->device_bytes = bytes_before;
        return rc;
      }
      d.t4_stride = stride; d.t4_hold_stride = hstride; d.t4_ev_shift = shift;
      d.t4_events = events; d.t4_hold = hold; d.t4_irr = irr; d.t4_irr_reach = im.irr_reach;
      idx->t4_rows_bytes = bytes + hbytes;
      idx->t4_alloc_bytes = idx->device_bytes - bytes_before;
    }
  }
  HIP_TRY(hipStreamSynchronize(idx->stream));
  return VS_OK;
}

static int build_device_image(vs_index* idx) {
  HostImage& im = idx->im;
  DevImage& d = idx->d;
  HIP_TRY(hipSetDevice(idx->device));
  HIP_TRY(hipStreamCreate(&idx->stream));
  for (auto& e : idx->ev) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(hipHostMalloc((void**)&idx->pinned, 16384, hipHostMallocCoherent | hipHostMallocMapped));
  memset(idx->pinned, 0, 16384);
  HIP_TRY(hipMalloc((void**)&idx->done_counter, 32));   // [0] blocks finished, [1..3] device time stamps of the latency kernel
  idx->image_allocs.push_back(idx->done_counter);
  HIP_TRY(hipMemsetAsync(idx->done_counter, 0, 32, idx->stream));
  d.ref_length = im.ref_length;
  d.nbits = (uint64_t)im.bits.size() * 64;
  d.num_samples = im.num_samples; d.wpc = im.wpc; d.use_bv = im.use_bit_vector;
  d.V = im.V; d.E = im.E; d.P = im.P; d.R = im.R; d.C = im.C;
  d.G = im.rp_cand_prefix[im.P];
  VS_TRY(upload_image(idx, im.bits, &d.bits));
  VS_TRY(upload_image(idx, im.blk_rank, &d.blk_rank));
  VS_TRY(upload_image(idx, im.idx_pos, &d.idx_pos));
  VS_TRY(upload_image(idx, im.rank_to_slot, &d.rank_to_slot));
  VS_TRY(upload_image(idx, im.rp_vid, &d.rp_vid));
  VS_TRY(upload_image(idx, im.rp_cand_prefix, &d.rp_cand_prefix));
  VS_TRY(upload_image(idx, im.row_ptr, &d.row_ptr));
  VS_TRY(upload_image(idx, im.col, &d.col));
  VS_TRY(upload_image(idx, im.v_off, &d.v_off));
  VS_TRY(upload_image(idx, im.v_len, &d.v_len));
  VS_TRY(upload_image(idx, im.v_ridx, &d.v_ridx));
  VS_TRY(upload_image(idx, im.v_class, &d.v_class));
  VS_TRY(upload_image(idx, im.v_src, &d.v_src));
  {
    const uint32_t *wv = nullptr, *we = nullptr;
    VS_TRY(upload_image(idx, im.w_vertex, &wv));
    VS_TRY(upload_image(idx, im.w_edge, &we));
    d.w_vertex = reinterpret_cast<const uint4*>(wv);
    d.w_edge = reinterpret_cast<const uint4*>(we);
    const uint32_t *wb = nullptr, *rb = nullptr;
    VS_TRY(upload_image(idx, im.wblob, &wb));
    VS_TRY(upload_image(idx, im.blob_of_slot, &d.blob_of_slot));
    VS_TRY(upload_image(idx, im.blob_row, &d.blob_row));
    VS_TRY(upload_image(idx, im.rk_back, &rb));
    VS_TRY(upload_image(idx, im.seq_breaks, &d.seq_breaks));
    const uint32_t* ra = nullptr;
    VS_TRY(upload_image(idx, im.rk_anc, &ra));
    d.rk_anc = reinterpret_cast<const uint2*>(ra);
    VS_TRY(upload_image(idx, im.slot_rank, &d.slot_rank));
    d.wblob = reinterpret_cast<const uint4*>(wb);
    d.rk_back = reinterpret_cast<const uint2*>(rb);
  }
  VS_TRY(upload_image(idx, im.v_ncar, &d.v_ncar));
  VS_TRY(upload_image(idx, im.v_nri, &d.v_nri));
  VS_TRY(upload_image(idx, im.v_car_begin, &d.v_car_begin));
  VS_TRY(upload_image(idx, im.class_rows, &d.class_rows));
  VS_TRY(upload_image(idx, im.cls_list_begin, &d.cls_list_begin));
  VS_TRY(upload_image(idx, im.cls_list_ids, &d.cls_list_ids));
  VS_TRY(upload_image(idx, im.cls_list16, &d.cls_list16));
  d.list_max = im.list_max;
  VS_TRY(upload_image(idx, im.gt_nibbles, &d.gt_nibbles));
  VS_TRY(upload_image(idx, im.gt_groups, &d.gt_groups));
  VS_TRY(upload_image(idx, im.car_sid, &d.car_sid));
  VS_TRY(upload_image(idx, im.car_index, &d.car_index));
  d.has_car_index = im.car_index.empty() ? 0u : 1u;
  d.class_cum = nullptr;
  if (d.has_car_index && d.use_bv && im.wpc && im.num_samples < 65536) {   // rank of a sample's bit in its class row without a pass over the row
    const uint64_t n_rows = im.class_rows.size() / im.wpc;
    uint16_t* cum = nullptr;
    VS_TRY(alloc_image(idx, n_rows * im.wpc, &cum));
    if (n_rows) hipLaunchKernelGGL(k_class_cum, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, idx->stream, d.class_rows, n_rows, (uint32_t)im.wpc, cum);
    HIP_TRY(hipGetLastError());
    d.class_cum = cum;
  }
  VS_TRY(upload_image(idx, im.seq_codes, &d.seq_codes));
  const uint64_t G = d.G;
  VS_TRY(alloc_image(idx, G, &d.s_pos));
  VS_TRY(alloc_image(idx, G, &d.s_ref_off));
  VS_TRY(alloc_image(idx, G, &d.s_ref_len));
  VS_TRY(alloc_image(idx, G, &d.s_alt_off));
  VS_TRY(alloc_image(idx, G, &d.s_alt_len));
  VS_TRY(alloc_image(idx, G, &d.s_vid));
  VS_TRY(alloc_image(idx, G, &d.s_ncar));
  VS_TRY(alloc_image(idx, G, &d.s_flags));
  VS_TRY(alloc_image(idx, G, &d.s_dup_prev));
  VS_TRY(alloc_image(idx, G + 1, &d.s_carpre));
  VS_TRY(alloc_image(idx, G + 1, &d.s_kpre));
  VS_TRY(alloc_image(idx, G, &d.s_class));
  VS_TRY(alloc_image(idx, G, &d.s_gt0));
  d.sus_g = nullptr; d.sus_prev = nullptr; d.n_sus = 0;

  // site table: the reference's per-node classification, once for the whole ref path
  if (im.P) {
    const uint64_t chunk = 1ull << 24;
    for (uint64_t s = 0; s < im.P; s += chunk) {
      uint64_t e = std::min<uint64_t>(im.P, s + chunk);
      hipLaunchKernelGGL(k_build_sites, dim3((unsigned)((e - s + 255) / 256)), dim3(256), 0, idx->stream, d, s, e);
    }
    HIP_TRY(hipGetLastError());
  }
  ScratchBufs scratch(idx);
  {  // arena offsets: prefix of the counts rounded up to the carrier alignment (kernels.hip.h: pad_car)
    void* padded = nullptr;
    VS_TRY(dev_alloc(idx, (G + 1) * 4, &padded, &scratch.bufs));
    if (G) hipLaunchKernelGGL(k_pad_counts, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, idx->stream, (const uint32_t*)d.s_ncar, (uint32_t*)padded, G);
    VS_TRY(exclusive_scan<uint32_t>(idx, (const uint32_t*)padded, G, d.s_carpre, &scratch.bufs));
    VS_TRY(exclusive_scan<uint32_t>(idx, (const uint32_t*)d.s_ncar, G, d.s_kpre, &scratch.bufs));
    {
      uint64_t *rc = nullptr, *rk = nullptr;
      VS_TRY(alloc_image(idx, im.P + 1, &rc));
      VS_TRY(alloc_image(idx, im.P + 1, &rk));
      hipLaunchKernelGGL(k_slot_prefixes, dim3((unsigned)((im.P + 256) / 256)), dim3(256), 0, idx->stream, d, rc, rk);
      HIP_TRY(hipGetLastError());
      d.rp_carpre = rc; d.rp_kpre = rk;
    }
    {
      VariantRow* rows = nullptr;
      VS_TRY(alloc_image(idx, G, &rows));
      if (G) hipLaunchKernelGGL(k_build_site_rows, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, idx->stream, d, rows);
      HIP_TRY(hipGetLastError());
      d.s_row = rows;
    }
    idx->h_carpre.resize(G + 1);
    HIP_TRY(hipMemcpyAsync(idx->h_carpre.data(), d.s_carpre, (G + 1) * 8, hipMemcpyDeviceToHost, idx->stream));
  }
  std::vector<uint32_t> h_dup(G), h_fl(G), h_ncar(G);
  if (G) {
    hipLaunchKernelGGL(k_mark_dups, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, idx->stream, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_dup.data(), d.s_dup_prev, G * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipMemcpyAsync(h_fl.data(), d.s_flags, G * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipMemcpyAsync(h_ncar.data(), d.s_ncar, G * 4, hipMemcpyDeviceToHost, idx->stream));
  }
  HIP_TRY(hipStreamSynchronize(idx->stream));
  scratch.release();
  {  // the dense sites (k_fill_dense)
    std::vector<uint32_t> dense;
    if (d.use_bv)
      for (uint64_t g = 0; g < G; ++g)
        if (h_ncar[g] > d.list_max) dense.push_back((uint32_t)g);
    d.n_dense = dense.size();
    VS_TRY(upload_image(idx, dense, &d.dense_site));
  }
  std::vector<uint32_t> sus_g, sus_prev;
  for (uint64_t g = 0; g < G; ++g) {
    if (h_fl[g] & kSiteAlwaysDrop) { sus_g.push_back((uint32_t)g); sus_prev.push_back(kNone); }
    else if (h_dup[g] != kNone) { sus_g.push_back((uint32_t)g); sus_prev.push_back(h_dup[g]); }
  }
  d.n_sus = (uint32_t)sus_g.size();
  {  // suspicious sites before each ref-path slot's first site
    std::vector<uint32_t> pre(im.P + 1, 0);
    size_t k = 0;
    for (uint64_t s = 0; s <= im.P; ++s) {
      while (k < sus_g.size() && sus_g[k] < im.rp_cand_prefix[s]) ++k;
      pre[s] = (uint32_t)k;
    }
    VS_TRY(upload_image(idx, pre, &d.rp_sus_prefix));
  }
  {  // the bounds' per-slot and per-rank records
    uint4* rp = nullptr; uint2* rk = nullptr;
    VS_TRY(alloc_image(idx, 2 * (im.P + 1), &rp));
    VS_TRY(alloc_image(idx, im.R + 1, &rk));
    hipLaunchKernelGGL(k_slot_records, dim3((unsigned)((im.P + 256) / 256)), dim3(256), 0, idx->stream, d, rp);
    hipLaunchKernelGGL(k_rank_records, dim3((unsigned)((im.R + 256) / 256)), dim3(256), 0, idx->stream, d, rk);
    HIP_TRY(hipGetLastError());
    d.rp_rec = rp; d.rk_rec = rk;
  }
  VS_TRY(upload_image(idx, sus_g, &d.sus_g));
  VS_TRY(upload_image(idx, sus_prev, &d.sus_prev));
  // ---- the per-sample rows of query type 4 (build_t4_rows below) ----
  d.t4_events = nullptr; d.t4_stride = 0; d.t4_hold = nullptr; d.t4_hold_stride = 0; d.t4_irr = nullptr; d.t4_ev_shift = 0; d.t4_irr_reach = 1;
  if (!idx->opts.no_t4_events) {
    // Budget: half of the free memory, at most 176 GB -- or VS_T4_ROWS_MAX_GB from the environment, or option t4_rows_max_mb
    // on the open handle; vs_index_get_info reports what was taken (t4_rows_bytes).
    uint64_t cap = 176ull << 30;
    if (const char* gb = getenv("VS_T4_ROWS_MAX_GB")) cap = (uint64_t)(std::max(0.0, atof(gb)) * (double)(1ull << 30));
    VS_TRY(build_t4_rows(idx, cap));
  }
  HIP_TRY(hipStreamSynchronize(idx->stream));
  return VS_OK;
}

// The environment switches of DESIGN.md section 7a, read once per handle.
static void read_env_opts(vs_index* idx) {
  EngineOpts& o = idx->opts;
  if (getenv("VS_NO_SERVER")) o.server = 0;
  else if (const char* sv = getenv("VS_SERVER")) o.server = std::max(0, std::min(2, atoi(sv)));
  if (const char* sb = getenv("VS_SRV_BLOCKS")) o.srv_blocks = (unsigned)std::max(1, std::min(64, atoi(sb)));
  if (getenv("VS_NO_SHARED_LISTS")) o.share_lists = false;
  o.async_fill = getenv("VS_ASYNC_FILL") != nullptr;
  if (getenv("VS_SYNC_SUBMIT")) o.async_submit = false;
  o.resident_lists = getenv("VS_RESIDENT_LISTS") != nullptr;   // (the arena itself is built at the end of finish_open)
  o.no_t4_events = getenv("VS_T4_NO_EVENTS") != nullptr;
  o.t4_exact_rows = getenv("VS_T4_EXACT_ROWS") != nullptr;
  if (const char* pp = getenv("VS_PLAN_STREAM_PRIORITY")) o.plan_stream_priority = !strcmp(pp, "high") ? 1 : (!strcmp(pp, "low") ? -1 : 0);
  if (const char* pi = getenv("VS_PLAN_ITEMS")) o.plan_items = (uint32_t)std::max(1, std::min(64, atoi(pi)));
  if (const char* lm = getenv("VS_LIST_MAX")) idx->im.list_max = (uint32_t)atoi(lm);   // tuning aid (default: kListMaxDefault)
#ifdef VS_TUNING
  o.lat_debug = getenv("VS_LAT_DEBUG") != nullptr;
#endif
}

static int build_resident_lists(vs_index* idx);
static int finish_open(vs_index* idx, int device) {
  try {
    read_env_opts(idx);
    build_host_image(idx->g, idx->im);
  } catch (const std::exception& e) {
    return fail(VS_ERR_FORMAT, "%s", e.what());
  }
  idx->seq_chars.resize(idx->g.seq.size());
  for (size_t i = 0; i < idx->g.seq.size(); ++i) idx->seq_chars[i] = map_int(idx->g.seq[i]);
  for (uint32_t i = 0; i < idx->g.sample_names.size(); ++i) idx->sample_ids.emplace(idx->g.sample_names[i], i);
  idx->device = device;
  if (device < 0) return VS_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VS_ERR_NO_DEVICE, "no HIP device is visible");
  if (device >= ndev) return fail(VS_ERR_NO_DEVICE, "device %d requested, %d visible", device, ndev);
  VS_TRY(build_device_image(idx));
  if (idx->opts.resident_lists) {   // VS_RESIDENT_LISTS: over budget is not an error at open, the lists are then expanded per batch
    idx->opts.resident_lists = false;
    if (build_resident_lists(idx) == VS_OK) idx->opts.resident_lists = true;
  }
  return VS_OK;
}

template <typename T>
static int ralloc(vs_result* r, size_t n, T** p) {
  void* q = nullptr;
  VS_TRY(dev_alloc(r->idx, n * sizeof(T), &q, &r->bufs));
  *p = (T*)q;
  return VS_OK;
}

// Several arrays out of ONE pooled buffer (every allocation is a search of the handle's pool, and a batch of a tenth of a
// millisecond makes sixteen of them): sizes are added up, the buffer is taken once, the arrays are carved on 256-byte bounds.
struct Slab {
  size_t bytes = 0;
  uint8_t* base = nullptr;
  size_t add(size_t b) { const size_t at = bytes; bytes += (b + 255) & ~(size_t)255; return at; }
  template <typename T> T* at(size_t off) const { return reinterpret_cast<T*>(base + off); }
};

// Strings of a type-7 batch (ref, alt per query) as one byte pool + [2n+1] offsets.
struct PointStrings {
  const std::vector<uint8_t>* chars;
  const std::vector<uint64_t>* off;
};

// Fill-kernel launch shared by every path: LDS per wave from the cohort width, WIDE from the row width.
static uint32_t fill_gt_words(const vs_index* idx) {
  uint32_t gt_words = ((std::min<uint32_t>(idx->d.num_samples, 4064) + 32 + 255) / 256) * 64;
  return gt_words < 448 ? 448 : gt_words;   // the medium path keeps 640 ids at word 256..
}
static size_t fill_lds_bytes(const vs_index* idx) {
  return 4 * (size_t)(idx->d.wpc <= 63 ? slice_lds_words(idx->d.num_samples) : fill_gt_words(idx) + kRingWords) * 4;
}

// one launch of the carrier expansion: over the rows of a private-row result or over the shared rows of a sorted batch
template <bool WIDE, uint32_t CH, bool TUNE>
static void launch_fill(vs_index* idx, const DevResult& d, bool share, const uint32_t* u_site, uint64_t n_fill, unsigned blocks, size_t lds_bytes,
                        uint32_t ablate, uint32_t gt_words, hipStream_t stream = nullptr) {
  if (!stream) stream = idx->stream;
  if (share) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_sites<WIDE, CH, TUNE>), dim3(blocks), dim3(256), lds_bytes, stream, idx->d, d, u_site, n_fill, ablate, gt_words);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_carriers<WIDE, CH, TUNE>), dim3(blocks), dim3(256), lds_bytes, stream, idx->d, d, ablate, gt_words);
}

static int ensure_fill_stream(vs_index* idx) {
  if (idx->fill_stream) return VS_OK;
  HIP_TRY(hipStreamCreateWithFlags(&idx->fill_stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&idx->fill_ev[0], hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&idx->fill_ev[1], hipEventDisableTiming));
  return VS_OK;
}
static int result_events(vs_result* r) {   // the result's own pair of timing events around its expansion, from the handle's pool
  vs_index* idx = r->idx;
  for (auto& e : r->ev_fill) {
    if (e) continue;
    if (!idx->ev_pool.empty()) { e = idx->ev_pool.back(); idx->ev_pool.pop_back(); }
    else HIP_TRY(hipEventCreate(&e));
  }
  return VS_OK;
}
static int pooled_event(vs_index* idx, hipEvent_t* e) {
  if (*e) return VS_OK;
  if (!idx->ev_pool.empty()) { *e = idx->ev_pool.back(); idx->ev_pool.pop_back(); return VS_OK; }
  HIP_TRY(hipEventCreate(e));
  return VS_OK;
}
static int ensure_plan_stream(vs_index* idx) {
  if (idx->plan_stream) return VS_OK;
  // (stream priorities, lowest and highest, were measured: the expansion beside a plan takes 0.031 ms longer than alone
  //  whatever the plan's priority -- the plan's 137 MB of scattered lines are what it shares, not wave slots)
  if (idx->opts.plan_stream_priority) {   // (tuning aid, VS_PLAN_STREAM_PRIORITY = high | low when the handle is opened)
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // (numerically: hi <= lo)
    HIP_TRY(hipStreamCreateWithPriority(&idx->plan_stream, hipStreamNonBlocking, idx->opts.plan_stream_priority > 0 ? hi : lo));
    return VS_OK;
  }
  HIP_TRY(hipStreamCreateWithFlags(&idx->plan_stream, hipStreamNonBlocking));
  return VS_OK;
}
// Regions and sample ids of the walking query types may be handed over in device memory (the copies below are
// hipMemcpyDefault: the runtime reads the direction off the pointers).
static bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // (plain host memory: not known to the runtime)
  return a.type == hipMemoryTypeDevice;
}
// The phase times of a batch from the handle's events (the batch's last event has completed or is waited for here).
static int collect_timing(vs_index* idx) {
  if (!idx->timing_pending) return VS_OK;
  idx->timing_pending = false;
  vs_timing& t = idx->timing;
  hipEvent_t* e = idx->timing_owner ? idx->tev : idx->ev;
  idx->timing_owner = nullptr;
  HIP_TRY(hipEventSynchronize(e[4]));
  HIP_TRY(hipEventElapsedTime(&t.ms_total, e[0], e[4]));
  if (idx->timing_total_only) {   // (a sequence batch records its first and last event only)
    idx->timing_total_only = false;
    t.ms_bounds = t.ms_scan = t.ms_emit = t.ms_fill = 0.f;
    t.fill_launches = 0;
    return VS_OK;
  }
  HIP_TRY(hipEventElapsedTime(&t.ms_bounds, e[0], e[1]));
  HIP_TRY(hipEventElapsedTime(&t.ms_scan, e[1], e[2]));
  HIP_TRY(hipEventElapsedTime(&t.ms_emit, e[2], e[3]));
  HIP_TRY(hipEventElapsedTime(&t.ms_fill, e[3], e[4]));
  t.fill_launches = idx->timing_fill_launches;
  return VS_OK;
}
// An asynchronous expansion (option "async_fill") must have finished before anything reads the result or returns its
// buffers to the pool.
static int result_ready(vs_result* r) {
  if (!r->pending) return VS_OK;
  r->pending = false;
  if (r->ev_done) HIP_TRY(hipEventSynchronize(r->ev_done));
  if (r->ev_fill[1]) {
    HIP_TRY(hipEventSynchronize(r->ev_fill[1]));
    HIP_TRY(hipEventElapsedTime(&r->fill_ms, r->ev_fill[0], r->ev_fill[1]));
  }
  return VS_OK;
}

// The carrier expansion of one result (or of the resident arena): one launch over n_fill rows.
static int fill_lists(vs_index* idx, const DevResult& d, bool share, const uint32_t* u_site, uint64_t n_fill, hipStream_t on = nullptr) {
  if (n_fill) {
    {
      // one task per wave, no grid-stride loop: task costs vary tenfold with the number of dense variants, and the
      // hardware's block scheduler balances that better than a static round-robin (8192-block grid: +8 % kernel time).
      // A task is 64 consecutive slots; batches of few, carrier-heavy variants (a type-4 batch: ~1 M variants of ~1300
      // carriers) take 16-slot tasks -- 17 k tasks of 64 would be two rounds of waves with a long tail.
      uint32_t chunk = idx->opts.fill_chunk;
      // (8-slot tasks for the carrier-heavy shape: 0.36 ms against 0.40 with 16 on the bench's type-4 leg, tools/run_t4.py)
      if (chunk == 0) chunk = share ? 16 : ((n_fill < 64ull * 8192 * 8 && d.S / n_fill >= 256) ? 8 : 64);
      const uint64_t nchunks = (n_fill + chunk - 1) / chunk;
      const uint64_t blocks = (nchunks + 3) / 4;
      if (blocks > 0x7FFFFFFFull) return fail(VS_ERR_ARG, "batch too large for one launch (%llu variant slots)", (unsigned long long)d.A);
      // per-wave LDS: one genotype byte per carrier of the widest variant the staged paths take, plus the ring
      const uint32_t gt_words = fill_gt_words(idx);
#ifdef VS_TUNING   // tuning builds: one regime of the kernel can be skipped, the LDS block padded (tools/exp_fill.py)
      const size_t lds_bytes = fill_lds_bytes(idx) + std::min<size_t>(idx->opts.fill_lds_pad, 96 << 10);
      const uint32_t ablate = idx->opts.fill_ablate;
      constexpr bool kTune = true;
#else
      const size_t lds_bytes = fill_lds_bytes(idx);
      const uint32_t ablate = 0;
      constexpr bool kTune = false;
#endif
      const bool wide = idx->d.wpc > 63;
      switch (chunk) {
        case 8:  wide ? launch_fill<true, 8, kTune>(idx, d, share, u_site, n_fill, (unsigned)blocks, lds_bytes, ablate, gt_words, on)
                      : launch_fill<false, 8, kTune>(idx, d, share, u_site, n_fill, (unsigned)blocks, lds_bytes, ablate, gt_words, on); break;
        case 16: wide ? launch_fill<true, 16, kTune>(idx, d, share, u_site, n_fill, (unsigned)blocks, lds_bytes, ablate, gt_words, on)
                      : launch_fill<false, 16, kTune>(idx, d, share, u_site, n_fill, (unsigned)blocks, lds_bytes, ablate, gt_words, on); break;
        case 32: wide ? launch_fill<true, 32, kTune>(idx, d, share, u_site, n_fill, (unsigned)blocks, lds_bytes, ablate, gt_words, on)
                      : launch_fill<false, 32, kTune>(idx, d, share, u_site, n_fill, (unsigned)blocks, lds_bytes, ablate, gt_words, on); break;
        default: wide ? launch_fill<true, 64, kTune>(idx, d, share, u_site, n_fill, (unsigned)blocks, lds_bytes, ablate, gt_words, on)
                      : launch_fill<false, 64, kTune>(idx, d, share, u_site, n_fill, (unsigned)blocks, lds_bytes, ablate, gt_words, on); break;
      }
    }
    HIP_TRY(hipGetLastError());
  }
  return VS_OK;
}

// Resident carrier lists (option "resident_lists"): every carrier list a query can report is a function of the index
// alone -- a site's list is its vertex's -- so it can be expanded ONCE, with the kernel the queries would run, into an
// arena that stays with the handle: 2 (4) bytes per carrier record of HBM bought back as the whole expansion of every
// later batch, and as 9/10 of what a result sends across PCIe (rows only; the host keeps one mirror of the arena).
// Layout: the sites' lists in site-table order at s_carpre[g] -- a region's lists stay ONE range, everything downstream
// of car_base is unchanged -- followed by the lists of vertices that only the walking query types report (no site, or
// a site the reference never reports).
static int build_resident_lists(vs_index* idx) {
  if (idx->res_arena) return VS_OK;
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device");
  HIP_TRY(hipSetDevice(idx->device));
  if (idx->srv_alive) VS_TRY(server_stop(idx));
  const HostImage& im = idx->im;
  const uint64_t G = idx->d.G, V = idx->d.V;
  std::vector<uint32_t> s_vid(G), s_ncar(G);
  if (G) {
    HIP_TRY(hipMemcpyAsync(s_vid.data(), idx->d.s_vid, G * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipMemcpyAsync(s_ncar.data(), idx->d.s_ncar, G * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
  }
  std::vector<uint64_t> v_abegin(V, ~0ull), x_begin;
  std::vector<uint32_t> x_vid;
  for (uint64_t g = 0; g < G; ++g) {
    const uint32_t v = s_vid[g];
    if (s_ncar[g] && s_ncar[g] == im.v_ncar[v] && v_abegin[v] == ~0ull) v_abegin[v] = idx->h_carpre[g];
  }
  uint64_t total = idx->h_carpre[G];
  idx->res_site_entries = total;
  for (uint64_t v = 0; v < V; ++v)
    if (im.v_ncar[v] && v_abegin[v] == ~0ull) {
      v_abegin[v] = total; x_vid.push_back((uint32_t)v); x_begin.push_back(total);
      total += pad_car(im.v_ncar[v]);
    }
  const uint64_t X = x_vid.size(), A = G + X;
  const uint32_t width = idx->d.wpc <= 63 ? 2 : 4;
  const uint64_t arena_bytes = total * width + 16, temp_bytes = A * (sizeof(VariantRow) + 12) + X * 12;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  if (arena_bytes + temp_bytes + V * 8 > free_b / 2)
    return fail(VS_ERR_UNSUPPORTED, "resident carrier lists need %.1f GB of HBM, %.1f GB are free", (arena_bytes + temp_bytes + V * 8) / 1e9, free_b / 1e9);
  void* arena = nullptr;
  HIP_TRY(hipMalloc(&arena, arena_bytes));
  idx->image_allocs.push_back(arena);
  const uint64_t* d_abegin = nullptr;
  VS_TRY(upload_image(idx, v_abegin, &d_abegin));
  if (A) {
    ScratchBufs tmp(idx);
    DevResult d{};
    uint32_t* dx_vid = nullptr; uint64_t* dx_begin = nullptr;
    VS_TRY(dev_alloc(idx, A * sizeof(VariantRow), (void**)&d.rows, &tmp.bufs));
    VS_TRY(dev_alloc(idx, A * 4, (void**)&d.r_class, &tmp.bufs));
    VS_TRY(dev_alloc(idx, A * 8, (void**)&d.r_gt0, &tmp.bufs));
    VS_TRY(dev_alloc(idx, X * 4, (void**)&dx_vid, &tmp.bufs));
    VS_TRY(dev_alloc(idx, X * 8, (void**)&dx_begin, &tmp.bufs));
    if (X) {
      HIP_TRY(hipMemcpyAsync(dx_vid, x_vid.data(), X * 4, hipMemcpyHostToDevice, idx->stream));
      HIP_TRY(hipMemcpyAsync(dx_begin, x_begin.data(), X * 8, hipMemcpyHostToDevice, idx->stream));
    }
    d.A = A; d.S = total; d.carriers = arena; d.car_width = width;
    hipLaunchKernelGGL(k_resident_params, dim3((unsigned)((A + 255) / 256)), dim3(256), 0, idx->stream, idx->d, (const uint32_t*)dx_vid, (const uint64_t*)dx_begin, X,
                       d.rows, d.r_class, d.r_gt0);
    HIP_TRY(hipGetLastError());
    VS_TRY(fill_lists(idx, d, false, nullptr, A));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    tmp.release();
  }
  idx->device_bytes += arena_bytes;
  idx->d.v_abegin = d_abegin;
  idx->res_entries = total;
  idx->res_arena = arena;
  return VS_OK;
}
// the arena's host mirror: page-locked, copied over once, shared by the handle's results
static int ensure_resident_mirror(vs_index* idx) {
  if (idx->res_mirror) return VS_OK;
  const uint32_t width = idx->d.wpc <= 63 ? 2 : 4;
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, idx->res_entries * width + 16, hipHostMallocDefault));
  const hipError_t e = hipMemcpy(p, idx->res_arena, idx->res_entries * width, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { (void)hipHostFree(p); return fail(VS_ERR_HIP, "copy of the resident carrier lists failed: %s", hipGetErrorString(e)); }
  idx->res_mirror = p;
  return VS_OK;
}

// Spin on a word in mapped host memory until the device has posted `seq` (the runtime's completion wait costs tens of
// microseconds more); a kernel that never posts -- a fault -- is caught by the synchronisation after the deadline.
static int wait_posted(vs_index* idx, volatile uint64_t* word, uint64_t seq, int deadline_ms) {
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(deadline_ms);
  bool posted = false;
  uint32_t spins = 0;
  while (!(posted = (*word == seq))) {
    __builtin_ia32_pause();
    if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() > deadline) break;
  }
  if (!posted) {
    if (idx->plan_stream) HIP_TRY(hipStreamSynchronize(idx->plan_stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    if (*word != seq) return fail(VS_ERR_INTERNAL, "a batch kernel finished without posting its sequence word");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return VS_OK;
}

// Up to three 64-bit device words for the host, behind everything queued on the handle's stream: a one-thread kernel
// copies them into mapped host memory and posts a sequence word the host spins on (a staged hipMemcpyAsync into pageable
// memory + hipStreamSynchronize costs 20 - 30 us more, and the walking query types do it two or three times per batch).
static int read_device_words(vs_index* idx, const uint64_t* a, uint64_t* va, const uint64_t* b = nullptr, uint64_t* vb = nullptr,
                             const uint64_t* c = nullptr, uint64_t* vc = nullptr) {
  uint64_t* dst = idx->pinned + vs_index::kPinWords;
  const uint64_t seq = ++idx->done_seq;
  hipLaunchKernelGGL(k_post_words, dim3(1), dim3(1), 0, work_stream(idx), dst, a, b, c, seq);
  HIP_TRY(hipGetLastError());
  VS_TRY(wait_posted(idx, dst + 3, seq, 2000));
  if (!idx->pre) idx->batch_in_flight = false;   // (the handle's own stream has drained)
  if (va) *va = ((volatile uint64_t*)dst)[0];
  if (vb) *vb = ((volatile uint64_t*)dst)[1];
  if (vc) *vc = ((volatile uint64_t*)dst)[2];
  return VS_OK;
}

// ---- speculative type-6 batches: hints, the totals mailbox, the result's sizes on first use ----
static void t6_hint_update(vs_index* idx, uint64_t n, uint64_t rows, uint64_t arena) {
  const uint64_t need_rows = rows + rows / 8 + 1024, need_arena = arena + arena / 8 + 4096;
  idx->t6_hint_n = n;
  if (need_rows > idx->t6_hint_rows || need_arena > idx->t6_hint_arena || idx->t6_hint_rows == 0) {   // grows at once (both: they belong to one shape of batch)
    idx->t6_hint_rows = std::max(idx->t6_hint_rows, need_rows); idx->t6_hint_arena = std::max(idx->t6_hint_arena, need_arena);
    idx->t6_small_runs = 0; idx->t6_small_rows = idx->t6_small_arena = 0;
    return;
  }
  if (need_rows >= idx->t6_hint_rows / 2 && need_arena >= idx->t6_hint_arena / 2) { idx->t6_small_runs = 0; idx->t6_small_rows = idx->t6_small_arena = 0; return; }
  // shrinks after kHintShrinkAfter consecutive batches that needed less than half of it, to the largest of those
  idx->t6_small_rows = std::max(idx->t6_small_rows, need_rows); idx->t6_small_arena = std::max(idx->t6_small_arena, need_arena);
  if (++idx->t6_small_runs >= kHintShrinkAfter) {
    idx->t6_hint_rows = idx->t6_small_rows; idx->t6_hint_arena = idx->t6_small_arena;
    idx->t6_small_runs = 0; idx->t6_small_rows = idx->t6_small_arena = 0;
  }
}
// The totals of a speculative batch move from its mailbox into the result (and the handle's hints follow them).  The plan is four short
// kernels in front of everything else of the batch: by the time anybody asks, the word is there.
static int capture_totals(vs_result* r) {
  if (!r->sizes_pending || r->totals_captured) return VS_OK;
  vs_index* idx = r->idx;
  volatile PlanTotals* pt = reinterpret_cast<volatile PlanTotals*>(idx->pinned + vs_index::kPinPlanRing + (size_t)r->plan_slot * 8);
  VS_TRY(wait_posted(idx, &pt->seq, r->plan_seq, 2000));
  r->plan_copy.rows = pt->rows; r->plan_copy.arena = pt->arena; r->plan_copy.shared_rows = pt->shared_rows; r->plan_copy.not_sorted = pt->not_sorted;
  r->plan_copy.reported = pt->reported; r->plan_copy.n_slow = pt->n_slow; r->plan_copy.n_runs = pt->n_runs; r->plan_copy.seq = r->plan_seq;
  r->totals_captured = true;
  if (idx->plan_slot_owner[r->plan_slot] == r) idx->plan_slot_owner[r->plan_slot] = nullptr;
  if (!r->plan_copy.not_sorted) t6_hint_update(idx, r->d.Q, r->plan_copy.rows, r->plan_copy.arena);
  else { idx->sort_hint = true; idx->sort_probe_in = 32; }   // (as after a batch that was sorted on the device: the next ones sort first -- also when nobody reads this one)
  return VS_OK;
}
static int result_ready(vs_result* r);
static void release_bufs(vs_index* idx, std::vector<DevBuf>& bufs);
static int run_type6_shared(vs_index* idx, const vs_region* regions, uint64_t n, vs_result* r, bool regions_on_device, const uint64_t* site_records,
                            bool allow_async, bool may_speculate);
// Every accessor of a result calls this first: a speculative batch's sizes become the plan's totals here -- or, when the plan refused the
// batch on the device (more rows or arena entries than were allocated, or regions that were not sorted), the batch is run again from the
// result's own copy of the regions, with the exact sizes, before anything is read.
static int result_sizes(vs_result* r) {
  if (!r->sizes_pending) return VS_OK;
  vs_index* idx = r->idx;
  VS_TRY(capture_totals(r));
  r->sizes_pending = false;
  const PlanTotals& t = r->plan_copy;
  if (!t.not_sorted && t.rows <= r->cap_rows && t.arena <= r->cap_arena) {
    r->d.A = t.rows; r->d.S = t.arena;
    r->n_rows_reported = t.reported; r->n_unique_sites = t.shared_rows;
    return VS_OK;
  }
  idx->t6_refused++;
  VS_TRY(result_ready(r));                         // (its kernels returned at once; nothing was written behind the plan)
  if (idx->timing_owner == r) { idx->timing_owner = nullptr; idx->timing_pending = false; }
  std::vector<DevBuf> old;
  old.swap(r->bufs);                               // (the result's copy of the regions lives in there: released when the redo has read it)
  const uint64_t n = r->d.Q;
  const vs_region* dreg = reinterpret_cast<const vs_region*>(r->d.regions);
  r->d = DevResult{};
  const int rc = run_type6_shared(idx, dreg, n, r, /*regions_on_device=*/true, nullptr, /*allow_async=*/false, /*may_speculate=*/false);
  release_bufs(idx, old);
  return rc;
}

// The expansion of a shared batch that writes the shared rows as well (k_fill_sites2).  mode 2 (tuning builds): split -- rows +
// listed variants here, the dense sites of the index the batch covers in k_fill_dense behind it.
template <bool WIDE, bool TUNE>
static void launch_fill2(vs_index* idx, const DevResult& d, const RunRec* runs, const uint32_t* coarse, uint64_t n_runs, uint64_t U, uint32_t chunk,
                         size_t lds_bytes, uint32_t ablate, uint32_t gt_words, unsigned long long* tstat, int mode, const PlanDev* pd = nullptr) {
  const unsigned blocks = (unsigned)(((U + chunk - 1) / chunk + 3) / 4);
#ifdef VS_TUNING
  if (mode == 2) {
    switch (chunk) {
      case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_sites2<WIDE, 16, TUNE, false>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, coarse, n_runs, U, ablate, gt_words, tstat, (const PlanDev*)nullptr); break;
      case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_sites2<WIDE, 32, TUNE, false>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, coarse, n_runs, U, ablate, gt_words, tstat, (const PlanDev*)nullptr); break;
      default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_sites2<WIDE, 64, TUNE, false>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, coarse, n_runs, U, ablate, gt_words, tstat, (const PlanDev*)nullptr); break;
    }
    const uint64_t nd = idx->d.n_dense;
    if (nd) {
      const uint32_t k = idx->opts.fill_dense_k;
      const unsigned db = (unsigned)(((nd + k - 1) / k + 3) / 4);
      switch (k) {
        case 8:  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_dense<WIDE, 8>), dim3(db), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, n_runs, U, gt_words); break;
        case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_dense<WIDE, 32>), dim3(db), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, n_runs, U, gt_words); break;
        case 64: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_dense<WIDE, 64>), dim3(db), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, n_runs, U, gt_words); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_dense<WIDE, 16>), dim3(db), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, n_runs, U, gt_words); break;
      }
    }
    return;
  }
#endif
  switch (chunk) {
    case 8:  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_sites2<WIDE, 8, TUNE>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, coarse, n_runs, U, ablate, gt_words, tstat, pd); break;
    case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_sites2<WIDE, 32, TUNE>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, coarse, n_runs, U, ablate, gt_words, tstat, pd); break;
    case 64: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_sites2<WIDE, 64, TUNE>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, coarse, n_runs, U, ablate, gt_words, tstat, pd); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill_sites2<WIDE, 16, TUNE>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, runs, coarse, n_runs, U, ablate, gt_words, tstat, pd); break;
  }
}

// Query type 6 over a batch whose regions SHARE rows and carrier lists (every batch of more than 64 regions unless
// option share_lists is 0).  Stages, each reading only what the stage before it left:
//   plan   k_t6_bounds / _mid / _apply: bounds, E_prev, the per-region arrays, the row deltas, the slow-region list;
//          the totals arrive in mapped host memory and the host spins on their sequence word (the one host wait of a batch)
//   (sort  a batch that turns out not to be sorted by first site is sorted on the device and planned again)
//   rows   k_t6_slow (private copies + the literal duplicate rule, only when the plan counted such regions)
//   fill   k_fill_sites2: the shared rows AND their carrier lists, one launch; with resident lists k_share_rows2 alone;
//          with async_fill k_share_rows2 here and k_fill_sites on the second stream
//   done   k_post_done: a word in mapped host memory, the host spins on it
static int run_type6_shared(vs_index* idx, const vs_region* regions, uint64_t n, vs_result* r, bool regions_on_device, const uint64_t* site_records,
                            bool allow_async, bool may_speculate = true) {
  if (idx->srv_alive) VS_TRY(server_stop(idx));   // a throughput batch does not share the GPU with a polling server
  idx->timing_pending = false; idx->timing_owner = nullptr; idx->timing_total_only = false;   // (an earlier lean batch's event pair is not this batch's)
  DevResult& d = r->d;
  d.Q = n;
  uint64_t* dreg = nullptr;
  {   // the per-region arrays of the result: one buffer
    Slab sl;
    const size_t o_reg = sl.add(2 * n * 8), o_fl = sl.add(n * sizeof(*d.q_flags)), o_g0 = sl.add(n * sizeof(*d.q_g0)), o_nv = sl.add(n * sizeof(*d.q_nvar)),
                 o_nc = sl.add(n * sizeof(*d.q_ncar)), o_vb = sl.add((n + 1) * sizeof(*d.var_begin)), o_cb = sl.add((n + 1) * sizeof(*d.car_base)),
                 o_vc = sl.add(n * sizeof(*d.var_count)), o_cl = sl.add(n * sizeof(*d.q_car_len));
    VS_TRY(ralloc(r, sl.bytes, &sl.base));
    dreg = sl.at<uint64_t>(o_reg);
    d.q_flags = sl.at<std::remove_pointer_t<decltype(d.q_flags)>>(o_fl); d.q_g0 = sl.at<std::remove_pointer_t<decltype(d.q_g0)>>(o_g0);
    d.q_nvar = sl.at<std::remove_pointer_t<decltype(d.q_nvar)>>(o_nv); d.q_ncar = sl.at<std::remove_pointer_t<decltype(d.q_ncar)>>(o_nc);
    d.var_begin = sl.at<std::remove_pointer_t<decltype(d.var_begin)>>(o_vb); d.car_base = sl.at<std::remove_pointer_t<decltype(d.car_base)>>(o_cb);
    d.var_count = sl.at<std::remove_pointer_t<decltype(d.var_count)>>(o_vc); d.q_car_len = sl.at<std::remove_pointer_t<decltype(d.q_car_len)>>(o_cl);
  }
  d.regions = dreg;
  // async_submit: the PLAN of this batch runs on a stream of its own -- beside the expansion of the batch before it, which
  // is still on the handle's stream when the caller submits back to back.  The plan reads the regions and the index and
  // writes this batch's own arrays; nothing in the pool is referenced by work in flight (a batch that returned when it
  // was enqueued keeps its temporaries until its completion event, vs_result::ev_done), so its buffers are its own.  The
  // rest of the batch (rows, expansion) goes to the handle's stream behind an event.  Not for a handle that sorts first,
  // not with async_fill (whose second stream plays the opposite game).
  const bool async_submit = allow_async && idx->opts.async_submit && !idx->opts.async_fill && !idx->opts.lat_debug;
  bool plan_aside = async_submit && !(idx->sort_hint && idx->sort_probe_in > 0);
  if (plan_aside) VS_TRY(ensure_plan_stream(idx));
  hipStream_t ps = plan_aside ? idx->plan_stream : idx->stream;
  // (regions already in device memory are copied by the kernel that reads them first: k_t6_bounds)
  const uint64_t* regions_dev = regions && regions_on_device && !site_records && (reinterpret_cast<uintptr_t>(regions) & 15) == 0
                                    ? reinterpret_cast<const uint64_t*>(regions) : nullptr;
  if (regions_dev) {}
  else if (regions) HIP_TRY(hipMemcpyAsync(dreg, regions, n * 16, regions_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ps));
  else HIP_TRY(hipMemsetAsync(dreg, 0, n * 16, ps));
  HIP_TRY(hipEventRecord(idx->ev[0], ps));
  ScratchBufs scratch(idx);
  const bool resident = idx->opts.resident_lists && idx->res_arena;
  // ---- plan ----
  uint32_t items = (uint32_t)((n + (uint64_t)kPlanBlock * kPlanMaxTiles - 1) / ((uint64_t)kPlanBlock * kPlanMaxTiles));
  if (n >= 65536 && items < idx->opts.plan_items) items = idx->opts.plan_items;   // (regions per thread of the plan's kernels: see EngineOpts::plan_items)
  const uint32_t ntiles = (uint32_t)((n + (uint64_t)kPlanBlock * items - 1) / ((uint64_t)kPlanBlock * items));
  ShareMax* tile_max = nullptr;
  Scan5* tile_sums = nullptr;
  uint32_t *e_prev = nullptr, *status = nullptr, *slow_list = nullptr;
  uint64_t* e_prev_c = nullptr;   // the arena prefix at E_prev, carried by the same scan (k_rows.hip.h: ShareMax)
  RunRec* runs = nullptr;
  uint32_t* coarse = nullptr;
  PlanDev* plan_dev = nullptr;
  {   // the plan's temporaries: one buffer
    Slab sl;
    const size_t o_co = sl.add((idx->d.G / kCoarseRows + 2) * 4), o_tm = sl.add(ntiles * sizeof(ShareMax)),
                 o_ts = sl.add((ntiles + 1) * sizeof(Scan5)),   // (+ the totals: k_t6_totals)
                 o_ep = sl.add(n * 4), o_st = sl.add(4), o_sl = sl.add(n * 4), o_ru = sl.add((n + 1) * sizeof(RunRec)), o_pd = sl.add(sizeof(PlanDev)),
                 o_ec = sl.add(n * 8);
    VS_TRY(dev_alloc(idx, sl.bytes, (void**)&sl.base, &scratch.bufs));
    coarse = sl.at<uint32_t>(o_co); tile_max = sl.at<ShareMax>(o_tm); tile_sums = sl.at<Scan5>(o_ts);
    e_prev = sl.at<uint32_t>(o_ep); status = sl.at<uint32_t>(o_st); slow_list = sl.at<uint32_t>(o_sl); runs = sl.at<RunRec>(o_ru);
    plan_dev = sl.at<PlanDev>(o_pd); e_prev_c = sl.at<uint64_t>(o_ec);
  }
  // SPECULATIVE (round 6, option t6_speculate): the default batch -- async_submit, rows and lists in one launch, regions given as regions --
  // on a handle whose previous shared batch was about this size does not wait for the plan's totals: table and arena are sized from that
  // batch (+ 1/8), k_t6_totals leaves the totals and its verdict in device memory for the kernels behind it (PlanDev) and in this batch's
  // own mailbox for the host, which reads it when the result is first asked for anything (result_sizes) -- and redoes a refused batch.
  const uint64_t want_rows = idx->t6_hint_rows, want_arena = idx->t6_hint_arena;
  const bool spec = may_speculate && idx->opts.t6_speculate && plan_aside && regions && !site_records && !resident && idx->opts.fill_fused && !idx->opts.fill_mode &&
                    !idx->opts.fill_stats && !idx->opts.lat_debug && want_rows > 0 && idx->t6_hint_n > 0 && n * 4 <= idx->t6_hint_n * 5 && n * 5 >= idx->t6_hint_n * 4;
  int slot = -1;
  if (spec) {
    slot = (int)(idx->plan_slot_next++ % vs_index::kPlanSlots);
    if (idx->plan_slot_owner[slot]) VS_TRY(capture_totals(idx->plan_slot_owner[slot]));   // (eight batches back: long done; its totals move into the result)
  }
  PlanTotals* pt = reinterpret_cast<PlanTotals*>(idx->pinned + (spec ? vs_index::kPinPlanRing + (size_t)slot * 8 : vs_index::kPinPlan));
  d.car_width = idx->d.wpc <= 63 ? 2 : 4;
  if (spec) {   // table and arena BEFORE the plan: the rows of the regions under the duplicate rule are written on the plan's stream (k_t6_slow below)
    d.A = want_rows; d.S = want_arena;
    VS_TRY(ralloc(r, d.A, &d.rows));
    uint8_t* arena = nullptr;
    VS_TRY(ralloc(r, d.S * d.car_width + 16, &arena));
    d.carriers = arena;
  }
  auto launch_bounds = [&](int src) {
    if (src == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_t6_bounds<0>), dim3(ntiles), dim3(kPlanBlock), 0, ps, idx->d, d, regions_dev, items, tile_max, status);
    else if (src == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_t6_bounds<1>), dim3(ntiles), dim3(kPlanBlock), 0, ps, idx->d, d, site_records, items, tile_max, status);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_t6_bounds<2>), dim3(ntiles), dim3(kPlanBlock), 0, ps, idx->d, d, (const uint64_t*)nullptr, items, tile_max, status);
  };
  // the plan over the regions as they stand in `d`; the totals arrive in mapped host memory
  bool plan_end_enqueued = false;   // ev[1] recorded behind the (last) plan and waited for by the handle's stream
  uint64_t plan_seq = 0;
  auto plan = [&](int src) -> int {
    plan_end_enqueued = false;
    launch_bounds(src);   // (block 0 clears `status`)
    hipLaunchKernelGGL(k_t6_mid, dim3(ntiles), dim3(kPlanBlock), 0, ps, idx->d, d, (const ShareMax*)tile_max, items, e_prev, e_prev_c, tile_sums, status);
    const uint64_t seq = ++idx->share_seq;
    hipLaunchKernelGGL(k_t6_totals, dim3(1), dim3(kPlanBlock), 0, ps, tile_sums, ntiles, (const uint32_t*)status, pt, seq, idx->res_entries, resident ? 1u : 0u,
                       spec ? plan_dev : (PlanDev*)nullptr, want_rows, want_arena);
    plan_seq = seq;
    if (resident) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_t6_apply<true>), dim3(ntiles), dim3(kPlanBlock), 0, ps, idx->d, d, (const uint32_t*)e_prev, (const uint64_t*)e_prev_c, (const Scan5*)tile_sums,
                                     ntiles, items, runs, coarse, slow_list, (const uint32_t*)status, idx->res_entries);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_t6_apply<false>), dim3(ntiles), dim3(kPlanBlock), 0, ps, idx->d, d, (const uint32_t*)e_prev, (const uint64_t*)e_prev_c, (const Scan5*)tile_sums,
                            ntiles, items, runs, coarse, slow_list, (const uint32_t*)status, (uint64_t)0);
    // speculative: the private rows of the regions under the duplicate rule (count and verdict from the plan's record) on the PLAN's stream,
    // beside the previous batch's expansion like the rest of the plan -- the handle's stream then carries the expansion alone
    if (spec) hipLaunchKernelGGL(k_t6_slow, dim3((unsigned)((std::min<uint64_t>(n, 1024) + 3) / 4)), dim3(256), 0, ps, idx->d, d, (const uint32_t*)slow_list, (uint64_t)0,
                                 (const PlanDev*)plan_dev);
    HIP_TRY(hipGetLastError());
    // the plan's end event, and the handle's stream waiting for it, are enqueued while the plan is still on its way to the totals:
    // two calls less between "the totals are here" and "the expansion is launched" (a batch of a tenth of a millisecond is the host's)
    if (plan_aside) {
      HIP_TRY(hipEventRecord(idx->ev[1], ps));
      HIP_TRY(hipStreamWaitEvent(idx->stream, idx->ev[1], 0));
      plan_end_enqueued = true;
    }
    return spec ? VS_OK : wait_posted(idx, &pt->seq, seq, 200);   // (a speculative batch: nobody waits here)
  };
  // A batch that is not sorted by first site: counting sort of the regions over the site index (k_sort_*), the batch
  // then works on sorted copies of its per-region arrays (`d` points at them from here on, `d_user` keeps the
  // caller's) and k_permute_out hands the outcome back at the end.
  DevResult d_user{};        // the caller's per-region arrays while an unsorted batch works on sorted copies
  uint32_t* perm = nullptr;  // sorted position -> region of the caller's batch (NULL: the batch is worked in the order given)
  auto sort_batch = [&]() -> int {
    const uint64_t G = idx->d.G;
    uint32_t* count = nullptr;
    unsigned long long* cursor = nullptr;
    VS_TRY(dev_alloc(idx, (G + 2) * 4, (void**)&count, &scratch.bufs));
    VS_TRY(dev_alloc(idx, (G + 2) * 8, (void**)&cursor, &scratch.bufs));
    VS_TRY(dev_alloc(idx, n * 4, (void**)&perm, &scratch.bufs));
    HIP_TRY(hipMemsetAsync(count, 0, (G + 1) * 4, idx->stream));
    hipLaunchKernelGGL(k_sort_hist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, d, count);
    VS_TRY(exclusive_scan<uint32_t>(idx, (const uint32_t*)count, G + 1, (uint64_t*)cursor, &scratch.bufs));
    hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, d, cursor, perm);
    d_user = d;
    uint64_t* sregions = nullptr;
    VS_TRY(dev_alloc(idx, 2 * n * 8, (void**)&sregions, &scratch.bufs));
    VS_TRY(dev_alloc(idx, n, (void**)&d.q_flags, &scratch.bufs));
    VS_TRY(dev_alloc(idx, n * 4, (void**)&d.q_g0, &scratch.bufs));
    VS_TRY(dev_alloc(idx, n * 8, (void**)&d.q_nvar, &scratch.bufs));
    VS_TRY(dev_alloc(idx, n * 8, (void**)&d.q_ncar, &scratch.bufs));
    VS_TRY(dev_alloc(idx, (n + 1) * 8, (void**)&d.var_begin, &scratch.bufs));
    VS_TRY(dev_alloc(idx, (n + 1) * 8, (void**)&d.car_base, &scratch.bufs));
    VS_TRY(dev_alloc(idx, n * 8, (void**)&d.q_car_len, &scratch.bufs));
    VS_TRY(dev_alloc(idx, n * 8, (void**)&d.var_count, &scratch.bufs));
    hipLaunchKernelGGL(k_permute_in, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, d_user, d, (const uint32_t*)perm, sregions);
    d.regions = sregions;
    HIP_TRY(hipGetLastError());
    return VS_OK;
  };
  const int src = site_records ? 1 : 0;
  // (a handle whose last batch needed sorting sorts first; it looks at the order as given again every 32nd batch)
  if (idx->sort_hint && idx->sort_probe_in > 0) {
    --idx->sort_probe_in;
    launch_bounds(src);
    VS_TRY(sort_batch());
    VS_TRY(plan(2));
  } else if (spec) {
    VS_TRY(plan(src));   // (sorted or not is the device's verdict)
  } else {
    VS_TRY(plan(src));
    if (pt->not_sorted) {
      if (plan_aside) {   // the sort and the second plan run on the handle's stream (behind the plan stream's work so far)
        HIP_TRY(hipStreamSynchronize(ps));
        plan_aside = false; ps = idx->stream;
      }
      VS_TRY(sort_batch());
      VS_TRY(plan(2));
      idx->sort_hint = true; idx->sort_probe_in = 32;
    } else idx->sort_hint = false;
  }
  if (!spec && pt->not_sorted) return fail(VS_ERR_INTERNAL, "the batch is not sorted by first site after the device-side sort");
  if (!plan_end_enqueued) HIP_TRY(hipEventRecord(idx->ev[1], ps));   // (a plan on the handle's own stream: the phase boundary only)
  // (speculative: every size below is what was ALLOCATED -- an upper bound the launches are made for; the kernels take the real ones from PlanDev)
  const uint64_t U = spec ? want_rows : pt->shared_rows, n_slow = spec ? 0 : pt->n_slow, n_runs = spec ? 0 : pt->n_runs;
  if (!spec) { d.A = pt->rows; d.S = pt->arena; }
  r->n_rows_reported = spec ? 0 : pt->reported;
  if (spec) {
    r->sizes_pending = true; r->totals_captured = false; r->plan_slot = slot; r->plan_seq = plan_seq; r->cap_rows = want_rows; r->cap_arena = want_arena;
    idx->plan_slot_owner[slot] = r;
    idx->t6_speculated++;
  } else t6_hint_update(idx, n, pt->rows, pt->arena);
  r->shared_lists = true;
  r->resident = resident;
  r->scattered_lists = false;
  r->n_unique_sites = resident ? 0 : U;
  if (!spec) {
    VS_TRY(ralloc(r, d.A, &d.rows));
    if (resident) d.carriers = idx->res_arena;
    else {
      uint8_t* arena = nullptr;
      VS_TRY(ralloc(r, d.S * d.car_width + 16, &arena));
      d.carriers = arena;
    }
  }
  // ---- shared rows + carrier lists: which form ----
  const uint64_t n_fill = resident ? 0 : U;
  const bool async_fill = allow_async && idx->opts.async_fill && n_fill > 0;
  const bool fused = idx->opts.fill_fused && !resident && !async_fill;
  // Every event on a stream is a packet the GPU works through (~3 us each): the default batch -- async_submit, rows and lists in one
  // launch, no permutation -- records TWO on the handle's stream, the result's own pair around the expansion, and its phase
  // times are read from those (round 3 recorded five per batch, and this round's completion event would have made it eight).
  const bool lean = async_submit && fused && n_fill > 0 && !perm;
  if (!lean) HIP_TRY(hipEventRecord(idx->ev[2], idx->stream));
  // ---- rows of the regions under the duplicate rule ----
  if (n_slow) {   // (a speculative batch launched the kernel behind its plan, with the count from the plan's record)
    const uint64_t waves = std::min<uint64_t>(n_slow, 16384);
    hipLaunchKernelGGL(k_t6_slow, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, idx->stream, idx->d, d, (const uint32_t*)slow_list, n_slow, (const PlanDev*)nullptr);
  }
  // ---- shared rows + carrier lists ----
  uint32_t* u_site = nullptr;
  if (U && !fused) {
    if (!resident) VS_TRY(dev_alloc(idx, U * 4 + 8, (void**)&u_site, &scratch.bufs));
    hipLaunchKernelGGL(k_share_rows2, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, idx->stream, idx->d, d, (const RunRec*)runs, (const uint32_t*)coarse, n_runs, U, u_site);
  }
  HIP_TRY(hipGetLastError());
  if (!lean) HIP_TRY(hipEventRecord(idx->ev[3], idx->stream));
  if (async_fill) {
    // the expansion goes to the handle's second stream behind an event and the call returns once the FIRST stream is done
    // (rows, per-region arrays); the next batch's plan and rows then run beside it.  The call's temporaries (the site
    // index the expansion reads) stay with the result until it is freed.
    VS_TRY(ensure_fill_stream(idx));
    VS_TRY(result_events(r));
    HIP_TRY(hipEventRecord(idx->fill_ev[0], idx->stream));
    HIP_TRY(hipStreamWaitEvent(idx->fill_stream, idx->fill_ev[0], 0));
    HIP_TRY(hipEventRecord(r->ev_fill[0], idx->fill_stream));
    VS_TRY(fill_lists(idx, d, true, u_site, n_fill, idx->fill_stream));
    HIP_TRY(hipEventRecord(r->ev_fill[1], idx->fill_stream));
    r->pending = true;
  } else if (n_fill && !fused) {
    VS_TRY(fill_lists(idx, d, true, u_site, n_fill));
  } else if (n_fill) {
    // rows per wave task: 32 (one look-up of the run records and one round of parameter loads per 32 rows; 16-row tasks -- round 3's
    // choice for the kernel that did not write rows -- measure the same to 5 % slower, 64 and 8 slower: tools/ab_t6.py)
    uint32_t chunk = idx->opts.fill_chunk ? idx->opts.fill_chunk : 32;
    const uint32_t gt_words = fill_gt_words(idx);
#ifdef VS_TUNING
    const size_t lds_bytes = fill_lds_bytes(idx) + std::min<size_t>(idx->opts.fill_lds_pad, 96 << 10);
    const uint32_t ablate = idx->opts.fill_ablate;
    constexpr bool kTune = true;
#else
    const size_t lds_bytes = fill_lds_bytes(idx);
    const uint32_t ablate = 0;
    constexpr bool kTune = false;
#endif
    if ((U + chunk - 1) / chunk / 4 > 0x7FFFFFF0ull) return fail(VS_ERR_ARG, "batch too large for one launch (%llu shared rows)", (unsigned long long)U);
    unsigned long long* tstat = nullptr;
#ifdef VS_TUNING
    if (idx->opts.fill_stats) {
      VS_TRY(dev_alloc(idx, ((U + chunk - 1) / chunk + 4) * 16, (void**)&tstat, &scratch.bufs));
      HIP_TRY(hipMemsetAsync(tstat, 0, ((U + chunk - 1) / chunk + 4) * 16, idx->stream));
    }
#endif
    const int mode = idx->opts.fill_mode;
    VS_TRY(result_events(r));   // the kernel's own duration, whenever the result is asked for it (vs_result_fill_ms)
    HIP_TRY(hipEventRecord(r->ev_fill[0], idx->stream));
    const PlanDev* pd = spec ? plan_dev : nullptr;
    if (idx->d.wpc > 63) launch_fill2<true, kTune>(idx, d, runs, coarse, n_runs, U, chunk, lds_bytes, ablate, gt_words, tstat, mode, pd);
    else launch_fill2<false, kTune>(idx, d, runs, coarse, n_runs, U, chunk, lds_bytes, ablate, gt_words, tstat, mode, pd);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(r->ev_fill[1], idx->stream));
    r->pending = true;
#ifdef VS_TUNING
    if (tstat) {
      const uint64_t nt = (U + chunk - 1) / chunk;
      std::vector<uint32_t> h(nt * 4);
      HIP_TRY(hipMemcpyAsync(h.data(), tstat, nt * 16, hipMemcpyDeviceToHost, idx->stream));
      HIP_TRY(hipStreamSynchronize(idx->stream));
      double sp = 0, sl = 0, sd = 0, st = 0, nd = 0; uint32_t mx = 0;
      for (uint64_t t = 0; t < nt; ++t) {
        const uint32_t tot = h[4 * t + 3] & 0xFFFFFFu;
        sp += h[4 * t]; sl += h[4 * t + 1]; sd += h[4 * t + 2]; st += tot; nd += h[4 * t + 3] >> 24; mx = std::max(mx, tot);
      }
      fprintf(stderr, "fill stats: %llu tasks of %u rows | mean ticks (10 ns) per task: parameters + rows %.1f, list phase %.1f, dense phase %.1f, whole task %.1f (max %u) | "
              "dense variants per task %.2f\n", (unsigned long long)nt, chunk, sp / nt, sl / nt, sd / nt, st / nt, mx, nd / nt);
    }
#endif
  }
  if (perm) {   // every region's outcome back to its place in the caller's order (rows and lists are shared: nothing else moves)
    const DevResult ds = d;
    d.regions = d_user.regions; d.q_flags = d_user.q_flags; d.q_g0 = d_user.q_g0; d.q_nvar = d_user.q_nvar; d.q_ncar = d_user.q_ncar;
    d.var_begin = d_user.var_begin; d.car_base = d_user.car_base; d.q_car_len = d_user.q_car_len; d.var_count = d_user.var_count;
    hipLaunchKernelGGL(k_permute_out, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, idx->stream, ds, d, (const uint32_t*)perm);
    HIP_TRY(hipGetLastError());
  }
  if (!lean) HIP_TRY(hipEventRecord(idx->ev[4], idx->stream));
  else {   // plan: the handle's two events on the plan stream; the rest: the result's pair (no separate rows kernel: ms_emit = 0)
    idx->tev[0] = idx->ev[0]; idx->tev[1] = idx->ev[1]; idx->tev[2] = r->ev_fill[0]; idx->tev[3] = r->ev_fill[0]; idx->tev[4] = r->ev_fill[1];
    idx->timing_owner = r;
  }
  idx->timing_pending = true;
  idx->timing_fill_launches = n_fill ? 1 : 0;   // (async_fill: ms_fill is what the first stream saw of it, ~0; vs_result_fill_ms has the kernel's time)
  // async_submit: the batch is enqueued, its sizes are known (the plan's totals; a speculative batch: result_sizes) and its buffers are the result's -- the call
  // returns here.  Whatever reads the result (copies, digests, packs, the next batch's kernels that reuse the temporaries
  // released below) is ordered behind the batch on the handle's stream; the timing events are read when asked for.
  if (!async_submit) {
    uint64_t* done = idx->pinned + vs_index::kPinDone;
    const uint64_t seq = ++idx->done_seq;
    hipLaunchKernelGGL(k_post_done, dim3(1), dim3(1), 0, idx->stream, done, seq);
    HIP_TRY(hipGetLastError());
    VS_TRY(wait_posted(idx, done, seq, 2000));
    idx->batch_in_flight = false;
  }
  if (async_submit) {
    // the batch's completion event (a lean batch's last kernel is its expansion: the result's own event behind that is it);
    // the call's temporaries stay with the result until it has happened (vs_result_free, result_ready): a buffer in the
    // pool is never referenced by work in flight, whatever stream takes it next
    if (!lean) {
      VS_TRY(pooled_event(idx, &r->ev_done));
      HIP_TRY(hipEventRecord(r->ev_done, idx->stream));
    }
    r->pending = true;
    idx->batch_in_flight = true;
  }
  if (async_fill || async_submit) { r->bufs.insert(r->bufs.end(), scratch.bufs.begin(), scratch.bufs.end()); scratch.bufs.clear(); }
  scratch.release();
  if (!async_submit) VS_TRY(collect_timing(idx));
  return VS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The batches whose regions own PRIVATE rows (query types 4 and 5, the point queries, type 6 with share_lists = 0 or from
// a handful of regions' records) run through the same four stages, each a function that reads only what the stage
// before it left in the batch's context -- a mode cannot reach into another mode's temporaries:
//   batch_setup       per-region arrays of the result, regions (and sample ids) on the device
//   <mode>: bounds    k_region_bounds / k_bounds_from_records / k_point_bounds -- or the recording walk of types 4 / 5
//   batch_sizes       offsets of rows and arena by one scan; the totals cross in mapped host memory
//   batch_tables      rows (+ the expansion's per-row parameters) and arena of exactly that size
//   <mode>: rows      k_emit_headers + k_dedup_slow (+ k_has_var_filter) -- or k_emit_from_walk / the emitting walk
//   batch_fill        k_fill_carriers over the rows (async_fill: on the handle's second stream)
//   batch_finish      synchronise, temporaries back to the pool, phase times from the handle's events
// (A type-6 batch of more than 64 regions shares rows and lists between its regions: run_type6_shared above.)
// ---------------------------------------------------------------------------------------------------------------------
struct BatchCtx {
  vs_index* idx;
  vs_result* r;
  uint64_t n;
  ScratchBufs scratch;
  uint32_t* dsids = nullptr;     // one sample per region (types 4 / 5), on the device
  uint64_t* bad_ids = nullptr;   // ids that arrived in device memory: set by k_walk_setup / k_check_sample_ids, read with the batch's first sizes
  uint64_t* words = nullptr;     // a walking batch's flag words (the handle's, zero when the batch starts: batch_words): [0] bad_ids, [1] the walk's overflow word
  bool caps_done = false;        // the capacities of the recording walk were written by the batch's first kernel (k_walk_setup)
  bool lean_events = false;      // first and last event only (every event is a packet between two kernels of a string of dependent launches)
  bool resident = false;         // rows point into the index's resident arena: nothing is expanded
  bool async_fill = false;
  BatchCtx(vs_index* i, vs_result* res, uint64_t nn) : idx(i), r(res), n(nn), scratch(i) {}
};

// (`bad`: a word the batch's memset has cleared)
static int check_device_ids(vs_index* idx, const uint32_t* dsids, uint64_t n, uint64_t* bad) {
  hipLaunchKernelGGL(k_check_sample_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, work_stream(idx), dsids, n, idx->g.num_samples, bad);
  HIP_TRY(hipGetLastError());
  return VS_OK;
}
// The flag words of a walking batch (BatchCtx::words: ids out of range, the walk's overflow word).  They belong to the HANDLE and
// are zero between batches: only the part of a batch in front of its host wait writes them, the host reads them in that wait, and
// a batch that read anything but zeros (or did not get that far) leaves them marked for the one memset the next batch then pays --
// a stream of healthy batches pays none.
static int batch_words(vs_index* idx, uint64_t** words) {
  if (!idx->walk_words) {
    HIP_TRY(hipMalloc((void**)&idx->walk_words, 32));
    idx->image_allocs.push_back(idx->walk_words);
    idx->walk_words_dirty = true;
  }
  if (idx->walk_words_dirty) HIP_TRY(hipMemsetAsync(idx->walk_words, 0, 32, work_stream(idx)));
  idx->walk_words_dirty = true;   // (until this batch's host wait has read zeros: words_read_clean)
  *words = idx->walk_words;
  return VS_OK;
}
static void words_read_clean(vs_index* idx, uint64_t overflow_seen) { if (!overflow_seen) idx->walk_words_dirty = false; }
static int bad_ids_error(vs_index* idx) {
  return fail(VS_ERR_UNKNOWN_SAMPLE, "a sample id of the batch is out of range (%u samples)", idx->g.num_samples);
}

// walk_caps: -1 none; 0 / 1: the batch is a recording walk of query type 4 / 5 -- when its regions and sample ids are in device
// memory the capacities come from the same kernel that takes the copies (k_walk_setup: BatchCtx::caps_done).
static int batch_setup(BatchCtx& c, const vs_region* regions, bool regions_on_device, const uint32_t* sample_ids, int walk_caps = -1) {
  vs_index* idx = c.idx;
  vs_result* r = c.r;
  const uint64_t n = c.n;
  if (idx->srv_alive) VS_TRY(server_stop(idx));   // a throughput batch does not share the GPU with a polling server
  idx->timing_pending = false; idx->timing_owner = nullptr; idx->timing_total_only = false;
  DevResult& d = r->d;
  d.Q = n;
  uint64_t* dreg = nullptr;
  VS_TRY(ralloc(r, 2 * n, &dreg));
  d.regions = dreg;
  VS_TRY(ralloc(r, n, &d.q_flags));
  VS_TRY(ralloc(r, n, &d.q_g0));
  VS_TRY(ralloc(r, n, &d.q_nvar));
  VS_TRY(ralloc(r, n, &d.q_ncar));
  VS_TRY(ralloc(r, n + 1, &d.var_begin));
  VS_TRY(ralloc(r, n + 1, &d.car_base));
  VS_TRY(ralloc(r, n, &d.var_count));
  static_assert(sizeof(vs_region) == 16, "vs_region layout");
  if (c.lean_events) HIP_TRY(hipEventRecord(idx->ev[0], work_stream(idx)));   // (in front of the first kernel, not between two)
  const bool ids_on_device = sample_ids && n && is_device_ptr(sample_ids);
  if (n && (walk_caps >= 0 || ids_on_device)) {
    VS_TRY(batch_words(idx, &c.words));
    if (ids_on_device) c.bad_ids = c.words;
  }
  if (sample_ids && n) VS_TRY(ralloc(r, n, &c.dsids));
  if (n && walk_caps >= 0 && regions && (ids_on_device || !sample_ids) && is_device_ptr(regions)) {
    const dim3 grid((unsigned)((n + 255) / 256));
    if (walk_caps == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_walk_setup<true>), grid, dim3(256), 0, work_stream(idx), idx->d, d, reinterpret_cast<const uint64_t*>(regions), sample_ids,
                                           c.dsids, idx->g.num_samples, c.words);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_walk_setup<false>), grid, dim3(256), 0, work_stream(idx), idx->d, d, reinterpret_cast<const uint64_t*>(regions), sample_ids,
                            c.dsids, idx->g.num_samples, c.words);
    HIP_TRY(hipGetLastError());
    c.caps_done = true;
  } else {
    if (n && regions) HIP_TRY(hipMemcpyAsync(dreg, regions, n * 16, regions_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDefault, work_stream(idx)));
    else if (n) HIP_TRY(hipMemsetAsync(dreg, 0, n * 16, work_stream(idx)));
    if (sample_ids && n) {
      HIP_TRY(hipMemcpyAsync(c.dsids, sample_ids, n * 4, hipMemcpyDefault, work_stream(idx)));
      if (ids_on_device) VS_TRY(check_device_ids(idx, c.dsids, n, c.bad_ids));
    }
  }
  if (!c.lean_events) HIP_TRY(hipEventRecord(idx->ev[0], work_stream(idx)));
  return VS_OK;
}

// Offsets of the rows and of the arena from the per-region counts the bounds / the walk left (q_nvar, q_ncar); the totals
// arrive in mapped host memory (read after the caller's synchronisation: batch_totals).
static int batch_sizes(BatchCtx& c) {
  DevResult& d = c.r->d;
  return scan_offsets(c.idx, d.q_nvar, d.q_ncar, c.n, d.var_begin, d.car_base, c.idx->pinned + vs_index::kPinBatch, &c.scratch.bufs);
}
static void batch_totals(const BatchCtx& c, uint64_t* rows, uint64_t* arena) {
  const volatile uint64_t* pin = c.idx->pinned + vs_index::kPinBatch;
  *rows = pin[0]; *arena = pin[1];
}

// The variant table (+ the per-row parameters of k_fill_carriers) and the arena, of exactly the sizes the scan found.
static int batch_tables(BatchCtx& c, uint64_t rows, uint64_t arena, bool shared_per_vertex, bool walking) {
  vs_index* idx = c.idx;
  vs_result* r = c.r;
  DevResult& d = r->d;
  d.A = rows;
  d.S = arena;
  VS_TRY(ralloc(r, d.A, &d.rows));
  if (!c.resident) {
    VS_TRY(ralloc(r, d.A, &d.r_class));
    VS_TRY(ralloc(r, d.A, &d.r_gt0));
  }
  r->n_rows_reported = d.A;
  r->shared_lists = shared_per_vertex;
  r->resident = c.resident;
  r->scattered_lists = shared_per_vertex || (c.resident && walking);   // a region's lists are not one arena range: texts come from a raw copy
  r->n_unique_sites = c.resident ? 0 : d.A;
  d.car_width = idx->d.wpc <= 63 ? 2 : 4;
  if (c.resident) d.carriers = idx->res_arena;
  else {
    uint8_t* a = nullptr;
    VS_TRY(ralloc(r, d.S * d.car_width + 16, &a));
    d.carriers = a;
  }
  if (!c.lean_events) HIP_TRY(hipEventRecord(idx->ev[2], idx->stream));
  return VS_OK;
}

// The carrier expansion over the rows; async_fill: on the handle's second stream behind an event, the call then returns
// once the FIRST stream is done and the call's temporaries stay with the result until it is freed.
static int batch_fill(BatchCtx& c, bool allow_async) {
  vs_index* idx = c.idx;
  vs_result* r = c.r;
  if (!c.lean_events) HIP_TRY(hipEventRecord(idx->ev[3], idx->stream));
  const uint64_t n_fill = c.resident ? 0 : r->d.A;
  c.async_fill = allow_async && idx->opts.async_fill && n_fill > 0;
  if (c.async_fill) {
    VS_TRY(ensure_fill_stream(idx));
    VS_TRY(result_events(r));
    HIP_TRY(hipEventRecord(idx->fill_ev[0], idx->stream));
    HIP_TRY(hipStreamWaitEvent(idx->fill_stream, idx->fill_ev[0], 0));
    HIP_TRY(hipEventRecord(r->ev_fill[0], idx->fill_stream));
    VS_TRY(fill_lists(idx, r->d, false, nullptr, n_fill, idx->fill_stream));
    HIP_TRY(hipEventRecord(r->ev_fill[1], idx->fill_stream));
    r->pending = true;
  } else VS_TRY(fill_lists(idx, r->d, false, nullptr, n_fill));
  idx->timing_fill_launches = n_fill ? 1 : 0;   // (async_fill: ms_fill is what the first stream saw of it, ~0; vs_result_fill_ms has the kernel's time)
  return VS_OK;
}

// enqueued: the batch returns when its last kernel is launched (the walking query types under async_submit): its temporaries stay
// with the result until the completion event recorded here, everything that reads the result is ordered behind it on the
// handle's stream, and the phase times are read from the handle's events when somebody asks (vs_index_last_timing)
static int batch_finish(BatchCtx& c, bool enqueued = false) {
  vs_index* idx = c.idx;
  HIP_TRY(hipEventRecord(idx->ev[4], idx->stream));
  if (c.lean_events) idx->timing_total_only = true;
  if (enqueued) {
    vs_result* r = c.r;
    VS_TRY(pooled_event(idx, &r->ev_done));
    HIP_TRY(hipEventRecord(r->ev_done, idx->stream));
    r->pending = true;
    idx->batch_in_flight = true;
    r->bufs.insert(r->bufs.end(), c.scratch.bufs.begin(), c.scratch.bufs.end());
    c.scratch.bufs.clear();
    idx->timing_pending = true;
    return VS_OK;
  }
  HIP_TRY(hipStreamSynchronize(idx->stream));
  idx->batch_in_flight = false;
  if (c.async_fill) { c.r->bufs.insert(c.r->bufs.end(), c.scratch.bufs.begin(), c.scratch.bufs.end()); c.scratch.bufs.clear(); }
  c.scratch.release();
  idx->timing_pending = true;
  return collect_timing(idx);
}

// Query type 6 with private rows (share_lists = 0, or few regions' records), and the point queries (point_mode 1 / 7: one
// next_variant_in_ref call per position -- closest_var / samples_has_var).
static int run_private_batch(vs_index* idx, const vs_region* regions, uint64_t n, vs_result* r, bool regions_on_device, const uint64_t* site_records,
                             uint32_t point_mode, const PointStrings* strings, bool allow_async) {
  BatchCtx c(idx, r, n);
  VS_TRY(batch_setup(c, regions, regions_on_device, nullptr));
  DevResult& d = r->d;
  if (n) {   // ---- bounds ----
    if (point_mode) hipLaunchKernelGGL(k_point_bounds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, idx->d, d, point_mode);
    else if (site_records) hipLaunchKernelGGL(k_bounds_from_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, idx->d, d, site_records);
    else hipLaunchKernelGGL(k_region_bounds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, idx->d, d);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(idx->ev[1], idx->stream));
  c.resident = idx->opts.resident_lists && idx->res_arena && !point_mode;
  VS_TRY(batch_sizes(c));
  if (c.resident) {   // the rows point into the index's arena; a region's lists ARE the arena range of its sites
    VS_TRY(ralloc(r, n, &d.q_car_len));   // (before the header kernels overwrite q_ncar with the reported carriers)
    hipLaunchKernelGGL(k_resident_bases, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, idx->stream, idx->d, d, idx->res_entries);
  }
  HIP_TRY(hipStreamSynchronize(idx->stream));
  uint64_t rows = 0, arena = 0;
  batch_totals(c, &rows, &arena);
  if (c.resident) arena = idx->res_entries;
  VS_TRY(batch_tables(c, rows, arena, false, false));
  if (n) {   // ---- rows ----
    if (c.resident) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_headers<false>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, idx->stream, idx->d, d);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_headers<true>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, idx->stream, idx->d, d);
    hipLaunchKernelGGL(k_dedup_slow, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, idx->stream, idx->d, d);
    if (strings) {
      uint8_t* dchars = nullptr;
      uint64_t* doff = nullptr;
      VS_TRY(ralloc(r, strings->chars->size() + 8, &dchars));
      VS_TRY(ralloc(r, strings->off->size(), &doff));
      if (!strings->chars->empty())
        HIP_TRY(hipMemcpyAsync(dchars, strings->chars->data(), strings->chars->size(), hipMemcpyHostToDevice, idx->stream));
      HIP_TRY(hipMemcpyAsync(doff, strings->off->data(), strings->off->size() * 8, hipMemcpyHostToDevice, idx->stream));
      hipLaunchKernelGGL(k_has_var_filter, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, idx->stream, idx->d, d,
                         (const uint8_t*)dchars, (const uint64_t*)doff);
    }
    HIP_TRY(hipGetLastError());
  }
  VS_TRY(batch_fill(c, allow_async && !point_mode));
  return batch_finish(c);
}

// Query type 6: more than 64 regions share rows and carrier lists (unless share_lists is 0).
static int run_type6(vs_index* idx, const vs_region* regions, uint64_t n, vs_result* r, bool regions_on_device, const uint64_t* site_records, bool allow_async) {
  if (idx->opts.share_lists && n > 64) return run_type6_shared(idx, regions, n, r, regions_on_device, site_records, allow_async);
  return run_private_batch(idx, regions, n, r, regions_on_device, site_records, 0, nullptr, allow_async);
}

// lanes per region of the one-chain walks of query types 2, 3 and 5 (k_sample_walk_sc, k_sample_seq): 1, or kScGroup running the same chain
template <int MODE>
static void launch_walk_sc(vs_index* idx, const DevResult& d, uint64_t n, const uint32_t* dsids, const WalkScratch& ws) {
  // the recording walk: cooperative (eight lanes per region, episodes in parallel) where the samples' event rows name slots
  if (MODE == 2 && idx->opts.t4_walk >= 2 && idx->d.t4_events && idx->d.seq_breaks && idx->d.t4_ev_shift == 0 && idx->opts.sc_group <= 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_walk_sc_coop<8>), dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, work_stream(idx), idx->d, d, dsids, ws);
    return;
  }
  if (idx->opts.sc_group > 1)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_walk_sc<MODE, kScGroup>), dim3((unsigned)((n * kScGroup + 255) / 256)), dim3(256), 0, work_stream(idx), idx->d, d, dsids, ws);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_walk_sc<MODE, 1>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, work_stream(idx), idx->d, d, dsids, ws);
}
template <int MODE, int PASS>
static void launch_sample_seq(vs_index* idx, const DevSeqResult& q, uint64_t n) {
  // the single walk: cooperative (eight lanes per region, episodes in parallel) where the samples' event rows name slots
  if (PASS == 2 && idx->opts.t4_walk >= 2 && idx->d.t4_events && idx->d.seq_breaks && idx->d.t4_ev_shift == 0 && idx->opts.sc_group <= 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_seq_coop<MODE, 8>), dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, work_stream(idx), idx->d, q);
    return;
  }
  if (idx->opts.sc_group > 1)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_seq<MODE, PASS, kScGroup>), dim3((unsigned)((n * kScGroup + 255) / 256)), dim3(256), 0, work_stream(idx), idx->d, q);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_seq<MODE, PASS, 1>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, work_stream(idx), idx->d, q);
}

// Query types 4 (walk_mode 4: get_sample_var_in_ref, one sample's path in reference coordinates) and 5 (walk_mode 5:
// get_sample_var_in_sample).  sample_id: the one sample of the batch, or kNone with one sample per region in sample_ids.
//   walk   capacities from the type-6 bounds of the same regions (k_walk_caps_sc for type 5), ONE recording walk
//          (cooperative, serial with jumps, or literal: option t4_walk); a region that outgrows its capacity -- not seen
//          in practice; option force_fallbacks -- sends the batch down the count-then-emit pair of walks
//   claims one carrier list per reported VERTEX, shared by the rows that report it (k_t4_claim)
static int run_walk_batch_once(vs_index* idx, const vs_region* regions, uint64_t n, vs_result* r, uint32_t sample_id, const uint32_t* sample_ids, int walk_mode,
                               bool speculate, bool* refused);
// ONE host wait per batch (round 5): the recording walk's scratch is sized from the handle's previous batch of the kind; a batch
// that does not fit is refused on the device (WalkAdmit: the walk's first look) and redone here with the exact size -- the first batch of a handle
// and a batch 12 % bigger than any before it pay the second wait, a steady stream of batches never does.
static int run_walk_batch(vs_index* idx, const vs_region* regions, uint64_t n, vs_result* r, uint32_t sample_id, const uint32_t* sample_ids, int walk_mode) {
  bool refused = false;
  VS_TRY(run_walk_batch_once(idx, regions, n, r, sample_id, sample_ids, walk_mode, true, &refused));
  if (!refused) return VS_OK;
  return run_walk_batch_once(idx, regions, n, r, sample_id, sample_ids, walk_mode, false, &refused);
}
static int run_walk_batch_once(vs_index* idx, const vs_region* regions, uint64_t n, vs_result* r, uint32_t sample_id, const uint32_t* sample_ids, int walk_mode,
                               bool speculate, bool* refused) {
  BatchCtx c(idx, r, n);
  // everything in front of the host wait on the handle's second stream, beside the previous batch's rows and expansion (PreStream)
  const bool aside = idx->opts.async_submit && speculate && n > 0 && !idx->opts.force_fallbacks && !idx->opts.walk_stats && !idx->opts.lat_debug;
  if (aside) VS_TRY(ensure_plan_stream(idx));
  PreStream pre(idx, aside ? idx->plan_stream : nullptr);
  c.lean_events = aside && !idx->opts.phase_events;
  VS_TRY(batch_setup(c, regions, false, sample_ids, idx->opts.force_fallbacks ? -1 : (walk_mode == 5 ? 1 : 0)));
  DevResult& d = r->d;
  ScratchBufs& scratch = c.scratch;
  const uint32_t* dsids = c.dsids;
  WalkScratch ws{};
  uint64_t ws_capacity = 0;
  bool single_walk = false, speculative = false;
  const uint64_t* cap_total_dev = nullptr;
  DevImage dwalk = idx->d;   // what the type-4 walk sees: with or without the event bitmaps
  if (idx->opts.t4_walk == 0) dwalk.t4_events = nullptr;
  auto counting_walk = [&]() {
    if (walk_mode == 5) launch_walk_sc<0>(idx, d, n, dsids, WalkScratch{});
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_walk<0>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, work_stream(idx), dwalk, d, sample_id, dsids, WalkScratch{});
  };
  if (c.bad_ids && idx->opts.force_fallbacks) {   // (otherwise read with the capacities' total below: nothing before that looks at an id)
    uint64_t bad = 0;
    VS_TRY(read_device_words(idx, c.bad_ids, &bad));
    if (bad) return bad_ids_error(idx);
  }
  if (n && !idx->opts.force_fallbacks) {
    if (c.caps_done) {}   // (k_walk_setup)
    else if (walk_mode == 5) hipLaunchKernelGGL(k_walk_caps_sc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, work_stream(idx), idx->d, d);
    else hipLaunchKernelGGL(k_region_bounds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, work_stream(idx), idx->d, d);
    uint64_t* cap_begin = nullptr;
    VS_TRY(dev_alloc(idx, (n + 1) * 8, (void**)&cap_begin, &scratch.bufs));
    VS_TRY(exclusive_scan<uint64_t>(idx, d.q_nvar, n, cap_begin, &scratch.bufs));
    uint64_t cap_total = 0, bad = 0;
    uint64_t& hint = idx->walk_cap_hint[walk_mode == 5 ? 1 : 0];
    speculative = speculate && hint > 0 && idx->opts.async_submit;
    if (speculative) cap_total = hint;   // (the walk itself holds the batch to it: WalkAdmit)
    else {
      VS_TRY(read_device_words(idx, cap_begin + n, &cap_total, c.bad_ids, &bad));
      if (bad) return bad_ids_error(idx);
      hint = cap_total + cap_total / 8 + 1024;
    }
    cap_total_dev = cap_begin + n;
    ws_capacity = cap_total;
    ws.cap_begin = cap_begin;
    VS_TRY(dev_alloc(idx, cap_total * 8 + 8, (void**)&ws.pos, &scratch.bufs));
    VS_TRY(dev_alloc(idx, cap_total * 4 + 8, (void**)&ws.cur, &scratch.bufs));
    VS_TRY(dev_alloc(idx, cap_total * 4 + 8, (void**)&ws.ro, &scratch.bufs));
    VS_TRY(dev_alloc(idx, cap_total * 4 + 8, (void**)&ws.rl, &scratch.bufs));
    VS_TRY(dev_alloc(idx, cap_total * 4 + 8, (void**)&ws.ao, &scratch.bufs));
    VS_TRY(dev_alloc(idx, cap_total * 4 + 8, (void**)&ws.al, &scratch.bufs));
    ws.overflow = c.words + 1;   // (zero: batch_words)
    if (speculative) ws.admit = WalkAdmit{cap_total_dev, cap_total, c.bad_ids};
#ifdef VS_TUNING
    if (idx->opts.walk_stats) {
      VS_TRY(dev_alloc(idx, 128, (void**)&ws.stats, &scratch.bufs));
      HIP_TRY(hipMemsetAsync(ws.stats, 0, 128, work_stream(idx)));
    }
#endif
    if (walk_mode == 5) launch_walk_sc<2>(idx, d, n, dsids, ws);
    else if (dwalk.t4_events && idx->opts.t4_walk == 2)   // 8 lanes per region: the episodes of a region run in parallel
    {
      if (dwalk.t4_hold) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_walk_coop<8, false>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, work_stream(idx), dwalk, d, sample_id, dsids, ws);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_walk_coop<8, true>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, work_stream(idx), dwalk, d, sample_id, dsids, ws);   // explicit ids
    }
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_walk<2>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, work_stream(idx), dwalk, d, sample_id, dsids, ws);
    HIP_TRY(hipGetLastError());
    single_walk = true;
  }
  uint64_t walk_overflow = 0;
#ifdef VS_TUNING
  if (single_walk && ws.stats) {
    unsigned long long h[16];
    HIP_TRY(hipMemcpyAsync(h, ws.stats, 128, hipMemcpyDeviceToHost, work_stream(idx)));
    HIP_TRY(hipStreamSynchronize(work_stream(idx)));
    const double nr = h[0] ? (double)h[0] : 1.0;
    fprintf(stderr, "walk stats: %llu regions | per region: search iterations %.1f (literal %.2f), jumps %.1f, steps %.1f, variants %.1f | "
            "mean ticks(10ns): search %.0f walk %.0f head %.0f | max: search iters %llu steps %llu search ticks %llu walk ticks %llu\n",
            h[0], h[1] / nr, h[2] / nr, h[3] / nr, h[4] / nr, h[7] / nr, h[5] / nr, h[6] / nr, h[12] / nr, h[8], h[9], h[10], h[11]);
  }
#endif
  if (!single_walk && n) { counting_walk(); HIP_TRY(hipGetLastError()); }
  if (!c.lean_events) HIP_TRY(hipEventRecord(idx->ev[1], work_stream(idx)));
  c.resident = idx->opts.resident_lists && idx->res_arena && single_walk;
  ListClaims lc{};
  bool share_t4 = false;   // one carrier list per reported VERTEX, shared by the rows that report it
  int claim_tab = -1;      // which of the handle's two claim tables the batch took
  uint64_t t4_arena = 0;
  VS_TRY(batch_sizes(c));

  if (single_walk && idx->opts.share_lists && n > 64 && !c.resident) {
    // (rows <= the scratch capacity the walk was given: the claim arrays can be sized before the row count is known)
    const uint64_t cap_rows = ws_capacity;
    const uint64_t gen = ++idx->t4_gen;
    const int tab = (int)(gen & 1);
    if (!idx->t4_claim[tab]) {
      HIP_TRY(hipMalloc((void**)&idx->t4_claim[tab], (idx->d.V + 1) * 8));
      idx->image_allocs.push_back(idx->t4_claim[tab]);
      HIP_TRY(hipMemsetAsync(idx->t4_claim[tab], 0, (idx->d.V + 1) * 8, work_stream(idx)));
    }
    // the table's last user (two batches back) has to be through with it: its completion event, if it returned when it was enqueued
    // (the event is the result's, pooled: should it have been recorded again since, this waits for something later -- never for less)
    if (idx->pre && idx->t4_claim_done[tab]) HIP_TRY(hipStreamWaitEvent(idx->pre, idx->t4_claim_done[tab], 0));
    idx->t4_claim_done[tab] = nullptr;
    claim_tab = tab;
    uint64_t* own_base = nullptr;
    VS_TRY(dev_alloc(idx, (cap_rows + 1) * 4, (void**)&lc.own_pad, &scratch.bufs));
    VS_TRY(dev_alloc(idx, (cap_rows + 1) * 8, (void**)&lc.own_off, &scratch.bufs));
    VS_TRY(dev_alloc(idx, (n + 1) * 8, (void**)&lc.q_own, &scratch.bufs));
    VS_TRY(dev_alloc(idx, (n + 2) * 8, (void**)&own_base, &scratch.bufs));
    lc.claim = idx->t4_claim[tab]; lc.gen = gen; lc.own_base = own_base; lc.rows_cap = cap_rows;
    hipLaunchKernelGGL(k_t4_claim, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, work_stream(idx), idx->d, d, ws, lc);
    if (n <= kScanSmallMax)   // the scan and the offsets in one launch
      hipLaunchKernelGGL(k_t4_offsets_small, dim3((unsigned)((n + kScanSmallTile - 1) / kScanSmallTile)), dim3(kScanSmallBlock), 0, work_stream(idx), d, lc, own_base);
    else {
      VS_TRY(exclusive_scan<uint64_t>(idx, (const uint64_t*)lc.q_own, n, own_base, &scratch.bufs));
      hipLaunchKernelGGL(k_t4_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, work_stream(idx), d, lc);
    }
    share_t4 = true;
  }
  // the walk's overflow flag, the claims' arena total and what the capacities added up to (the scan's totals are in mapped memory
  // already): THE host wait of a batch whose scratch was sized from the one before
  uint64_t cap_seen = 0;
  VS_TRY(read_device_words(idx, single_walk ? ws.overflow : nullptr, &walk_overflow, share_t4 ? lc.own_base + n : nullptr, &t4_arena,
                           speculative ? cap_total_dev : nullptr, &cap_seen));
  pre.done();   // (that part is over: the host has seen its last kernel's words; the rest goes on the handle's stream)
  if (c.words) words_read_clean(idx, walk_overflow);   // (ids out of range: verdict 3 in the same word, or the early return above)
  if (speculative) {
    uint64_t& hint = idx->walk_cap_hint[walk_mode == 5 ? 1 : 0];
    if (walk_overflow == 3) return bad_ids_error(idx);
    if (walk_overflow == 2) {   // refused: the caller redoes the batch with the exact size (and the next batches get the bigger scratch)
      hint = cap_seen + cap_seen / 8 + 1024;
      *refused = true;
      return VS_OK;
    }
    hint_after_batch(idx, walk_mode == 5 ? 1 : 0, hint, cap_seen);   // (batches that have become much smaller for a while: so does the scratch)
  }
  uint64_t rows = 0, arena = 0;
  batch_totals(c, &rows, &arena);
  if (share_t4 && !walk_overflow) arena = t4_arena;
  if (c.resident && !walk_overflow) arena = idx->res_entries;
  if (single_walk && walk_overflow) {   // redo the sizes with a counting walk; the emitting walk follows below
    single_walk = false;
    share_t4 = false;
    c.resident = false;
    counting_walk();
    VS_TRY(batch_sizes(c));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    batch_totals(c, &rows, &arena);
  }
  VS_TRY(batch_tables(c, rows, arena, share_t4, true));
  if (n) {   // ---- rows ----
    const dim3 g16((unsigned)((n + 15) / 16)), g64((unsigned)((n + 63) / 64));
    if (single_walk && walk_mode == 5 && c.resident) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_from_walk<false, 2>), g16, dim3(256), 0, idx->stream, idx->d, d, ws, lc);
    else if (single_walk && walk_mode == 5 && share_t4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_from_walk<false, 1>), g16, dim3(256), 0, idx->stream, idx->d, d, ws, lc);
    else if (single_walk && walk_mode == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_from_walk<false, 0>), g16, dim3(256), 0, idx->stream, idx->d, d, ws, lc);
    else if (single_walk && c.resident) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_from_walk<true, 2>), g16, dim3(256), 0, idx->stream, idx->d, d, ws, lc);
    else if (single_walk && share_t4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_from_walk<true, 1>), g16, dim3(256), 0, idx->stream, idx->d, d, ws, lc);
    else if (single_walk) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_from_walk<true, 0>), g16, dim3(256), 0, idx->stream, idx->d, d, ws, lc);
    else if (walk_mode == 5) launch_walk_sc<1>(idx, d, n, dsids, WalkScratch{});
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_walk<1>), g64, dim3(64), 0, idx->stream, dwalk, d, sample_id, dsids, WalkScratch{});
    HIP_TRY(hipGetLastError());
  }
  VS_TRY(batch_fill(c, false));
  // the batch is enqueued: return (async_submit), unless a tuning aid wants the host in the loop
  const bool enqueued = idx->opts.async_submit && single_walk && !idx->opts.walk_stats && !idx->opts.lat_debug;
  VS_TRY(batch_finish(c, enqueued));
  if (claim_tab >= 0 && enqueued) idx->t4_claim_done[claim_tab] = r->ev_done;   // (not enqueued: the stream has been synchronised)
  return VS_OK;
}

template <typename T>
static int fetch(vs_index* idx, std::vector<T>& h, const T* dptr, size_t n) {
  h.resize(n);
  if (n) HIP_TRY(hipMemcpyAsync(h.data(), dptr, n * sizeof(T), hipMemcpyDeviceToHost, idx->stream));
  return VS_OK;
}

// carriers [first, first + n) of the arena as 32-bit words (id | gt << 29), whatever the arena's width
static int fetch_carriers(vs_result* r, uint64_t first, uint64_t n, std::vector<uint32_t>& out) {
  VS_TRY(result_sizes(r));
  vs_index* idx = r->idx;
  VS_TRY(result_ready(r));
  out.resize(n);
  if (n == 0) return VS_OK;
  if (r->d.car_width == 4) {
    HIP_TRY(hipMemcpyAsync(out.data(), (const uint32_t*)r->d.carriers + first, n * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    return VS_OK;
  }
  // 16-bit arena: copy into the upper half of the output buffer, widen in place from the front
  uint16_t* narrow = reinterpret_cast<uint16_t*>(out.data()) + n;
  HIP_TRY(hipMemcpyAsync(narrow, (const uint16_t*)r->d.carriers + first, n * 2, hipMemcpyDeviceToHost, idx->stream));
  HIP_TRY(hipStreamSynchronize(idx->stream));
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t c = narrow[i];
    out[i] = (c & 0x1FFFu) | ((c >> 13) << 29);
  }
  return VS_OK;
}

// the per-region arrays (Q-sized: a few MB for the largest batches)
static int fetch_region_meta(vs_result* r) {
  VS_TRY(result_sizes(r));
  if (r->have_meta) return VS_OK;
  vs_index* idx = r->idx;
  HIP_TRY(hipSetDevice(idx->device));
  const DevResult& d = r->d;
  VS_TRY(fetch(idx, r->h_flags, (const uint8_t*)d.q_flags, d.Q));
  VS_TRY(fetch(idx, r->h_var_begin, (const uint64_t*)d.var_begin, d.Q + 1));
  VS_TRY(fetch(idx, r->h_car_base, (const uint64_t*)d.car_base, d.Q + 1));
  VS_TRY(fetch(idx, r->h_var_count, (const uint64_t*)d.var_count, d.Q));
  VS_TRY(fetch(idx, r->h_nvar, (const uint64_t*)d.q_nvar, d.Q));
  if (d.q_car_len) VS_TRY(fetch(idx, r->h_car_len, (const uint64_t*)d.q_car_len, d.Q));
  HIP_TRY(hipStreamSynchronize(idx->stream));
  if (!d.q_car_len) {
    r->h_car_len.resize(d.Q);
    for (uint64_t q = 0; q < d.Q; ++q) r->h_car_len[q] = r->h_car_base[q + 1] - r->h_car_base[q];
  }
  for (auto& f : r->h_flags) f &= (uint8_t)~kRegionSlow;
  r->have_meta = true;
  return VS_OK;
}

// The whole variant table on the host, and the VIEW built from it: every region's rows expanded back to back (shared
// rows once per region that reports them) as the structure-of-arrays vs_result_view promises.
static int fetch_headers(vs_result* r) {
  VS_TRY(result_sizes(r));
  if (r->have_headers) return VS_OK;
  vs_index* idx = r->idx;
  VS_TRY(fetch_region_meta(r));
  const DevResult& d = r->d;
  VS_TRY(fetch(idx, r->h_rows, (const VariantRow*)d.rows, d.A));
  HIP_TRY(hipStreamSynchronize(idx->stream));
  r->h_view_begin.assign(d.Q + 1, 0);
  for (uint64_t q = 0; q < d.Q; ++q) r->h_view_begin[q + 1] = r->h_view_begin[q] + r->h_nvar[q];
  const uint64_t ns = r->h_view_begin[d.Q];
  r->h_pos.resize(ns); r->h_car_begin.resize(ns); r->h_car_begin_view.resize(ns);
  r->h_ref_off.resize(ns); r->h_ref_len.resize(ns); r->h_alt_off.resize(ns); r->h_alt_len.resize(ns);
  r->h_vflags.resize(ns); r->h_car_count.resize(ns);
  uint64_t acc = 0;
  for (uint64_t q = 0; q < d.Q; ++q) {
    const VariantRow* src = r->h_rows.data() + r->h_var_begin[q];
    for (uint64_t j = 0, a = r->h_view_begin[q]; j < r->h_nvar[q]; ++j, ++a) {
      const VariantRow& v = src[j];
      const uint32_t cnt = v.count_flags & ~kRowDropped;
      r->h_pos[a] = v.pos; r->h_ref_off[a] = v.ref_off; r->h_ref_len[a] = v.ref_len; r->h_alt_off[a] = v.alt_off; r->h_alt_len[a] = v.alt_len;
      r->h_vflags[a] = (v.count_flags & kRowDropped) ? VS_VAR_DROPPED : 0u;
      r->h_car_count[a] = cnt; r->h_car_begin[a] = v.car_begin;
      r->h_car_begin_view[a] = acc; acc += cnt;
    }
  }
  r->n_view_carriers = acc;
  r->have_headers = true;
  return VS_OK;
}

// Query types 2 and 3: count pieces and bytes per region, scan, emit the piece list, decode it.  ONE host wait per batch
// (the byte total): the piece list is sized from the handle's previous batch of the type, as the walking types' scratch is.
static int run_sample_seq_once(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids, int mode, vs_result* r, bool speculate, bool* refused);
static int run_sample_seq(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids, int mode, vs_result* r) {
  bool refused = false;
  VS_TRY(run_sample_seq_once(idx, regions, n, sample_ids, mode, r, true, &refused));
  if (!refused) return VS_OK;
  return run_sample_seq_once(idx, regions, n, sample_ids, mode, r, false, &refused);
}
static int run_sample_seq_once(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids, int mode, vs_result* r, bool speculate, bool* refused) {
  DevSeqResult& q = r->sq;
  // everything in front of the host wait on the handle's second stream, beside the previous batch's k_copy_segments (PreStream)
  const bool aside = idx->opts.async_submit && speculate && n > 0 && !idx->opts.force_fallbacks && !idx->opts.lat_debug;
  if (aside) VS_TRY(ensure_plan_stream(idx));
  PreStream pre(idx, aside ? idx->plan_stream : nullptr);
  if (idx->srv_alive) VS_TRY(server_stop(idx));
  idx->timing_pending = false; idx->timing_owner = nullptr; idx->timing_total_only = false;
  q.Q = n;
  r->d.Q = n;
  uint64_t* dreg = nullptr;
  uint32_t* dsids = nullptr;
  VS_TRY(ralloc(r, 2 * n, &dreg));
  VS_TRY(ralloc(r, n, &dsids));
  q.regions = dreg; q.sids = dsids;
  VS_TRY(ralloc(r, n, &q.q_flags));
  VS_TRY(ralloc(r, n, &q.q_nseg));
  VS_TRY(ralloc(r, n, &q.q_nbytes));
  VS_TRY(ralloc(r, n + 1, &q.seg_begin));
  VS_TRY(ralloc(r, n + 1, &q.byte_begin));
  HIP_TRY(hipEventRecord(idx->ev[0], work_stream(idx)));   // (in front of the first kernel, not between two)
  uint64_t *bad_ids = nullptr, *words = nullptr;   // (the batch's flag words, batch_words: [0] ids out of range, [1] the walk's overflow word)
  bool caps_done = false;
  if (n) {
    const bool ids_on_device = is_device_ptr(sample_ids);
    VS_TRY(batch_words(idx, &words));
    if (ids_on_device) bad_ids = words;
    if (ids_on_device && !idx->opts.force_fallbacks && is_device_ptr(regions)) {   // copies, id check and piece capacities in one launch
      hipLaunchKernelGGL(k_seq_setup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, work_stream(idx), idx->d, q, reinterpret_cast<const uint64_t*>(regions), sample_ids,
                         idx->g.num_samples, words, (uint32_t)(mode == 3));
      HIP_TRY(hipGetLastError());
      caps_done = true;
    } else {
      HIP_TRY(hipMemcpyAsync(dreg, regions, n * 16, hipMemcpyDefault, work_stream(idx)));
      HIP_TRY(hipMemcpyAsync(dsids, sample_ids, n * 4, hipMemcpyDefault, work_stream(idx)));
      if (ids_on_device) VS_TRY(check_device_ids(idx, dsids, n, bad_ids));
    }
  }
  ScratchBufs scratch(idx);
  uint64_t totals[2] = {0, 0};
  if (bad_ids && idx->opts.force_fallbacks) {
    uint64_t bad = 0;
    VS_TRY(read_device_words(idx, bad_ids, &bad));
    if (bad) return bad_ids_error(idx);
  }
  // Single walk: piece capacities from the reference range of each region, one recording walk, then the byte
  // offsets.  A region that outgrows its capacity sends the batch down the count-then-emit path.
  bool single_walk = n > 0 && !idx->opts.force_fallbacks;
  if (single_walk) {
    q.overflow = words + 1;
    q.admit = WalkAdmit{};   // (a batch that is redone with the exact size keeps its result: nothing to admit then)
    if (!caps_done) hipLaunchKernelGGL(k_seq_caps, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, work_stream(idx), idx->d, q, (uint32_t)(mode == 3));
    VS_TRY(exclusive_scan<uint64_t>(idx, q.q_nseg, n, q.seg_begin, &scratch.bufs));
    uint64_t bad = 0;
    uint64_t& hint = idx->seq_cap_hint[mode == 3 ? 1 : 0];
    const bool speculative = speculate && hint > 0 && idx->opts.async_submit;
    if (speculative) {
      totals[0] = hint;
      q.admit = WalkAdmit{q.seg_begin + n, hint, bad_ids};   // (the walk's first look: seq_void)
    } else {
      VS_TRY(read_device_words(idx, q.seg_begin + n, &totals[0], bad_ids, &bad));
      if (bad) return bad_ids_error(idx);
      hint = totals[0] + totals[0] / 8 + 1024;
    }
    VS_TRY(ralloc(r, totals[0], &q.seg_src));
    VS_TRY(ralloc(r, totals[0], &q.seg_len));
    VS_TRY(ralloc(r, totals[0], &q.seg_dst));
    q.relative = 1;
    if (mode == 2) launch_sample_seq<2, 2>(idx, q, n);
    else launch_sample_seq<3, 2>(idx, q, n);
    HIP_TRY(hipGetLastError());
    VS_TRY(exclusive_scan<uint64_t>(idx, q.q_nbytes, n, q.byte_begin, &scratch.bufs));
    uint64_t over = 0, cap_seen = 0;
    VS_TRY(read_device_words(idx, q.byte_begin + n, &totals[1], q.overflow, &over, speculative ? q.seg_begin + n : nullptr, &cap_seen));
    pre.done();   // (the rest goes on the handle's stream)
    words_read_clean(idx, over);
    if (speculative) {
      if (over == 3) return bad_ids_error(idx);
      if (over == 2) { hint = cap_seen + cap_seen / 8 + 1024; *refused = true; return VS_OK; }
      hint_after_batch(idx, mode == 3 ? 3 : 2, hint, cap_seen);
    }
    if (over) { single_walk = false; q.relative = 0; }
    else {
      VS_TRY(ralloc(r, totals[1], &q.chars));
      r->seq_bytes = totals[1];
      hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, idx->stream, idx->d, q);
      HIP_TRY(hipGetLastError());
    }
  }
  if (!single_walk) {
    if (n) {
      if (mode == 2) launch_sample_seq<2, 0>(idx, q, n);
      else launch_sample_seq<3, 0>(idx, q, n);
      HIP_TRY(hipGetLastError());
    }
    VS_TRY(exclusive_scan<uint64_t>(idx, q.q_nseg, n, q.seg_begin, &scratch.bufs));
    VS_TRY(exclusive_scan<uint64_t>(idx, q.q_nbytes, n, q.byte_begin, &scratch.bufs));
    VS_TRY(read_device_words(idx, q.seg_begin + n, &totals[0], q.byte_begin + n, &totals[1]));
    VS_TRY(ralloc(r, totals[0], &q.seg_src));
    VS_TRY(ralloc(r, totals[0], &q.seg_len));
    VS_TRY(ralloc(r, totals[0], &q.seg_dst));
    VS_TRY(ralloc(r, totals[1], &q.chars));
    r->seq_bytes = totals[1];
    if (n) {
      if (mode == 2) launch_sample_seq<2, 1>(idx, q, n);
      else launch_sample_seq<3, 1>(idx, q, n);
      hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, idx->stream, idx->d, q);
      HIP_TRY(hipGetLastError());
    }
  }
  HIP_TRY(hipEventRecord(idx->ev[4], idx->stream));
  if (idx->opts.async_submit && single_walk) {   // the batch is enqueued: return; its temporaries stay with the result until its completion event
    VS_TRY(pooled_event(idx, &r->ev_done));
    HIP_TRY(hipEventRecord(r->ev_done, idx->stream));
    r->pending = true;
    idx->batch_in_flight = true;
    r->bufs.insert(r->bufs.end(), scratch.bufs.begin(), scratch.bufs.end());
    scratch.bufs.clear();
    idx->timing_pending = true; idx->timing_total_only = true;
    return VS_OK;
  }
  HIP_TRY(hipStreamSynchronize(idx->stream));
  scratch.release();
  vs_timing& t = idx->timing;
  t.ms_bounds = t.ms_scan = t.ms_emit = t.ms_fill = 0.f;
  HIP_TRY(hipEventElapsedTime(&t.ms_total, idx->ev[0], idx->ev[4]));
  t.fill_launches = 0;
  return VS_OK;
}

// the per-region records of any result on the handle's stream: site ranges + counts (k_pack_regions: query types 6, 4, 5, 1, 7)
// or pieces + bytes (k_pack_seq_regions: types 2, 3)
static int launch_pack_regions(vs_result* r, uint64_t* dst, uint64_t region_base) {
  VS_TRY(result_sizes(r));
  vs_index* idx = r->idx;
  const uint64_t n = r->d.Q;
  if (!n) return VS_OK;
  if (r->kind == 2 || r->kind == 3) hipLaunchKernelGGL(k_pack_seq_regions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, r->sq, dst, region_base);
  else hipLaunchKernelGGL(k_pack_regions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, r->d, dst, region_base);
  HIP_TRY(hipGetLastError());
  return VS_OK;
}

static int fetch_sequences(vs_result* r) {
  if (r->have_seq) return VS_OK;
  vs_index* idx = r->idx;
  HIP_TRY(hipSetDevice(idx->device));
  VS_TRY(fetch(idx, r->h_flags, (const uint8_t*)r->sq.q_flags, r->sq.Q));
  VS_TRY(fetch(idx, r->h_byte_begin, (const uint64_t*)r->sq.byte_begin, r->sq.Q + 1));
  VS_TRY(fetch(idx, r->h_chars, (const uint8_t*)r->sq.chars, r->seq_bytes));
  HIP_TRY(hipStreamSynchronize(idx->stream));
  r->h_chars.push_back(0);
  r->have_seq = true;
  return VS_OK;
}

// Host-side SIZES of a region's answer: the arithmetic of k_region_bounds (kernels.hip.h: region_bounds_of) on the
// host copies of the rank structure, the ref-path tables and the arena prefix -- slot count and padded carrier count,
// nothing else.  The latency path allocates exactly this much and launches exactly the waves it needs; the device
// works the bounds out again for itself and refuses to write if it ever disagrees.
static uint32_t host_rank1(const HostImage& im, uint64_t p) {
  const uint64_t nbits = (uint64_t)im.bits.size() * 64;
  if (p > nbits) p = nbits;
  const uint64_t blk = p >> 9, w1 = p >> 6;
  uint32_t r = im.blk_rank[blk];
  for (uint64_t w = blk << 3; w < w1; ++w) r += (uint32_t)__builtin_popcountll(im.bits[w]);
  const uint32_t rem = (uint32_t)(p & 63);
  if (rem) r += (uint32_t)__builtin_popcountll(im.bits[w1] & ((1ULL << rem) - 1));
  return r;
}
static void host_region_size(const vs_index* idx, uint64_t x, uint64_t y, uint64_t* slots, uint64_t* arena_entries) {
  const HostImage& im = idx->im;
  *slots = 0; *arena_entries = 0;
  if (x < 1 || x > im.ref_length || !(x < y)) return;
  const uint32_t rx = host_rank1(im, x);
  if (rx >= im.R || !((uint64_t)im.idx_pos[rx] - 1 <= y)) return;   // Index::is_empty
  const uint64_t rf = (x >= im.ref_length) ? im.R - 1 : (uint64_t)rx - 1;
  const uint32_t s0 = im.rank_to_slot[rf];
  uint32_t s1 = im.rank_to_slot[host_rank1(im, y - 1)] - 1;
  if (s1 < s0) s1 = s0;
  const uint32_t g0 = im.rp_cand_prefix[s0], g1 = im.rp_cand_prefix[s1];
  *slots = g1 - g0;
  *arena_entries = idx->h_carpre[g1] - idx->h_carpre[g0];
}

// ---- resident query server (kernels.hip.h: k_query_server) ----
constexpr uint64_t kSrvMaxTasks = 2 * 16 * 4;       // (16 blocks = 64 waves share a request's 4-slot tasks) larger requests are better
                                                    // off with a launch sized for them
constexpr uint32_t kSrvStreak = 4;                  // opts.server == 1: this many small queries back to back start the server
constexpr auto kSrvStreakGap = std::chrono::microseconds(150);   // "back to back": the caller came back within this of the last answer
constexpr uint64_t kSrvLifeTicks = 2000000;         // device clock, 100 MHz: 20 ms, then the kernel leaves by itself
constexpr uint64_t kSrvIdleTicks = 100000;          // ... or 1 ms after the last request
constexpr auto kSrvHostLife = std::chrono::milliseconds(14);     // the host replaces a server older than this
constexpr auto kSrvHostIdle = std::chrono::microseconds(600);    // ... and does not trust one that has been idle this long

static ServerRequest* srv_request(vs_index* idx) { return reinterpret_cast<ServerRequest*>(idx->pinned + vs_index::kPinSrvReq); }

static int server_stop(vs_index* idx) {
  if (!idx->srv_stream) return VS_OK;
  if (idx->srv_alive) {
    volatile uint64_t* head = &srv_request(idx)->head;
    *head = ~0ULL;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    idx->srv_alive = false;
  }
  HIP_TRY(hipStreamSynchronize(idx->srv_stream));   // bounded: the kernel leaves on the exit word, or by its own clock
  return VS_OK;
}

static int server_ensure(vs_index* idx) {
  const auto now = std::chrono::steady_clock::now();
  if (idx->srv_alive && now - idx->srv_started < kSrvHostLife && now - idx->srv_last < kSrvHostIdle) return VS_OK;
  if (!idx->srv_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&idx->srv_stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void**)&idx->srv_counter, 8));
    idx->image_allocs.push_back(idx->srv_counter);
  }
  VS_TRY(server_stop(idx));
  ServerRequest* rq = srv_request(idx);
  memset(rq, 0, 64);
  std::atomic_thread_fence(std::memory_order_seq_cst);
  HIP_TRY(hipMemsetAsync(idx->srv_counter, 0, 8, idx->srv_stream));
  const uint32_t gt_words = fill_gt_words(idx);
  const size_t lds_bytes = fill_lds_bytes(idx);
  volatile uint64_t* resp = idx->pinned + vs_index::kPinSrvResp;
  const unsigned srv_blocks = std::max(1u, idx->opts.srv_blocks);
  if (idx->d.wpc <= 63)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_server<false>), dim3(srv_blocks), dim3(256), lds_bytes, idx->srv_stream, idx->d,
                       (const ServerRequest*)rq, idx->srv_counter, resp, idx->srv_seq, gt_words, kSrvLifeTicks, kSrvIdleTicks);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_server<true>), dim3(srv_blocks), dim3(256), lds_bytes, idx->srv_stream, idx->d,
                       (const ServerRequest*)rq, idx->srv_counter, resp, idx->srv_seq, gt_words, kSrvLifeTicks, kSrvIdleTicks);
  HIP_TRY(hipGetLastError());
  idx->srv_alive = true;
  idx->srv_started = idx->srv_last = std::chrono::steady_clock::now();
  return VS_OK;
}

// Latency path for type-6 batches of at most 64 regions: ONE launch (k_query_small), regions in the kernel arguments,
// one exact-size pooled slab for the whole result, completion through a mailbox in mapped host memory the caller
// spins on.  Returns 1 if the device found the slab too small (cannot happen unless host and device disagree; the
// caller then takes the general path).
static int run_small_type6(vs_index* idx, const vs_region* regions, uint64_t n, vs_result* r) {
  if (idx->batch_in_flight) {   // the latency path's server runs on a stream of its own and takes its slab from the pool a batch in
    idx->batch_in_flight = false;   // flight has already returned its temporaries to: that batch first
    HIP_TRY(hipStreamSynchronize(idx->stream));
  }
  const auto host_enter = std::chrono::steady_clock::now();
  idx->timing_pending = false; idx->timing_owner = nullptr; idx->timing_total_only = false;
  DevResult& d = r->d;
  uint64_t capA = 0, capS = 0, ntasks = 0;
  for (uint64_t q = 0; q < n; ++q) {
    uint64_t a = 0, c = 0;
    host_region_size(idx, regions[q].x, regions[q].y, &a, &c);
    capA += a; capS += c; ntasks += (a + kFillChunkSmall - 1) / kFillChunkSmall;
  }
  const uint32_t car_width = idx->d.wpc <= 63 ? 2 : 4;
  uint8_t* slab = nullptr;
  {  // one pooled slab for everything (two dozen pool look-ups cost microseconds at this scale)
    DevResult probe{};
    const size_t bytes = small_result_layout(probe, nullptr, n, capA, capS, car_width);
    VS_TRY(ralloc(r, bytes, &slab));
    small_result_layout(d, slab, n, capA, capS, car_width);
  }
  r->n_rows_reported = capA; r->n_unique_sites = capA; r->shared_lists = false;
  const bool lat_debug = idx->opts.lat_debug;
  vs_timing& t = idx->timing;
  const auto host_prep = std::chrono::steady_clock::now();
  // The resident server pays off for a client that asks again the moment it has its answer, and costs everybody else
  // (it holds CUs for up to 1 ms after a request; a caller slower than its idle clock pays stop + restart).  So by
  // default (opts.server == 1) it is started only once kSrvStreak small queries have arrived back to back, and a
  // caller that pauses for longer than kSrvStreakGap goes back to the single launch.
  if (idx->small_streak && host_enter - idx->last_small_done > kSrvStreakGap) idx->small_streak = 0;
  idx->small_streak++;
  const bool use_server = idx->opts.server == 2 || (idx->opts.server == 1 && idx->small_streak > kSrvStreak);
  struct StreakStamp {   // every exit of this function is "the last small query was answered now"
    vs_index* i;
    ~StreakStamp() { i->last_small_done = std::chrono::steady_clock::now(); }
  } streak_stamp{idx};
  if (!use_server && idx->srv_alive) VS_TRY(server_stop(idx));

  // ---- resident server: no launch at all (requests small enough for its 64 waves) ----
  if (use_server && ntasks <= kSrvMaxTasks) {
    VS_TRY(server_ensure(idx));
    ServerRequest* rq = srv_request(idx);
    const uint64_t seq = idx->srv_seq;
    volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(rq);
    if (n > 1) memcpy((void*)rq->xy, regions, n * 16);
    const uint64_t body[6] = {(uint64_t)slab, capA, capS, n | (lat_debug ? 1ull << 16 : 0ull) | ((uint64_t)car_width << 32),
                              regions[0].x, regions[0].y};
    for (int i = 0; i < 6; ++i) w[1 + i] = body[i];
    std::atomic_thread_fence(std::memory_order_release);
    w[7] = server_request_tail(seq, body);        // the tail seals the body (sequence number ^ checksum): a reader that
    std::atomic_thread_fence(std::memory_order_release);   // caught a half-written line sees a tail that does not match
    w[0] = seq;
    const auto posted_at = std::chrono::steady_clock::now();
    volatile uint64_t* resp = idx->pinned + vs_index::kPinSrvResp;
    const auto deadline = posted_at + std::chrono::microseconds(400);
    bool answered = false;
    while (!(answered = ((*resp & 0x3FFFFFFFFFFFFFFFull) == seq))) {
      __builtin_ia32_pause();
      if (std::chrono::steady_clock::now() > deadline) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (answered) {
      idx->srv_seq = seq + 1;
      const auto host_done = std::chrono::steady_clock::now();
      idx->srv_last = host_done;
      if (*resp >> 63) return 1;   // the device wanted more than the host computed: nothing was written
      if (lat_debug)
        fprintf(stderr, "server request: seen -> tasks done %.2f us, release fence %.2f us, -> last block posts %.2f us (wait on host %.2f us)\n",
                (resp[2] - resp[1]) * 0.01, (resp[3] - resp[2]) * 0.01, (double)(int64_t)(resp[4] - resp[3]) * 0.01,
                std::chrono::duration<double, std::micro>(host_done - posted_at).count());
      t.ms_bounds = std::chrono::duration<float, std::milli>(host_prep - host_enter).count();
      t.ms_scan = std::chrono::duration<float, std::milli>(posted_at - host_prep).count();
      t.ms_emit = std::chrono::duration<float, std::milli>(host_done - posted_at).count();
      t.ms_fill = 0.f;
      t.ms_total = std::chrono::duration<float, std::milli>(host_done - host_enter).count();
      t.fill_launches = 0;
      return VS_OK;
    }
    // Not answered in time: the server left between the host's check and the request (its own clock), or some of its
    // blocks did.  Whatever a block wrote is what the launch below writes again; make sure the server is gone first.
    VS_TRY(server_stop(idx));
    idx->srv_seq = seq + 1;   // a sequence number is never reused
  }

  // ---- one launch ----
  d.done_counter = idx->done_counter;
  d.done_flag = idx->pinned + vs_index::kPinFlag;
  d.done_seq = ++idx->lat_seq;
  d.host_totals = lat_debug ? idx->pinned + vs_index::kPinTotals : nullptr;
  const auto host_t0 = std::chrono::steady_clock::now();   // no HIP events here: each one is a packet on the critical path
  {
    const unsigned blocks = (unsigned)std::max<uint64_t>(1, (ntasks + 3) / 4);
    const uint32_t gt_words = fill_gt_words(idx);
    const size_t lds_bytes = fill_lds_bytes(idx);
    const bool wide = idx->d.wpc > 63;
    if (n <= 8) {
      SmallRegions<8> regs;
      memset(&regs, 0, sizeof(regs));
      memcpy(regs.xy, regions, n * 16);
      if (!wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_small<false, 8>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, regs, gt_words, capA, capS);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_small<true, 8>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, regs, gt_words, capA, capS);
    } else {
      SmallRegions<64> regs;
      memset(&regs, 0, sizeof(regs));
      memcpy(regs.xy, regions, n * 16);
      if (!wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_small<false, 64>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, regs, gt_words, capA, capS);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_query_small<true, 64>), dim3(blocks), dim3(256), lds_bytes, idx->stream, idx->d, d, regs, gt_words, capA, capS);
    }
  }
  HIP_TRY(hipGetLastError());
  const auto host_launched = std::chrono::steady_clock::now();
  {  // spin on the mailbox (the runtime's completion wait costs several microseconds more); a kernel that never posts
     // -- a fault -- is caught by the stream synchronisation after the deadline
    volatile uint64_t* flag = idx->pinned + vs_index::kPinFlag;
    const auto deadline = host_t0 + std::chrono::microseconds(300);
    bool posted = false;
    while (!(posted = ((*flag & 0x3FFFFFFFFFFFFFFFull) == d.done_seq))) {
      __builtin_ia32_pause();
      if (std::chrono::steady_clock::now() > deadline) break;
    }
    if (!posted) HIP_TRY(hipStreamSynchronize(idx->stream));
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  d.done_flag = nullptr;
  d.host_totals = nullptr;
  const uint64_t posted_word = *(volatile uint64_t*)(idx->pinned + vs_index::kPinFlag);
  if ((posted_word & 0x3FFFFFFFFFFFFFFFull) != d.done_seq) return fail(VS_ERR_INTERNAL, "the latency kernel finished without posting its completion word");
  if (posted_word >> 63) return 1;  // the device wanted more than the host computed: nothing was written
  // d.A / d.S keep the host's figures: the device arrived at the same ones or it would have said so
  const volatile uint64_t* tot = idx->pinned + vs_index::kPinTotals;
  // latency path: host clock.  ms_bounds = sizing + slab, ms_scan = the launch call (or posting the request), ms_emit =
  // waiting for the mailbox, ms_fill = the kernel's own duration by the device clock (VS_LAT_DEBUG only)
  const auto host_done = std::chrono::steady_clock::now();
  t.ms_bounds = std::chrono::duration<float, std::milli>(host_prep - host_enter).count();
  t.ms_scan = std::chrono::duration<float, std::milli>(host_launched - host_t0).count();
  t.ms_emit = std::chrono::duration<float, std::milli>(host_done - host_launched).count();
  t.ms_fill = lat_debug ? (float)tot[0] * 1e-5f : 0.f;
  if (lat_debug) fprintf(stderr, "latency kernel: %.2f us total, bounds %.2f us, tasks %.2f us\n", tot[0] * 0.01, tot[1] * 0.01, tot[2] * 0.01);
  t.ms_total = std::chrono::duration<float, std::milli>(host_done - host_enter).count();   // call -> completion
  t.fill_launches = 1;
  return VS_OK;
}

// A failed call: kernels of the attempt may still be queued, so the stream is drained before the result's buffers
// go back to the pool.
static int drop_result(vs_result* r, int rc) {
  if (r->idx && r->idx->plan_stream) (void)hipStreamSynchronize(r->idx->plan_stream);
  if (r->idx && r->idx->stream) (void)hipStreamSynchronize(r->idx->stream);
  vs_result_free(r);
  return rc;
}

extern "C" {

const char* vs_strerror(int code) {
  switch (code) {
    case VS_OK: return "ok";
    case VS_ERR_IO: return "i/o error";
    case VS_ERR_FORMAT: return "malformed index";
    case VS_ERR_NO_DEVICE: return "no GPU device (the query path has no CPU fallback)";
    case VS_ERR_HIP: return "HIP runtime error";
    case VS_ERR_ARG: return "bad argument";
    case VS_ERR_UNKNOWN_SAMPLE: return "unknown sample";
    case VS_ERR_UNSUPPORTED: return "unsupported";
    case VS_ERR_INTERNAL: return "internal error";
    default: return "unknown error code";
  }
}
const char* vs_last_error(void) { return g_last_error.c_str(); }

void vs_index_close(vs_index* idx) {
  if (!idx) return;
  if (idx->live_results > 0 || idx->live_comms > 0) {  // results / communicators still use this handle: the last vs_result_free / vs_comm_destroy closes it
    idx->close_pending = true;
    return;
  }
  if (idx->device >= 0) {
    (void)hipSetDevice(idx->device);
    (void)server_stop(idx);
    if (idx->srv_stream) (void)hipStreamDestroy(idx->srv_stream);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    for (auto p : idx->image_allocs) (void)hipFree(p);
    for (auto& b : idx->pool) (void)hipFree(b.p);
    for (auto& b : idx->pin_pool) (void)hipHostFree(b.p);
    if (idx->res_mirror) (void)hipHostFree(idx->res_mirror);
    if (idx->copy_stream) (void)hipStreamDestroy(idx->copy_stream);
    if (idx->fill_stream) (void)hipStreamDestroy(idx->fill_stream);
    if (idx->plan_stream) (void)hipStreamDestroy(idx->plan_stream);
    for (auto& e : idx->fill_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : idx->ev_pool) if (e) (void)hipEventDestroy(e);
    for (auto& e : idx->ev) if (e) (void)hipEventDestroy(e);
    if (idx->pinned) (void)hipHostFree(idx->pinned);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
  }
  delete idx;
}

int vs_index_from_vcf(const char* fasta, const char* vcf, int device, vs_construct_stats* stats, vs_index** out) {
  if (!fasta || !vcf || !out) return fail(VS_ERR_ARG, "null argument");
  vs_index* idx = new vs_index();
  try {
    uint64_t nk = 0, ne = 0, sl = 0;
    ConstructStats st = construct_from_files(fasta, vcf, idx->g, &nk, &ne, &sl);
    idx->cstats = vs_construct_stats{st.num_vars, st.num_mutations, st.num_mutations_samples, nk, ne, sl,
                                     idx->g.num_classes, (uint32_t)st.use_bit_vector};
  } catch (const std::exception& e) {
    delete idx;
    return fail(VS_ERR_IO, "%s", e.what());
  }
  if (stats) *stats = idx->cstats;
  int rc = finish_open(idx, device);
  if (rc != VS_OK) { vs_index_close(idx); return rc; }
  *out = idx;
  return VS_OK;
}

int vs_index_synthetic(const vs_synth_params* p, int device, vs_construct_stats* stats, vs_index** out) {
  if (!p || !out) return fail(VS_ERR_ARG, "null argument");
  vs_index* idx = new vs_index();
  try {
    SynthParams sp;
    sp.ref_length = p->ref_length; sp.num_variants = p->num_variants; sp.num_samples = p->num_samples;
    sp.seed = p->seed; sp.first_pos = p->first_pos; sp.frac_ins = p->frac_ins; sp.frac_del = p->frac_del;
    sp.frac_multi = p->frac_multi; sp.max_indel = p->max_indel ? p->max_indel : 1; sp.af_exponent = p->af_exponent;
    sp.sample_coordinates = p->sample_coordinates != 0;
    sp.max_af = (p->max_af > 0 && p->max_af < 0.5) ? p->max_af : 0.5;
    uint64_t nk = 0, ne = 0, sl = 0;
    SynthStats st = construct_synthetic(sp, idx->g, &nk, &ne, &sl);
    idx->cstats = vs_construct_stats{st.num_vars, st.num_mutations, st.num_mutations_samples, nk, ne, sl,
                                     idx->g.num_classes, (uint32_t)st.use_bit_vector};
  } catch (const std::exception& e) {
    delete idx;
    return fail(VS_ERR_INTERNAL, "%s", e.what());
  }
  if (stats) *stats = idx->cstats;
  int rc = finish_open(idx, device);
  if (rc != VS_OK) { vs_index_close(idx); return rc; }
  *out = idx;
  return VS_OK;
}

int vs_index_open(const char* prefix, int device, vs_index** out) {
  if (!prefix || !out) return fail(VS_ERR_ARG, "null argument");
  vs_index* idx = new vs_index();
  try {
    load_index_dir(prefix, idx->g);
  } catch (const std::exception& e) {
    delete idx;
    return fail(VS_ERR_IO, "%s", e.what());
  }
  int rc = finish_open(idx, device);
  if (rc != VS_OK) { vs_index_close(idx); return rc; }
  *out = idx;
  return VS_OK;
}

int vs_index_save(const vs_index* idx, const char* prefix) {
  if (!idx || !prefix) return fail(VS_ERR_ARG, "null argument");
  try {
    save_index_dir(idx->g, prefix);
  } catch (const std::exception& e) {
    return fail(VS_ERR_IO, "%s", e.what());
  }
  return VS_OK;
}

int vs_index_get_info(const vs_index* idx, vs_index_info* info) {
  if (!idx || !info) return fail(VS_ERR_ARG, "null argument");
  info->ref_length = idx->g.ref_length;
  info->num_vertices = idx->im.V;
  info->num_edges_csr = idx->im.E;
  info->ref_path_nodes = idx->im.P;
  info->index_nodes = idx->im.R;
  info->num_classes = idx->im.C;
  info->num_sites = idx->im.rp_cand_prefix.empty() ? 0 : idx->im.rp_cand_prefix[idx->im.P];
  info->num_carriers = idx->g.car_flags.size();
  info->seq_length = idx->g.seq.size();
  info->num_samples = idx->g.num_samples;
  info->use_bit_vector = idx->g.use_bit_vector;
  info->device_bytes = idx->device_bytes;
  info->device = idx->device;
  info->num_topology_keys = 0;
  for (uint32_t v : idx->g.topo_val) info->num_topology_keys += v != 0;
  info->list_max = idx->im.use_bit_vector ? idx->im.list_max : 0;
  info->reserved_ = 0;
  info->t4_rows_bytes = idx->t4_rows_bytes;
  info->pool_mallocs = idx->pool_mallocs;
  info->pool_frees = idx->pool_frees;
  info->t6_speculated = idx->t6_speculated;
  info->t6_refused = idx->t6_refused;
  return VS_OK;
}

int vs_index_sample_id(const vs_index* idx, const char* name, uint32_t* id) {
  if (!idx || !name || !id) return fail(VS_ERR_ARG, "null argument");
  auto it = idx->sample_ids.find(name);
  if (it == idx->sample_ids.end()) return fail(VS_ERR_UNKNOWN_SAMPLE, "Sample not found: %s", name);
  *id = it->second;
  return VS_OK;
}
const char* vs_index_sample_name(const vs_index* idx, uint32_t id) {
  if (!idx || id >= idx->g.sample_names.size()) return nullptr;
  return idx->g.sample_names[id].c_str();
}
const char* vs_index_chr(const vs_index* idx) { return idx ? idx->g.chr.c_str() : nullptr; }

int vs_index_export_plain(const vs_index* idx, const char* path) {
  if (!idx || !path) return fail(VS_ERR_ARG, "null argument");
  try {
    idx->g.write_plain(path);
  } catch (const std::exception& e) {
    return fail(VS_ERR_IO, "%s", e.what());
  }
  return VS_OK;
}

int64_t vs_index_out_neighbors(const vs_index* idx, uint32_t v, uint32_t* out, uint64_t cap) {
  if (!idx || v >= idx->im.V) return -1;
  uint32_t b = idx->im.row_ptr[v], e = idx->im.row_ptr[v + 1];
  for (uint32_t i = b; i < e && (uint64_t)(i - b) < cap; ++i) out[i - b] = idx->im.col[i];
  return (int64_t)(e - b);
}

int vs_index_set_option(vs_index* idx, const char* key, int64_t value) {
  if (!idx || !key) return fail(VS_ERR_ARG, "null argument");
  EngineOpts& o = idx->opts;
  const std::string k(key);
  if (k == "latency_server") {
    if (value < 0 || value > 2) return fail(VS_ERR_ARG, "latency_server takes 0 (never), 1 (back-to-back streaks) or 2 (always)");
    if (idx->device >= 0 && idx->srv_alive) { HIP_TRY(hipSetDevice(idx->device)); VS_TRY(server_stop(idx)); }
    o.server = (int)value; idx->small_streak = 0;
  } else if (k == "server_blocks") {
    if (value < 1 || value > 64) return fail(VS_ERR_ARG, "server_blocks takes 1..64");
    if (idx->device >= 0 && idx->srv_alive) { HIP_TRY(hipSetDevice(idx->device)); VS_TRY(server_stop(idx)); }
    o.srv_blocks = (unsigned)value;
  } else if (k == "share_lists") {
    if (value < 0 || value > 1) return fail(VS_ERR_ARG, "share_lists takes 0 (private rows and lists per region) or 1 (shared, default)");
    o.share_lists = value != 0;
  } else if (k == "resident_lists") {
    if (value != 0 && value != 1) return fail(VS_ERR_ARG, "resident_lists takes 0 or 1");
    if (value) VS_TRY(build_resident_lists(idx));   // (kept once built: switching back to 0 only stops results from using it)
    o.resident_lists = value != 0;
  } else if (k == "async_fill") o.async_fill = value != 0;
  else if (k == "async_submit") o.async_submit = value != 0;
  else if (k == "t4_walk") {
    if (value < 0 || value > 2) return fail(VS_ERR_ARG, "t4_walk takes 0 (literal), 1 (one lane per region, jumping) or 2 (cooperative, default)");
    o.t4_walk = (int)value;
  } else if (k == "t4_rows_max_mb") {
    // the per-sample rows of query type 4 on an open handle: dropped when they take more than `value` MiB (0: always), built
    // when they are absent and fit it (and half of the free memory); vs_index_get_info().t4_rows_bytes says what is there.
    // An explicit request on the open handle overrides VS_T4_NO_EVENTS of the environment (that switch only decides what open builds).
    if (value < 0) return fail(VS_ERR_ARG, "t4_rows_max_mb takes a size in MiB (0: drop the rows)");
    if (idx->device < 0) return fail(VS_ERR_ARG, "the handle has no device image");
    HIP_TRY(hipSetDevice(idx->device));
    const uint64_t cap = (uint64_t)value << 20;
    if (idx->d.t4_events && idx->t4_rows_bytes > cap) VS_TRY(drop_t4_rows(idx));
    else if (!idx->d.t4_events && cap) VS_TRY(build_t4_rows(idx, cap));
  } else if (k == "phase_events") o.phase_events = value != 0;
  else if (k == "force_fallbacks") o.force_fallbacks = value != 0;
  else if (k == "t6_speculate") o.t6_speculate = value != 0;
  else if (k == "lat_debug" || k == "sc_group" || k == "fill_fused" || k == "fill_chunk" || k == "fill_mode" || k == "fill_dense_k" || k == "fill_stats" || k == "walk_stats" || k == "fill_ablate" ||
           k == "fill_lds_pad") {
#ifdef VS_TUNING
    if (value < 0) return fail(VS_ERR_ARG, "%s takes a non-negative value", key);
    if (k == "lat_debug") o.lat_debug = value != 0;
    else if (k == "fill_fused") o.fill_fused = value != 0;
    else if (k == "sc_group") o.sc_group = value > 1 ? 8 : 1;
    else if (k == "fill_chunk") {
      if (value != 0 && value != 8 && value != 16 && value != 32 && value != 64) return fail(VS_ERR_ARG, "fill_chunk takes 0 (by the batch's shape), 8, 16, 32 or 64");
      o.fill_chunk = (uint32_t)value;
    }
    else if (k == "fill_mode") { if (value != 0 && value != 2) return fail(VS_ERR_ARG, "fill_mode takes 0 or 2"); o.fill_mode = (int)value; }
    else if (k == "fill_dense_k") { if (value != 8 && value != 16 && value != 32 && value != 64) return fail(VS_ERR_ARG, "fill_dense_k takes 8, 16, 32 or 64"); o.fill_dense_k = (uint32_t)value; }
    else if (k == "fill_stats") o.fill_stats = value != 0;
    else if (k == "walk_stats") o.walk_stats = value != 0;
    else if (k == "fill_ablate") o.fill_ablate = (uint32_t)value & 7u;
    else o.fill_lds_pad = (size_t)value;
#else
    return fail(VS_ERR_UNSUPPORTED, "%s exists in tuning builds only (VS_BUILD_TUNING=1 python -m variantstore_amd.build --force)", key);
#endif
  } else return fail(VS_ERR_ARG, "unknown option %s", key);
  return VS_OK;
}

int vs_index_last_timing(const vs_index* cidx, vs_timing* t) {
  vs_index* idx = const_cast<vs_index*>(cidx);
  if (!idx || !t) return fail(VS_ERR_ARG, "null argument");
  if (idx->timing_pending) {   // (the last batch returned when it was enqueued: its events are read -- waited for -- here)
    HIP_TRY(hipSetDevice(idx->device));
    VS_TRY(collect_timing(idx));
  }
  *t = idx->timing;
  return VS_OK;
}

// --------------------------------------------------------------------- queries
void vs_result_free(vs_result* r) {
  if (!r) return;
  if (r->idx) {
    (void)hipSetDevice(r->idx->device);
    (void)result_ready(r);
    if (r->sizes_pending) {   // a speculative batch nobody asked anything of: its totals still feed the handle's hints, its mailbox goes back
      (void)capture_totals(r);
      if (r->idx->plan_slot_owner[r->plan_slot] == r) r->idx->plan_slot_owner[r->plan_slot] = nullptr;
      r->sizes_pending = false;
    }
    if (r->idx->timing_owner == r) (void)collect_timing(r->idx);   // (its events go back to the pool below)
    for (auto& e : r->ev_fill) if (e) { r->idx->ev_pool.push_back(e); e = nullptr; }
    if (r->ev_done) { r->idx->ev_pool.push_back(r->ev_done); r->ev_done = nullptr; }
    release_bufs(r->idx, r->bufs);
    pin_release(r->idx, r->raw_pin);
    for (auto& b : r->old_pins) pin_release(r->idx, b);
    r->idx->live_results--;
    if (r->idx->close_pending && r->idx->live_results == 0 && r->idx->live_comms == 0) vs_index_close(r->idx);
  }
  delete r;
}

// ids in host memory are checked before anything is launched (ids in device memory: k_check_sample_ids, inside the batch)
static int check_host_ids(vs_index* idx, const uint32_t* sample_ids, uint64_t n) {
  if (!n || is_device_ptr(sample_ids)) return VS_OK;
  uint32_t top = 0;
  for (uint64_t i = 0; i < n; ++i) top = sample_ids[i] > top ? sample_ids[i] : top;
  if (top < idx->g.num_samples) return VS_OK;
  for (uint64_t i = 0; i < n; ++i)
    if (sample_ids[i] >= idx->g.num_samples) return fail(VS_ERR_UNKNOWN_SAMPLE, "sample id %u out of range (%u samples)", sample_ids[i], idx->g.num_samples);
  return VS_OK;
}

int vs_query_var_in_ref(vs_index* idx, const vs_region* regions, uint64_t n, vs_result** out) {
  if (!idx || !out || (n && !regions)) return fail(VS_ERR_ARG, "null argument");
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device; queries run on the GPU only");
  HIP_TRY(hipSetDevice(idx->device));
  vs_result* r = new vs_result();
  r->idx = idx;
  idx->live_results++;
  int rc = 1;
  if (n > 0 && n <= 64) {
    rc = run_small_type6(idx, regions, n, r);
    if (rc == 1) {  // the device asked for more than the host had sized (host and device disagree: should not happen)
      release_bufs(idx, r->bufs);
      r->d = DevResult{};
    }
  }
  if (rc == 1) rc = run_type6(idx, regions, n, r, false, nullptr, /*allow_async=*/true);
  if (rc != VS_OK) return drop_result(r, rc);
  *out = r;
  return VS_OK;
}

int vs_query_var_in_ref_device(vs_index* idx, const vs_region* device_regions, uint64_t n, vs_result** out) {
  if (!idx || !out || (n && !device_regions)) return fail(VS_ERR_ARG, "null argument");
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device; queries run on the GPU only");
  HIP_TRY(hipSetDevice(idx->device));
  vs_result* r = new vs_result();
  r->idx = idx;
  idx->live_results++;
  const int rc = run_type6(idx, device_regions, n, r, /*regions_on_device=*/true, nullptr, /*allow_async=*/true);
  if (rc != VS_OK) return drop_result(r, rc);
  *out = r;
  return VS_OK;
}

int vs_query_expand_site_ranges(vs_index* idx, const void* device_records, uint64_t n, vs_result** out) {
  if (!idx || !out || (n && !device_records)) return fail(VS_ERR_ARG, "null argument");
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device; queries run on the GPU only");
  HIP_TRY(hipSetDevice(idx->device));
  vs_result* r = new vs_result();
  r->idx = idx;
  idx->live_results++;
  const int rc = run_type6(idx, nullptr, n, r, false, (const uint64_t*)device_records, /*allow_async=*/false);
  if (rc != VS_OK) return drop_result(r, rc);
  *out = r;
  return VS_OK;
}

int vs_query_sample_var_in_ref(vs_index* idx, const vs_region* regions, uint64_t n, uint32_t sample_id, vs_result** out) {
  if (!idx || !out || (n && !regions)) return fail(VS_ERR_ARG, "null argument");
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device; queries run on the GPU only");
  if (sample_id >= idx->g.num_samples) return fail(VS_ERR_UNKNOWN_SAMPLE, "sample id %u out of range (%u samples)", sample_id, idx->g.num_samples);
  HIP_TRY(hipSetDevice(idx->device));
  vs_result* r = new vs_result();
  r->idx = idx;
  idx->live_results++;
  int rc = run_walk_batch(idx, regions, n, r, sample_id, nullptr, 4);
  if (rc != VS_OK) return drop_result(r, rc);
  *out = r;
  return VS_OK;
}

int vs_query_samples_var_in_ref(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids,
                                vs_result** out) {
  if (!idx || !out || (n && (!regions || !sample_ids))) return fail(VS_ERR_ARG, "null argument");
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device; queries run on the GPU only");
  HIP_TRY(hipSetDevice(idx->device));
  VS_TRY(check_host_ids(idx, sample_ids, n));
  vs_result* r = new vs_result();
  r->idx = idx;
  idx->live_results++;
  int rc = run_walk_batch(idx, regions, n, r, kNone, sample_ids, 4);
  if (rc != VS_OK) return drop_result(r, rc);
  *out = r;
  return VS_OK;
}

static int run_point_batch(vs_index* idx, const uint64_t* positions, uint64_t n, uint32_t mode, const PointStrings* strings,
                           vs_result** out) {
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device; queries run on the GPU only");
  HIP_TRY(hipSetDevice(idx->device));
  std::vector<vs_region> regions(n);
  for (uint64_t i = 0; i < n; ++i) regions[i] = vs_region{positions[i], 0};
  vs_result* r = new vs_result();
  r->idx = idx;
  r->kind = mode == 7 ? 7 : 0;
  idx->live_results++;
  int rc = run_private_batch(idx, regions.data(), n, r, false, nullptr, mode, strings, false);
  if (rc != VS_OK) return drop_result(r, rc);
  *out = r;
  return VS_OK;
}

int vs_query_closest_var(vs_index* idx, const uint64_t* positions, uint64_t n, vs_result** out) {
  if (!idx || !out || (n && !positions)) return fail(VS_ERR_ARG, "null argument");
  return run_point_batch(idx, positions, n, 1, nullptr, out);
}

int vs_query_samples_has_var(vs_index* idx, const uint64_t* positions, const char* const* refs, const char* const* alts,
                             uint64_t n, vs_result** out) {
  if (!idx || !out || (n && (!positions || !refs || !alts))) return fail(VS_ERR_ARG, "null argument");
  std::vector<uint8_t> chars;
  std::vector<uint64_t> off(2 * n + 1, 0);
  for (uint64_t i = 0; i < n; ++i) {
    if (!refs[i] || !alts[i]) return fail(VS_ERR_ARG, "null ref/alt string at query %llu", (unsigned long long)i);
    off[2 * i] = chars.size();
    chars.insert(chars.end(), (const uint8_t*)refs[i], (const uint8_t*)refs[i] + strlen(refs[i]));
    off[2 * i + 1] = chars.size();
    chars.insert(chars.end(), (const uint8_t*)alts[i], (const uint8_t*)alts[i] + strlen(alts[i]));
  }
  off[2 * n] = chars.size();
  PointStrings ps{&chars, &off};
  return run_point_batch(idx, positions, n, 7, &ps, out);
}

static int check_sample_batch(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids, vs_result** out,
                              bool need_index) {
  if (!idx || !out || (n && (!regions || !sample_ids))) return fail(VS_ERR_ARG, "null argument");
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device; queries run on the GPU only");
  HIP_TRY(hipSetDevice(idx->device));
  VS_TRY(check_host_ids(idx, sample_ids, n));
  if (need_index && !idx->d.has_car_index)
    return fail(VS_ERR_ARG, "this index holds no sample coordinates (built without them); query types 2, 3 and 5 need them");
  return VS_OK;
}

int vs_query_sample_seq(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids, int sample_coordinates,
                        vs_result** out) {
  VS_TRY(check_sample_batch(idx, regions, n, sample_ids, out, true));
  HIP_TRY(hipSetDevice(idx->device));
  vs_result* r = new vs_result();
  r->idx = idx;
  r->kind = sample_coordinates ? 3 : 2;
  idx->live_results++;
  int rc = run_sample_seq(idx, regions, n, sample_ids, r->kind, r);
  if (rc != VS_OK) return drop_result(r, rc);
  *out = r;
  return VS_OK;
}

int vs_query_sample_var_in_sample(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids, vs_result** out) {
  VS_TRY(check_sample_batch(idx, regions, n, sample_ids, out, true));
  HIP_TRY(hipSetDevice(idx->device));
  vs_result* r = new vs_result();
  r->idx = idx;
  idx->live_results++;
  static const uint32_t none = 0;
  int rc = run_walk_batch(idx, regions, n, r, kNone, n ? sample_ids : &none, 5);
  if (rc != VS_OK) return drop_result(r, rc);
  *out = r;
  return VS_OK;
}

int vs_result_get_sequences(vs_result* r, uint64_t* n_regions, const uint8_t** region_flags, const uint64_t** seq_begin,
                            const char** chars) {
  if (!r) return fail(VS_ERR_ARG, "null argument");
  if (r->kind != 2 && r->kind != 3) return fail(VS_ERR_ARG, "not a sequence result");
  VS_TRY(fetch_sequences(r));
  if (n_regions) *n_regions = r->sq.Q;
  if (region_flags) *region_flags = r->h_flags.data();
  if (seq_begin) *seq_begin = r->h_byte_begin.data();
  if (chars) *chars = (const char*)r->h_chars.data();
  return VS_OK;
}

int vs_index_draw_subgraph(const vs_index* idx, uint64_t pos, uint64_t radius, const char* sample, const char* outfile) {
  if (!idx || !outfile) return fail(VS_ERR_ARG, "null argument");
  uint32_t sid = 0;
  if (sample && *sample && std::string(sample) != "ref") {
    auto it = idx->sample_ids.find(sample);
    if (it == idx->sample_ids.end()) return fail(VS_ERR_UNKNOWN_SAMPLE, "Sample not found: %s", sample);
    sid = it->second;
  }
  try {
    const std::string text = dot_text(idx->g, draw_start_vertex(idx->g, pos, sid), radius);
    FILE* f = fopen(outfile, "wb");
    if (!f) return fail(VS_ERR_IO, "Can't open the file graph output: %s", outfile);
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
  } catch (const std::exception& e) {
    return fail(VS_ERR_INTERNAL, "%s", e.what());
  }
  return VS_OK;
}

int vs_index_find(vs_index* idx, const uint64_t* pos, uint64_t n, uint32_t* vertex_out) {
  if (!idx || (n && (!pos || !vertex_out))) return fail(VS_ERR_ARG, "null argument");
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device");
  HIP_TRY(hipSetDevice(idx->device));
  if (n == 0) return VS_OK;
  ScratchBufs tmp(idx);
  void *dp = nullptr, *dout = nullptr;
  VS_TRY(dev_alloc(idx, n * 8, &dp, &tmp.bufs));
  VS_TRY(dev_alloc(idx, n * 4, &dout, &tmp.bufs));
  HIP_TRY(hipMemcpyAsync(dp, pos, n * 8, hipMemcpyHostToDevice, idx->stream));
  hipLaunchKernelGGL(k_find, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, idx->d, (const uint64_t*)dp, n, (uint32_t*)dout);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(vertex_out, dout, n * 4, hipMemcpyDeviceToHost, idx->stream));
  HIP_TRY(hipStreamSynchronize(idx->stream));
  tmp.release();
  return VS_OK;
}

// ---------------------------------------------------------------- result access
// Start (stream) the raw copy of a result: rows and -- on request -- the arena go into one page-locked block.
static int raw_copy_begin(vs_result* r, bool with_carriers, hipStream_t stream) {
  VS_TRY(result_sizes(r));
  vs_index* idx = r->idx;
  const DevResult& d = r->d;
  const size_t row_bytes = (size_t)d.A * sizeof(VariantRow), arena_bytes = with_carriers && !r->resident ? (size_t)d.S * d.car_width : 0;
  if (r->raw_rows && (r->raw_arena || !with_carriers)) return VS_OK;
  if (with_carriers) VS_TRY(result_ready(r));
  if (with_carriers && r->resident) VS_TRY(ensure_resident_mirror(idx));   // the rows point into the handle's mirror of the resident arena
  if (r->raw_pin.p) { r->old_pins.push_back(r->raw_pin); r->raw_pin = DevBuf{nullptr, 0}; r->raw_rows = nullptr; r->raw_arena = nullptr; }
  VS_TRY(pin_alloc(idx, row_bytes + arena_bytes + 64, &r->raw_pin));
  uint8_t* base = (uint8_t*)r->raw_pin.p;
  if (row_bytes) HIP_TRY(hipMemcpyAsync(base, d.rows, row_bytes, hipMemcpyDeviceToHost, stream));
  const size_t arena_at = (row_bytes + 63) & ~(size_t)63;
  if (arena_bytes) HIP_TRY(hipMemcpyAsync(base + arena_at, d.carriers, arena_bytes, hipMemcpyDeviceToHost, stream));
  r->raw_rows = (const VariantRow*)base;
  r->raw_arena = !with_carriers ? nullptr : r->resident ? (const uint8_t*)idx->res_mirror : base + arena_at;
  return VS_OK;
}
static void fill_raw(vs_result* r, vs_result_raw* raw) {
  raw->n_regions = r->d.Q;
  raw->region_flags = r->h_flags.data();
  raw->row_begin = r->h_var_begin.data();
  raw->row_count = r->h_nvar.data();
  raw->var_count = r->h_var_count.data();
  raw->car_base = r->h_car_base.data();
  raw->car_len = r->h_car_len.data();
  raw->n_rows = r->d.A;
  raw->rows = (const vs_variant_row*)r->raw_rows;
  raw->arena_entries = r->d.S;
  raw->carrier_bytes = r->d.car_width;
  raw->arena = r->raw_arena;
  raw->seq_pool = r->idx->seq_chars.data();
  raw->shared = (r->shared_lists ? 1 : 0) | (r->resident ? 2 : 0) | (r->scattered_lists ? 4 : 0);
  if (r->scattered_lists) { raw->car_base = nullptr; raw->car_len = nullptr; }   // a region's lists are not one arena range: rows[].car_begin alone addresses them
}

int vs_result_get_raw(vs_result* r, int with_carriers, vs_result_raw* raw) {
  if (r) { VS_NOT_SEQ(r); }
  if (!r || !raw) return fail(VS_ERR_ARG, "null argument");
  static_assert(sizeof(vs_variant_row) == sizeof(VariantRow), "row layout of the ABI");
  vs_index* idx = r->idx;
  HIP_TRY(hipSetDevice(idx->device));
  VS_TRY(raw_copy_begin(r, with_carriers != 0, idx->stream));
  VS_TRY(fetch_region_meta(r));   // (synchronises the stream)
  HIP_TRY(hipStreamSynchronize(idx->stream));
  fill_raw(r, raw);
  return VS_OK;
}

// Type 6 over a large batch with the results leaving the GPU as they are produced: the batch is cut into chunks of
// `chunk_regions`; while chunk k + 1 is computed, the raw copy of chunk k (rows + carrier arena) crosses PCIe on a
// second stream into one of two page-locked buffers, and `fn` is called with it.  The regions must be sorted for the
// chunks to share rows and lists (each chunk is a batch of its own).
int vs_query_var_in_ref_stream(vs_index* idx, const vs_region* regions, uint64_t n, uint64_t chunk_regions, int with_carriers,
                               vs_chunk_fn fn, void* user) {
  if (!idx || (n && !regions) || !fn) return fail(VS_ERR_ARG, "null argument");
  if (idx->device < 0) return fail(VS_ERR_NO_DEVICE, "index handle was opened without a device; queries run on the GPU only");
  if (chunk_regions == 0) return fail(VS_ERR_ARG, "chunk_regions must be positive");
  HIP_TRY(hipSetDevice(idx->device));
  if (!idx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&idx->copy_stream, hipStreamNonBlocking));
  vs_result* in_flight = nullptr;   // chunk whose copy is under way
  uint64_t in_flight_first = 0;
  int rc = VS_OK;
  auto deliver = [&]() -> int {
    if (!in_flight) return VS_OK;
    int rcd = VS_OK;
    if (hipStreamSynchronize(idx->copy_stream) != hipSuccess) rcd = fail(VS_ERR_HIP, "copy of a result chunk failed");
    if (rcd == VS_OK) {
      vs_result_raw raw;
      fill_raw(in_flight, &raw);
      if (fn(user, in_flight_first, &raw) != 0) rcd = fail(VS_ERR_ARG, "the chunk callback asked to stop");
    }
    vs_result_free(in_flight);
    in_flight = nullptr;
    return rcd;
  };
  for (uint64_t first = 0; first < n && rc == VS_OK; first += chunk_regions) {
    const uint64_t cn = std::min<uint64_t>(chunk_regions, n - first);
    vs_result* r = new vs_result();
    r->idx = idx;
    idx->live_results++;
    rc = run_type6(idx, regions + first, cn, r, false, nullptr, /*allow_async=*/false);   // (returns with the chunk's kernels complete; the previous chunk's copy ran beside them)
    if (rc == VS_OK) rc = fetch_region_meta(r);
    if (rc != VS_OK) { drop_result(r, rc); break; }
    const int rcd = deliver();                           // the previous chunk
    if (rcd != VS_OK) { vs_result_free(r); rc = rcd; break; }
    rc = raw_copy_begin(r, with_carriers != 0, idx->copy_stream);
    if (rc != VS_OK) { vs_result_free(r); break; }
    in_flight = r;
    in_flight_first = first;
  }
  const int rcd = deliver();
  return rc != VS_OK ? rc : rcd;
}

int vs_result_get_view(vs_result* r, int with_carriers, vs_result_view* view) {
  if (r) { VS_NOT_SEQ(r); }
  if (!r || !view) return fail(VS_ERR_ARG, "null argument");
  VS_TRY(fetch_headers(r));
  if (with_carriers && !r->have_carriers) {
    // the arena pads every variant's range; the view packs the lists back to back
    r->h_carriers.resize(r->n_view_carriers);
    const uint64_t ns = r->h_view_begin[r->d.Q];
    if (r->resident) {   // from the handle's mirror of the resident arena
      HIP_TRY(hipSetDevice(r->idx->device));
      VS_TRY(ensure_resident_mirror(r->idx));
      const uint16_t* m16 = (const uint16_t*)r->idx->res_mirror;
      const uint32_t* m32 = (const uint32_t*)r->idx->res_mirror;
      for (uint64_t a = 0; a < ns; ++a) {
        uint32_t* dst = r->h_carriers.data() + r->h_car_begin_view[a];
        const uint64_t src = r->h_car_begin[a];
        if (r->d.car_width == 4) { if (r->h_car_count[a]) memcpy(dst, m32 + src, (size_t)r->h_car_count[a] * 4); }
        else for (uint32_t k = 0; k < r->h_car_count[a]; ++k) { const uint32_t c = m16[src + k]; dst[k] = (c & 0x1FFFu) | ((c >> 13) << 29); }
      }
    } else {
    std::vector<uint32_t> arena;
    VS_TRY(fetch_carriers(r, 0, r->d.S, arena));
    for (uint64_t a = 0; a < ns; ++a)
      if (r->h_car_count[a])
        memcpy(r->h_carriers.data() + r->h_car_begin_view[a], arena.data() + r->h_car_begin[a], (size_t)r->h_car_count[a] * 4);
    }
    r->have_carriers = true;
  }
  view->n_regions = r->d.Q;
  view->region_flags = r->h_flags.data();
  view->var_begin = r->h_view_begin.data();
  view->var_count = r->h_var_count.data();
  view->n_slots = r->h_view_begin[r->d.Q];
  view->pos = r->h_pos.data();
  view->ref_off = r->h_ref_off.data(); view->ref_len = r->h_ref_len.data();
  view->alt_off = r->h_alt_off.data(); view->alt_len = r->h_alt_len.data();
  view->var_flags = r->h_vflags.data();
  view->car_begin = r->h_car_begin_view.data();
  view->car_count = r->h_car_count.data();
  view->n_carriers = r->n_view_carriers;
  view->carriers = (with_carriers || r->have_carriers) ? r->h_carriers.data() : nullptr;
  view->seq_pool = r->idx->seq_chars.data();
  return VS_OK;
}

int vs_result_totals(const vs_result* cr, uint64_t* n_regions, uint64_t* n_variants, uint64_t* n_carriers, uint64_t* n_bases) {
  vs_result* r = const_cast<vs_result*>(cr);
  if (!r) return fail(VS_ERR_ARG, "null argument");
  if (r->kind == 2 || r->kind == 3) {  // sequences: only regions and bases
    if (n_regions) *n_regions = r->sq.Q;
    if (n_variants) *n_variants = 0;
    if (n_carriers) *n_carriers = 0;
    if (n_bases) *n_bases = r->seq_bytes;
    return VS_OK;
  }
  if (!r->have_totals) {   // reduced on the device: nothing but three words crosses PCIe
    vs_index* idx = r->idx;
    HIP_TRY(hipSetDevice(idx->device));
    VS_TRY(result_sizes(r));
    ScratchBufs tmp(idx);
    void* dt = nullptr;
    VS_TRY(dev_alloc(idx, 24, &dt, &tmp.bufs));
    HIP_TRY(hipMemsetAsync(dt, 0, 24, idx->stream));
    if (r->d.Q) {
      hipLaunchKernelGGL(k_result_totals, dim3((unsigned)((r->d.Q + 3) / 4)), dim3(256), 0, idx->stream, r->d, (unsigned long long*)dt);
      HIP_TRY(hipGetLastError());
    }
    uint64_t h[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h, dt, 24, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    tmp.release();
    r->n_variants = h[0]; r->n_carriers_kept = h[1]; r->n_bases = h[2]; r->have_totals = true;
  }
  if (n_regions) *n_regions = r->d.Q;
  if (n_variants) *n_variants = r->n_variants;
  if (n_carriers) *n_carriers = r->n_carriers_kept;
  if (n_bases) *n_bases = r->n_bases;
  return VS_OK;
}

int vs_result_format_region(vs_result* r, uint64_t q, const char** text, uint64_t* len) {
  if (!r || !text) return fail(VS_ERR_ARG, "null argument");
  VS_TRY(result_sizes(r));
  if (r->kind == 2 || r->kind == 3) {  // `out << seq << std::endl`, query.h:182-187 / :252-257
    VS_TRY(fetch_sequences(r));
    if (q >= r->sq.Q) return fail(VS_ERR_ARG, "region %llu out of range", (unsigned long long)q);
    r->text.clear();
    if (!(r->h_flags[q] & (VS_REGION_INVALID | VS_REGION_ENDLESS))) {
      r->text.assign((const char*)r->h_chars.data() + r->h_byte_begin[q], r->h_byte_begin[q + 1] - r->h_byte_begin[q]);
      r->text += '\n';
    }
    *text = r->text.c_str();
    if (len) *len = r->text.size();
    return VS_OK;
  }
  VS_TRY(fetch_region_meta(r));
  if (q >= r->d.Q) return fail(VS_ERR_ARG, "region %llu out of range", (unsigned long long)q);
  vs_index* idx = r->idx;
  const uint64_t a0 = r->h_var_begin[q], a1 = a0 + r->h_nvar[q];
  const uint64_t c0 = r->h_car_base[q], c1 = c0 + r->h_car_len[q];
  const uint32_t* car = nullptr;
  const bool from_view = r->have_carriers;
  if (!from_view && r->scattered_lists && !(r->raw_rows && r->raw_arena)) {   // lists all over the arena: bring it over once
    HIP_TRY(hipSetDevice(idx->device));
    VS_TRY(raw_copy_begin(r, true, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
  }
  const bool from_raw = !from_view && r->raw_rows && r->raw_arena;   // a raw copy holds rows and arena: nothing to fetch
  // the rows: the whole table when a view or a raw copy has already brought it over, otherwise this region's slice of it
  const VariantRow* rows;
  if (r->have_headers) rows = r->h_rows.data() + a0;
  else if (r->raw_rows) rows = r->raw_rows + a0;
  else {
    VS_TRY(fetch(idx, r->sl_rows, (const VariantRow*)r->d.rows + a0, (size_t)(a1 - a0)));
    HIP_TRY(hipStreamSynchronize(idx->stream));   // the rows are read below whether or not any carrier is fetched
    rows = r->sl_rows.data();
  }
  const uint64_t vbase = r->have_headers ? r->h_view_begin[q] : 0;   // slot of the region's first row in the view
  if (!from_view && !from_raw) {
    VS_TRY(fetch_carriers(r, c0, c1 - c0, r->slice_carriers));
    car = r->slice_carriers.data();
  }
  const uint16_t* raw16 = reinterpret_cast<const uint16_t*>(r->raw_arena);
  const uint32_t* raw32 = reinterpret_cast<const uint32_t*>(r->raw_arena);
  const bool narrow = r->d.car_width == 2;
  // carrier k of row v as id | gt << 29, from whichever host copy there is
  auto carrier = [&](const VariantRow& v, uint64_t slot, uint32_t k) -> uint32_t {
    if (from_view) return r->h_carriers[r->h_car_begin_view[slot] + k];
    if (from_raw) {
      if (!narrow) return raw32[v.car_begin + k];
      const uint32_t c = raw16[v.car_begin + k];
      return (c & 0x1FFFu) | ((c >> 13) << 29);
    }
    return car[(v.car_begin - c0) + k];
  };
  std::string& out = r->text;
  out.clear();
  if (r->kind == 7) {  // samples_has_var's output line, query.h:811-816: `name gt` pairs with nothing between them
    for (uint64_t a = a0; a < a1; ++a) {
      const VariantRow& v = rows[a - a0];
      if (v.count_flags & kRowDropped) continue;
      for (uint32_t k = 0; k < v.count_flags; ++k) {
        const uint32_t cw = carrier(v, vbase + (a - a0), k);
        const uint32_t id = VS_CARRIER_ID(cw), gt = VS_CARRIER_GT(cw);
        out += id < idx->g.sample_names.size() ? idx->g.sample_names[id] : std::string("?");
        out += ' ';
        out += (gt & GT_1) ? '1' : '0';
        out += (gt & GT_PHASE) ? '|' : '/';
        out += (gt & GT_2) ? '1' : '0';
      }
      out += '\n';
    }
    *text = out.c_str();
    if (len) *len = out.size();
    return VS_OK;
  }
  if (r->h_flags[q] & VS_REGION_NOT_FOUND) {  // closest_var returned false: the reference writes no file
    *text = out.c_str();
    if (len) *len = 0;
    return VS_OK;
  }
  out += "Pos\tRef\tAlt\tSamples\n";  // print_header, query.h:38-41
  for (uint64_t a = a0; a < a1; ++a) {
    const VariantRow& v = rows[a - a0];
    if (v.count_flags & kRowDropped) continue;
    out += std::to_string(v.pos);  // print_var, query.h:43-50
    out += '\t';
    out.append(idx->seq_chars, v.ref_off, v.ref_len);
    out += '\t';
    out.append(idx->seq_chars, v.alt_off, v.alt_len);
    out += '\t';
    for (uint32_t k = 0; k < v.count_flags; ++k) {
      const uint32_t cw = carrier(v, vbase + (a - a0), k);
      const uint32_t id = VS_CARRIER_ID(cw), gt = VS_CARRIER_GT(cw);
      out += id < idx->g.sample_names.size() ? idx->g.sample_names[id] : std::string("?");
      out += '(';
      out += (gt & GT_1) ? '1' : '0';   // get_sample_phasing, variant_graph.h:882-900
      out += (gt & GT_PHASE) ? '|' : '/';
      out += (gt & GT_2) ? '1' : '0';
      out += ") ";
    }
    out += '\n';
  }
  *text = out.c_str();
  if (len) *len = out.size();
  return VS_OK;
}

/* Duration of the result's carrier expansion by HIP events on the stream it ran on, when it ran asynchronously (option
 * "async_fill"; waits for it); -1 otherwise (vs_index_last_timing().ms_fill has it then). */
int vs_result_fill_ms(vs_result* r, float* ms) {
  if (!r || !ms) return fail(VS_ERR_ARG, "null argument");
  if (r->idx && r->idx->device >= 0) HIP_TRY(hipSetDevice(r->idx->device));
  VS_TRY(result_sizes(r));   // (a refused speculative batch is run now: the time asked for is that of the expansion that produced the result)
  VS_TRY(result_ready(r));
  *ms = r->fill_ms;
  return VS_OK;
}

int vs_result_layout(const vs_result* r, uint64_t* n_slots, uint64_t* table_rows, uint64_t* arena_entries, uint64_t* lists_expanded, int* shared) {
  if (!r) return fail(VS_ERR_ARG, "null argument");
  VS_NOT_SEQ(r);
  if (r->sizes_pending) { if (r->idx && r->idx->device >= 0) HIP_TRY(hipSetDevice(r->idx->device)); VS_TRY(result_sizes(const_cast<vs_result*>(r))); }
  if (n_slots) *n_slots = r->n_rows_reported;
  if (table_rows) *table_rows = r->d.A;
  if (arena_entries) *arena_entries = r->resident ? 0 : r->d.S;   // (a result over resident lists owns no arena)
  if (lists_expanded) *lists_expanded = r->n_unique_sites;
  if (shared) *shared = r->shared_lists ? 1 : 0;
  return VS_OK;
}

int vs_result_digest(vs_result* r, uint64_t* digest) {
  if (!r || !digest) return fail(VS_ERR_ARG, "null argument");
  VS_NOT_SEQ(r);
  vs_index* idx = r->idx;
  HIP_TRY(hipSetDevice(idx->device));
  VS_TRY(result_sizes(r));
  VS_TRY(result_ready(r));
  ScratchBufs tmp(idx);
  void* dd = nullptr;
  VS_TRY(dev_alloc(idx, 8, &dd, &tmp.bufs));
  HIP_TRY(hipMemsetAsync(dd, 0, 8, idx->stream));
  if (r->d.A && r->d.Q) {   // the carrier part of every table row once, then every (region, row) pair
    void* rh = nullptr;
    VS_TRY(dev_alloc(idx, r->d.A * 8, &rh, &tmp.bufs));
    const uint64_t blocks = std::min<uint64_t>((r->d.A + 3) / 4, 8192);
    hipLaunchKernelGGL(k_digest_rows, dim3((unsigned)blocks), dim3(256), 0, idx->stream, r->d, (uint64_t*)rh);
    hipLaunchKernelGGL(k_digest, dim3((unsigned)((r->d.Q + 3) / 4)), dim3(256), 0, idx->stream, idx->d, r->d, (const uint64_t*)rh, (uint64_t*)dd);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(digest, dd, 8, hipMemcpyDeviceToHost, idx->stream));
  HIP_TRY(hipStreamSynchronize(idx->stream));
  tmp.release();
  return VS_OK;
}

int vs_result_pack_headers(vs_result* r, void* device_dst, uint64_t capacity_records, uint64_t region_base,
                           uint64_t* n_records) {
  if (!r) return fail(VS_ERR_ARG, "null argument");
  VS_NOT_SEQ(r);
  if (r->sizes_pending) { HIP_TRY(hipSetDevice(r->idx->device)); VS_TRY(result_sizes(r)); }
  if (n_records) *n_records = r->n_rows_reported;   // rows over all regions (a shared row once per region reporting it)
  if (!device_dst) return VS_OK;  // size query
  if (capacity_records < r->n_rows_reported) return fail(VS_ERR_ARG, "destination holds %llu records, %llu needed",
                                                        (unsigned long long)capacity_records, (unsigned long long)r->n_rows_reported);
  vs_index* idx = r->idx;
  HIP_TRY(hipSetDevice(idx->device));
  if (r->d.Q) {
    ScratchBufs tmp(idx);
    uint64_t* slot_begin = nullptr;
    VS_TRY(dev_alloc(idx, (r->d.Q + 1) * 8, (void**)&slot_begin, &tmp.bufs));
    VS_TRY(exclusive_scan<uint64_t>(idx, r->d.q_nvar, r->d.Q, slot_begin, &tmp.bufs));
    hipLaunchKernelGGL(k_pack_headers, dim3((unsigned)((r->d.Q + 3) / 4)), dim3(256), 0, idx->stream, r->d,
                       (uint64_t*)device_dst, (const uint64_t*)slot_begin, region_base);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(idx->stream));
    tmp.release();
  }
  HIP_TRY(hipStreamSynchronize(idx->stream));
  return VS_OK;
}

int vs_result_pack_regions(vs_result* r, void* device_dst, uint64_t capacity_records, uint64_t region_base,
                           uint64_t* n_records) {
  if (!r) return fail(VS_ERR_ARG, "null argument");
  if (n_records) *n_records = r->d.Q;
  if (!device_dst) return VS_OK;
  if (r->sizes_pending) { HIP_TRY(hipSetDevice(r->idx->device)); VS_TRY(result_sizes(r)); }   // (a refused batch is redone before its records travel)
  if (capacity_records < r->d.Q) return fail(VS_ERR_ARG, "destination holds %llu records, %llu needed",
                                             (unsigned long long)capacity_records, (unsigned long long)r->d.Q);
  vs_index* idx = r->idx;
  HIP_TRY(hipSetDevice(idx->device));
  VS_TRY(launch_pack_regions(r, (uint64_t*)device_dst, region_base));
  HIP_TRY(hipStreamSynchronize(idx->stream));
  return VS_OK;
}

}  // extern "C"

#include "comm.hip.h"
