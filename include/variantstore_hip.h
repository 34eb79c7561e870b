/* variantstore_hip.h -- C ABI of the MI355X variant-query engine.
 *
 * The reference (Kingsford-Group/variantstore) has no plugin / FFI interface;
 * its query path is entered from `query_main` (reference src/commands.cc:114-215)
 * through two C++ calls on two objects built from an index directory:
 *
 *   Index idx(prefix);                         include/index.h:108-117
 *   VariantGraph vg(prefix, mode);             include/variant_graph.h:366-446
 *   get_var_in_ref(&vg,&idx,x,y,print,file)    include/query.h:736-784   (query type 6)
 *   get_sample_var_in_ref(...,sample,...)      include/query.h:618-729   (query type 4)
 *
 * This header is that seam as a C ABI: plain pointers and sizes, int error
 * codes, no exceptions, no C++ or torch types.  One vs_index per device; calls on
 * one handle must be serialised by the caller; different handles may be used
 * from different host threads.  Every entry point that computes runs on the GPU:
 * there is no CPU fallback, and a handle opened without a device refuses queries
 * with VS_ERR_NO_DEVICE.
 */
#ifndef VARIANTSTORE_HIP_H
#define VARIANTSTORE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vs_index vs_index;   /* replaces the (Index, VariantGraph) pair: host copy + HBM image */
typedef struct vs_result vs_result; /* replaces std::vector<Variant> per region (query.h:30-36), batched */

enum {
  VS_OK = 0,
  VS_ERR_IO = -1,          /* index directory / file unreadable or malformed   */
  VS_ERR_FORMAT = -2,      /* on-disk structure violates the layout contract   */
  VS_ERR_NO_DEVICE = -3,   /* no usable GPU / handle opened host-only          */
  VS_ERR_HIP = -4,         /* a HIP runtime call failed                        */
  VS_ERR_ARG = -5,         /* bad argument                                     */
  VS_ERR_UNKNOWN_SAMPLE = -6,
  VS_ERR_UNSUPPORTED = -7,
  VS_ERR_INTERNAL = -8
};
const char* vs_strerror(int code);
/* message of the last failure on the calling thread (empty string if none) */
const char* vs_last_error(void);

/* ---- region ------------------------------------------------------------ */
typedef struct { uint64_t x, y; } vs_region; /* [pos_x, pos_y), 1-based: std::get<0>/<1> of commands.cc:64-93 */

/* ---- construction: `variantstore construct` (commands.cc:33-60) ---------- */
typedef struct {
  uint64_t num_vars, num_mutations, num_mutations_samples;  /* the "Num mutations" log line */
  uint64_t num_vertices, num_edges, seq_length;              /* the "Graph stats" log line  */
  uint64_t num_classes;                                       /* "Number of sample vector classes" */
  uint32_t use_bit_vector;
} vs_construct_stats;

/* Build a graph from FASTA + VCF entirely in memory and open it on `device`
 * (device < 0: host-only handle, for inspection/export; queries are refused). */
int vs_index_from_vcf(const char* fasta, const char* vcf, int device, vs_construct_stats* stats, vs_index** out);

/* Deterministic synthetic cohort (bench / scale tests): a random reference of
 * `ref_length` bases, `num_variants` sites (SNP / insertion / deletion mix, some
 * multi-allelic), `num_samples` samples with a skewed allele-frequency spectrum,
 * fed record by record through the same constructor as a VCF would be. */
typedef struct {
  uint64_t ref_length;
  uint64_t num_variants;
  uint32_t num_samples;
  uint64_t seed;
  uint64_t first_pos;        /* variants are placed in [first_pos, ref_length) */
  double frac_ins, frac_del; /* remaining fraction are SNPs                    */
  double frac_multi;         /* fraction of SNP sites with a second ALT        */
  uint32_t max_indel;        /* indel length 1..max_indel                      */
  double af_exponent;        /* AF = min(0.5, 10^(-af_exponent * U))           */
  uint32_t sample_coordinates; /* != 0: also compute the per-carrier sample-coordinate indexes (the reference's
                                * "Fixing sample indexes" pass, variant_graph.h:1919-1997) that query types 2, 3 and 5
                                * read; 4 bytes per carrier record                                                      */
  double max_af;             /* cap of the allele frequency; 0 = 0.5.  Small caps give somatic-like cohorts that the
                              * constructor stores with explicit sample ids instead of class bit vectors          */
} vs_synth_params;
int vs_index_synthetic(const vs_synth_params* p, int device, vs_construct_stats* stats, vs_index** out);

/* Open an index directory written by `variantstore construct` (Index(prefix) +
 * VariantGraph(prefix, mode): index.h:108-117, variant_graph.h:366-446). */
int vs_index_open(const char* prefix, int device, vs_index** out);
/* Write the index directory (VariantGraph::serialize + Index::serialize). */
int vs_index_save(const vs_index* idx, const char* prefix);
/* Releases the handle.  If results of this handle are still alive the release is deferred to the vs_result_free of
 * the last one (their arrays live in the handle's HBM pool); the handle must not be used for new calls meanwhile. */
void vs_index_close(vs_index* idx);

typedef struct {
  uint64_t ref_length, num_vertices, num_edges_csr, ref_path_nodes, index_nodes;
  uint64_t num_classes, num_sites, num_carriers, seq_length;
  uint32_t num_samples;      /* includes "ref" */
  uint32_t use_bit_vector;
  uint64_t device_bytes;     /* HBM held by the image */
  int device;
  uint64_t num_topology_keys; /* Graph::get_num_vertices(): vertices with an adjacency entry (graph.h:318-320) */
  uint32_t list_max;         /* classes of at most this many carriers are expanded from decoded id lists (16-bit entries up
                              * to 4032 samples, else 32-bit), denser ones from their bit row; 0: explicit-id cohort   */
  uint32_t reserved_;
  uint64_t t4_rows_bytes;    /* of device_bytes: the per-sample event and hold rows of query type 4 (O(samples x ref-path slots):
                              * taken when they fit half of the free HBM and 176 GB -- VS_T4_ROWS_MAX_GB in the environment
                              * lowers the cap, option "t4_rows_max_mb" drops / rebuilds them on the open handle; 0: not built, the walks
                              * then visit every vertex).  Explicit-id cohorts keep coarse event rows (a bit per 8 slots) and no hold rows;
                              * VS_T4_EXACT_ROWS=1 in the environment gives them the exact form when it fits the same budget */
  uint64_t pool_mallocs;     /* hipMalloc / hipFree calls the handle's pool of batch buffers has made since it was opened: a loop */
  uint64_t pool_frees;       /* of like batches makes none once it is warm (hipFree waits for the whole device)                  */
  uint64_t t6_speculated;    /* type-6 batches submitted without waiting for their plan's totals (option "t6_speculate") ...       */
  uint64_t t6_refused;       /* ... and those of them the device refused (did not fit what was allocated, or unsorted) and the host  */
                             /* ran again with the exact sizes when the result was first asked for anything                          */
} vs_index_info;
int vs_index_get_info(const vs_index* idx, vs_index_info* info);
/* sampleid_map / idsample_map lookups (variant_graph.h:1230-1236, 1327-1339) */
int vs_index_sample_id(const vs_index* idx, const char* name, uint32_t* id);
const char* vs_index_sample_name(const vs_index* idx, uint32_t id);
const char* vs_index_chr(const vs_index* idx);
/* Dump the decoded index content in the flat format the test oracle reads. */
int vs_index_export_plain(const vs_index* idx, const char* path);
/* Host-side inspection used by structure tests (no GPU needed):
 * out-neighbours of v in the reference's iteration order; returns the degree. */
int64_t vs_index_out_neighbors(const vs_index* idx, uint32_t v, uint32_t* out, uint64_t cap);

/* ---- queries ------------------------------------------------------------ */
/* type 6: get_var_in_ref for each of the n regions (query.h:736-784).  Batches of at most 64 regions take the latency
 * path: one kernel for the whole query, or -- while the handle's resident query server is alive -- no launch at all
 * (by default only for back-to-back streaks of small queries -- vs_index_set_option
 * "latency_server"; VS_NO_SERVER=1 in the environment when the handle is opened keeps it to one launch per call). */
int vs_query_var_in_ref(vs_index* idx, const vs_region* regions, uint64_t n, vs_result** out);
/* The same with the regions already in DEVICE memory of the handle's GPU (e.g. produced there, or uploaded once and
 * queried repeatedly): no host buffer crosses PCIe inside the call.  Always the batch pipeline, whatever n. */
int vs_query_var_in_ref_device(vs_index* idx, const vs_region* device_regions, uint64_t n, vs_result** out);
/* The receiving side of the hit-list collective (vs_result_pack_regions below): n compact region records in DEVICE
 * memory -- this rank's own or gathered from other ranks holding the same index -- expanded into a full type-6 result
 * (variant rows + carrier lists), exactly what the producing rank holds.  Records whose site range does not fit this
 * index come back as VS_REGION_INVALID. */
int vs_query_expand_site_ranges(vs_index* idx, const void* device_records, uint64_t n, vs_result** out);
/* type 4: get_sample_var_in_ref for one sample over n regions (query.h:618-729) */
int vs_query_sample_var_in_ref(vs_index* idx, const vs_region* regions, uint64_t n, uint32_t sample_id,
                               vs_result** out);
/* type 4 with one sample id per region (a batch mixing samples, e.g. the cohort round-robin of the bench).
 * Here and in vs_query_sample_seq / vs_query_sample_var_in_sample, `regions` and `sample_ids` may each lie in host
 * memory or in DEVICE memory of the handle's GPU (the engine asks the runtime which): device arrays are neither read
 * on the host nor copied over the link, and an id out of range is found by the batch's first kernel instead of the
 * host loop -- the call fails with VS_ERR_UNKNOWN_SAMPLE either way. */
int vs_query_samples_var_in_ref(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids,
                                vs_result** out);
/* Query type 1, closest_var (include/query.h:441-483; called from src/commands.cc:151-155): the variants of the
 * site nearest to each position, found exactly as the reference does (one next_variant_in_ref call forwards, one
 * mirrored call backwards, or the one-position-at-a-time step back when nothing lies ahead).  Each position gives one
 * "region" of the result; VS_REGION_NOT_FOUND marks a call for which the reference returns false (it then writes
 * no output file and vs_result_format_region gives an empty text). */
int vs_query_closest_var(vs_index* idx, const uint64_t* positions, uint64_t n, vs_result** out);
/* Query type 7, samples_has_var (include/query.h:792-823; src/commands.cc:181-189): the carriers of the variant
 * (positions[i], refs[i], alts[i]) among the variants ONE next_variant_in_ref(positions[i]) call reports.
 * refs/alts are NUL-terminated strings compared byte for byte with the index's sequences.  A region of the result
 * holds one variant when found (vs_result_format_region then gives the reference's output line: `name gt` pairs
 * with no separator, then a newline) and carries VS_REGION_NOT_FOUND otherwise ("There is no such variant!"). */
int vs_query_samples_has_var(vs_index* idx, const uint64_t* positions, const char* const* refs, const char* const* alts,
                             uint64_t n, vs_result** out);
/* Query types 2 and 3, query_sample_from_ref / query_sample_from_sample (include/query.h:118-190, :196-261;
 * src/commands.cc:156-165): the sequence of sample sample_ids[i] over regions[i] = [x, y) in reference coordinates
 * (sample_coordinates == 0) or in the sample's own coordinates (!= 0).  The result is a SEQUENCE result: read it with
 * vs_result_get_sequences or vs_result_format_region (sequence + '\n', the reference's output file).  Regions on
 * which the reference dies of an uncaught std::out_of_range carry VS_REGION_INVALID, regions on which its backward
 * search never ends carry VS_REGION_ENDLESS; both give an empty sequence.  Needs an index with sample coordinates
 * (every index built from a VCF or loaded from disk has them). */
int vs_query_sample_seq(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids, int sample_coordinates,
                        vs_result** out);
/* Query type 5, get_sample_var_in_sample (include/query.h:490-612; src/commands.cc:171-175): the variants on the path
 * of sample_ids[i] over regions[i] in the SAMPLE's coordinates; an ordinary variant-table result.  var_pos follows the
 * reference: the sample-coordinate index for substitutions and deletions, the reference index for insertions. */
int vs_query_sample_var_in_sample(vs_index* idx, const vs_region* regions, uint64_t n, const uint32_t* sample_ids, vs_result** out);
/* host view of a sequence result: region i is chars[seq_begin[i] .. seq_begin[i+1]) */
int vs_result_get_sequences(vs_result* r, uint64_t* n_regions, const uint8_t** region_flags, const uint64_t** seq_begin,
                            const char** chars);
/* `variantstore draw` (src/commands.cc:217-242 -> draw_subgraph, include/query.h:825-842 -> createDotGraph,
 * include/dot_graph.h:71-132): the vertices within `radius` hops of the vertex at `pos` (on the path of `sample`;
 * NULL or "ref" = the reference) as a Graphviz file.  Host-only; works on handles opened without a device. */
int vs_index_draw_subgraph(const vs_index* idx, uint64_t pos, uint64_t radius, const char* sample, const char* outfile);
/* batched Index::find (index.h:119-133): vertex id of the ref node covering each position */
int vs_index_find(vs_index* idx, const uint64_t* pos, uint64_t n, uint32_t* vertex_out);

/* Result of one batch.  Arrays live in HBM until a view is requested. */
enum { VS_REGION_EMPTY = 1,   /* Index::is_empty early-out fired (query.h:745-756 prints the other label) */
       VS_REGION_INVALID = 2, /* pos_x < 1: the reference aborts (index.h:151-154) */
       VS_REGION_NOT_FOUND = 4, /* types 1 and 7: closest_var returned false / "There is no such variant!" */
       VS_REGION_ENDLESS = 8   /* types 3 and 5: the reference's backward search (query.h:213-218) never terminates */ };
enum { VS_VAR_DROPPED = 1 };  /* suppressed by the reference's "already seen" rule, query.h:397-414 */
#define VS_CARRIER_ID(c) ((c) & 0x1FFFFFFFu)
#define VS_CARRIER_GT(c) ((c) >> 29)         /* bit0 phase ('|'), bit1 gt_1, bit2 gt_2 */

typedef struct {
  uint64_t n_regions;
  const uint8_t* region_flags;   /* [n_regions] */
  const uint64_t* var_begin;     /* [n_regions+1] slot range of each region */
  const uint64_t* var_count;     /* [n_regions]   variants the reference reports (slots minus dropped) */
  uint64_t n_slots;
  const uint64_t* pos;           /* [n_slots] Variant::var_pos */
  const uint32_t* ref_off;       /* [n_slots] Variant::ref = seq_pool[ref_off, ref_off+ref_len) */
  const uint32_t* ref_len;
  const uint32_t* alt_off;       /* Variant::alt likewise */
  const uint32_t* alt_len;
  const uint32_t* var_flags;     /* VS_VAR_* */
  const uint64_t* car_begin;     /* [n_slots] first carrier of the slot */
  const uint32_t* car_count;     /* [n_slots] Variant::samples.size() */
  uint64_t n_carriers;
  const uint32_t* carriers;      /* sample id | gt << 29, s_info order; NULL when carriers were not fetched.  (In HBM the
                                  * arena holds 16-bit words for cohorts of at most 4032 samples; the copy widens them.) */
  const char* seq_pool;          /* index-owned: one character per base */
} vs_result_view;

/* The RAW host copy: the result exactly as it lies in HBM -- per-region arrays, the variant table (32-byte rows) and, on
 * request, the carrier arena -- copied with two transfers into page-locked memory owned by the result.  No repacking:
 * region q reports rows [row_begin[q], row_begin[q] + row_count[q]) of the table (ranges of different regions overlap
 * when `shared`), a row's carriers are arena[car_begin .. car_begin + VS_ROW_COUNT) in units of carrier_bytes
 * (2: id | gt << 13, VS_CARRIER16_*; 4: id | gt << 29, VS_CARRIER_*).  This is the form to use when results leave the GPU
 * in bulk; vs_result_get_view below expands it per region into the structure-of-arrays of round 1. */
typedef struct {
  uint32_t pos;                 /* Variant::var_pos */
  uint32_t ref_off, ref_len;    /* Variant::ref = seq_pool[ref_off, ref_off + ref_len) */
  uint32_t alt_off, alt_len;    /* Variant::alt likewise */
  uint32_t count_flags;         /* VS_ROW_COUNT carriers | VS_ROW_DROPPED */
  uint64_t car_begin;           /* first carrier of the row in the arena */
} vs_variant_row;
#define VS_ROW_COUNT(row) ((row).count_flags & 0x7FFFFFFFu)
#define VS_ROW_DROPPED(row) ((row).count_flags >> 31)
#define VS_CARRIER16_ID(c) ((uint32_t)(c) & 0x1FFFu)
#define VS_CARRIER16_GT(c) ((uint32_t)(c) >> 13)
typedef struct {
  uint64_t n_regions;
  const uint8_t* region_flags;   /* [n_regions] */
  const uint64_t* row_begin;     /* [n_regions] first table row of each region */
  const uint64_t* row_count;     /* [n_regions] rows it reports (including dropped ones) */
  const uint64_t* var_count;     /* [n_regions] variants the reference reports */
  const uint64_t* car_base;      /* [n_regions] arena offset of the region's first row (NULL: shared bit 2) */
  const uint64_t* car_len;       /* [n_regions] arena extent of the region's rows (NULL: shared bit 2) */
  uint64_t n_rows;
  const vs_variant_row* rows;    /* [n_rows], page-locked */
  uint64_t arena_entries;
  uint32_t carrier_bytes;        /* 2 or 4 */
  const void* arena;             /* [arena_entries], page-locked; NULL when carriers were not copied */
  const char* seq_pool;
  int shared;                    /* bit 2: lists shared per VERTEX (query types 4 / 5): a region's carriers are not one arena
                                  * range -- car_base / car_len are NULL, rows[].car_begin alone addresses the lists;
                                  * bit 0: rows and lists shared between regions (vs_result_layout); bit 1: `arena` is the
                                  * handle's host mirror of the index's RESIDENT carrier lists ("resident_lists"): nothing
                                  * but the rows was copied for this result, and the pointer stays valid until the index
                                  * is closed */
} vs_result_raw;
int vs_result_get_raw(vs_result* r, int with_carriers, vs_result_raw* raw);

/* Duration of the result's carrier expansion by its OWN pair of HIP events on the stream the kernel ran on (waits for the
 * kernel): every type-6 batch that shares rows and lists carries one, as does every "async_fill" batch; -1 for the other
 * result forms (vs_index_last_timing().ms_fill has it then).
 * "async_submit" (default 1): a type-6 batch of more than 64 regions returns when it is ENQUEUED -- sizes known (or, option "t6_speculate", taken from the handle's previous batch and settled when the result is first asked for anything), buffers
 * allocated, last kernel launched.  Everything that reads the result is ordered behind the batch on the handle's stream
 * (copies, digests, packs, the collective) or waits for it (this call, vs_index_last_timing); freeing it early is safe.
 * "async_fill" (default 0): the call returns as soon as rows and per-region arrays are in HBM while the expansion runs
 * on the handle's second stream beside the next batch's plan and rows; accessors that read carriers wait by themselves. */
int vs_result_fill_ms(vs_result* r, float* ms);

/* Type 6 with delivery: the (sorted) batch is answered in chunks of `chunk_regions`, and while one chunk is computed the
 * raw copy of the previous one crosses PCIe on a second stream into page-locked memory; `fn(user, first_region, raw)` is
 * called once per chunk, in order, with a raw view that is valid during the call (return non-zero to stop). */
typedef int (*vs_chunk_fn)(void* user, uint64_t first_region, const vs_result_raw* chunk);
int vs_query_var_in_ref_stream(vs_index* idx, const vs_region* regions, uint64_t n, uint64_t chunk_regions, int with_carriers,
                               vs_chunk_fn fn, void* user);

/* Copy the result to host memory (owned by the vs_result).  with_carriers = 0
 * leaves the carrier lists in HBM (view->carriers == NULL). */
int vs_result_get_view(vs_result* r, int with_carriers, vs_result_view* view);
/* Totals without any copy of the arrays. */
int vs_result_totals(const vs_result* r, uint64_t* n_regions, uint64_t* n_variants, uint64_t* n_carriers,
                     uint64_t* n_bases);
/* How the result lies in HBM: rows the regions report in total, rows of the variant table, arena entries in use (lists
 * padded to groups of 8), carrier lists actually expanded, and whether rows and lists are SHARED: a type-6 batch of
 * overlapping regions holds one row and one carrier list per site it covers, and every region reporting the site
 * refers to them (its rows are a range of the shared table) -- the way REF / ALT are references into the sequence
 * pool.  (A batch that is not sorted by start is sorted on the device for this and answered in the caller's order.)
 * Views, texts, digests and totals are per region and unaffected.  With resident carrier lists ("resident_lists") a
 * result owns no arena: arena_entries and lists_expanded are 0. */
int vs_result_layout(const vs_result* r, uint64_t* n_slots, uint64_t* table_rows, uint64_t* arena_entries, uint64_t* lists_expanded,
                     int* shared);
/* The `-o` file of region q (print_header + print_var, query.h:38-50) as text owned by the result. */
int vs_result_format_region(vs_result* r, uint64_t q, const char** text, uint64_t* len);
/* Order-independent 64-bit digest of (region, pos, ref, alt, carriers) computed
 * on the device -- the "checksum of checksums" used by full-size property tests. */
int vs_result_digest(vs_result* r, uint64_t* digest);
/* Hit-list records for a collective over the shards of a batch (all-gatherv):
 * writes n_slots records of 4 x uint64 into DEVICE memory at device_dst
 *   {pos | dropped<<63, ref_off | ref_len<<32, alt_off | alt_len<<32, (region_base+region) | car_count<<32}.
 * device_dst == NULL only reports the record count. */
int vs_result_pack_headers(vs_result* r, void* device_dst, uint64_t capacity_records, uint64_t region_base,
                           uint64_t* n_records);
/* Compact hit lists (query type 6): because every rank holds the same index, a region's variant list is
 * its range of the position-ordered site table.  n_regions records of 4 x uint64 in DEVICE memory:
 *   {region_base+q, first_site | region_flags<<32 | has_dropped<<40, sites | variants reported<<32, carriers}.
 * (A region with has_dropped set lost entries to the reference's duplicate rule: fewer variants reported than
 * sites; vs_query_expand_site_ranges applies the rule again.)  device_dst == NULL only reports the record count.
 * Results of the other query types pack per-region SUMMARIES in the same 32 bytes -- what the reference's driver prints
 * per region (src/commands.cc:150-193) and what a sharded run gathers: types 4, 5, 1, 7 the same fields (flags incl.
 * VS_REGION_NOT_FOUND of the point queries, variants reported, carriers; the site fields mean nothing there); types 2, 3
 *   {region_base+q, region_flags<<32, pieces, bytes of the sequence}. */
int vs_result_pack_regions(vs_result* r, void* device_dst, uint64_t capacity_records, uint64_t region_base,
                           uint64_t* n_records);
void vs_result_free(vs_result* r);

/* ---- multi-GPU: the hit-list collective (one process per GPU, RCCL over xGMI) ----
 * north_star: "a batch of thousands of independent region queries shards trivially across the 8 GPUs of one node with
 * an RCCL all-gatherv of hit lists over xGMI".  The reference's loop over the regions is serial and single-process
 * (src/commands.cc:145); these entry points are what its query_main would call once per batch after sharding the sorted
 * region list into contiguous pieces.  RCCL is loaded at run time: VS_ERR_UNSUPPORTED when
 * no librccl.so can be found.
 *   vs_comm_unique_id   rank 0 makes the id (VS_COMM_ID_BYTES bytes) and hands it to the other ranks by whatever means
 *                       the host program has (a file, a pipe, MPI, a torch broadcast)
 *   vs_comm_init        every rank, with the handle whose device it computes on (ncclCommInitRank: collective)
 *   vs_comm_allgather_regions   packs the result's per-region records (vs_result_pack_regions' format) and all-gathers
 *                       them, padded to max_count records per rank: device_dst receives world x max_count records, rank k's
 *                       at record k * max_count, of which the first (regions of rank k) are valid.  async_op != 0: returns
 *                       once the collective is enqueued (on the communicator's own stream, behind the result's kernels);
 *                       vs_comm_wait -- or the next vs_comm_allgather_regions -- waits for it.
 *   Any rank can then rebuild rows and carrier lists of the whole batch from the records: vs_query_expand_site_ranges. */
#define VS_COMM_ID_BYTES 128
typedef struct vs_comm vs_comm;
int vs_comm_unique_id(void* id_out);
int vs_comm_init(vs_index* idx, int rank, int world, const void* id, vs_comm** out);
int vs_comm_allgather_regions(vs_comm* c, vs_result* r, uint64_t region_base, uint64_t max_count, void* device_dst, int async_op);
/* the same into HOST memory (world x max_count records; synchronous) for callers that hold no device memory of their own */
int vs_comm_allgather_regions_host(vs_comm* c, vs_result* r, uint64_t region_base, uint64_t max_count, void* host_dst);
int vs_comm_wait(vs_comm* c);
/* what the communicator is: the rank and world size it was made with and the number of ranks RCCL itself reports for it
 * (ncclCommCount) -- a self-check for launchers: the three agree or the ranks did not all join the same communicator */
int vs_comm_info(vs_comm* c, int* rank, int* world, int* rccl_ranks);
void vs_comm_destroy(vs_comm* c);

/* ---- switches of one handle ----
 * The environment is read once when a handle is opened; afterwards only this call changes a
 * switch.  The production library has ELEVEN keys:
 *   "latency_server"  0 never / 1 for back-to-back streaks of small queries (default) / 2 from the first small query
 *   "server_blocks"   1..64 blocks of the resident server
 *   "share_lists"     1 (default): a type-6 batch of more than 64 regions holds one row and one carrier list per covered
 *                     site, shared by the regions that report it (type 4 / 5: one list per reported vertex); 0: private rows
 *                     and lists per region
 *   "resident_lists"  1 = expand every carrier list of the index ONCE into an arena that stays in HBM with the handle (2 or
 *                     4 bytes per carrier record; VS_ERR_UNSUPPORTED when that does not fit): batches of query types 6 and 4
 *                     then emit rows that point into it, expand nothing and own no arena, and a raw copy moves the rows only
 *                     (default 0; VS_RESIDENT_LISTS=1 in the environment builds it when the handle is opened)
 *   "async_submit"    see vs_result_fill_ms (default 1)
 *   "async_fill"      see vs_result_fill_ms (default 0)
 *   "t6_speculate"    1 (default): a type-6 batch that returns when it is enqueued (async_submit) does not wait for its plan's totals
 *                     either when the handle's previous shared batch had about as many regions (4/5 .. 5/4): variant table and arena
 *                     are sized from that batch (+ 1/8), the kernels behind the plan read the totals in device memory, and a batch
 *                     that does not fit (or whose regions are not sorted) is refused on the device and run again with the exact
 *                     sizes the first time its result is asked for anything -- vs_index_info.t6_speculated / t6_refused count
 *                     them.  Same answers; the host never waits between two batches of a stream of like batches.  0: every batch
 *                     waits for its totals (rounds 1-5)
 *   "t4_walk"         the walk of the query types that follow one sample's path: 2 cooperative (8 lanes per region, a region's
 *                     events walked in parallel: types 4, 2 and 3; default), 1 one lane per region jumping over uneventful
 *                     ref-path runs, 0 literal (type 4: every vertex of the sample's path; types 2 / 3 / 5 as 1)
 *   "t4_rows_max_mb"  the per-sample event and hold rows of query type 4 (vs_index_info.t4_rows_bytes: 7.6 GB for 2504
 *                     samples over chr1) on an OPEN handle: rows larger than the value in MiB are dropped (0: always -- the walks
 *                     then visit every vertex, same answers), absent rows are built when they fit it and half of the free
 *                     memory.  Several handles on one GPU: the caller decides which of them keeps its rows.  Waits for the
 *                     device before it frees anything.
 *   "phase_events"    1 = batches of query types 4 and 5 record all five phase events, so that vs_index_last_timing reports their
 *                     phases (walk, sizes, rows, expansion); default 0: first and last event only -- ms_total, phases 0 -- because
 *                     every event is a packet between two kernels of a string of dependent launches (~3 us each)
 *   "force_fallbacks" 1 = query types 2 - 5 take the count-then-emit pair of walks they fall back to when a region outgrows
 *                     the capacity of its recording walk (tests of that path)
 * Tuning builds (VS_BUILD_TUNING=1 python -m variantstore_amd.build --force) add "lat_debug", "fill_fused", "fill_chunk",
 * "fill_stats", "walk_stats", "fill_ablate", "fill_lds_pad"; VS_ERR_UNSUPPORTED in the production library, whose kernels
 * do not carry the code. */
int vs_index_set_option(vs_index* idx, const char* key, int64_t value);

/* ---- timing of the last batch on this handle ----
 * Batches of more than 64 regions and every other query type: HIP events on the engine's stream.
 * Type-6 batches of at most 64 regions (latency path: no events on the critical path): host clock --
 *   ms_total call -> result resident, ms_bounds sizing + result slab, ms_scan posting the request / the launch call,
 *   ms_emit waiting for the completion word, ms_fill 0 (the kernel's duration by the device clock under VS_LAT_DEBUG),
 *   fill_launches 0 when the resident query server answered, 1 when a kernel was launched for the call. */
typedef struct {
  float ms_total;    /* first launch to last kernel completion */
  float ms_bounds;   /* rank / region-bounds kernel            */
  float ms_scan;     /* offset scan (+ the host round trip for the sizes) */
  float ms_emit;     /* variant-header kernel                  */
  float ms_fill;     /* carrier-expansion kernel (dominant)    */
  uint64_t fill_launches;
} vs_timing;
int vs_index_last_timing(const vs_index* idx, vs_timing* t);

#ifdef __cplusplus
}
#endif
#endif
