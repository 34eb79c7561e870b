// kernels.hip.h -- gfx950 device code of the region-query path.
//
// All integer / index work: the roofline that bounds these kernels is HBM
// bandwidth (and memory latency for the gather phases), never MFMA.
// Wavefront width is 64 throughout; one workgroup = 256 threads = 4 waves.
//
// Kernel                replaces (reference file:line)
// --------------------  ------------------------------------------------------
// k_build_sites         next_variant_in_ref's per-node branch classification
//                       (include/query.h:308-393) + the radius-1 neighbour walk
//                       (include/graph.h:394-431), run ONCE over the whole ref
//                       path when an index is opened (the "site table")
// k_mark_dups           the candidates the "already seen" rule can ever hit
//                       (include/query.h:397-414)
// k_region_bounds       Index::is_empty / Index::find (include/index.h:119-166)
//                       + the stop rule ref_index+length >= end (query.h:312)
// k_emit_headers        building std::vector<Variant> (query.h:736-771)
// k_dedup_slow          the literal "already seen" rule for the rare regions
//                       that contain a repeated (pos, alt)
// k_t6_bounds / _mid /  the plan of a SORTED type-6 batch whose regions share rows and carrier lists (the reference's
//  _totals / _apply     driver sorts its regions, src/commands.cc:91): is_empty / find / stop rule per region, the sites each
//                       region is the first to cover, run records, totals for the host (k_rows.hip.h)
// k_t6_slow             private rows + the literal "already seen" rule for the regions under it
// k_fill_sites2         the shared rows AND their carrier lists: get_samples -> get_sample_id / get_sample_phasing
//                       (query.h:268-285, variant_graph.h:875-1006) once per covered site.  THE DOMINANT KERNEL
//                       (DESIGN.md section 6); k_fill_dense: its dense variants alone (tuning builds, for profilers)
// k_fill_carriers       the same expansion over private rows (share_lists = 0, point queries, the walking types' lists):
//                       decoded class id lists / class bit rows + genotype bits into carrier lists
// k_query_small         a whole get_var_in_ref batch of <= 64 regions in one launch
//                       (bounds, offsets, headers, carriers, dedup): the latency path
// k_query_server        the same, resident: polls requests in mapped host memory
// k_point_bounds        one next_variant_in_ref call from a position (query.h:297-436)
//                       as used by closest_var (query.h:441-483, type 1) and
//                       samples_has_var (query.h:792-823, type 7)
// k_has_var_filter      the (pos, ref, alt) match of samples_has_var (query.h:802-803)
// k_sample_walk_sc      get_sample_var_in_sample (query.h:490-612, type 5), one lane per region;
// k_sample_walk_sc_coop  eight lanes per region, the sample's events walked as episodes in parallel
// k_sample_seq          query_sample_from_ref / query_sample_from_sample
//                       (query.h:118-261, types 2 and 3): the walk emits (offset, length)
//                       pieces of the sequence pool; k_copy_segments decodes them;
// k_sample_seq_coop      the cooperative form (window logic of query.h:160-177 / :236-247 at the hand-over)
// k_sample_walk_coop    get_prev_vertex_with_sample + get_sample_var_in_ref (query.h:618-729, type 4), eight lanes per region;
//  (k_sample_walk)       WalkAdmit holds a batch to the scratch its predecessor needed; k_t4_claim / _offsets(_small): one carrier
//                       list per reported vertex; k_emit_from_walk: the rows (query.h:680-704)
// k_walk_setup /        the first kernel of a walking batch whose regions and sample ids are in device memory: the result's
//  k_seq_setup           copies, the range check of the ids, the capacities of the recording walk (no reference counterpart:
//                       the reference reads its regions from a file, src/commands.cc:100-140)
// k_scan_*, k_scan2_*,  offsets of rows, arena and scratch (exclusive scans over the regions of a batch; up to 16 k regions: one launch)
//  k_scan_small
// k_pack_regions /      what a sharded run gathers per region (the reference's loop, src/commands.cc:145-193, prints counts and
//  k_pack_seq_regions    flags per region): site range + counts, or pieces + bytes of a sequence; k_bounds_from_records: the
//                       receiving side (vs_query_expand_site_ranges)
// k_find                Index::find batched
#pragma once
#include "k_image.hip.h"
#include "k_sites.hip.h"
#include "k_scan.hip.h"
#include "k_rows.hip.h"
#include "k_expand.hip.h"
#include "k_latency.hip.h"
#include "k_walk.hip.h"
#include "k_points.hip.h"
#include "k_sample_coords.hip.h"
#include "k_digest.hip.h"
