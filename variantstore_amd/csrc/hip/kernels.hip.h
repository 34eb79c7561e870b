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
// k_fill_carriers       get_samples -> get_sample_id / get_sample_phasing
//                       (query.h:268-285, variant_graph.h:875-942): expansion of
//                       decoded class id lists / class bit rows + genotype bits into
//                       carrier lists.  This is the dominant kernel.
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
