#!/usr/bin/env python3
"""profiles/traffic_<workload>.json from the rocprofv3 --pmc passes of tools/profile_round.sh:

   python tools/make_traffic_json.py <workload> <tag> gpurun_out/pmc_<tag>_1 gpurun_out/pmc_<tag>_2 ...

Per kernel (fill, header, walk, ...): mean FETCH_SIZE / WRITE_SIZE per launch, the corrected HBM traffic
(2 x FETCH_SIZE + WRITE_SIZE: this image's rocprofv3 reports half the bytes of wide coalesced reads on gfx950,
MI355X_MICROARCH.md "HBM"), TCC hit rate and the SQ counters of the same passes.  The file records the git blob
hash of kernels.hip.h: bench.py only attaches these figures to a run whose kernels are the profiled ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

workload, tag, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
# the hash of the kernels the counters were taken on: recorded by the profiling run itself (profile_round.sh writes
# kernels_blob.txt next to its outputs on the GPU box); only a summary made on that same tree may fall back to hashing
blob = None
if dirs and dirs[0].startswith("--blob="):
    blob, dirs = dirs[0].split("=", 1)[1].strip(), dirs[1:]
raw = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + dirs))
NAMES = ["k_fill_carriers", "k_fill_sites2", "k_fill_sites", "k_share_rows2", "k_t6_bounds", "k_t6_mid", "k_t6_apply", "k_t6_slow", "k_emit_headers", "k_region_bounds",
         "k_sample_walk_coop", "k_sample_walk_sc",
         "k_sample_walk", "k_emit_from_walk", "k_t4_claim", "k_point_bounds", "k_has_var_filter", "k_sample_seq", "k_copy_segments"]
NAMES.sort(key=len, reverse=True)   # (longest first: k_sample_walk is a prefix of two others)
kernels = {}
for name, ctr in raw.items():
    short = next((k for k in NAMES if k in name), None)
    if short is None or "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr:
        continue
    grid = int(name.rsplit("[grid ", 1)[1].rstrip("]"))
    if short in kernels and (kernels[short]["_launches"], kernels[short]["grid"]) >= (ctr["_launches"], grid):
        continue  # (a kernel that ran in several shapes: the headline batch is the one launched most often -- warm-up, timed
                  #  steps and the comparison legs -- and, among equals, the largest)
    fetch, write = ctr["FETCH_SIZE"] * 1024, ctr["WRITE_SIZE"] * 1024
    e = {"full_name": name, "grid": grid, "_launches": ctr["_launches"], "avg_us_under_pmc": ctr["_avg_us"], "FETCH_SIZE_KiB": ctr["FETCH_SIZE"], "WRITE_SIZE_KiB": ctr["WRITE_SIZE"],
         "traffic_bytes_per_launch": int(2 * fetch + write), "traffic_bytes_per_launch_uncorrected": int(fetch + write)}
    hit, miss = ctr.get("TCC_HIT_sum", ctr.get("TCC_HIT")), ctr.get("TCC_MISS_sum", ctr.get("TCC_MISS"))
    if hit is not None and miss is not None and hit + miss > 0:
        e["tcc_hit_rate"] = hit / (hit + miss)
    for k, v in ctr.items():
        if k.startswith("SQ_") or k.startswith("GRBM_"):
            e[k] = v
    kernels[short] = e
out = {"workload": workload, "tag": tag, "kernels_blob": blob or bench.kernels_hash(),
       "note": "separate rocprofv3 --pmc passes of `python3 bench.py --steps 6 --warmup 2 --extras t4,points,sc` (tools/profile_round.sh); "
               "means over all launches of a kernel; traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of the guide; an "
               "upper bound where reads are narrow)",
       "kernels": kernels}
path = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(path, {k: v["traffic_bytes_per_launch"] for k, v in kernels.items()})
