#!/usr/bin/env python3
"""The measured table of DESIGN.md section 7 / README from the committed bench lines of a profiling round:
   python tools/final_table.py r05f   (reads profiles/<tag>_bench.json, <tag>c_bench.json, <tag>t_bench.json and friends)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05f"


def load(name):
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    with open(p) as f:
        txt = f.read().strip()
    return json.loads(txt.splitlines()[-1]) if txt else None


def M(x):
    return "—" if x is None else f"{x / 1e6:.1f} M"


rows = []
for wl, suffix in (("chr1-2504", ""), ("chr22-100", "c"), ("tcga-10k", "t")):
    d = load(f"{tag}{suffix}_bench.json")
    if not d:
        continue
    r = d["roofline"]
    bc = r.get("box_ceilings") or {}
    sc = d.get("sample_coordinate_queries") or {}
    t4 = d.get("type4") or {}
    pq = d.get("point_queries") or {}
    print(f"## {wl}  ({tag}{suffix}_bench.json)")
    print(f"value {M(d['value'])} regions/s, ms_per_step {d['ms_per_step']:.4f}")
    print(f"expansion {r['kernel']}: in loop {r['avg_launch_ms']:.4f} ms, alone {r.get('avg_launch_ms_alone') or 0:.4f} ms; traffic {(r.get('traffic') or 0) / 1e9:.3f} GB ({r['basis']}); "
          f"frac {r['frac']:.3f}, frac_alone {r.get('frac_alone') or 0:.3f}; box mix ceiling {bc.get('mix_2to3_GBps')}, copy {bc.get('copy_1to1_GBps')}; "
          f"frac_of_box_mix {r.get('frac_of_box_mix_ceiling') or 0:.3f} (alone {r.get('frac_alone_of_box_mix_ceiling') or 0:.3f})")
    print(f"type4 {M(t4.get('queries_per_s'))} ({t4.get('ms_per_batch')}) | types 2/3/5 {M(sc.get('type2_queries_per_s'))} / {M(sc.get('type3_queries_per_s'))} / {M(sc.get('type5_queries_per_s'))} | "
          f"type1 {pq.get('type1_queries_per_s')}, type7 {pq.get('type7_queries_per_s_device')}")
    print(f"p50 {d.get('p50_latency_us')} us (paced {d.get('p50_latency_paced_1ms_us')}); unsorted {M((d.get('unsorted_batch') or {}).get('queries_per_s'))}; "
          f"private {M(((d['config'].get('result_layout') or {}).get('private_rows_and_lists_for_comparison') or {}).get('queries_per_s'))}; "
          f"delivered {M(d.get('delivered_queries_per_s'))}; resident {M((d.get('resident_lists') or {}).get('queries_per_s'))}")
    cb = d.get("cpu_baseline")
    if cb:
        print(f"cpu_baseline {cb['value']:.1f} q/s on {cb['cores']} core; all_cores {cb.get('all_cores', {}).get('value')} on {cb.get('all_cores', {}).get('cores')}; parity_checked_regions {d.get('parity_checked_regions')}")
    print(f"digest {d.get('result_digest')}")
    print()
