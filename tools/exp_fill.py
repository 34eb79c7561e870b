#!/usr/bin/env python3
"""Same-process experiments on the type-6 throughput path: one index build, many configurations.
python tools/exp_fill.py [config ...]; a config is NAME:option=VAL,...[,len=N][,lm=N] where option is a key of
vs_index_set_option (fill_ablate / fill_lds_pad need a tuning build: VS_BUILD_TUNING=1 python -m variantstore_amd.build --force)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from variantstore_amd import VariantStore

w = dict(bench.WORKLOADS[os.environ.get("VS_BENCH_WORKLOAD", "chr1-2504")])
vs = None
cur_lm = None


def reopen(lm):
    """lm=N in a config: (re)build the index with VS_LIST_MAX=N (the threshold is fixed when an index is opened)"""
    global vs, cur_lm
    if vs is not None and lm == cur_lm:
        return
    if vs is not None:
        vs.close()
    if lm is None:
        os.environ.pop("VS_LIST_MAX", None)
    else:
        os.environ["VS_LIST_MAX"] = str(lm)
    vs = VariantStore.synthetic(device=0, **bench.synth_kwargs(w))
    cur_lm = lm
OPTIONS = {"fill_ablate": 0, "fill_lds_pad": 0, "fill_fused": 1, "fill_chunk": 0, "fill_stats": 0}   # the switches an experiment may set, with their defaults


def run(name, env, rlen, steps=8):
    unknown = set(env) - set(OPTIONS)
    if unknown:
        raise SystemExit(f"unknown option(s) in config {name}: {sorted(unknown)}")
    for k, dflt in OPTIONS.items():
        vs.set_option(k, int(env.get(k, dflt)))
    ww = dict(w, region_len=rlen)
    regions = bench.make_regions(ww, 0, w["regions"])
    for _ in range(2):
        vs.get_var_in_ref(regions).close()
    fill = tot = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        res = vs.get_var_in_ref(regions)
        t = vs.last_timing()
        fill += t.ms_fill
        tot += t.ms_total
        dig = res.digest() if _ == steps - 1 else 0
        res.close()
    wall = (time.perf_counter() - t0) / steps * 1e3
    print(f"{name:>28}: len {rlen:6d}  fill {fill / steps:.3f} ms  emit {t.ms_emit:.3f}  scan {t.ms_scan:.3f}  pipeline {tot / steps:.3f} ms  "
          f"wall {wall:.3f} ms  digest {dig:016x}", flush=True)


configs = sys.argv[1:] or ["default:"]
for rep in range(int(os.environ.get("EXP_REPS", "2"))):
    for c in configs:
        name, _, rest = c.partition(":")
        env, rlen, lm = {}, w["region_len"], None
        for kv in filter(None, rest.split(",")):
            k, v = kv.split("=")
            if k == "len":
                rlen = int(v)
            elif k == "lm":
                lm = int(v)
            else:
                env[k] = v
        reopen(lm)
        run(name, env, rlen)
