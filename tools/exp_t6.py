#!/usr/bin/env python3
"""Type-6 throughput experiments on the bench cohort: options given as key=value on the command line are applied
before the timed batches (vs_index_set_option); prints fill / rows / pipeline times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from variantstore_amd import VariantStore

w = dict(bench.WORKLOADS[os.environ.get("VS_BENCH_WORKLOAD", "chr1-2504")])
vs = VariantStore.synthetic(device=0, **bench.synth_kwargs(w))
nreg = w["regions"]
regions = bench.make_regions(w, 0, nreg)
dev = torch.from_numpy(regions.astype(np.int64)).cuda().contiguous()
torch.cuda.synchronize()
configs = sys.argv[1:] or [""]
for rep in range(2):
    for cfg in configs:
        opts = dict(kv.split("=") for kv in cfg.split(",") if kv)
        for k, v in opts.items():
            vs.set_option(k, int(v))
        for _ in range(3):
            vs.get_var_in_ref_device(dev.data_ptr(), nreg).close()
        fill = emit = tot = 0.0
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            r = vs.get_var_in_ref_device(dev.data_ptr(), nreg)
            t = vs.last_timing()
            fill += t.ms_fill; emit += t.ms_emit; tot += t.ms_total
            r.close()
        wall = (time.perf_counter() - t0) / n * 1e3
        print(f"{cfg or 'default':>28}: fill {fill / n:.3f} rows {emit / n:.3f} pipeline {tot / n:.3f} wall {wall:.3f} ms  -> {nreg / wall / 1e3:.1f} M regions/s", flush=True)
        for k in opts:
            vs.set_option(k, {"share_lists": 1, "fill_chunk": 0}.get(k, 0))
