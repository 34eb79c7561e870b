#!/usr/bin/env python3
"""Type-4 walk experiment on the bench cohort (tuning build: VS_BUILD_TUNING=1 python -m variantstore_amd.build --force):
iteration counts and device-clock ticks of k_sample_walk, then timing of the whole type-4 batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from variantstore_amd import VariantStore

w = dict(bench.WORKLOADS[os.environ.get("VS_BENCH_WORKLOAD", "chr1-2504")])
vs = VariantStore.synthetic(device=0, **bench.synth_kwargs(w))
vs.set_option("phase_events", 1)   # the phase times of walking batches are read below
nreg = w["regions"]
regions = bench.make_regions(w, 0, nreg)
ns = vs.info().num_samples
sids16 = [1 + (i * 157) % (ns - 1) for i in range(16)]
per_region = np.array([sids16[i % 16] for i in range(nreg)], dtype=np.uint32)
digests = {}
for skip, coop in ((1, 16), (1, 8), (1, 0), (0, 0)):
    vs.set_option("t4_skip", skip)
    vs.set_option("t4_walk", 2 if coop else 1)
    try:
        vs.set_option("walk_stats", 1)
    except Exception as e:
        print("no walk_stats:", e)
    vs.get_sample_var_in_ref(regions, per_region).close()
    try:
        vs.set_option("walk_stats", 0)
    except Exception:
        pass
    t0 = time.perf_counter()
    for _ in range(5):
        r = vs.get_sample_var_in_ref(regions, per_region)
        t = vs.last_timing()
        r_tot, r_dig = r.totals(), r.digest()
        r.close()
    dt = (time.perf_counter() - t0) / 5
    digests[(skip, coop)] = (r_tot, r_dig)
    print(f"t4_skip={skip} coop={coop}: {nreg / dt / 1e6:.1f} M regions/s, {dt * 1e3:.3f} ms per batch; phases ms: walk {t.ms_bounds:.3f} scan {t.ms_scan:.3f} emit {t.ms_emit:.3f} fill {t.ms_fill:.3f}")
print("digests agree:", len(set(digests.values())) == 1, digests)
