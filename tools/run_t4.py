#!/usr/bin/env python3
"""Five type-4 batches of the bench's shape (16 samples round-robin over 100 k sorted regions) -- the thing to profile
when the walk kernels are the subject.  VS_T4_COOP=0 / VS_T4_SKIP=0 select the serial / literal forms."""
import os
import sys

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
vs.set_option("t4_walk", int(os.environ.get("VS_T4_WALK", "2")))
if os.environ.get("VS_FILL_CHUNK"):
    vs.set_option("fill_chunk", int(os.environ["VS_FILL_CHUNK"]))   # (tuning builds)
for _ in range(5):
    r = vs.get_sample_var_in_ref(regions, per_region)
    t = vs.last_timing()
    r.close()
print(f"phases ms: walk {t.ms_bounds:.3f} scan {t.ms_scan:.3f} emit {t.ms_emit:.3f} fill {t.ms_fill:.3f}")
