import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench
from variantstore_amd import VariantStore
w = dict(bench.WORKLOADS["chr1-2504"])
vs = VariantStore.synthetic(device=0, **bench.synth_kwargs(w))
vs.set_option("phase_events", 1)   # the phase times of walking batches are read below
nreg = w["regions"]; regions = bench.make_regions(w, 0, nreg)
ns = vs.info().num_samples
sids16 = [1 + (i * 157) % (ns - 1) for i in range(16)]
per = np.array([sids16[i % 16] for i in range(nreg)], dtype=np.uint32)
for _ in range(3):
    r = vs.get_sample_var_in_ref(regions, per); t = vs.last_timing()
    lay = r.layout(); tot = r.totals(); r.close()
print("layout (slots, table, arena, lists, shared):", lay, "totals", tot)
print(f"phases ms: total {t.ms_total:.3f} walk {t.ms_bounds:.3f} scan {t.ms_scan:.3f} emit {t.ms_emit:.3f} fill {t.ms_fill:.3f}")
