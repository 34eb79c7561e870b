import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from variantstore_amd import VariantStore
w = bench.WORKLOADS["chr1-2504"]
vs = VariantStore.synthetic(device=0, **bench.synth_kwargs(w))
nreg = w["regions"]; regions = bench.make_regions(w, 0, nreg)
ns = vs.info().num_samples
sids16 = [1 + (i * 157) % (ns - 1) for i in range(16)]
per = np.array([sids16[i % 16] for i in range(nreg)], dtype=np.uint32)
r = vs.get_sample_var_in_ref(regions, per)
raw = r.raw(with_carriers=False)
rows = raw["rows"]
key = rows["pos"].astype(np.uint64) << np.uint64(32) | rows["alt_off"].astype(np.uint64)
u, idx = np.unique(key, return_index=True)
cnt = (rows["count_flags"] & 0x7FFFFFFF).astype(np.int64)
print("type-4 rows", len(rows), "unique (pos, alt)", len(u), "carriers total", cnt.sum(), "carriers unique", cnt[idx].sum())
