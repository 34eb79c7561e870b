import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from variantstore_amd import VariantStore
KW = dict(ref_length=10_000_000, num_variants=200_000, num_samples=200, seed=3, first_pos=10_000,
          frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6, af_exponent=3.0)
vs = VariantStore.synthetic(device=0, **KW)
rng = np.random.default_rng(5)
def run(regs, tag):
    res = vs.get_var_in_ref(regs)
    v = res.view(with_carriers=False)
    bad = []
    for q in range(len(regs)):
        one = vs.get_var_in_ref(regs[q:q + 1])
        if res.region_text(q) != one.region_text(0):
            bad.append(q)
        one.close()
    print(tag, "regions", len(regs), "layout", res.layout(), "mismatches", len(bad), bad[:10], flush=True)
    for q in bad[:3]:
        one = vs.get_var_in_ref(regs[q:q + 1]); v1 = one.view(with_carriers=False)
        print("   q", q, regs[q], {k: (v[k][q] if len(v[k]) > q else None) for k in ("region_flags", "var_begin", "var_count", "car_base") if k in v},
              "| single", {k: v1[k][0] for k in ("region_flags", "var_begin", "var_count", "car_base") if k in v1})
        print("      batch :", repr(res.region_text(q)[:160])); print("      single:", repr(one.region_text(0)[:160]))
        one.close()
    res.close()
    return bad
for n in (100, 300, 1000, 4000):
    starts = np.sort(rng.integers(20_000, KW["ref_length"] - 20_000, size=n)).astype(np.uint64)
    run(np.stack([starts, starts + 30], axis=1), f"scattered n={n} len=30")
    run(np.stack([starts, starts + 120], axis=1), f"scattered n={n} len=120")
