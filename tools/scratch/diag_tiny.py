"""Diagnosis (round 6): thousands of tiny sorted regions on the full-size index -- batch answer against the single-region answer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from variantstore_amd import VariantStore
KW = dict(ref_length=249_250_621, num_variants=5_000_000, num_samples=2504, seed=1, first_pos=10_000,
          frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6, af_exponent=11.0)
vs = VariantStore.synthetic(device=0, **KW)
rng = np.random.default_rng(5)
for trial, (nwin, per) in enumerate([(6, 600), (40, 100), (1, 3000), (200, 20)]):
    starts = []
    for c in np.sort(rng.integers(1_000_000, KW["ref_length"] - 1_000_000, size=nwin)):
        starts += [int(c) + 50 * j + int(rng.integers(0, 20)) for j in range(per)]
    starts = np.array(sorted(starts), dtype=np.uint64)
    regs = np.stack([starts, starts + 7], axis=1)
    res = vs.get_var_in_ref(regs)
    bad = 0
    for q in range(len(regs)):
        one = vs.get_var_in_ref(regs[q:q + 1])
        a, b = res.region_text(q), one.region_text(0)
        one.close()
        if a != b:
            bad += 1
            if bad <= 5:
                print("MISMATCH trial", trial, "q", q, regs[q], repr(a[:200]), "single:", repr(b[:200]))
    print("trial", trial, "regions", len(regs), "layout", res.layout(), "mismatches", bad, flush=True)
    # private rows for comparison
    vs.set_option("share_lists", 0)
    priv = vs.get_var_in_ref(regs)
    vs.set_option("share_lists", 1)
    print("   shared digest == private digest:", res.digest() == priv.digest(), res.totals(), priv.totals(), flush=True)
    priv.close(); res.close()
