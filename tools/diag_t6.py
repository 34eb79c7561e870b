"""Diagnostic: a shared type-6 batch against the private form and the latency path, region by region (raw arrays)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from variantstore_amd import VariantStore

KW = dict(ref_length=249_250_621, num_variants=5_000_000, num_samples=2504, seed=1, first_pos=10_000,
          frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6, af_exponent=11.0)
vs = VariantStore.synthetic(device=0, **KW)
rng = np.random.default_rng(2000)
starts = np.sort(rng.integers(1, KW["ref_length"] - 10_000, size=100_000))
regions = np.stack([starts, starts + 10_000], axis=1).astype(np.uint64)
a = vs.get_var_in_ref(regions)
print("layout", a.layout(), "totals", a.totals(), "digest %016x" % a.digest())
ra = a.raw(with_carriers=False)
vs.set_option("share_lists", 0)
p = vs.get_var_in_ref(regions)
vs.set_option("share_lists", 1)
rp = p.raw(with_carriers=False)
print("private digest %016x" % p.digest())
for k in ("row_count", "var_count", "car_len", "region_flags"):
    bad = np.nonzero(ra[k] != rp[k])[0]
    print(k, "differs in", len(bad), "regions", bad[:10])
nbad = 0
for q in list(range(0, 100_000, 997)) + [0, 1, 31_337, 99_999]:
    ta, tp = a.region_text(q), p.region_text(q)
    if ta != tp:
        nbad += 1
        i = next((i for i in range(min(len(ta), len(tp))) if ta[i] != tp[i]), min(len(ta), len(tp)))
        print("region", q, "text differs at", i, "of", len(ta), len(tp), "| rows", ra["row_begin"][q], ra["row_count"][q], "car_base", ra["car_base"][q],
              "car_len", ra["car_len"][q], "private car_len", rp["car_len"][q])
        rows = ra["rows"][int(ra["row_begin"][q]):int(ra["row_begin"][q] + ra["row_count"][q])]
        print("   first/last car_begin", rows["car_begin"][0], rows["car_begin"][-1], "last count", rows["count_flags"][-1])
        if nbad > 5:
            break
print("bad regions:", nbad)
