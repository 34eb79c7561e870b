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
ta, da = a.totals(), a.digest()
b = vs.get_var_in_ref(regions)
assert (b.totals(), b.digest()) == (ta, da)
b.close()
parts = [vs.get_var_in_ref(regions[i::4]) for i in range(4)]
assert tuple(sum(p.totals()[k] for p in parts) for k in range(4)) == ta
for p in parts:
    p.close()
for q in (0, 1, 31_337, 99_999):
    single = vs.get_var_in_ref(regions[q:q + 1])
    ts = single.region_text(0)
    tq = a.region_text(q)
    print("q", q, "same:", ts == tq, len(ts), len(tq), flush=True)
    if ts != tq:
        i = next((i for i in range(min(len(ts), len(tq))) if ts[i] != tq[i]), -1)
        print("   first diff at", i, "\n   single:", repr(ts[i-60:i+80]), "\n   batch :", repr(tq[i-60:i+80]))
        print("   a.region_text again same as before:", a.region_text(q) == tq, " digest same:", a.digest() == da)
        s2 = vs.get_var_in_ref(regions[q:q + 1])
        print("   second single == first single:", s2.region_text(0) == ts, " == batch:", s2.region_text(0) == tq)
        ra = a.raw(True)
        print("   batch raw: row_begin", ra["row_begin"][q], "rows", ra["row_count"][q], "car_base", ra["car_base"][q], "car_len", ra["car_len"][q], "arena", len(ra["arena"]))
        rows = ra["rows"][int(ra["row_begin"][q]):int(ra["row_begin"][q] + ra["row_count"][q])]
        print("   last rows", rows[-3:])
        rs = single.raw(True)
        print("   single raw: car_base", rs["car_base"], "car_len", rs["car_len"], "last rows", rs["rows"][-3:], "arena", len(rs["arena"]))
    single.close()
