"""Diagnostic: the sequence of test_determinism_and_additivity with a check of `a` after every step."""
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
da = a.digest()
qs = (0, 1, 31_337, 99_999)
ta = {q: a.region_text(q) for q in qs}
print("a digest %016x" % da)
def check(label):
    d = a.digest()
    t = {q: a.region_text(q) for q in qs}
    print(label, "digest same:", d == da, "texts same:", [t[q] == ta[q] for q in qs], flush=True)
b = vs.get_var_in_ref(regions); b.close(); check("after b")
parts = [vs.get_var_in_ref(regions[i::4]) for i in range(4)]; check("after parts")
for p in parts: p.close()
check("after parts closed")
vs.set_option("share_lists", 0)
pv = vs.get_var_in_ref(regions)
vs.set_option("share_lists", 1)
for q in qs:
    single = vs.get_var_in_ref(regions[q:q + 1])
    ts = single.region_text(0)
    tp = pv.region_text(q)
    print("q", q, "single == a:", ts == ta[q], "single == private:", ts == tp, "a == private:", ta[q] == tp, "len", len(ts), len(ta[q]), flush=True)
    if ts != ta[q]:
        i = next((i for i in range(min(len(ts), len(ta[q]))) if ts[i] != ta[q][i]), -1)
        print("   first diff at", i, repr(ts[i-40:i+60]), "|||", repr(ta[q][i-40:i+60]))
        rs = single.raw(True)
        print("   single rows", rs["row_begin"], rs["row_count"], rs["car_base"], rs["car_len"], "last rows", rs["rows"][-3:])
    single.close()
    check("after single %d" % q)
