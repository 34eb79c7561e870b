"""Resident carrier lists (option "resident_lists", engine.hip: build_resident_lists): every carrier list of the index
expanded once into an arena that stays in HBM; batches of query types 6 and 4 emit rows that point into it.  Same
answers as the per-batch expansion, text-exact against the oracle, through every way a result is read."""
import os

import numpy as np
import pytest

from helpers import oracle_texts, random_regions, write_random_cohort
from oracle.oracle import Oracle
from variantstore_amd import VariantStore

pytestmark = pytest.mark.gpu

COHORTS = [
    (701, dict(n_rows=400, ref_len=5000, n_samples=70, carrier_p=0.4)),                       # class rows wider than one word
    (702, dict(n_rows=400, ref_len=4000, p_near=0.7, p_multi=0.3, p_same=0.25)),               # crowded: the duplicate rule fires
    (703, dict(n_rows=300, ref_len=5000, n_samples=130, carrier_p=0.004)),                     # explicit sample ids
    (704, dict(n_rows=300, ref_len=6000, n_samples=900, carrier_p=0.5, p_ins=0.2, p_del=0.2)),   # dense rows: the row path
]


def _open(tmp_path, seed, kw):
    fasta, vcf, _ = write_random_cohort(str(tmp_path), seed, **kw)
    vs = VariantStore.from_vcf(fasta, vcf, device=0)
    plain = os.path.join(tmp_path, "plain.bin")
    vs.export_plain(plain)
    return vs, Oracle(plain)


def _raw_text(vs, raw, q, names, pool):
    out = ["Pos\tRef\tAlt\tSamples\n"]
    b, n = int(raw["row_begin"][q]), int(raw["row_count"][q])
    for v in raw["rows"][b:b + n]:
        if int(v["count_flags"]) >> 31:
            continue
        cnt, cb = int(v["count_flags"]) & 0x7FFFFFFF, int(v["car_begin"])
        cars = raw["arena"][cb:cb + cnt].astype(np.uint32)
        ids, gts = (cars & 0x1FFF, cars >> 13) if raw["carrier_bytes"] == 2 else (cars & 0x1FFFFFFF, cars >> 29)
        ref = pool[int(v["ref_off"]):int(v["ref_off"]) + int(v["ref_len"])]
        alt = pool[int(v["alt_off"]):int(v["alt_off"]) + int(v["alt_len"])]
        samples = "".join(f"{names[i]}({(g >> 1) & 1}{'|' if g & 1 else '/'}{(g >> 2) & 1}) " for i, g in zip(ids, gts))
        out.append(f"{int(v['pos'])}\t{ref}\t{alt}\t{samples}\n")
    return "".join(out)


@pytest.mark.parametrize("seed,kw", COHORTS)
def test_type6_over_resident_lists(seed, kw, tmp_path):
    vs, orc = _open(tmp_path, seed, kw)
    rng = np.random.default_rng(seed)
    L = vs.info().ref_length
    regions = random_regions(rng, L, 400, max_len=900)
    batches = {"unsorted": regions, "sorted": sorted(regions)}
    before = {k: vs.get_var_in_ref(b) for k, b in batches.items()}
    vs.set_option("resident_lists", 1)
    for k, b in batches.items():
        res = vs.get_var_in_ref(b)
        slots, table, arena, lists, is_shared = res.layout()
        assert arena == 0 and lists == 0 and is_shared == before[k].layout()[4]          # nothing expanded, no arena of its own
        assert res.digest() == before[k].digest() and res.totals() == before[k].totals()
        want = oracle_texts(orc, b)
        for q, (n, early, text) in enumerate(want):
            if n >= 0:
                assert res.region_text(q) == text, (k, q, b[q])
        va, vb = res.view(True), before[k].view(True)
        for key in va:
            assert np.array_equal(va[key], vb[key]), (k, key)
        res.close()
    # the raw form: rows only cross PCIe, the arena is the handle's mirror of the resident lists
    res = vs.get_var_in_ref(batches["sorted"])
    raw = res.raw(with_carriers=True)
    assert raw["resident"] and raw["shared"]
    names = [vs.sample_name(i) for i in range(vs.info().num_samples)]
    import ctypes as C
    from variantstore_amd._lib import ResultRaw
    rr = ResultRaw()
    assert vs._lib.vs_result_get_raw(res._h, 1, C.byref(rr)) == 0
    rws = raw["rows"]
    pool_len = int(max((rws["ref_off"].astype(np.int64) + rws["ref_len"]).max(), (rws["alt_off"].astype(np.int64) + rws["alt_len"]).max()))
    pool = C.string_at(rr.seq_pool, pool_len).decode("latin-1")
    for q in (0, 1, 57, 200, len(regions) - 1):
        n, _, text = orc.get_var_in_ref(*batches["sorted"][q])
        if n >= 0:
            assert _raw_text(vs, raw, q, names, pool) == text
    res2 = vs.get_var_in_ref(batches["unsorted"])
    raw2 = res2.raw(with_carriers=True)
    assert raw2["resident"] and raw2["arena"].ctypes.data == raw["arena"].ctypes.data   # one mirror per handle
    # small batches (latency path) and the point queries keep their own arenas; answers unchanged
    few = regions[:9]
    small = vs.get_var_in_ref(few)
    for q, (n, early, text) in enumerate(oracle_texts(orc, few)):
        if n >= 0:
            assert small.region_text(q) == text
    # switching the option off again: private arenas, same digest
    vs.set_option("resident_lists", 0)
    again = vs.get_var_in_ref(batches["sorted"])
    assert again.layout()[2] > 0 and again.digest() == before["sorted"].digest()


@pytest.mark.parametrize("seed,kw", COHORTS)
def test_type4_over_resident_lists(seed, kw, tmp_path):
    vs, orc = _open(tmp_path, seed, kw)
    rng = np.random.default_rng(seed + 1)
    L = vs.info().ref_length
    ns = vs.info().num_samples
    regions = sorted(random_regions(rng, L, 300, max_len=900))
    names = [vs.sample_name(int(i)) for i in rng.integers(1, ns, size=len(regions))]
    before = vs.get_sample_var_in_ref(regions, names)
    vs.set_option("resident_lists", 1)
    for coop in (2, 1):
        vs.set_option("t4_walk", coop)
        res = vs.get_sample_var_in_ref(regions, names)
        assert res.layout()[2] == 0
        assert res.digest() == before.digest() and res.totals() == before.totals()
        for q, (x, y) in enumerate(regions):
            n, _, text = orc.get_sample_var_in_ref(x, y, names[q])
            if n >= 0:
                assert res.region_text(q) == text, (coop, q, x, y, names[q])
        va, vb = res.view(True), before.view(True)
        for key in va:
            assert np.array_equal(va[key], vb[key]), (coop, key)
        res.close()
    vs.set_option("t4_walk", 2)
    # one sample for the whole batch, and the count-then-emit fallback (keeps a private arena)
    one = vs.get_sample_var_in_ref(regions, names[0])
    for q, (x, y) in enumerate(regions[:60]):
        n, _, text = orc.get_sample_var_in_ref(x, y, names[0])
        if n >= 0:
            assert one.region_text(q) == text
    vs.set_option("force_fallbacks", 1)
    two = vs.get_sample_var_in_ref(regions, names)
    vs.set_option("force_fallbacks", 0)
    assert two.layout()[2] > 0 and two.digest() == before.digest()


@pytest.mark.parametrize("seed,kw", COHORTS[:3])
def test_type5_over_resident_lists(seed, kw, tmp_path):
    """get_sample_var_in_sample (sample coordinates) with its rows pointing into the resident arena: the oracle's text."""
    vs, orc = _open(tmp_path, seed, kw)
    rng = np.random.default_rng(seed + 2)
    L = vs.info().ref_length
    ns = vs.info().num_samples
    regions = sorted(random_regions(rng, L, 120, max_len=700))
    names = [vs.sample_name(int(i)) for i in rng.integers(1, ns, size=len(regions))]
    before = vs.get_sample_var_in_sample(regions, names)
    vs.set_option("resident_lists", 1)
    res = vs.get_sample_var_in_sample(regions, names)
    assert res.layout()[2] == 0 and res.totals() == before.totals() and res.digest() == before.digest()
    flags = res.view(False)["region_flags"]
    checked = 0
    for q, (x, y) in enumerate(regions):
        n, text = orc.get_sample_var_in_sample(x, y, names[q])
        if n == -1:
            assert flags[q] & 8
            continue
        assert res.region_text(q) == text == before.region_text(q), (q, x, y, names[q])
        checked += 1
    assert checked > 20
