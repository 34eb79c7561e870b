"""Full BASELINE.json size (configs[2]: 5 M sites x 2504 samples, 100 k regions of 10 kb) on the GPU,
checked through size-independent properties -- the CPU oracle cannot hold this index in reasonable
time, so exactness at this size rests on: determinism, additivity over batch splits, equality of a
region's result whatever batch it travels in, and structural invariants of every carrier list
(ascending sample ids in range, a non-zero allele on every carrier, positions inside the region)."""
import numpy as np
import pytest

from variantstore_amd import VariantStore

pytestmark = pytest.mark.gpu

KW = dict(ref_length=249_250_621, num_variants=5_000_000, num_samples=2504, seed=1, first_pos=10_000,
          frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6, af_exponent=11.0)


@pytest.fixture(scope="module")
def big():
    vs = VariantStore.synthetic(device=0, **KW)
    rng = np.random.default_rng(2000)
    starts = np.sort(rng.integers(1, KW["ref_length"] - 10_000, size=100_000))
    regions = np.stack([starts, starts + 10_000], axis=1).astype(np.uint64)
    yield vs, regions
    vs.close()


def test_determinism_and_additivity(big):
    vs, regions = big
    a = vs.get_var_in_ref(regions)
    ta, da = a.totals(), a.digest()
    assert ta[0] == 100_000 and ta[1] > 15_000_000 and ta[2] > 2_000_000_000
    b = vs.get_var_in_ref(regions)
    assert (b.totals(), b.digest()) == (ta, da)
    b.close()
    parts = [vs.get_var_in_ref(regions[i::4]) for i in range(4)]
    assert tuple(sum(p.totals()[k] for p in parts) for k in range(4)) == ta
    for p in parts:
        p.close()
    # a region's rows do not depend on the batch around it
    for q in (0, 1, 31_337, 99_999):
        single = vs.get_var_in_ref(regions[q:q + 1])
        assert single.region_text(0) == a.region_text(q)
        single.close()
    rev = vs.get_var_in_ref(regions[::-1].copy())
    for q in (5, 50_000, 99_990):
        assert rev.region_text(100_000 - 1 - q) == a.region_text(q)
    rev.close()
    a.close()


def test_structure_of_every_carrier_list(big):
    vs, regions = big
    sub = regions[40_000:40_600]
    res = vs.get_var_in_ref(sub)
    v = res.view(with_carriers=True)
    car = v["carriers"]
    ids = (car & np.uint32(0x1FFFFFFF)).astype(np.int64)
    gt = car >> np.uint32(29)
    assert len(car) == int(v["car_count"].sum()) > 10_000_000
    assert ids.min() >= 1 and ids.max() <= 2504            # sample ids, "ref" (0) is never a carrier
    assert ((gt & 6) != 0).all()                            # every carrier holds a non-zero allele
    assert ((gt & 1) == 1).all()                            # the synthetic cohort is fully phased
    # ascending ids inside every variant (class bit order == get_sample_ids order)
    begin = v["car_begin"].astype(np.int64)
    cnt = v["car_count"].astype(np.int64)
    assert (begin[1:] == begin[:-1] + cnt[:-1]).all() and begin[0] == 0
    d = np.diff(ids)
    boundaries = (begin[1:] - 1)[cnt[:-1] > 0]
    mask = np.ones(len(d), dtype=bool)
    mask[boundaries[boundaries < len(d)]] = False
    assert (d[mask] > 0).all()
    # positions lie inside their region and are sorted up to the insertion off-by-one
    vb = v["var_begin"].astype(np.int64)
    for q in range(len(sub)):
        p = v["pos"][vb[q]:vb[q + 1]].astype(np.int64)
        if len(p):
            assert p.min() >= int(sub[q, 0]) and p.max() < int(sub[q, 1])
            assert (np.diff(p) >= -1).all()
    res.close()


def test_type4_agrees_with_type6_on_membership(big):
    """A sample's type-4 variants are the type-6 variants of the same region whose carrier list holds
    that sample (true for this cohort: no variant lies within a deletion of the same sample's path
    often enough to matter is NOT assumed -- only inclusion is checked)."""
    vs, regions = big
    sub = regions[70_000:70_050]
    sid = vs.sample_id("S01234")
    r6 = vs.get_var_in_ref(sub)
    v6 = r6.view(with_carriers=True)
    r4 = vs.get_sample_var_in_ref(sub, "S01234")
    v4 = r4.view(with_carriers=True)
    ids6 = v6["carriers"] & np.uint32(0x1FFFFFFF)
    b6, c6 = v6["car_begin"].astype(np.int64), v6["car_count"].astype(np.int64)
    vb6, vb4 = v6["var_begin"].astype(np.int64), v4["var_begin"].astype(np.int64)
    total4 = 0
    for q in range(len(sub)):
        have = set()
        for a in range(vb6[q], vb6[q + 1]):
            if (ids6[b6[a]:b6[a] + c6[a]] == sid).any():
                have.add((int(v6["pos"][a]), int(v6["alt_off"][a]), int(v6["alt_len"][a])))
        for a in range(vb4[q], vb4[q + 1]):
            key = (int(v4["pos"][a]), int(v4["alt_off"][a]), int(v4["alt_len"][a]))
            assert key in have, (q, key)
            total4 += 1
    assert total4 > 50
    r6.close()
    r4.close()


def test_one_million_region_batch_on_one_gpu(big):
    """BASELINE.json configs[3]'s batch (1,000,000 sorted 10 kb regions, seed 3) as ONE single-GPU batch: 200 M
    variant rows and 28 G carriers (56 GB of arena).  Totals equal the sum over ten 100 k-region pieces, a region's
    rows do not depend on the batch it travels in, two runs give one digest, and the batch's compact hit-list
    records (what an 8-GPU run all-gathers) expand back into a result with the same digest."""
    import torch
    vs, _ = big
    rng = np.random.default_rng(3000)
    starts = np.sort(rng.integers(1, KW["ref_length"] - 10_000, size=1_000_000))
    regions = np.stack([starts, starts + 10_000], axis=1).astype(np.uint64)
    a = vs.get_var_in_ref(regions)
    ta, da = a.totals(), a.digest()
    assert ta[0] == 1_000_000 and ta[1] > 150_000_000 and ta[2] > 20_000_000_000
    texts = {q: a.region_text(q) for q in (0, 123_456, 500_000, 999_999)}
    recs = torch.empty((1_000_000, 4), dtype=torch.int64, device="cuda")
    assert a.pack_regions_into(recs.data_ptr(), 1_000_000, 0) == 1_000_000
    a.close()
    sums = [0, 0, 0, 0]
    for i in range(10):
        p = vs.get_var_in_ref(regions[i * 100_000:(i + 1) * 100_000])
        for k, t in enumerate(p.totals()):
            sums[k] += t
        for q, text in texts.items():
            if i * 100_000 <= q < (i + 1) * 100_000:
                assert p.region_text(q - i * 100_000) == text
        p.close()
    assert tuple(sums) == ta
    b = vs.expand_site_ranges(recs.data_ptr(), 1_000_000)
    assert (b.totals(), b.digest()) == (ta, da)
    b.close()


def test_type4_event_bitmap_walk_equals_the_literal_walk_at_full_size(big):
    """Query type 4 over the whole 100 k-region batch, 16 samples round-robin (the bench's leg): the walk that jumps
    over uneventful ref-path runs with the per-sample event bitmaps gives the digest, the totals and the rows of the
    walk that visits every vertex."""
    vs, regions = big
    assert vs.info().num_samples == 2505
    sids = np.array([1 + ((i % 16) * 157) % 2504 for i in range(len(regions))], dtype=np.uint32)
    fast = vs.get_sample_var_in_ref(regions, sids)
    tf, df = fast.totals(), fast.digest()
    assert tf[1] > 500_000
    vs.set_option("t4_walk", 1)               # the one-lane-per-region form of the same walk (jumps, no groups)
    try:
        other = vs.get_sample_var_in_ref(regions, sids)
    finally:
        vs.set_option("t4_walk", 2)
    assert (other.totals(), other.digest()) == (tf, df)
    other.close()
    vs.set_option("t4_walk", 0)               # the literal walk: every vertex of the sample's path
    try:
        slow = vs.get_sample_var_in_ref(regions, sids)
    finally:
        vs.set_option("t4_walk", 2)
    assert (slow.totals(), slow.digest()) == (tf, df)
    for q in (0, 17, 50_001, 99_999):
        assert fast.region_text(q) == slow.region_text(q)
    vf, vl = fast.view(False), slow.view(False)
    for k in ("region_flags", "var_begin", "var_count", "pos", "ref_off", "ref_len", "alt_off", "alt_len", "car_count"):
        assert np.array_equal(vf[k], vl[k]), k
    fast.close()
    slow.close()


def test_shared_and_private_carrier_lists_agree_at_full_size(big):
    """The bench batch both ways: one list per covered site shared by the regions that report it (the default for a
    sorted batch) against a private copy per region -- same digest, totals and rows; a quarter of the lists."""
    vs, regions = big
    shared = vs.get_var_in_ref(regions)
    slots, table, arena, lists, is_shared = shared.layout()
    assert is_shared and slots > 15_000_000 and 4_000_000 < lists <= table < 5_100_000
    ts, ds = shared.totals(), shared.digest()
    texts = {q: shared.region_text(q) for q in (0, 7, 55_555, 99_999)}
    shared.close()
    vs.set_option("share_lists", 0)
    try:
        private = vs.get_var_in_ref(regions)
    finally:
        vs.set_option("share_lists", 1)
    p_slots, p_table, p_arena, p_lists, p_shared = private.layout()
    assert not p_shared and p_slots == p_table == p_lists == slots and p_arena > 3 * arena
    assert (private.totals(), private.digest()) == (ts, ds)
    for q, text in texts.items():
        assert private.region_text(q) == text
    private.close()


def test_unsorted_and_resident_forms_agree_at_full_size(big):
    """The bench batch four ways: sorted (shared rows and lists), in random order (sorted on the device, outcomes handed
    back in the caller's order), and both again over resident carrier lists (nothing expanded): same totals, the same
    per-region counts and texts, digests equal where the region order is."""
    vs, regions = big
    try:
        base = vs.get_var_in_ref(regions)
        tb, db = base.totals(), base.digest()
        vb = base.view(False)
        perm = np.random.default_rng(77).permutation(len(regions))
        shuffled = np.ascontiguousarray(regions[perm])
        mixed = vs.get_var_in_ref(shuffled)
        assert mixed.layout()[4] and mixed.layout()[:4] == base.layout()[:4] and mixed.totals() == tb
        vm = mixed.view(False)
        assert np.array_equal(vm["var_count"], vb["var_count"][perm]) and np.array_equal(vm["region_flags"], vb["region_flags"][perm])
        probes = (0, 3, 12_345, 77_777, 99_999)
        for j in probes:
            assert mixed.region_text(j) == base.region_text(int(perm[j]))
        dm = mixed.digest()
        vs.set_option("resident_lists", 1)
        res_sorted, res_mixed = vs.get_var_in_ref(regions), vs.get_var_in_ref(shuffled)
        assert res_sorted.layout()[2] == 0 and res_mixed.layout()[2] == 0
        assert (res_sorted.totals(), res_sorted.digest()) == (tb, db) and (res_mixed.totals(), res_mixed.digest()) == (tb, dm)
        for j in probes:
            assert res_sorted.region_text(j) == base.region_text(j) and res_mixed.region_text(j) == base.region_text(int(perm[j]))
        # type 4 over resident lists: 16 samples round-robin, against the per-batch expansion
        per = np.array([1 + (i * 157) % 2503 for i in range(16)], dtype=np.uint32)[np.arange(len(regions)) % 16]
        r4 = vs.get_sample_var_in_ref(regions, per)
        vs.set_option("resident_lists", 0)
        e4 = vs.get_sample_var_in_ref(regions, per)
        assert r4.layout()[2] == 0 and e4.layout()[2] > 0 and (r4.totals(), r4.digest()) == (e4.totals(), e4.digest())
        for j in probes:
            assert r4.region_text(j) == e4.region_text(j)
        for x in (base, mixed, res_sorted, res_mixed, r4, e4):
            x.close()
    finally:
        vs.set_option("resident_lists", 0)
        vs.set_option("share_lists", 1)


def test_a_loop_of_batches_in_flight_neither_allocates_nor_frees_device_memory(big):
    """The bench's loop form -- a batch call returns when the batch is enqueued, its result is closed one step late -- after
    batches of many other shapes have left their buffers in the handle's pool: once warm, no step calls hipMalloc or hipFree
    (hipFree waits for the whole device: a pool that trimmed itself inside the loop turned 0.6 ms steps into 2 ms ones).
    Counters: vs_index_info.pool_mallocs / pool_frees."""
    import torch
    vs, regions = big
    for n in (70, 300, 1_000, 3_000, 7_000, 11_000, 20_000, 33_000, 50_000, 64_000, 80_000, 90_000):   # clutter: twelve other shapes
        vs.get_var_in_ref(regions[:n]).close()
    per = np.full(5_000, 17, dtype=np.uint32)
    vs.get_sample_var_in_ref(regions[:5_000], per).close()
    dev = torch.from_numpy(regions.astype(np.int64)).cuda()
    n = len(regions)

    def loop(steps):
        prev, digests = None, set()
        for _ in range(steps):
            r = vs.get_var_in_ref_device(dev.data_ptr(), n)
            if prev is not None:
                prev.close()
            prev = r
        digests.add(prev.digest())
        prev.close()
        return digests

    loop(4)                                    # warm: the pool now holds what two batches alive at a time need
    before = vs.info()
    assert len(loop(25)) == 1
    after = vs.info()
    assert (after.pool_mallocs, after.pool_frees) == (before.pool_mallocs, before.pool_frees)


def test_oracle_spot_checks_through_windows(big, tmp_path):
    """The CPU oracle on the FULL-size index, through windows (exactness at full size used to rest on properties
    alone).  The generator's records inside a window are written as FASTA + VCF relative to the window
    (tests/native/synth_windows.cpp), built into a small index through the product's VCF path on the host, and the oracle's
    rows over that index -- shifted back by the window's origin -- are compared with what the GPU answers on the whole
    5 M-site index: query type 6 on 90 + 12 regions, type 4 on 24 (region, sample) pairs whose walk back to the sample's
    previous vertex stays inside a 150 kb margin.  tests/test_windows.py checks the windowing itself against the oracle on
    a cohort it can hold whole."""
    from helpers import parse_rows, synth_windows, window_oracle
    vs, _regions = big
    rng = np.random.default_rng(77)
    m6, m4 = 20_000, 150_000
    wins, plan = [], []   # plan: (window, kind, [(x, y), ...])
    for c in np.sort(rng.integers(1_000_000, KW["ref_length"] - 1_000_000, size=30)):
        c = int(c)
        wins.append((c - m6, c + 30_000 + m6))
        plan.append((len(wins) - 1, 6, [(c + j * 10_000, c + (j + 1) * 10_000) for j in range(3)]))
    for c in np.sort(rng.integers(1_000_000, KW["ref_length"] - 1_000_000, size=6)):
        c = int(c)
        wins.append((c - m4, c + 20_000 + m4))
        plan.append((len(wins) - 1, 4, [(c, c + 10_000), (c + 10_000, c + 20_000)]))
    counts = synth_windows(KW, wins, tmp_path)
    assert min(counts) > 500
    # the full-size answers, one batch per query type (sorted by start: shared rows and lists, the bench's own path)
    q6 = sorted(r for _w, _k, rs in plan for r in rs)
    res6 = vs.get_var_in_ref(np.array(q6, dtype=np.uint64))
    assert res6.layout()[4]
    text6 = {r: res6.region_text(i) for i, r in enumerate(q6)}
    # type 4: two samples per region that carry something within 100 kb before it
    q4 = []
    for _w, kind, rs in plan:
        if kind != 4:
            continue
        for (x, y) in rs:
            before = vs.get_var_in_ref(np.array([[x - 100_000, x]], dtype=np.uint64))
            names = []
            for row in reversed(parse_rows(before.region_text(0))):
                for s in row[3].split():
                    nm = s.split("(")[0]
                    if nm not in names:
                        names.append(nm)
                if len(names) >= 2:
                    break
            before.close()
            assert len(names) >= 2
            q4 += [((x, y), names[0]), ((x, y), names[1])]
    res4 = vs.get_sample_var_in_ref(np.array([r for r, _s in q4], dtype=np.uint64), [s for _r, s in q4])
    text4 = {(r, s): res4.region_text(i) for i, (r, s) in enumerate(q4)}
    n6 = n4 = rows6 = rows4 = 0
    for w, kind, rs in plan:
        lo, _hi = wins[w]
        orc = window_oracle(tmp_path, w)
        for (x, y) in rs:
            n, _, t = orc.get_var_in_ref(x - lo + 1, y - lo + 1)
            assert n >= 0 and parse_rows(text6[(x, y)]) == parse_rows(t, lo - 1), ("type 6", x, y)
            n6 += 1
            rows6 += n
            if kind == 4:
                for (r, s) in [k for k in text4 if k[0] == (x, y)]:
                    n, _, t = orc.get_sample_var_in_ref(x - lo + 1, y - lo + 1, s)
                    assert n >= 0 and parse_rows(text4[(r, s)]) == parse_rows(t, lo - 1), ("type 4", x, y, s)
                    n4 += 1
                    rows4 += n
        orc.close()
    res6.close()
    res4.close()
    assert n6 == 102 and n4 == 24 and rows6 > 15_000 and rows4 > 0


def test_vcf_text_truth_at_full_size(big, tmp_path):
    """The one check that bypasses the product's constructor, at FULL size: the generator's records inside six
    windows are written as VCF text (tests/native/synth_windows.cpp), the expected type-6 rows of the ISOLATED records follow from
    that text alone (tests/vcf_truth.py: VCF semantics + the reference's reporting convention; no from_vcf, no oracle), and they are
    compared with what the GPU answers on the whole 5 M-site index: every such row is reported, with exactly that carrier text, by a
    small region around it and by the 10 kb region that holds it; query type 4 for one of its carriers reports a substitution's row."""
    from helpers import synth_windows, vcf_truth_cases
    vs, _regions = big
    rng = np.random.default_rng(99)
    wins = [(int(c), int(c) + 30_000) for c in np.sort(rng.integers(1_000_000, KW["ref_length"] - 1_000_000, size=6))]
    synth_windows(KW, wins, tmp_path)
    cases = []
    for k, (lo, _hi) in enumerate(wins):
        _names, rows = vcf_truth_cases(tmp_path, k, lo)
        cases += rows
    assert len(cases) >= 2000, len(cases)
    cases.sort(key=lambda c: c[0])
    small = np.array([(p - 3, p + 4) for p, _t, _r in cases], dtype=np.uint64)
    res = vs.get_var_in_ref(small)
    wide = np.array([(p - 5_000, p + 5_000) for p, _t, _r in cases[::25]], dtype=np.uint64)
    resw = vs.get_var_in_ref(wide)
    assert res.layout()[4] and resw.layout()[4]
    for q, (p, text, _rec) in enumerate(cases):
        assert text in res.region_text(q), (p, text[:60])
    for q, (p, text, _rec) in enumerate(cases[::25]):
        assert text in resw.region_text(q), (p, "inside its 10 kb region")
    res.close()
    resw.close()
    # query type 4: a carrier of a substitution reports the record's row (indels follow type 4's own rule, query.h:682-698)
    subs = [(p, t, r) for p, t, r in cases if len(r[1]) == len(r[2][0])]
    pick = [subs[i] for i in rng.choice(len(subs), size=200, replace=False)]
    pick.sort(key=lambda c: c[0])
    carriers = []
    for _p, t, _r in pick:
        cs = t.split("\t")[3].split()
        carriers.append(cs[len(cs) // 2].split("(")[0])
    r4 = vs.get_sample_var_in_ref(np.array([(p - 3, p + 4) for p, _t, _r in pick], dtype=np.uint64), carriers)
    for q, (p, text, _rec) in enumerate(pick):
        assert text in r4.region_text(q), (p, carriers[q])
    r4.close()
