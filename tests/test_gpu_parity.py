"""GPU parity (the parity tests proper): the HIP path, called through the C ABI,
against the golden vectors and against the CPU oracle on the same seeded inputs.
Bar: bit-exact text of every region (positions, REF/ALT strings, sample sets with
phasing), plus the early-out flag."""
import os

import numpy as np
import pytest

from helpers import oracle_texts, random_regions, write_random_cohort
from oracle.oracle import Oracle
from variantstore_amd import VariantStore

pytestmark = pytest.mark.gpu


def _open_gpu(fasta, vcf, tmp_path):
    vs = VariantStore.from_vcf(fasta, vcf, device=0)
    plain = os.path.join(tmp_path, "plain.bin")
    vs.export_plain(plain)
    return vs, Oracle(plain)


def _compare_t6(vs, orc, regions):
    res = vs.get_var_in_ref(regions)
    view = res.view(with_carriers=False)
    want = oracle_texts(orc, regions)
    checked = 0
    for q, (n, early, text) in enumerate(want):
        if n < 0:
            continue  # the reference does not terminate on this region; nothing to be identical to
        assert res.region_text(q) == text, (q, regions[q])
        assert int(view["var_count"][q]) == n
        assert bool(view["region_flags"][q] & 1) == early, (q, regions[q])
        checked += 1
    assert checked > 0
    res.close()
    return checked


@pytest.mark.parametrize("key", ["G1", "G3", "G4"])
def test_golden_vectors_on_gpu(key, golden_dir, survey_vectors, tmp_path):
    g = survey_vectors[key]
    vs, _ = _open_gpu(os.path.join(golden_dir, g["fasta"]), os.path.join(golden_dir, g["vcf"]), tmp_path)
    res = vs.get_var_in_ref([tuple(g["region"])])
    assert res.region_text(0) == g["text"]


def test_golden_g2_and_probes_on_gpu(golden_dir, survey_vectors, tmp_path):
    g = survey_vectors["G2"]
    vs, _ = _open_gpu(os.path.join(golden_dir, g["fasta"]), os.path.join(golden_dir, g["vcf"]), tmp_path)
    res = vs.get_var_in_ref([tuple(g["region"]), (10, 105)])
    lines = res.region_text(0).split("\n")[1:-1]
    assert len(lines) == g["count"] and lines[-1] == g["last_row"]
    for frag in g["contains"]:
        assert any(l.startswith(frag) for l in lines)
    assert res.totals()[1] == g["count"] + survey_vectors["readme"]["x"]["t6_10_105"]
    p = survey_vectors["G4_probes"]
    vs, _ = _open_gpu(os.path.join(golden_dir, p["fasta"]), os.path.join(golden_dir, p["vcf"]), tmp_path)
    regions = [tuple(x["region"]) for x in p["probes"]]
    res = vs.get_var_in_ref(regions)
    flags = res.view(False)["region_flags"]
    for q, x in enumerate(p["probes"]):
        assert res.region_text(q) == x["text"]
        assert bool(flags[q] & 1) == x["early_out"]


def test_index_find_batched(golden_dir, survey_vectors, tmp_path):
    g = survey_vectors["x_small_graph"]
    vs, orc = _open_gpu(os.path.join(golden_dir, g["fasta"]), os.path.join(golden_dir, g["vcf"]), tmp_path)
    pos = list(range(1, 120))
    got = vs.find(pos)
    assert [int(v) for v in got] == [orc.find(p) for p in pos]
    for p, v in g["find"].items():
        assert int(vs.find([int(p)])[0]) == v


@pytest.mark.parametrize("seed,kw", [
    (101, dict()),                                                             # SNP/indel/MNP/two-ALT mix
    (102, dict(sample_names=["S2", "S10", "S1", "b", "a", "Z", "m"])),          # name order != column order
    (103, dict(p_near=0.7, p_multi=0.3, n_rows=300, ref_len=3000)),             # crowded sites, many dummies
    (104, dict(p_ins=0.3, p_del=0.3, n_rows=250)),                              # indel heavy, overlapping deletions
    (105, dict(n_samples=70, carrier_p=0.4)),                                   # class rows wider than one word
    (106, dict(n_samples=130, carrier_p=0.004, n_rows=200)),                    # sparse -> explicit sample ids
    (107, dict(unphased_p=0.5, missing_p=0.2, haploid_p=0.2)),                  # '/', './.', haploid GT
    (108, dict(n_samples=1, carrier_p=1.0)),                                    # single sample
])
def test_random_cohorts_match_oracle(seed, kw, tmp_path):
    fasta, vcf, _ = write_random_cohort(str(tmp_path), seed, **kw)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(seed)
    ref_len = vs.info().ref_length
    regions = random_regions(rng, ref_len, 400)
    _compare_t6(vs, orc, regions)
    if seed == 106:
        assert vs.info().use_bit_vector == 0
    # regions handed over unsorted and duplicated must not matter
    perm = rng.permutation(len(regions))
    _compare_t6(vs, orc, [regions[i] for i in perm][:120] + regions[:5])


def test_regions_resident_in_device_memory(tmp_path):
    """vs_query_var_in_ref_device: the same batch handed over as a device buffer answers with the same digest and text."""
    import torch
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 31, n_rows=300, ref_len=5000, n_samples=70, carrier_p=0.4)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    regions = random_regions(np.random.default_rng(31), vs.info().ref_length, 300)
    arr = np.asarray(regions, dtype=np.uint64).reshape(-1, 2)
    dev = torch.from_numpy(arr.astype(np.int64)).cuda().contiguous()
    torch.cuda.synchronize()
    a = vs.get_var_in_ref(arr)
    b = vs.get_var_in_ref_device(dev.data_ptr(), arr.shape[0])
    assert a.digest() == b.digest() and a.totals() == b.totals()
    for q in (0, 7, 150, 299, len(regions) - 1):
        assert b.region_text(q) == orc.get_var_in_ref(*regions[q])[2]
    small = vs.get_var_in_ref_device(dev.data_ptr(), 3)      # small batches too take the batch pipeline here
    assert [small.region_text(q) for q in range(3)] == [a.region_text(q) for q in range(3)]
    empty = vs.get_var_in_ref_device(dev.data_ptr(), 0)
    assert empty.totals() == (0, 0, 0, 0)


def test_invalid_and_empty_batches(tmp_path):
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 7)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    res = vs.get_var_in_ref([])
    assert res.totals() == (0, 0, 0, 0)
    res = vs.get_var_in_ref([(0, 50), (1, 50)])
    v = res.view(False)
    assert v["region_flags"][0] & 2 and int(v["var_count"][0]) == 0   # pos_x < 1: the reference aborts
    assert res.region_text(1) == orc.get_var_in_ref(1, 50)[2]


def test_synthetic_midsize_all_regions(tmp_path):
    """20k variants x 200 samples: every region of a 2000-region batch, text-exact."""
    vs = VariantStore.synthetic(device=0, ref_length=2_000_000, num_variants=20_000, num_samples=200, seed=21,
                                first_pos=500, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=6,
                                af_exponent=3.0)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(5)
    starts = rng.integers(1, 1_990_000, size=2000)
    regions = [(int(s), int(s) + 5000) for s in starts]
    n = _compare_t6(vs, orc, regions)
    assert n == 2000 and orc.ub_events() == 0


@pytest.mark.parametrize("length", [7, 30, 120])
def test_many_short_scattered_regions_many_runs(length, tmp_path):
    """Round 6: thousands of SHORT regions far apart -- the shared table then consists of hundreds of RUNS of one or two rows (a run: a
    maximal stretch of covered sites), and a wave task of k_fill_sites2 looks its rows' runs up in a window of 64 run records
    (shared_row_run).  A row that was the last of 63 single-row runs took the 64th run's site: the window's 64th record was fetched by a
    shuffle under a divergent condition, with its source lane switched off.  Batches of long overlapping regions (the bench, every
    earlier test) have a handful of runs and never reached that code; the VCF-text check at full size did.  Every region against the
    oracle, shared against private rows, 300 to 4000 regions."""
    vs = VariantStore.synthetic(device=0, ref_length=10_000_000, num_variants=200_000, num_samples=200, seed=3, first_pos=10_000,
                                frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6, af_exponent=3.0)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(5)
    for n in (300, 1000, 4000):
        starts = np.sort(rng.integers(20_000, 9_980_000, size=n))
        regions = [(int(x), int(x) + length) for x in starts]
        res = vs.get_var_in_ref(regions)
        assert res.layout()[4]
        for q, (x, y) in enumerate(regions):
            c, _, text = orc.get_var_in_ref(x, y)
            if c >= 0:
                assert res.region_text(q) == text, (n, q, x, y)
        vs.set_option("share_lists", 0)
        private = vs.get_var_in_ref(regions)
        vs.set_option("share_lists", 1)
        assert (private.totals(), private.digest()) == (res.totals(), res.digest()), n
        private.close()
        res.close()
    vs.close()


def test_type6_batches_that_outgrow_the_previous_batch_are_redone(tmp_path):
    """Round 6 (option t6_speculate, default on): a type-6 batch of more than 64 regions on a handle whose previous shared batch had about
    as many regions is SUBMITTED without waiting for its plan's totals -- table and arena are sized from that batch (+ 1/8), the kernels
    read the totals in device memory, and a batch that does not fit (or is not sorted) is refused on the device and run again with the
    exact sizes when its result is first asked for anything.  Short regions, then as many long ones (refused, redone), the same again
    (fits), an unsorted batch (refused: the device-side sort runs in the redo), results freed unread, eight and more batches in flight --
    every answer against the oracle and against the same handle with the option off."""
    vs = VariantStore.synthetic(device=0, ref_length=2_000_000, num_variants=20_000, num_samples=200, seed=21,
                                first_pos=500, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=6, af_exponent=3.0)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(9)
    n = 1500
    starts = np.sort(rng.integers(1, 1_980_000, size=n))
    short = np.stack([starts, starts + 300], axis=1).astype(np.uint64)
    long_ = np.stack([starts, starts + 9000], axis=1).astype(np.uint64)
    shuffled = long_[rng.permutation(n)]

    def check(res, regions, step=7):
        for q in range(0, len(regions), step):
            c, _, text = orc.get_var_in_ref(int(regions[q, 0]), int(regions[q, 1]))
            if c >= 0:
                assert res.region_text(q) == text, (q, regions[q])

    vs.set_option("t6_speculate", 0)
    want = {}
    for name, regs in (("short", short), ("long", long_)):   # (an unsorted batch would make the handle sort first for its next 32 batches)
        r = vs.get_var_in_ref(regs)
        want[name] = (r.totals(), r.digest(), r.layout())
        r.close()
    assert vs.info().t6_speculated == 0
    vs.set_option("t6_speculate", 1)
    a = vs.get_var_in_ref(short)            # (the first batch after the switch sets the handle's expectation: it has one already, from the runs above)
    a.close()
    s0 = vs.info()
    b = vs.get_var_in_ref(short)            # speculative, fits
    assert vs.info().t6_speculated == s0.t6_speculated + 1
    assert (b.totals(), b.digest(), b.layout()) == want["short"]
    check(b, short)
    b.close()
    # the handle expects the long batches' sizes now only if the short ones shrank it -- they did not (eight small batches in a row would):
    # force the small expectation with a fresh handle state: nine short batches, unread
    for _ in range(9):
        vs.get_var_in_ref(short).close()
    r0 = vs.info().t6_refused
    c = vs.get_var_in_ref(long_)            # as many regions, thirty times the rows: refused on the device, redone at first use
    assert (c.totals(), c.digest(), c.layout()) == want["long"]
    assert vs.info().t6_refused == r0 + 1
    check(c, long_)
    d = vs.get_var_in_ref(long_)            # fits now
    assert (d.totals(), d.digest(), d.layout()) == want["long"] and vs.info().t6_refused == r0 + 1
    check(d, long_, step=11)
    c.close(); d.close()
    e = vs.get_var_in_ref(shuffled)         # not sorted: the device refuses, the redo sorts on the device
    shuffled_answer = (e.totals(), e.digest())
    assert e.totals() == want["long"][0] and vs.info().t6_refused == r0 + 2
    check(e, shuffled, step=11)
    e.close()
    f = vs.get_var_in_ref(long_)            # (a handle that has just sorted sorts first: not speculative; still the same answer)
    assert (f.totals(), f.digest(), f.layout()) == want["long"]
    f.close()
    # many batches in flight, results read in another order than they were submitted, some never read
    for _ in range(40):
        vs.get_var_in_ref(long_).close()    # (until the handle looks at the order as given again and finds it sorted)
    flight = [vs.get_var_in_ref(long_) for _ in range(12)]
    for k in (11, 0, 5):
        assert (flight[k].totals(), flight[k].digest(), flight[k].layout()) == want["long"], k
    for r in flight:
        r.close()
    g = vs.get_var_in_ref(long_)
    assert (g.totals(), g.digest()) == want["long"][:2]
    check(g, long_, step=13)
    g.close()
    assert vs.info().t6_speculated >= s0.t6_speculated + 10
    vs.set_option("t6_speculate", 0)
    h = vs.get_var_in_ref(shuffled)
    assert (h.totals(), h.digest()) == shuffled_answer
    h.close()
    vs.close()


@pytest.mark.parametrize("shape", ["wide", "explicit"])
def test_many_short_scattered_regions_wide_and_explicit_cohorts(shape, tmp_path):
    """The many-runs shape of a shared batch (test_many_short_scattered_regions_many_runs) through the OTHER instantiation of
    k_fill_sites2 -- 32-bit carrier words: a class-row cohort of 4,100 samples, and an explicit-id cohort of 10,000 (lane per group for
    every variant) -- 2,500 regions of 25 bases, every region against the oracle, shared against private rows."""
    kw = dict(ref_length=1_500_000, num_variants=30_000, seed=9, first_pos=2_000, frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6)
    if shape == "wide":
        kw.update(num_samples=4_100, af_exponent=3.0)
    else:
        kw.update(num_samples=10_000, af_exponent=2.0, max_af=0.0004)
    vs = VariantStore.synthetic(device=0, **kw)
    info = vs.info()
    assert bool(info.use_bit_vector) == (shape == "wide")
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(6)
    starts = np.sort(rng.integers(3_000, 1_495_000, size=2_500))
    regions = [(int(x), int(x) + 25) for x in starts]
    res = vs.get_var_in_ref(regions)
    assert res.layout()[4] and res.layout()[1] > 200     # (hundreds of table rows: hundreds of runs)
    for q, (x, y) in enumerate(regions):
        c, _, text = orc.get_var_in_ref(x, y)
        if c >= 0:
            assert res.region_text(q) == text, (q, x, y)
    vs.set_option("share_lists", 0)
    private = vs.get_var_in_ref(regions)
    vs.set_option("share_lists", 1)
    assert (private.totals(), private.digest()) == (res.totals(), res.digest())
    private.close(); res.close(); vs.close()


def test_run_table_sizes_around_the_window_of_64(tmp_path):
    """A wave task of the shared expansion finds its rows' runs in a window of 64 run records; batches of at most 64 runs load the whole
    table, larger ones go through the coarse index (shared_row_run).  Batches with EXACTLY 2, 63, 64, 65, 66, 128, 129 and 200 runs --
    that many far-apart regions with sites, plus a hundred repeats of the first one (more than 64 regions: the throughput path; repeats
    cover nothing new) -- every region against the oracle, shared against private rows."""
    vs = VariantStore.synthetic(device=0, ref_length=10_000_000, num_variants=200_000, num_samples=200, seed=3, first_pos=10_000,
                                frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6, af_exponent=3.0)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    for runs in (2, 63, 64, 65, 66, 128, 129, 200):
        spaced = [(20_000 + 40_000 * k, 20_000 + 40_000 * k + 700 + 13 * (k % 7)) for k in range(runs)]   # ~14 sites each, 40 kb apart
        regions = sorted(spaced + [spaced[0]] * 100)
        res = vs.get_var_in_ref(regions)
        assert res.layout()[4]
        seen = {}
        for q, (x, y) in enumerate(regions):
            if (x, y) not in seen:
                seen[(x, y)] = orc.get_var_in_ref(x, y)
            c, _, text = seen[(x, y)]
            assert c > 0 and res.region_text(q) == text, (runs, q, x, y)
        vs.set_option("share_lists", 0)
        private = vs.get_var_in_ref(regions)
        vs.set_option("share_lists", 1)
        assert (private.totals(), private.digest()) == (res.totals(), res.digest()), runs
        private.close(); res.close()
    vs.close()


def test_plan_with_several_regions_per_thread(tmp_path, monkeypatch):
    """The plan's kernels take `items` regions per thread when a batch has more than 256 x 4096 regions -- a path no other test reaches
    (the 1 M-region batch of the full-size tests is just below it).  VS_PLAN_ITEMS (read when a handle is opened) asks for it from 64 k
    regions on: a 70,000-region batch, sorted and shuffled, speculative and not, on a handle opened with 4 regions per thread against a
    handle opened without -- same totals, digests and layout, and 300 of the regions against the oracle."""
    kw = dict(ref_length=2_000_000, num_variants=20_000, num_samples=200, seed=21, first_pos=500, frac_ins=0.05, frac_del=0.05,
              frac_multi=0.02, max_indel=6, af_exponent=3.0)
    rng = np.random.default_rng(17)
    starts = np.sort(rng.integers(1, 1_995_000, size=70_000))
    regions = np.stack([starts, starts + rng.integers(1, 3000, size=70_000)], axis=1).astype(np.uint64)
    shuffled = regions[rng.permutation(len(regions))]
    answers = []
    for items in (None, "4"):
        if items:
            monkeypatch.setenv("VS_PLAN_ITEMS", items)
        else:
            monkeypatch.delenv("VS_PLAN_ITEMS", raising=False)
        vs = VariantStore.synthetic(device=0, **kw)
        got = []
        for regs in (regions, regions, shuffled):          # (the second one is submitted speculatively)
            r = vs.get_var_in_ref(regs)
            got.append((r.totals(), r.digest(), r.layout()))
            if items and regs is regions:
                plain = os.path.join(tmp_path, "p.bin")
                if not os.path.exists(plain):
                    vs.export_plain(plain)
                orc = Oracle(plain)
                for q in range(0, len(regs), 233):
                    c, _, text = orc.get_var_in_ref(int(regs[q, 0]), int(regs[q, 1]))
                    if c >= 0:
                        assert r.region_text(q) == text, q
            r.close()
        assert got[0] == got[1]
        answers.append(got)
        vs.close()
    assert answers[0] == answers[1]


def test_digest_properties(tmp_path):
    """Size-independent properties used at full scale: the device digest is a function of the result
    only (same batch twice, and any permutation of the batch re-indexed, give the same per-region
    digests), and totals are additive over a split of the batch."""
    vs = VariantStore.synthetic(device=0, ref_length=1_000_000, num_variants=30_000, num_samples=300, seed=9,
                                first_pos=500, frac_ins=0.05, frac_del=0.05, frac_multi=0.01, af_exponent=3.0)
    rng = np.random.default_rng(1)
    starts = rng.integers(1, 990_000, size=5000)
    regions = [(int(s), int(s) + 4000) for s in starts]
    a = vs.get_var_in_ref(regions)
    b = vs.get_var_in_ref(regions)
    assert a.digest() == b.digest()
    assert a.totals() == b.totals()
    half = len(regions) // 2
    t1 = vs.get_var_in_ref(regions[:half]).totals()
    t2 = vs.get_var_in_ref(regions[half:]).totals()
    assert tuple(x + y for x, y in zip(t1, t2)) == a.totals()
    # a single-region batch holds the same variants as that region inside the big batch
    for q in (0, 17, 4999):
        single = vs.get_var_in_ref([regions[q]])
        assert single.region_text(0) == a.region_text(q)


# ---------------------------------------------------------------- query type 4
def _compare_t4(vs, orc, regions, sample):
    res = vs.get_sample_var_in_ref(regions, sample)
    view = res.view(with_carriers=False)
    want = oracle_texts(orc, regions, sample=sample)
    checked = 0
    for q, (n, early, text) in enumerate(want):
        if n < 0:
            continue
        assert res.region_text(q) == text, (q, regions[q], sample)
        assert int(view["var_count"][q]) == n
        assert bool(view["region_flags"][q] & 1) == early
        checked += 1
    res.close()
    return checked


@pytest.mark.parametrize("key", ["G1_t4", "G3_t4"])
def test_golden_type4_on_gpu(key, golden_dir, survey_vectors, tmp_path):
    g = survey_vectors[key]
    vs, _ = _open_gpu(os.path.join(golden_dir, g["fasta"]), os.path.join(golden_dir, g["vcf"]), tmp_path)
    res = vs.get_sample_var_in_ref([tuple(g["region"])], g["sample"])
    assert res.region_text(0) == g["text"]


@pytest.mark.parametrize("seed,kw", [
    (201, dict()),
    (202, dict(sample_names=["S2", "S10", "S1", "b", "a", "Z", "m"])),
    (203, dict(p_near=0.7, p_multi=0.3, n_rows=300, ref_len=3000)),
    (204, dict(p_ins=0.3, p_del=0.3, n_rows=250)),
    (205, dict(n_samples=70, carrier_p=0.4)),
    (206, dict(n_samples=130, carrier_p=0.004, n_rows=200)),
])
def test_random_cohorts_type4_match_oracle(seed, kw, tmp_path):
    fasta, vcf, names = write_random_cohort(str(tmp_path), seed, **kw)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(seed)
    regions = random_regions(rng, vs.info().ref_length, 150)
    for sample in [names[0], names[-1], names[len(names) // 2]]:
        assert _compare_t4(vs, orc, regions, sample) > 0


def test_type4_mixed_samples_in_one_batch(tmp_path):
    """vs_query_samples_var_in_ref: one sample id per region."""
    fasta, vcf, names = write_random_cohort(str(tmp_path), 207, n_samples=9, p_near=0.5, p_multi=0.2)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(207)
    regions = random_regions(rng, vs.info().ref_length, 120)
    per = [names[i % len(names)] for i in range(len(regions))]
    res = vs.get_sample_var_in_ref(regions, per)
    for q, (x, y) in enumerate(regions):
        n, early, text = orc.get_sample_var_in_ref(x, y, per[q])
        if n >= 0:
            assert res.region_text(q) == text, (q, per[q])


def test_type4_synthetic_midsize(tmp_path):
    vs = VariantStore.synthetic(device=0, ref_length=2_000_000, num_variants=20_000, num_samples=200, seed=21,
                                first_pos=500, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=6,
                                af_exponent=3.0)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(6)
    starts = rng.integers(1, 1_990_000, size=400)
    regions = [(int(s), int(s) + 5000) for s in starts]
    for sample in ("S00001", "S00100", "S00200"):
        assert _compare_t4(vs, orc, regions, sample) == 400


def test_wide_cohort_dense_variants(tmp_path):
    """1500 samples with common alleles: variants with hundreds to > 1000 carriers drive the
    wave-per-variant regimes of k_fill_carriers (medium and dense), checked text-exact."""
    vs = VariantStore.synthetic(device=0, ref_length=400_000, num_variants=3000, num_samples=1500, seed=77,
                                first_pos=200, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=4,
                                af_exponent=3.5)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(8)
    starts = rng.integers(1, 395_000, size=120)
    regions = [(int(s), int(s) + 3000) for s in starts]
    res = vs.get_var_in_ref(regions)
    cc = res.view(False)["car_count"]
    assert (cc > 640).sum() > 50 and ((cc > 32) & (cc <= 640)).sum() > 50 and (cc <= 32).sum() > 50
    n = _compare_t6(vs, orc, regions)
    assert n == len(regions)
    assert _compare_t4(vs, orc, regions[:40], "S00750") == 40


@pytest.mark.parametrize("list_max", [0, 3, 17, 64, 200])
@pytest.mark.parametrize("n_samples", [70, 150, 1500, 3900])
def test_list_threshold_sweep(list_max, n_samples, tmp_path, monkeypatch):
    """k_fill_carriers takes a variant either from its class's decoded 16-bit id list (<= list_max carriers) or from
    its bit row (two rounds of half a row).  With the production threshold small cohorts never reach the row path,
    so the threshold is swept here (VS_LIST_MAX is read when an index is opened): both paths, row widths of 2, 3
    (odd), 24 and 61 (odd, 31-bit slices) words, text-exact against the oracle for query types 6 and 4."""
    monkeypatch.setenv("VS_LIST_MAX", str(list_max))
    small = n_samples <= 150
    vs = VariantStore.synthetic(device=0, ref_length=60_000 if small else 120_000, num_variants=1500 if small else 700,
                                num_samples=n_samples, seed=500 + n_samples, first_pos=100, frac_ins=0.06, frac_del=0.06,
                                frac_multi=0.03, max_indel=4, af_exponent=2.5)
    assert vs.info().list_max == list_max
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(list_max * 31 + n_samples)
    L = vs.info().ref_length
    starts = rng.integers(1, L - 4000, size=60 if small else 30)
    regions = [(int(s), int(s) + int(rng.integers(1, 4000))) for s in starts] + [(1, 3000), (L - 2000, L + 5)]
    cc = vs.get_var_in_ref(regions).view(False)["car_count"]
    if list_max * 4 < n_samples:
        assert (cc > list_max).sum() > 20, "the row path must be exercised"
    assert _compare_t6(vs, orc, regions) == len(regions)
    assert _compare_t6(vs, orc, regions[:3]) == 3          # latency path (4-slot tasks)
    name = vs.sample_name(1 + n_samples // 2)
    assert _compare_t4(vs, orc, regions[:12], name) == 12
    vs.close()


def test_cohort_wider_than_one_wave_of_row_words(tmp_path):
    """4500 samples: class rows of 71 words (> 64) take the generic expansion path."""
    vs = VariantStore.synthetic(device=0, ref_length=60_000, num_variants=400, num_samples=4500, seed=78,
                                first_pos=200, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=4,
                                af_exponent=3.0)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(9)
    starts = rng.integers(1, 57_000, size=40)
    regions = [(int(s), int(s) + 2500) for s in starts]
    cc = vs.get_var_in_ref(regions).view(False)["car_count"]
    assert (cc > 32).sum() > 20 and cc.max() > 1000
    assert _compare_t6(vs, orc, regions) == len(regions)
    assert _compare_t4(vs, orc, regions[:10], "S02250") == 10


def test_hit_list_records_for_the_collective(tmp_path):
    """vs_result_pack_headers / vs_result_pack_regions write what the view holds (device -> torch tensor)."""
    import torch
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 301, n_rows=300, ref_len=3000, p_near=0.7, p_multi=0.3)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(3)
    regions = random_regions(rng, vs.info().ref_length, 200)
    res = vs.get_var_in_ref(regions)
    v = res.view(with_carriers=False)
    n = res.num_header_records()
    assert n == len(v["pos"])
    buf = torch.zeros((n + 3, 4), dtype=torch.int64, device="cuda")
    assert res.pack_headers_into(buf.data_ptr(), n + 3, region_base=1000) == n
    rec = buf.cpu().numpy().view(np.uint64)[:n]
    assert np.array_equal(rec[:, 0] & np.uint64((1 << 63) - 1), v["pos"])
    assert np.array_equal((rec[:, 0] >> np.uint64(63)).astype(np.uint32), v["var_flags"] & 1)
    assert np.array_equal(rec[:, 1], v["ref_off"].astype(np.uint64) | (v["ref_len"].astype(np.uint64) << np.uint64(32)))
    assert np.array_equal(rec[:, 2], v["alt_off"].astype(np.uint64) | (v["alt_len"].astype(np.uint64) << np.uint64(32)))
    assert np.array_equal(rec[:, 3] >> np.uint64(32), v["car_count"].astype(np.uint64))
    vb = v["var_begin"].astype(np.int64)
    reg_of_slot = np.repeat(np.arange(len(regions)), np.diff(vb))
    assert np.array_equal(rec[:, 3] & np.uint64(0xFFFFFFFF), (reg_of_slot + 1000).astype(np.uint64))
    # compact records: one per region
    q = res.num_region_records()
    assert q == len(regions)
    rbuf = torch.zeros((q, 4), dtype=torch.int64, device="cuda")
    assert res.pack_regions_into(rbuf.data_ptr(), q, region_base=7) == q
    rr = rbuf.cpu().numpy().view(np.uint64)
    assert np.array_equal(rr[:, 0], np.arange(q, dtype=np.uint64) + np.uint64(7))
    assert np.array_equal(rr[:, 2] >> np.uint64(32), v["var_count"])
    assert np.array_equal(rr[:, 2] & np.uint64(0xFFFFFFFF), np.diff(vb).astype(np.uint64))
    car_per_region = np.add.reduceat(np.concatenate([v["car_count"], [0]]).astype(np.uint64), np.minimum(vb[:-1], len(v["car_count"])))
    car_per_region[np.diff(vb) == 0] = 0
    assert np.array_equal(rr[:, 3], car_per_region)
    assert np.array_equal((rr[:, 1] >> np.uint64(32)) & np.uint64(3), v["region_flags"].astype(np.uint64) & np.uint64(3))
    has_dropped = ((rr[:, 1] >> np.uint64(40)) & np.uint64(1)).astype(bool)
    assert np.array_equal(has_dropped, v["var_count"] != np.diff(vb).astype(np.uint64))
    # the site range really is the region's variant list: consecutive regions with equal first sites
    # and counts have equal rows
    first = (rr[:, 1] & np.uint64(0xFFFFFFFF))
    for a in range(q - 1):
        for b in range(a + 1, min(q, a + 4)):
            if first[a] == first[b] and rr[a, 2] == rr[b, 2] and not has_dropped[a] and not has_dropped[b] and rr[a, 2] > 0:
                assert res.region_text(a) == res.region_text(b)
    # ... and the receiving side of the collective expands the records back into the very same result
    back = vs.expand_site_ranges(rbuf.data_ptr(), q)
    assert back.totals() == res.totals()
    for a in range(q):
        assert back.region_text(a) == res.region_text(a), a
    vb_back = back.view(False)
    assert np.array_equal(vb_back["var_count"], v["var_count"]) and np.array_equal(vb_back["region_flags"], v["region_flags"])
    # records of two shards of the batch, concatenated as an all-gather would deliver them (region_base = shard offset)
    half = q // 2
    both = torch.zeros((q, 4), dtype=torch.int64, device="cuda")
    ra, rb = vs.get_var_in_ref(regions[:half]), vs.get_var_in_ref(regions[half:])
    ra.pack_regions_into(both.data_ptr(), half, region_base=0)
    rb.pack_regions_into(both[half:].data_ptr(), q - half, region_base=half)
    assert np.array_equal(both.cpu().numpy().view(np.uint64)[:, 0], np.arange(q, dtype=np.uint64))
    merged = vs.expand_site_ranges(both.data_ptr(), q)
    assert merged.digest() == res.digest() and merged.totals() == res.totals()
    # a record that does not fit this index's site table is refused, not followed
    bad = rbuf.clone()
    bad[0, 1] = int(vs.info().num_sites)          # first site == G with a non-zero count
    bad[0, 2] = 5
    worst = vs.expand_site_ranges(bad.data_ptr(), q)
    assert worst.view(False)["region_flags"][0] & 2 and worst.region_text(1) == res.region_text(1)


def test_latency_path_with_large_answers(tmp_path):
    """Batches of <= 64 regions take the single-launch latency path (k_query_small), whose result slab is sized on the
    host from the same bounds arithmetic the device repeats: regions holding thousands of variants and millions of
    carriers, mixed with tiny ones, and a 512-region batch (general path) all give the oracle's text."""
    vs = VariantStore.synthetic(device=0, ref_length=600_000, num_variants=9000, num_samples=1200, seed=91,
                                first_pos=200, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=4,
                                af_exponent=2.5)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    whole = [(1, 599_000)]
    res = vs.get_var_in_ref(whole)
    nv, ncar = int(res.view(False)["var_begin"][1]), int(res.totals()[2])
    assert nv > 4096 and ncar > (1 << 20), (nv, ncar)
    assert _compare_t6(vs, orc, whole) == 1
    # within the variant guess but not the carrier guess, and the other way round
    assert _compare_t6(vs, orc, [(1, 250_000)]) == 1
    assert _compare_t6(vs, orc, [(1, 599_000), (300_000, 301_000), (5, 200)]) == 3
    # and right after a fallback the latency path still answers small regions
    assert _compare_t6(vs, orc, [(300_000, 301_000)]) == 1
    batch = [(int(s), int(s) + 700) for s in np.random.default_rng(3).integers(1, 598_000, size=512)]
    assert _compare_t6(vs, orc, batch) == 512


def test_resident_server_and_launch_forms_of_the_latency_path_agree(tmp_path):
    """Small batches are answered by the resident query server (no launch per request) while one is alive, by a single
    launch otherwise: same text either way, across server expiry (it leaves 1 ms after the last request), restart, a
    general-path batch in between, and requests too large for the server."""
    import time
    vs = VariantStore.synthetic(device=0, ref_length=400_000, num_variants=6000, num_samples=900, seed=93,
                                first_pos=200, frac_ins=0.05, frac_del=0.05, frac_multi=0.03, max_indel=4,
                                af_exponent=2.5)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(17)
    singles = [[(int(s), int(s) + int(rng.integers(1, 3000)))] for s in rng.integers(1, 396_000, size=40)]
    pairs = [[(int(s), int(s) + 900), (int(s) + 5000, int(s) + 5400), (3, 2)] for s in rng.integers(1, 390_000, size=10)]
    want = {tuple(b): [orc.get_var_in_ref(x, y)[2] for x, y in b] for b in singles + pairs}

    def check(batches):
        for b in batches:
            res = vs.get_var_in_ref(b)
            assert [res.region_text(q) for q in range(len(b))] == want[tuple(b)], b
            res.close()

    vs.set_option("latency_server", 2)               # the server starts with the first small request and stays
    check(singles[:20] + pairs[:5])
    assert vs.last_timing().fill_launches == 0       # ... and answers without a launch
    time.sleep(0.01)                                 # ... leaves by its idle clock; the next request restarts it
    check(singles[20:30])
    big = [(int(s), int(s) + 700) for s in rng.integers(1, 398_000, size=300)]
    assert _compare_t6(vs, orc, big) == 300          # general path (stops a live server first)
    check(pairs[5:])
    assert _compare_t6(vs, orc, [(1, 300_000)]) == 1  # too many tasks for the server: one launch sized for it
    vs.set_option("latency_server", 0)
    check(singles[30:] + pairs[:3])                  # launch form
    assert vs.last_timing().fill_launches == 1
    vs.set_option("latency_server", 1)               # default: only a back-to-back streak of small queries starts it
    check(singles[:2])
    assert vs.last_timing().fill_launches == 1
    for _ in range(3):                               # (back to back as seen from C: keep Python out of the gaps)
        fast = [np.asarray(b, dtype=np.uint64).reshape(-1, 2) for b in singles[:12]]
        answered_by_server = 0
        for arr in fast:
            vs.get_var_in_ref(arr).close()
            answered_by_server += vs.last_timing().fill_launches == 0
        if answered_by_server:
            break
    assert answered_by_server > 0
    check(singles[:12])
    time.sleep(0.002)                                # a caller that pauses goes back to the single launch
    check(singles[5:6])
    assert vs.last_timing().fill_launches == 1
    vs.close()


def _parse_rows(text):
    rows = []
    for line in text.split("\n")[1:-1]:
        pos, ref, alt, _ = line.split("\t")
        rows.append((int(pos), ref, alt))
    return rows


@pytest.mark.parametrize("seed,kw", [
    (301, dict()),
    (302, dict(p_same=0.25, p_near=0.6)),
    (303, dict(n_samples=70, carrier_p=0.2, ref_len=3000, n_rows=90)),
    (304, dict(n_samples=40, carrier_p=0.004, ref_len=2500, n_rows=60)),   # explicit sample ids
    # crowded rows that stop a third into the reference, repeated positions (zero-length dummy ref nodes at the
    # last sites): the step back from beyond the last variant must land on the FIRST reportable slot at or after
    # the find() image, not on the last slot (found by tools/stress_parity.py, cohort 12199)
    (305, dict(ref_len=1352, n_rows=370, n_samples=5, p_ins=0.15, p_del=0.27, p_multi=0.36, p_mnp=0.1, p_near=0.86,
               carrier_p=0.05, p_same=0.5, max_indel=2)),
])
def test_type1_closest_var_matches_oracle(seed, kw, tmp_path):
    """Query type 1 at EVERY position of the reference (and past its end), text-exact, including the
    calls the reference answers with `false` (no file written)."""
    fasta, vcf, _ = write_random_cohort(str(tmp_path), seed, **kw)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    L = kw.get("ref_len", 4000)
    # the last two exercise the reference's `int cur_pos` truncation (a position like 2^31 - 1 is left out: the
    # reference -- and the literal oracle -- would step back two billion positions one call at a time)
    positions = list(range(0, L + 40)) + [2 ** 31 + 5, 2 ** 32 + 7]
    res = vs.closest_var(positions)
    flags = res.view(False)["region_flags"]
    for q, p in enumerate(positions):
        n, text = orc.closest_var(p)
        if n < 0:
            assert flags[q] & 4 and res.region_text(q) == "", p
        else:
            assert not (flags[q] & 4) and res.region_text(q) == text, p
    res.close()


def test_type1_on_an_index_without_variants_ahead(golden_dir, tmp_path):
    vs, orc = _open_gpu(os.path.join(golden_dir, "x.small.fa"), os.path.join(golden_dir, "g4.vcf"), tmp_path)
    positions = list(range(0, 120))
    res = vs.closest_var(positions)
    for q, p in enumerate(positions):
        n, text = orc.closest_var(p)
        assert n >= 0 and res.region_text(q) == text, p


@pytest.mark.parametrize("seed,kw", [
    (311, dict()),
    (312, dict(p_same=0.25, p_near=0.6)),
    (313, dict(n_samples=70, carrier_p=0.2, ref_len=3000, n_rows=90)),
])
def test_type7_samples_has_var_matches_oracle(seed, kw, tmp_path):
    """Query type 7 for every variant type 6 reports (at its own position and the neighbouring ones), for
    wrong REF/ALT strings, and for positions with nothing to find."""
    fasta, vcf, _ = write_random_cohort(str(tmp_path), seed, **kw)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    L = kw.get("ref_len", 4000)
    rows = _parse_rows(orc.get_var_in_ref(1, L)[2])
    assert len(rows) > 50
    qs = []
    for pos, ref, alt in rows:
        for d in (-2, -1, 0, 1):
            if pos + d >= 0:
                qs.append((pos + d, ref, alt))
        qs.append((pos, ref + "A", alt))
        qs.append((pos, ref, alt + "C"))
        qs.append((pos, alt, ref))
        qs.append((pos, "", ""))
    qs += [(L + 50, "A", "C"), (0, "A", "C"), (1, "", "")]
    res = vs.samples_has_var([q[0] for q in qs], [q[1] for q in qs], [q[2] for q in qs])
    flags = res.view(False)["region_flags"]
    found = 0
    for q, (pos, ref, alt) in enumerate(qs):
        want = orc.samples_has_var(pos, ref, alt)
        if want is None:
            assert flags[q] & 4 and res.region_text(q) == "", qs[q]
        else:
            found += 1
            assert not (flags[q] & 4) and res.region_text(q) == want, qs[q]
    assert found > 20
    res.close()


# ---- sample-coordinate queries (types 2, 3, 5): SURVEY.md §8(f) rank 3 ----
SC_COHORTS = [
    (401, dict()),
    (402, dict(p_same=0.2, p_near=0.6, p_multi=0.2)),
    (403, dict(n_samples=70, carrier_p=0.2, ref_len=3000, n_rows=90)),
    (404, dict(n_samples=40, carrier_p=0.004, ref_len=2500, n_rows=60)),   # explicit sample ids
    (405, dict(sample_names=["S2", "S10", "S1", "x", "A9"], p_ins=0.25, p_del=0.25)),  # column order != name order
]


def _sc_regions(rng, L):
    regions = random_regions(rng, L, 60, max_len=400)
    regions += [(0, 10), (0, 0), (1, 1), (3, 2), (L - 3, L + 30)]
    regions += [(x, x + 1) for x in range(1, 40)]
    return regions


@pytest.mark.parametrize("seed,kw", SC_COHORTS)
@pytest.mark.parametrize("coords", [0, 1])
def test_types_2_and_3_sample_sequences_match_oracle(seed, kw, coords, tmp_path):
    """query_sample_from_ref / query_sample_from_sample: the sequence string of every sample (and of "ref")
    over random and edge-shaped regions; inputs on which the reference throws or never returns are flagged."""
    fasta, vcf, names = write_random_cohort(str(tmp_path), seed, **kw)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    L = vs.info().ref_length
    rng = np.random.default_rng(seed)
    regions = _sc_regions(rng, L)
    samples = ["ref"] + [names[i] for i in rng.choice(len(names), size=min(4, len(names)), replace=False)]
    ok = 0
    for smp in samples:
        res = vs.query_sample_seq(regions, smp, sample_coordinates=bool(coords))
        flags, seqs = res.sequences()
        for q, (x, y) in enumerate(regions):
            n, seq = (orc.query_sample_from_sample if coords else orc.query_sample_from_ref)(x, y, smp)
            if n == -1:
                assert flags[q] & 8, (smp, x, y)
            elif n == -3:
                assert flags[q] & 2, (smp, x, y)
            else:
                assert not flags[q] and seqs[q] == seq, (smp, x, y)
                assert res.region_text(q) == seq + "\n"
                ok += 1
        res.close()
    assert ok > 200


@pytest.mark.parametrize("seed,kw", SC_COHORTS)
def test_type5_sample_variants_in_sample_coordinates_match_oracle(seed, kw, tmp_path):
    fasta, vcf, names = write_random_cohort(str(tmp_path), seed, **kw)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    L = vs.info().ref_length
    rng = np.random.default_rng(seed + 1)
    regions = _sc_regions(rng, L)
    samples = ["ref"] + [names[i] for i in rng.choice(len(names), size=min(4, len(names)), replace=False)]
    ok = nvar = 0
    for smp in samples:
        res = vs.get_sample_var_in_sample(regions, smp)
        view = res.view(False)
        for q, (x, y) in enumerate(regions):
            n, text = orc.get_sample_var_in_sample(x, y, smp)
            if n == -1:
                assert view["region_flags"][q] & 8, (smp, x, y)
                continue
            assert not view["region_flags"][q] and res.region_text(q) == text, (smp, x, y)
            assert int(view["var_count"][q]) == n
            ok += 1
            nvar += n
        res.close()
    assert ok > 200 and nvar > 100


def test_sample_sequence_batch_with_one_sample_per_region(tmp_path):
    fasta, vcf, names = write_random_cohort(str(tmp_path), 410, n_samples=9, p_near=0.5)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(410)
    regions = random_regions(rng, vs.info().ref_length, 150, max_len=800)
    per = [names[i % len(names)] for i in range(len(regions))]
    for coords in (False, True):
        res = vs.query_sample_seq(regions, per, sample_coordinates=coords)
        flags, seqs = res.sequences()
        for q, (x, y) in enumerate(regions):
            n, seq = (orc.query_sample_from_sample if coords else orc.query_sample_from_ref)(x, y, per[q])
            if n >= 0:
                assert seqs[q] == seq, (q, per[q], coords)
        assert res.totals()[3] == sum(len(s) for s in seqs)


def test_type4_rows_can_be_dropped_and_rebuilt_on_an_open_handle(tmp_path):
    """Option t4_rows_max_mb (variantstore_hip.h): the per-sample event / hold rows of query type 4 leave and come back on an
    open handle, vs_index_get_info accounts for them, and every walking query type answers the same as the oracle either way."""
    fasta, vcf, names = write_random_cohort(str(tmp_path), 977, n_samples=60, n_rows=260, ref_len=5000, p_near=0.5)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(977)
    regions = np.array(random_regions(rng, vs.info().ref_length, 150, max_len=900), dtype=np.uint64)
    ids = rng.integers(1, vs.info().num_samples, size=len(regions)).astype(np.uint32)
    want = [orc.get_sample_var_in_ref(int(x), int(y), vs.sample_name(int(i)))[2] for (x, y), i in zip(regions, ids)]
    info0 = vs.info()
    assert info0.t4_rows_bytes > 0

    def answers():
        a = vs.get_sample_var_in_ref(regions, ids)
        b = vs.get_sample_var_in_sample(regions, ids)
        c = vs.query_sample_seq(regions, ids)
        out = ([a.region_text(q) for q in range(len(regions))], a.digest(), b.digest(), c.sequences()[1])
        a.close(); b.close(); c.close()
        return out

    with_rows = answers()
    assert with_rows[0] == want
    vs.set_option("t4_rows_max_mb", 0)
    info1 = vs.info()
    assert info1.t4_rows_bytes == 0 and info1.device_bytes < info0.device_bytes
    assert answers() == with_rows
    vs.set_option("t4_rows_max_mb", 0)            # (nothing to drop: no-op)
    vs.set_option("t4_rows_max_mb", 1 << 20)
    info2 = vs.info()
    assert info2.t4_rows_bytes == info0.t4_rows_bytes and info2.device_bytes == info0.device_bytes
    assert answers() == with_rows
    vs.set_option("t4_rows_max_mb", 1 << 20)      # (they fit the cap: kept)
    assert vs.info().t4_rows_bytes == info0.t4_rows_bytes
    vs.close()


def test_walking_batches_around_the_one_launch_scans_tile_sizes(tmp_path):
    """Batches of 4095 ... 16385 regions: one launch of 1 - 4 blocks scans their offsets (k_scan_small, k_scan2_small: tiles of 4096,
    at most 16384 elements), one more region takes the three-launch scan.  Query types 4, 5, 2 and 3 with inputs in device memory:
    totals = the sum over the batch cut in two uneven parts, same digest as from host arrays, oracle text on a sample of regions."""
    import torch
    from variantstore_amd import DeviceArray
    fasta, vcf, names = write_random_cohort(str(tmp_path), 1733, n_samples=40, n_rows=220, ref_len=4000, p_near=0.4)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(1733)
    for n in (4095, 4096, 4097, 8192, 12289, 16384, 16385):
        regions = np.array(random_regions(rng, vs.info().ref_length, n, max_len=300), dtype=np.uint64)[:n]   # (the helper adds edge cases)
        assert len(regions) == n
        ids = rng.integers(1, vs.info().num_samples, size=n).astype(np.uint32)
        reg_t, ids_t = torch.from_numpy(regions.view(np.int64)).cuda(), torch.from_numpy(ids.view(np.int32)).cuda()
        dreg, dids = DeviceArray(reg_t.data_ptr(), n), DeviceArray(ids_t.data_ptr(), n)
        cut = n // 3
        for call in (vs.get_sample_var_in_ref, vs.get_sample_var_in_sample):
            whole, host = call(dreg, dids), call(regions, ids)
            parts = [call(regions[:cut], ids[:cut]), call(regions[cut:], ids[cut:])]
            assert whole.totals() == host.totals() == tuple(sum(p.totals()[k] for p in parts) for k in range(4)), n
            assert whole.digest() == host.digest()
            if call == vs.get_sample_var_in_ref:
                for q in (0, cut - 1, cut, 4094, n - 1):
                    assert whole.region_text(q) == orc.get_sample_var_in_ref(int(regions[q, 0]), int(regions[q, 1]), vs.sample_name(int(ids[q])))[2], (n, q)
            for r in [whole, host] + parts:
                r.close()
        for coords in (False, True):
            whole, host = vs.query_sample_seq(dreg, dids, sample_coordinates=coords), vs.query_sample_seq(regions, ids, sample_coordinates=coords)
            (fa, sa), (fb, sb) = whole.sequences(), host.sequences()
            assert np.array_equal(fa, fb) and sa == sb and whole.totals() == host.totals(), (n, coords)
            whole.close(); host.close()
    vs.close()


def test_walking_batches_report_their_phases_on_request(tmp_path):
    """Option phase_events (variantstore_hip.h): a batch of query type 4 / 5 records its first and last event only -- ms_total, phases 0 --
    unless the handle is asked for all five; the answers do not depend on it."""
    fasta, vcf, names = write_random_cohort(str(tmp_path), 611, n_samples=30, n_rows=200, ref_len=4000, p_near=0.4)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(611)
    regions = np.array(random_regions(rng, vs.info().ref_length, 300, max_len=500), dtype=np.uint64)
    ids = rng.integers(1, vs.info().num_samples, size=len(regions)).astype(np.uint32)
    seen = {}
    for on in (0, 1, 0):
        vs.set_option("phase_events", on)
        for _k in range(2):   # (the first batch of a handle sizes its scratch exactly and waits twice; the second is the steady state)
            r = vs.get_sample_var_in_ref(regions, ids)
        t = vs.last_timing()
        assert t.ms_total > 0
        if on:
            assert t.ms_bounds > 0 and t.ms_fill > 0 and abs(t.ms_bounds + t.ms_scan + t.ms_emit + t.ms_fill - t.ms_total) < 0.05 * t.ms_total + 0.01
        else:
            assert t.ms_bounds == t.ms_scan == t.ms_emit == t.ms_fill == 0
        seen[on] = (r.totals(), r.digest())
        r.close()
    assert seen[0] == seen[1]
    vs.close()


def test_walking_queries_take_regions_and_sample_ids_in_device_memory(tmp_path):
    """Query types 4 (one sample per region), 2, 3 and 5 with their inputs already on the GPU (variantstore_hip.h:
    vs_query_samples_var_in_ref): the same answers as from host arrays, in both walk forms; an id out of range in a device
    array is refused by the batch itself."""
    import torch
    from variantstore_amd import DeviceArray, VariantStoreError
    fasta, vcf, names = write_random_cohort(str(tmp_path), 415, n_samples=70, n_rows=300, ref_len=5000, p_near=0.5)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(415)
    regions = np.array(random_regions(rng, vs.info().ref_length, 200, max_len=900), dtype=np.uint64)
    ids = rng.integers(1, vs.info().num_samples, size=len(regions)).astype(np.uint32)
    reg_t = torch.from_numpy(regions.view(np.int64)).cuda()
    ids_t = torch.from_numpy(ids.view(np.int32)).cuda()
    dreg, dids = DeviceArray(reg_t.data_ptr(), len(regions)), DeviceArray(ids_t.data_ptr(), len(ids))
    for form in (2, 1):
        vs.set_option("t4_walk", form)
        for call in (lambda r, i: vs.get_sample_var_in_ref(r, i), lambda r, i: vs.get_sample_var_in_sample(r, i)):
            a, b, c = call(regions, ids), call(dreg, dids), call(dreg, ids)
            assert a.totals() == b.totals() == c.totals() and a.digest() == b.digest() == c.digest()
            assert [a.region_text(q) for q in range(0, 200, 7)] == [b.region_text(q) for q in range(0, 200, 7)]
            a.close(); b.close(); c.close()
        for coords in (False, True):
            a, b = vs.query_sample_seq(regions, ids, sample_coordinates=coords), vs.query_sample_seq(dreg, dids, sample_coordinates=coords)
            (fa, sa), (fb, sb) = a.sequences(), b.sequences()
            assert np.array_equal(fa, fb) and sa == sb and a.totals() == b.totals()
            a.close(); b.close()
    vs.set_option("t4_walk", 2)
    bad = ids.copy()
    bad[137] = vs.info().num_samples
    bad_t = torch.from_numpy(bad.view(np.int32)).cuda()
    dbad = DeviceArray(bad_t.data_ptr(), len(bad))
    for call in (lambda: vs.get_sample_var_in_ref(dreg, dbad), lambda: vs.get_sample_var_in_sample(dreg, dbad),
                 lambda: vs.query_sample_seq(dreg, dbad), lambda: vs.get_sample_var_in_ref(regions, bad)):
        with pytest.raises(VariantStoreError) as e:
            call()
        assert e.value.code == -6, e.value   # VS_ERR_UNKNOWN_SAMPLE
    r = vs.get_sample_var_in_ref(dreg, dids)     # the handle is fine afterwards
    assert r.region_text(3) == orc.get_sample_var_in_ref(int(regions[3, 0]), int(regions[3, 1]), vs.sample_name(int(ids[3])))[2]
    r.close()


def test_walking_batches_that_outgrow_the_previous_batch_are_redone(tmp_path):
    """Round 5: a walking batch waits for the host once -- its recording walk's scratch (types 4, 5) or piece list (types 2, 3) is
    sized from the handle's PREVIOUS batch of the kind, and a batch that needs more is refused on the device (k_walk_admit) and
    redone with the exact size.  A small batch, then one several times its size (refused and redone), then the same again (fits):
    the same answers every time, and the oracle's."""
    fasta, vcf, names = write_random_cohort(str(tmp_path), 416, n_samples=50, n_rows=350, ref_len=6000, p_near=0.5, carrier_p=0.3)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(416)
    big = np.array(sorted(random_regions(rng, vs.info().ref_length, 400, max_len=1500)), dtype=np.uint64)
    small = big[:30].copy()
    small[:, 1] = small[:, 0] + 20
    ids_big = rng.integers(1, vs.info().num_samples, size=len(big)).astype(np.uint32)
    for name, call in (("4", lambda r, i: vs.get_sample_var_in_ref(r, i)), ("5", lambda r, i: vs.get_sample_var_in_sample(r, i))):
        call(small, ids_big[:30]).close()                       # (sets the handle's expectation: a few rows)
        a = call(big, ids_big)                                   # outgrows it: refused, redone
        b = call(big, ids_big)                                   # fits
        assert a.totals() == b.totals() and a.digest() == b.digest(), name
        for q in range(0, len(big), 9):
            smp = vs.sample_name(int(ids_big[q]))
            want = (orc.get_sample_var_in_ref if name == "4" else orc.get_sample_var_in_sample)(int(big[q, 0]), int(big[q, 1]), smp)
            if want[0] >= 0:
                assert a.region_text(q) == want[-1], (name, q)   # ((n, early, text) for type 4, (n, text) for type 5)
        a.close(); b.close()
    for coords in (False, True):
        vs.query_sample_seq(small, ids_big[:30], sample_coordinates=coords).close()
        a = vs.query_sample_seq(big, ids_big, sample_coordinates=coords)
        b = vs.query_sample_seq(big, ids_big, sample_coordinates=coords)
        (fa, sa), (fb, sb) = a.sequences(), b.sequences()
        assert np.array_equal(fa, fb) and sa == sb
        for q in range(0, len(big), 9):
            smp = vs.sample_name(int(ids_big[q]))
            n, seq = (orc.query_sample_from_sample if coords else orc.query_sample_from_ref)(int(big[q, 0]), int(big[q, 1]), smp)
            if n >= 0:
                assert sa[q] == seq, (coords, q)
        a.close(); b.close()


def test_sample_coordinate_queries_synthetic_midsize(tmp_path):
    """20k variants x 200 samples (bit-vector classes), 300 regions of 5 kb, three samples: types 2, 3 and 5."""
    vs = VariantStore.synthetic(device=0, ref_length=2_000_000, num_variants=20_000, num_samples=200, seed=23,
                                first_pos=500, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=6,
                                af_exponent=3.0, sample_coordinates=True)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(6)
    starts = rng.integers(1, 1_990_000, size=300)
    regions = [(int(s), int(s) + 5000) for s in starts]
    for smp in ("S00001", "S00100", "S00200"):
        for coords in (False, True):
            res = vs.query_sample_seq(regions, smp, sample_coordinates=coords)
            flags, seqs = res.sequences()
            good = 0
            for q, (x, y) in enumerate(regions):
                n, seq = (orc.query_sample_from_sample if coords else orc.query_sample_from_ref)(x, y, smp)
                if n == -3:   # pos_x inside a segment this sample deletes: substr throws in the reference
                    assert flags[q] & 2, (smp, coords, x, y)
                    continue
                if n == -1:   # sample_pos of the start vertex >= pos_x: the reference's backward search repeats itself
                    assert flags[q] & 8, (smp, coords, x, y)
                    continue
                assert n >= 0 and not flags[q] and seqs[q] == seq, (smp, coords, x, y)
                good += 1
            assert good > 270
            res.close()
        res = vs.get_sample_var_in_sample(regions, smp)
        rflags = res.view(False)["region_flags"]
        nv = 0
        for q, (x, y) in enumerate(regions):
            n, text = orc.get_sample_var_in_sample(x, y, smp)
            if n == -1:
                assert rflags[q] & 8, (smp, x, y)
                continue
            assert n >= 0 and res.region_text(q) == text, (smp, x, y)
            nv += n
        assert nv > 300
        res.close()
    # the sample "ref" read in reference coordinates is the reference itself: pieces concatenate
    res = vs.query_sample_seq([(1000, 9000), (1000, 5000), (5000, 9000)], "ref")
    _, seqs = res.sequences()
    assert len(seqs[0]) == 8000 and seqs[0] == seqs[1] + seqs[2]


def test_sample_coordinate_queries_need_the_indexes(tmp_path):
    vs = VariantStore.synthetic(device=0, ref_length=100_000, num_variants=500, num_samples=50, seed=3)
    with pytest.raises(Exception, match="sample coordinates"):
        vs.query_sample_seq([(100, 200)], "S00001")
    with pytest.raises(Exception, match="sample coordinates"):
        vs.get_sample_var_in_sample([(100, 200)], "S00001")


@pytest.mark.parametrize("rows", ["exact", "coarse"])
def test_tcga_shaped_cohort_mixed_types(rows, tmp_path, monkeypatch):
    """BASELINE config #5's shape at reduced scale: 10,000 samples, somatic-like sparse carriers (explicit sample ids,
    not class bit vectors), 10 % indels; mixed query types 3 / 6 / 7 / 4 / 5 (the code's numbering) against the oracle.
    Both forms of the per-sample rows of an explicit-id cohort: COARSE (round 4, the default: a bit per eight slots, hold tests from the
    carrier lists) and EXACT (round 6, VS_T4_EXACT_ROWS: a bit per slot and sample and a hold row from the carrier records, when they fit
    the HBM budget -- the walks then take the class-row cohort's kernels)."""
    if rows == "exact":
        monkeypatch.setenv("VS_T4_EXACT_ROWS", "1")
    else:
        monkeypatch.delenv("VS_T4_EXACT_ROWS", raising=False)
    vs = VariantStore.synthetic(device=0, ref_length=3_000_000, num_variants=30_000, num_samples=10_000, seed=55,
                                first_pos=500, frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6,
                                af_exponent=2.0, max_af=0.0004, sample_coordinates=True)
    info = vs.info()
    assert not info.use_bit_vector and info.num_samples == 10_001
    # (exact rows: 10,001 samples x (slots + vertices) bits; coarse: an eighth of the slots' part and no hold rows)
    exact_bytes = 10_001 * ((info.ref_path_nodes + 63) // 64 + 1 + (info.num_vertices + 63) // 64 + 1) * 8
    assert (info.t4_rows_bytes == exact_bytes) == (rows == "exact"), (info.t4_rows_bytes, exact_bytes)
    plain = os.path.join(tmp_path, "p.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rng = np.random.default_rng(55)
    starts = rng.integers(1, 2_990_000, size=400)
    regions = [(int(s), int(s) + 8000) for s in starts]
    assert _compare_t6(vs, orc, regions) == 400
    # type 7 for every variant type 6 reports in the first regions (and a miss beside each)
    rows = []
    for x, y in regions[:60]:
        rows += _parse_rows(orc.get_var_in_ref(x, y)[2])
    qs = [(p, r, a) for p, r, a in rows] + [(p + 1, r, a) for p, r, a in rows]
    res = vs.samples_has_var([q[0] for q in qs], [q[1] for q in qs], [q[2] for q in qs])
    fl = res.view(False)["region_flags"]
    hits = 0
    for q, (p, r, a) in enumerate(qs):
        want = orc.samples_has_var(p, r, a)
        assert (want is None) == bool(fl[q] & 4) and (want is None or res.region_text(q) == want), qs[q]
        hits += want is not None
    assert hits > 5
    # type 3 (a sample's sequence in its own coordinates) for carriers of some variants and for "ref"
    carriers = sorted({s.split("(")[0] for x, y in regions[:40] for line in orc.get_var_in_ref(x, y)[2].split("\n")[1:-1]
                       for s in line.split("\t")[3].split()})[:6]
    assert len(carriers) >= 3
    for smp in ["ref"] + carriers:
        sub = regions[:80]
        rs = vs.query_sample_seq(sub, smp, sample_coordinates=True)
        fl3, seqs = rs.sequences()
        good = 0
        for q, (x, y) in enumerate(sub):
            n, seq = orc.query_sample_from_sample(x, y, smp)
            if n == -1:
                assert fl3[q] & 8
            elif n == -3:
                assert fl3[q] & 2
            else:
                assert not fl3[q] and seqs[q] == seq, (smp, x, y)
                good += 1
        assert good > 60
        rs.close()
    # types 4 and 5 with one carrier sample per region (explicit sample ids; a carrier has a variant every ~100 kb here:
    # the backward searches run for thousands of ranks -- the hop phase of the cooperative and of the one-lane search)
    sub = sorted(regions[:240])
    per = [carriers[i % len(carriers)] for i in range(len(sub))]
    want4 = [orc.get_sample_var_in_ref(x, y, sm) for (x, y), sm in zip(sub, per)]
    for walk in (2, 1):
        vs.set_option("t4_walk", walk)
        r4 = vs.get_sample_var_in_ref(sub, per)
        for q, (n, _, text) in enumerate(want4):
            if n >= 0:
                assert r4.region_text(q) == text, (walk, q, sub[q], per[q])
        r4.close()
    vs.set_option("t4_walk", 2)
    r5 = vs.get_sample_var_in_sample(sub, per)
    f5 = r5.view(False)["region_flags"]
    for q, ((x, y), sm) in enumerate(zip(sub, per)):
        n, text = orc.get_sample_var_in_sample(x, y, sm)
        if n == -1:
            assert f5[q] & 8
        else:
            assert r5.region_text(q) == text, (q, x, y, sm)


def test_saved_and_reloaded_index_answers_identically(tmp_path):
    """vs_index_save -> vs_index_open (sdsl vectors, gzip-framed protobuf vertex blocks, CQF, sampleid_map): the
    reloaded handle gives the same digests for all result kinds as the one that built the graph in memory."""
    vs = VariantStore.synthetic(device=0, ref_length=2_000_000, num_variants=20_000, num_samples=200, seed=29,
                                first_pos=500, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=6,
                                af_exponent=3.0, sample_coordinates=True)
    d = os.path.join(tmp_path, "ser")
    os.makedirs(d)
    vs.save(d)
    vs2 = VariantStore.open(d, device=0)
    ia, ib = vs.info(), vs2.info()
    assert (ia.num_vertices, ia.num_sites, ia.num_carriers, ia.num_classes, ia.num_samples) == (
        ib.num_vertices, ib.num_sites, ib.num_carriers, ib.num_classes, ib.num_samples)
    rng = np.random.default_rng(29)
    starts = rng.integers(1, 1_990_000, size=3000)
    regions = [(int(s), int(s) + 5000) for s in starts]
    a, b = vs.get_var_in_ref(regions), vs2.get_var_in_ref(regions)
    assert a.digest() == b.digest() and a.totals() == b.totals() and a.totals()[1] > 100_000
    for q in (0, 1234, 2999):
        assert a.region_text(q) == b.region_text(q)
    a4, b4 = vs.get_sample_var_in_ref(regions, "S00077"), vs2.get_sample_var_in_ref(regions, "S00077")
    assert a4.digest() == b4.digest() and a4.totals() == b4.totals()
    a5, b5 = vs.get_sample_var_in_sample(regions[:500], "S00077"), vs2.get_sample_var_in_sample(regions[:500], "S00077")
    assert a5.digest() == b5.digest()
    fa, sa = vs.query_sample_seq(regions[:300], "S00150", sample_coordinates=True).sequences()
    fb, sb = vs2.query_sample_seq(regions[:300], "S00150", sample_coordinates=True).sequences()
    assert list(fa) == list(fb) and sa == sb


def test_positions_far_beyond_the_reference(tmp_path):
    """64-bit region bounds far past the reference end (and past 2^32) behave as in the oracle for types 6, 4 and 1."""
    fasta, vcf, names = write_random_cohort(str(tmp_path), 77)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    L = vs.info().ref_length
    regions = [(2 ** 40, 2 ** 40 + 10), (5, 2 ** 40), (L - 5, 2 ** 33), (2 ** 62, 2 ** 62 + 5), (1, 2 ** 63)]
    assert _compare_t6(vs, orc, regions) == len(regions)
    assert _compare_t4(vs, orc, regions, names[0]) == len(regions)
    res = vs.closest_var([2 ** 32 + 7, 2 ** 40 + 3])
    flags = res.view(False)["region_flags"]
    for q, p in enumerate([2 ** 32 + 7, 2 ** 40 + 3]):
        n, text = orc.closest_var(p)
        assert (n < 0) == bool(flags[q] & 4) and (n < 0 or res.region_text(q) == text), p


def test_barely_overlapping_regions_share_rows_all_the_same(tmp_path):
    """Round 4: the plan of a shared batch is three launches -- fewer than the private form takes -- so every batch of
    more than 64 regions shares its rows and lists, whether or not its regions overlap (rounds 2-3 learnt per handle
    whether sharing paid: a batch's path then depended on the batches before it).  `share_lists` 0 gives private rows."""
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 611, n_rows=600, ref_len=30000, n_samples=20)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    L = vs.info().ref_length
    regions = [(x, x + 260) for x in range(1, L - 300, 250)]           # neighbours share 10 of 260 bases
    assert len(regions) > 64
    first = vs.get_var_in_ref(regions)
    second = vs.get_var_in_ref(regions)
    assert first.layout()[4] and second.layout()[4] and first.layout() == second.layout()
    assert first.digest() == second.digest() and first.totals() == second.totals()
    for q, (x, y) in enumerate(regions):
        n, _, text = orc.get_var_in_ref(x, y)
        if n >= 0:
            assert first.region_text(q) == text and second.region_text(q) == text, (x, y)
    vs.set_option("share_lists", 0)
    private = vs.get_var_in_ref(regions)
    assert not private.layout()[4] and private.digest() == first.digest() and private.totals() == first.totals()
    vs.set_option("share_lists", 1)
    wide = sorted((x, x + 2000) for x in range(1, L - 2100, 150))       # 13 regions over every base
    a, b = vs.get_var_in_ref(wide), vs.get_var_in_ref(wide)
    assert a.layout()[4] and b.layout()[4] and a.digest() == b.digest()
    for q, (x, y) in enumerate(wide):
        n, _, text = orc.get_var_in_ref(x, y)
        if n >= 0:
            assert a.region_text(q) == text, (x, y)


@pytest.mark.parametrize("seed,kw", [
    (601, dict(n_rows=400, ref_len=5000, n_samples=70, carrier_p=0.4)),                      # class rows wider than one word
    (602, dict(n_rows=400, ref_len=4000, p_near=0.7, p_multi=0.3, p_same=0.25)),              # crowded: the duplicate rule fires
    (603, dict(n_rows=300, ref_len=5000, n_samples=130, carrier_p=0.004)),                    # explicit sample ids
    (604, dict(n_rows=300, ref_len=6000, n_samples=900, carrier_p=0.5, p_ins=0.2, p_del=0.2)),  # dense rows: the row path
])
def test_shared_carrier_lists_of_a_sorted_batch(seed, kw, tmp_path):
    """A sorted batch of overlapping regions expands every covered site once and lets the regions share the list
    (kernels.hip.h: k_share_*): same text as the oracle, same digest and view as the private-list form; an unsorted
    batch takes the private form by itself."""
    fasta, vcf, _ = write_random_cohort(str(tmp_path), seed, **kw)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(seed)
    ref_len = vs.info().ref_length
    regions = sorted(random_regions(rng, ref_len, 500, max_len=900))
    shared = vs.get_var_in_ref(regions)
    slots, table, arena, lists, is_shared = shared.layout()
    assert is_shared and lists < slots / 2 and arena > 0 and lists <= table   # (regions under the duplicate rule keep private rows: table = shared + private)
    want = oracle_texts(orc, regions)
    for q, (n, early, text) in enumerate(want):
        if n >= 0:
            assert shared.region_text(q) == text, (q, regions[q])
    vs.set_option("share_lists", 0)
    private = vs.get_var_in_ref(regions)
    vs.set_option("share_lists", 1)
    p_slots, p_table, p_arena, p_lists, p_shared = private.layout()
    assert not p_shared and p_lists == p_table == p_slots == slots and p_arena > 2 * arena
    assert private.digest() == shared.digest() and private.totals() == shared.totals()
    va, vb = shared.view(True), private.view(True)
    for k in va:
        assert np.array_equal(va[k], vb[k]), k
    # not sorted: the engine sorts the regions by first site on the device (k_sort_*), shares rows and lists all the
    # same and hands every region's outcome back in the caller's order -- twice (the second batch is sorted first, on the
    # handle's hint), then a sorted batch again
    perm = rng.permutation(len(regions))
    for _round in range(2):
        mixed = vs.get_var_in_ref([regions[i] for i in perm])
        assert mixed.layout()[4] and mixed.layout()[:4] == shared.layout()[:4]
        assert mixed.digest() != 0 and mixed.totals() == shared.totals()
        for j in range(len(regions)):
            assert mixed.region_text(j) == shared.region_text(int(perm[j])), j
        vm = mixed.view(True)
        assert np.array_equal(vm["var_count"], va["var_count"][perm]) and np.array_equal(vm["region_flags"], va["region_flags"][perm])
        mixed.close()
    again = vs.get_var_in_ref(regions)
    assert again.layout() == shared.layout() and again.digest() == shared.digest()
    # duplicates of one region, regions without sites in between, one region swallowing many others
    odd = sorted([(1, ref_len)] * 3 + regions[::7] + [(ref_len + 5, ref_len + 9), (0, 5)] + [(40, 41)] * 70)
    a = vs.get_var_in_ref(odd)
    assert a.layout()[4]
    for q, (x, y) in enumerate(odd):
        n, _, text = orc.get_var_in_ref(x, y) if x >= 1 else (-1, None, None)
        if n >= 0:
            assert a.region_text(q) == text, (q, x, y)


def test_raw_copy_and_streamed_delivery(tmp_path):
    """vs_result_get_raw hands the result over as it lies in HBM (variant table + arena, page-locked, no repacking);
    vs_query_var_in_ref_stream delivers a sorted batch chunk by chunk while the next chunk is computed.  Both decode to
    the rows of the structure-of-arrays view and to the oracle's text."""
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 707, n_rows=400, ref_len=6000, n_samples=90, carrier_p=0.35,
                                        p_near=0.5, p_multi=0.2, p_same=0.2)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    names = [vs.sample_name(i) for i in range(vs.info().num_samples)]
    regions = sorted(random_regions(np.random.default_rng(8), vs.info().ref_length, 600, max_len=700))
    want = [orc.get_var_in_ref(x, y) for x, y in regions]

    def decode(raw, k, seq_pool):
        """text of region k of a raw view (python dict or ctypes struct fields as arrays)"""
        rows = raw["rows"][int(raw["row_begin"][k]):int(raw["row_begin"][k]) + int(raw["row_count"][k])]
        out = ["Pos\tRef\tAlt\tSamples\n"]
        for v in rows:
            if v["count_flags"] >> 31:
                continue
            cnt, cb = int(v["count_flags"]) & 0x7FFFFFFF, int(v["car_begin"])
            cars = raw["arena"][cb:cb + cnt].astype(np.uint32)
            ids, gts = (cars & 0x1FFF, cars >> 13) if raw["carrier_bytes"] == 2 else (cars & 0x1FFFFFFF, cars >> 29)
            ref = seq_pool[int(v["ref_off"]):int(v["ref_off"]) + int(v["ref_len"])]
            alt = seq_pool[int(v["alt_off"]):int(v["alt_off"]) + int(v["alt_len"])]
            samples = "".join(f"{names[i]}({(g >> 1) & 1}{'|' if g & 1 else '/'}{(g >> 2) & 1}) " for i, g in zip(ids, gts))
            out.append(f"{int(v['pos'])}\t{ref}\t{alt}\t{samples}\n")
        return "".join(out)

    res = vs.get_var_in_ref(regions)
    assert res.layout()[4]
    raw = res.raw(with_carriers=True)
    assert raw["shared"] and raw["carrier_bytes"] == 2 and len(raw["rows"]) == res.layout()[1]
    # the sequence pool: decode REF/ALT through the texts of the view for a few regions instead of exposing it twice
    import ctypes as C
    from variantstore_amd._lib import ResultRaw
    rr = ResultRaw()
    assert vs._lib.vs_result_get_raw(res._h, 1, C.byref(rr)) == 0
    rws = raw["rows"]
    pool_len = int(max((rws["ref_off"].astype(np.int64) + rws["ref_len"]).max(), (rws["alt_off"].astype(np.int64) + rws["alt_len"]).max()))
    pool = C.string_at(rr.seq_pool, pool_len).decode("latin-1")
    for k in range(0, len(regions), 7):
        if want[k][0] >= 0:
            assert decode(raw, k, pool) == want[k][2] == res.region_text(k), k
    assert np.array_equal(raw["var_count"], res.view(False)["var_count"])
    # streamed: chunks of 100 regions, each decoded inside its callback
    seen = []

    def on_chunk(first, chunk):
        q, a, s = int(chunk.n_regions), int(chunk.n_rows), int(chunk.arena_entries)
        view = {"row_begin": np.ctypeslib.as_array(chunk.row_begin, shape=(q,)), "row_count": np.ctypeslib.as_array(chunk.row_count, shape=(q,)),
                "rows": np.ctypeslib.as_array(C.cast(chunk.rows, C.POINTER(C.c_uint8)), shape=(a * 32,)).view(res.ROW_DTYPE),
                "arena": np.ctypeslib.as_array(C.cast(chunk.arena, C.POINTER(C.c_uint16)), shape=(max(s, 1),)), "carrier_bytes": int(chunk.carrier_bytes)}
        for k in range(q):
            if want[first + k][0] >= 0:
                assert decode(view, k, pool) == want[first + k][2], first + k
        seen.append((first, q))

    vs.stream_var_in_ref(regions, 100, on_chunk)
    assert seen == [(i, min(100, len(regions) - i)) for i in range(0, len(regions), 100)]
    seen.clear()
    vs.stream_var_in_ref(regions[:250], 100, on_chunk)
    assert seen == [(0, 100), (100, 100), (200, 50)]


def test_type4_regions_that_outgrow_their_scratch_capacity(tmp_path):
    """The recording walk of type 4 gets the region's type-6 row count as scratch capacity; tiny regions report more than
    that (their head episode lies in front of x's node), the batch then takes the two-walk path -- and the list claims of
    the shared form must not read what the overflowing walk did not record (found by the round-3 stress campaign)."""
    fasta, vcf, names = write_random_cohort(str(tmp_path), 20027, ref_len=3352, n_rows=186, n_samples=130, p_ins=0.08, p_del=0.04,
                                            p_multi=0.16, p_near=0.34, carrier_p=0.3)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(20027)
    L = vs.info().ref_length
    regions = [(int(x), int(x) + int(rng.integers(0, 6))) for x in rng.integers(1, L, size=300)]
    per = [names[int(i)] for i in rng.integers(0, len(names), size=len(regions))]
    for walk in (2, 1):
        vs.set_option("t4_walk", walk)
        res = vs.get_sample_var_in_ref(regions, per)
        for q, (x, y) in enumerate(regions):
            n, _, text = orc.get_sample_var_in_ref(x, y, per[q])
            if n >= 0:
                assert res.region_text(q) == text, (walk, q, x, y, per[q])
        res.close()
    vs.set_option("t4_walk", 2)
    # and a batch of ordinary regions right afterwards (the claims' generation moves on)
    regions = sorted(random_regions(rng, L, 200, max_len=700))
    per = [names[int(i)] for i in rng.integers(0, len(names), size=len(regions))]
    res = vs.get_sample_var_in_ref(regions, per)
    assert res.layout()[4]
    for q, (x, y) in enumerate(regions):
        n, _, text = orc.get_sample_var_in_ref(x, y, per[q])
        if n >= 0:
            assert res.region_text(q) == text, (q, x, y)


def test_async_fill_results_are_the_same(tmp_path):
    """Option "async_fill": a type-6 batch call returns while its carrier expansion still runs on the engine's second
    stream; every accessor that reads carriers waits by itself.  Several batches in flight, read and freed in any order,
    small (latency-path) queries in between: the oracle's text everywhere, the synchronous digests."""
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 631, n_rows=500, ref_len=8000, n_samples=300, carrier_p=0.3, p_ins=0.1, p_del=0.1)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(631)
    L = vs.info().ref_length
    batches = [sorted(random_regions(rng, L, 300, max_len=1500)), random_regions(rng, L, 200, max_len=900), sorted(random_regions(rng, L, 120, max_len=4000))]
    vs.set_option("async_fill", 0)
    sync = [vs.get_var_in_ref(b) for b in batches]
    want = [(r.digest(), r.totals()) for r in sync]
    vs.set_option("async_fill", 1)
    flight = [vs.get_var_in_ref(b) for b in batches]          # three expansions queued behind each other
    small = vs.get_var_in_ref(batches[0][:5])                   # the latency path does not wait for them
    for q, (x, y) in enumerate(batches[0][:5]):
        n, _, text = orc.get_var_in_ref(x, y)
        if n >= 0:
            assert small.region_text(q) == text
    assert flight[2].fill_ms() >= 0 and sync[2].fill_ms() >= 0   # (round 4: every shared batch carries its own pair of events around its expansion)
    for k in (2, 0, 1):
        assert (flight[k].digest(), flight[k].totals()) == want[k]
        for q in range(0, len(batches[k]), 9):
            n, _, text = orc.get_var_in_ref(*batches[k][q])
            if n >= 0:
                assert flight[k].region_text(q) == text, (k, q)
        va, vb = flight[k].view(True), sync[k].view(True)
        for key in va:
            assert np.array_equal(va[key], vb[key]), (k, key)
    for r in flight:
        r.close()
    # freed while still in flight; then a raw copy straight after the call
    vs.get_var_in_ref(batches[0]).close()
    r = vs.get_var_in_ref(batches[0])
    raw = r.raw(with_carriers=True)
    assert raw["rows"].shape[0] == sync[0].layout()[1] and int(raw["arena"].astype(np.uint64).sum()) == int(sync[0].raw(True)["arena"].astype(np.uint64).sum())
    vs.set_option("async_fill", 0)
    assert vs.get_var_in_ref(batches[0]).fill_ms() >= 0   # (every shared batch times its own expansion)
    vs.set_option("share_lists", 0)
    assert vs.get_var_in_ref(batches[0]).fill_ms() == -1  # (private rows: the handle's events, vs_index_last_timing)
    vs.set_option("share_lists", 1)


@pytest.mark.parametrize("seed,n_samples,carrier_p", [(641, 40, 0.002), (642, 150, 0.004), (643, 9, 0.01)])
def test_type4_long_backward_searches(seed, n_samples, carrier_p, tmp_path):
    """Samples with a handful of variants on a reference with thousands of nodes: get_prev_vertex_with_sample runs for
    hundreds of ranks, so the cooperative walk's search leaves its rank windows for the set bits of the sample's event
    row (k_sample_walk_coop: hop) -- every walk form against the oracle, and the same for types 2, 3 and 5."""
    fasta, vcf, names = write_random_cohort(str(tmp_path), seed, n_rows=1500, ref_len=24000, n_samples=n_samples, carrier_p=carrier_p,
                                            p_ins=0.1, p_del=0.1, p_multi=0.1)
    vs, orc = _open_gpu(fasta, vcf, tmp_path)
    rng = np.random.default_rng(seed)
    L = vs.info().ref_length
    regions = sorted(random_regions(rng, L, 260, max_len=1200))
    per = [names[int(i)] for i in rng.integers(0, len(names), size=len(regions))]
    want = [orc.get_sample_var_in_ref(x, y, sm) for (x, y), sm in zip(regions, per)]
    assert sum(1 for n, _, _ in want if n >= 0) > 200
    for walk in (2, 1, 0):                        # cooperative, one lane per region with jumps, literal
        vs.set_option("t4_walk", walk)
        res = vs.get_sample_var_in_ref(regions, per)
        for q, (n, _, text) in enumerate(want):
            if n >= 0:
                assert res.region_text(q) == text, (walk, q, regions[q], per[q])
        res.close()
    vs.set_option("t4_walk", 2)
    r5 = vs.get_sample_var_in_sample(regions, per)
    f5 = r5.view(False)["region_flags"]
    for q, ((x, y), sm) in enumerate(zip(regions, per)):
        n, text = orc.get_sample_var_in_sample(x, y, sm)
        if n == -1:
            assert f5[q] & 8
        else:
            assert r5.region_text(q) == text, (q, x, y, sm)
    for coords in (False, True):
        flags, seqs = vs.query_sample_seq(regions, per, sample_coordinates=coords).sequences()
        for q, ((x, y), sm) in enumerate(zip(regions, per)):
            n, seq = (orc.query_sample_from_sample if coords else orc.query_sample_from_ref)(x, y, sm)
            if n == -1:
                assert flags[q] & 8, (coords, q)
            elif n == -3:
                assert flags[q] & 2, (coords, q)
            else:
                assert not flags[q] and seqs[q] == seq, (coords, q, x, y, sm)
