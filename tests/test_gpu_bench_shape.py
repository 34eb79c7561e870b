"""Oracle parity on the cohort SHAPES the bench runs (BASELINE.json configs #2 and #3), text-exact.

* config #3's shape -- 2504 samples, af_exponent 11 (class rows of 40 words: the `sb = 20` form of the fill kernel's
  row path, decoded lists up to 640 carriers), 5 % / 5 % / 1 % insertions / deletions / two-ALT sites -- on the
  1/25-length slice of the chromosome that bench.py's cpu_baseline leg builds (same generator, same seed): the oracle
  holds this slice in seconds; types 6 and 4 (16 samples round-robin, exactly the bench's type-4 leg), batch pipeline
  and latency path.
* config #2 at FULL size (chr22-100: 100 samples, 100 k SNPs, all 10,000 regions of 1 kb).
The full-size config #3 index is covered through size-independent properties in test_gpu_full_size.py.
"""
import os

import numpy as np
import pytest

import bench
from oracle.oracle import Oracle
from variantstore_amd import VariantStore

pytestmark = pytest.mark.gpu


def _slice_kwargs(w, scale=25):
    kw = bench.synth_kwargs(w)
    kw["ref_length"] = max(200_000, w["ref_length"] // scale)
    kw["num_variants"] = max(1000, w["num_variants"] // scale)
    kw["first_pos"] = min(w["first_pos"], kw["ref_length"] // 10)
    return kw


@pytest.fixture(scope="module")
def chr1_slice(tmp_path_factory):
    w = bench.WORKLOADS["chr1-2504"]
    kw = _slice_kwargs(w)
    vs = VariantStore.synthetic(device=0, **kw)
    plain = os.path.join(tmp_path_factory.mktemp("slice"), "slice.plain")
    vs.export_plain(plain)
    orc = Oracle(plain)
    yield vs, orc, dict(w, **kw)
    orc.close()
    vs.close()


def test_bench_cohort_shape_type6_matches_oracle(chr1_slice):
    vs, orc, sub = chr1_slice
    info = vs.info()
    assert info.num_samples == 2505 and info.use_bit_vector == 1 and info.list_max == 640
    regions = bench.make_regions(sub, 4242, 4000)
    res = vs.get_var_in_ref(regions)
    view = res.view(with_carriers=False)
    dense = int((view["car_count"] > info.list_max).sum())
    assert dense > 1000                      # the row path of k_fill_carriers is exercised, not only the lists
    nvar = 0
    for q, (x, y) in enumerate(regions):
        n, early, text = orc.get_var_in_ref(int(x), int(y))
        assert n >= 0
        assert res.region_text(q) == text, (q, int(x), int(y))
        assert int(view["var_count"][q]) == n and bool(view["region_flags"][q] & 1) == early
        nvar += n
    assert nvar > 700_000 and orc.ub_events() == 0
    res.close()


def test_bench_cohort_shape_latency_path_matches_oracle(chr1_slice):
    """Batches of at most 64 regions (k_query_small / the resident server) on the same cohort."""
    vs, orc, sub = chr1_slice
    regions = bench.make_regions(sub, 99, 400)
    want = [orc.get_var_in_ref(int(x), int(y))[2] for x, y in regions]
    for mode in (0, 2):
        vs.set_option("latency_server", mode)
        at = 0
        for size in (1, 1, 1, 2, 5, 17, 64, 64, 33, 1, 1, 64):
            batch = regions[at:at + size]
            res = vs.get_var_in_ref(batch)
            for i in range(len(batch)):
                assert res.region_text(i) == want[at + i], (mode, at + i)
            res.close()
            at += size
    vs.set_option("latency_server", 1)


@pytest.mark.parametrize("skip", [1, 0])
def test_bench_cohort_shape_type4_matches_oracle(chr1_slice, skip):
    """The bench's type-4 leg: 16 fixed samples round-robin over the sorted regions; with the event-bitmap shortcut
    (the default) and as the literal vertex-by-vertex walk."""
    vs, orc, sub = chr1_slice
    ns = vs.info().num_samples
    sids16 = [1 + (i * 157) % (ns - 1) for i in range(16)]
    names = [vs.sample_name(s) for s in sids16]
    nreg = 4000 if skip else 1500
    regions = bench.make_regions(sub, 777, nreg)
    per_region = np.array([sids16[i % 16] for i in range(nreg)], dtype=np.uint32)
    vs.set_option("t4_walk", 2 if skip else 0)     # the cooperative walk over the event bitmaps / the literal walk
    res = vs.get_sample_var_in_ref(regions, per_region)
    vs.set_option("t4_walk", 2)
    view = res.view(with_carriers=False)
    nvar = 0
    for q, (x, y) in enumerate(regions):
        n, early, text = orc.get_sample_var_in_ref(int(x), int(y), names[q % 16])
        assert n >= 0
        assert res.region_text(q) == text, (q, int(x), int(y), names[q % 16])
        assert int(view["var_count"][q]) == n and bool(view["region_flags"][q] & 1) == early
        nvar += n
    assert nvar > 8 * nreg
    res.close()
    # one sample for the whole batch (the CLI's form) on a piece of it
    one = vs.get_sample_var_in_ref(regions[:300], names[3])
    for q in range(300):
        assert one.region_text(q) == orc.get_sample_var_in_ref(int(regions[q, 0]), int(regions[q, 1]), names[3])[2]
    one.close()


def test_config2_chr22_100_full_size_every_region(tmp_path):
    """BASELINE.json configs[1] as the bench runs it (--workload chr22-100), every one of its 10,000 regions."""
    w = bench.WORKLOADS["chr22-100"]
    vs = VariantStore.synthetic(device=0, **bench.synth_kwargs(w))
    plain = os.path.join(tmp_path, "chr22.plain")
    vs.export_plain(plain)
    orc = Oracle(plain)
    regions = bench.make_regions(w, 0, w["regions"])       # rank 0's batch of the bench
    res = vs.get_var_in_ref(regions)
    view = res.view(with_carriers=False)
    nvar = 0
    for q, (x, y) in enumerate(regions):
        n, early, text = orc.get_var_in_ref(int(x), int(y))
        assert res.region_text(q) == text, (q, int(x), int(y))
        assert int(view["var_count"][q]) == n and bool(view["region_flags"][q] & 1) == early
        nvar += n
    assert nvar == res.totals()[1] > 10_000
    res.close()
    # type 4 on the same index: 16 samples round-robin over the first 3000 regions
    names = [vs.sample_name(1 + (i * 7) % 100) for i in range(16)]
    per_region = [names[i % 16] for i in range(3000)]
    r4 = vs.get_sample_var_in_ref(regions[:3000], per_region)
    for q in range(3000):
        assert r4.region_text(q) == orc.get_sample_var_in_ref(int(regions[q, 0]), int(regions[q, 1]), per_region[q])[2], q
    r4.close()
    orc.close()
    vs.close()
