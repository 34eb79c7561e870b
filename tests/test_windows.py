"""The windowing method of the full-size oracle spot-checks (tests/test_gpu_full_size.py, test_zz_gpu_full_size_tcga.py),
checked on the CPU on a cohort the oracle CAN hold whole: a window's records, written as a VCF relative to the window and
built through the product's VCF path, give the oracle the same rows -- shifted by the window's origin -- as the whole
cohort gives it, for regions a margin away from the window's ends.  (The reference's walk is local: a region's rows
depend on the ref-path nodes it covers, whose boundaries are set by the variants around them.)"""
import os

import numpy as np

from helpers import parse_rows, synth_windows, window_oracle
from oracle.oracle import Oracle
from variantstore_amd import VariantStore

KW = dict(ref_length=3_000_000, num_variants=60_000, num_samples=300, seed=9, first_pos=1000, frac_ins=0.05, frac_del=0.05,
          frac_multi=0.02, max_indel=6, af_exponent=2.0)


def test_windowed_reconstruction_matches_the_whole_cohort(tmp_path):
    full = VariantStore.synthetic(device=-1, **KW)
    plain = os.path.join(tmp_path, "full.bin")
    full.export_plain(plain)
    names = [full.sample_name(i) for i in range(1, 6)]
    full.close()
    orc = Oracle(plain)
    rng = np.random.default_rng(1)
    margin = 20_000
    wins = [(c - margin, c + 30_000 + margin) for c in (int(x) for x in rng.integers(200_000, 2_700_000, size=8))]
    counts = synth_windows(KW, wins, tmp_path)
    assert min(counts) > 1000
    checked = rows_seen = 0
    for k, (lo, hi) in enumerate(wins):
        ow = window_oracle(tmp_path, k)
        for j in range(3):
            x = lo + margin + j * 10_000
            y = x + 10_000
            nf, _, tf = orc.get_var_in_ref(x, y)
            nw, _, tw = ow.get_var_in_ref(x - lo + 1, y - lo + 1)
            assert nf == nw and parse_rows(tf) == parse_rows(tw, lo - 1), (k, j, x, y)
            rows_seen += nf
            # type 4 as well: a sample's variants in the region (the walk back to the sample's previous vertex stays inside the margin
            # for a sample that carries something there)
            smp = names[(k + j) % len(names)]
            n4f, _, t4f = orc.get_sample_var_in_ref(x, y, smp)
            n4w, _, t4w = ow.get_sample_var_in_ref(x - lo + 1, y - lo + 1, smp)
            if n4f >= 0 and n4w >= 0:
                assert parse_rows(t4f) == parse_rows(t4w, lo - 1), (k, j, smp)
            checked += 1
        ow.close()
    assert checked == 24 and rows_seen > 3000
    orc.close()
