"""Constructor evidence that does not pass through the product: expected type-6 rows and type-2 sequences are derived
in Python from the VCF / FASTA TEXT (tests/vcf_truth.py) and compared with

  * the CPU oracle reading the graph the product's constructor built   (this file's CPU tests), and
  * the HIP path on the same cohorts                                    (the -m gpu tests below).

A constructor fault on an isolated SNP / MNP / insertion / deletion -- wrong vertex, wrong carrier set, wrong genotype
bits, wrong sequence offsets -- is invisible to oracle-vs-GPU parity (both read the same graph) and visible here."""
import os
import random

import pytest

import vcf_truth as vt
from helpers import write_random_cohort
from oracle.oracle import Oracle
from variantstore_amd import VariantStore

COHORTS = [
    (200, dict(n_samples=5, p_near=0.0, carrier_p=0.3)),
    (201, dict(n_samples=40, p_near=0.3, carrier_p=0.3)),
    (202, dict(n_samples=70, p_near=0.0, carrier_p=0.02)),      # class rows wider than one word
    (203, dict(n_samples=130, p_near=0.3, carrier_p=0.02)),
    (204, dict(n_samples=130, p_near=0.0, carrier_p=0.004)),    # sparse: explicit sample ids
]


def _cohort(tmp_path, seed, kw):
    fasta, vcf, _ = write_random_cohort(str(tmp_path), seed, ref_len=6000, n_rows=150, p_multi=0.05, p_mnp=0.08,
                                        unphased_p=0.2, missing_p=0.05, haploid_p=0.03, **kw)
    _name, ref = vt.read_fasta(fasta)
    names, recs = vt.read_vcf(vcf)
    assert names == sorted(names)          # (sample ids then follow the columns: no name-order permutation in play)
    return fasta, vcf, ref, names, recs


def _sequence_cases(ref, names, recs, seed, n=250):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        x = rng.randint(1, len(ref) - 50)
        y = min(len(ref), x + rng.randint(1, 300))
        smp = rng.choice(names)
        want = vt.sample_sequence(ref, names, recs, smp, x, y)
        if want is not None:
            out.append((x, y, smp, want))
    return out


def test_x_vcf_rows_follow_from_the_vcf_text(golden_dir, survey_vectors, tmp_path):
    """The reference's own data set: all 75 records of x.vcf are simple and their rows follow from the VCF lines; the
    count and the fragments the reference publishes (G2, README) are a subset of that."""
    fasta, vcf = os.path.join(golden_dir, "x.fa"), os.path.join(golden_dir, "x.vcf")
    _n, ref = vt.read_fasta(fasta)
    names, recs = vt.read_vcf(vcf)
    want = "Pos\tRef\tAlt\tSamples\n" + "".join(t for _i, t in vt.expected_type6_rows(names, recs, only=range(len(recs))))
    assert want.count("\n") == 76
    g2 = survey_vectors["G2"]
    assert want.split("\n")[-2] == g2["last_row"] and all(any(l.startswith(f) for l in want.split("\n")) for f in g2["contains"])
    vs = VariantStore.from_vcf(fasta, vcf, device=-1)
    plain = os.path.join(tmp_path, "x.plain")
    vs.export_plain(plain)
    orc = Oracle(plain)
    assert orc.get_var_in_ref(1, len(ref) + 1)[2] == want
    g1 = survey_vectors["G1"]
    in_g1 = [t for _i, t in vt.expected_type6_rows(names, recs, only=range(len(recs))) if g1["region"][0] <= int(t.split("\t")[0]) < g1["region"][1]]
    assert "Pos\tRef\tAlt\tSamples\n" + "".join(in_g1) == g1["text"]      # the reference's published 8 rows, from the VCF alone


@pytest.mark.parametrize("seed,kw", COHORTS)
def test_oracle_on_the_constructed_graph_matches_the_vcf_text(seed, kw, tmp_path):
    fasta, vcf, ref, names, recs = _cohort(tmp_path, seed, kw)
    vs = VariantStore.from_vcf(fasta, vcf, device=-1)
    plain = os.path.join(tmp_path, "c.plain")
    vs.export_plain(plain)
    orc = Oracle(plain)
    rows = set(orc.get_var_in_ref(1, len(ref) + 1)[2].split("\n")[1:-1])
    exp = vt.expected_type6_rows(names, recs)
    assert len(exp) > 40
    for i, text in exp:
        assert text[:-1] in rows, (i, recs[i][:3], text)
        pos = int(text.split("\t")[0])       # and a small region around the row reports it
        assert text in orc.get_var_in_ref(max(1, pos - 3), pos + 4)[2]
    cases = _sequence_cases(ref, names, recs, seed)
    assert len(cases) > 30
    for x, y, smp, want in cases:
        assert orc.query_sample_from_ref(x, y, smp)[1] == want, (x, y, smp)
    if seed == 204:
        assert vs.info().use_bit_vector == 0


@pytest.mark.gpu
def test_gpu_x_vcf_rows_follow_from_the_vcf_text(golden_dir):
    fasta, vcf = os.path.join(golden_dir, "x.fa"), os.path.join(golden_dir, "x.vcf")
    _n, ref = vt.read_fasta(fasta)
    names, recs = vt.read_vcf(vcf)
    want = "Pos\tRef\tAlt\tSamples\n" + "".join(t for _i, t in vt.expected_type6_rows(names, recs, only=range(len(recs))))
    vs = VariantStore.from_vcf(fasta, vcf, device=0)
    res = vs.get_var_in_ref([(1, len(ref) + 1), (10, 105)])
    assert res.region_text(0) == want                       # all 75 rows of G2, row by row, from the VCF text
    assert res.region_text(1).count("\n") == 9              # README: 8 variants for 10:105
    # the sample's whole sequence (type 2) is the reference with its alleles applied
    cases = _sequence_cases(ref, names, recs, 5, n=300)
    assert len(cases) > 25
    rs = vs.query_sample_seq([(x, y) for x, y, _s, _w in cases], [s for _x, _y, s, _w in cases])
    flags, seqs = rs.sequences()
    assert [s for s in seqs] == [w for _x, _y, _s, w in cases] and not flags.any()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", COHORTS)
def test_gpu_matches_the_vcf_text(seed, kw, tmp_path):
    fasta, vcf, ref, names, recs = _cohort(tmp_path, seed, kw)
    vs = VariantStore.from_vcf(fasta, vcf, device=0)
    exp = vt.expected_type6_rows(names, recs)
    regions = [(1, len(ref) + 1)] + [(max(1, int(t.split("\t")[0]) - 3), int(t.split("\t")[0]) + 4) for _i, t in exp]
    res = vs.get_var_in_ref(regions)
    rows = set(res.region_text(0).split("\n")[1:-1])
    for k, (i, text) in enumerate(exp):
        assert text[:-1] in rows, (i, recs[i][:3])
        assert text in res.region_text(k + 1)
    # type 4: a carrier's rows include the record's row (substitutions: type 4 reports indels by its own rule, query.h:682-698);
    # type 2: its sequence is the reference with its alleles applied
    rng = random.Random(seed)
    subs = [(i, t) for i, t in exp if len(recs[i][1]) == len(recs[i][2][0])]
    for i, text in rng.sample(subs, 25):
        carrier = text.split("\t")[3].split("(")[0]
        pos = int(text.split("\t")[0])
        r4 = vs.get_sample_var_in_ref([(max(1, pos - 3), pos + 4)], carrier)
        assert text in r4.region_text(0), (i, carrier)
    cases = _sequence_cases(ref, names, recs, seed)
    rs = vs.query_sample_seq([(x, y) for x, y, _s, _w in cases], [s for _x, _y, s, _w in cases])
    flags, seqs = rs.sequences()
    assert seqs == [w for _x, _y, _s, w in cases] and not flags.any()
