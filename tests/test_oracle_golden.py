"""The CPU oracle (and the constructor feeding it) against the golden vectors:
the reference README's published outputs and the reference outputs recorded in
SURVEY.md §4.3/§4.4/§7.3.  No GPU needed: handles are opened host-only."""
import os

import pytest

from oracle.oracle import Oracle
from variantstore_amd import VariantStore


def _open(golden_dir, fasta, vcf, tmp_path):
    vs = VariantStore.from_vcf(os.path.join(golden_dir, fasta), os.path.join(golden_dir, vcf), device=-1)
    plain = os.path.join(tmp_path, f"{fasta}.{vcf}.plain")
    vs.export_plain(plain)
    return vs, Oracle(plain)


def test_readme_construct_stats(golden_dir, survey_vectors, tmp_path):
    vs, orc = _open(golden_dir, "x.fa", "x.vcf", tmp_path)
    want = survey_vectors["readme"]["x"]
    st = vs.construct_stats
    assert (st.num_mutations, st.num_mutations_samples) == (want["num_mutations"], want["num_mutations_samples"])
    assert (st.num_vertices, st.num_edges, st.seq_length) == (want["vertices"], want["edges"], want["seq_length"])
    assert st.num_classes == want["classes"]
    n, early, _ = orc.get_var_in_ref(10, 105)
    assert n == want["t6_10_105"] and not early
    assert orc.ub_events() == 0


@pytest.mark.parametrize("key", ["G1", "G3", "G4", "G1_t4", "G3_t4"])
def test_golden_texts(key, golden_dir, survey_vectors, tmp_path):
    g = survey_vectors[key]
    vs, orc = _open(golden_dir, g["fasta"], g["vcf"], tmp_path)
    x, y = g["region"]
    if g["type"] == 6:
        n, early, text = orc.get_var_in_ref(x, y)
    else:
        n, early, text = orc.get_sample_var_in_ref(x, y, g["sample"])
    assert text == g["text"]
    assert n == g["text"].count("\n") - 1
    if "stats" in g:
        st = vs.construct_stats
        assert (st.num_vertices, st.num_edges, st.seq_length, st.num_classes) == (
            g["stats"]["vertices"], g["stats"]["edges"], g["stats"]["seq_length"], g["stats"]["classes"])
    assert orc.ub_events() == 0


def test_golden_g2(golden_dir, survey_vectors, tmp_path):
    g = survey_vectors["G2"]
    _, orc = _open(golden_dir, g["fasta"], g["vcf"], tmp_path)
    n, early, text = orc.get_var_in_ref(*g["region"])
    assert n == g["count"]
    lines = text.split("\n")[1:-1]
    for frag in g["contains"]:
        assert any(l.startswith(frag) for l in lines), frag
    assert lines[-1] == g["last_row"]


def test_golden_region_start_probes(golden_dir, survey_vectors, tmp_path):
    g = survey_vectors["G4_probes"]
    _, orc = _open(golden_dir, g["fasta"], g["vcf"], tmp_path)
    for p in g["probes"]:
        n, early, text = orc.get_var_in_ref(*p["region"])
        assert (text, early) == (p["text"], p["early_out"]), p["region"]


def test_x_small_graph_dump(golden_dir, survey_vectors, tmp_path):
    """Constructor output == the reference's graph for x.small, including the
    query-time neighbour order, which the oracle derives with the real std::unordered_set."""
    g = survey_vectors["x_small_graph"]
    vs, orc = _open(golden_dir, g["fasta"], g["vcf"], tmp_path)
    assert vs.info().num_vertices == len(g["vertices"])
    for v, (off, length, cls, ridx, out) in g["vertices"].items():
        assert orc.out_neighbors(int(v)) == out, v
        assert vs.out_neighbors(int(v)) == out, v
    for pos, v in g["find"].items():
        assert orc.find(int(pos)) == v


def test_x_small_sample_indexes(golden_dir, survey_vectors, tmp_path):
    """fix_sample_indexes (variant_graph.h:1883-1997): per-carrier sample-coordinate indexes of the dump."""
    from helpers import read_plain
    g = survey_vectors["x_small_graph"]
    vs, _ = _open(golden_dir, g["fasta"], g["vcf"], tmp_path)
    plain = os.path.join(tmp_path, "idx.plain")
    vs.export_plain(plain)
    pg = read_plain(plain)
    got = {}
    for v in range(len(pg["off"])):
        cs = [int(pg["car_index"][c]) for c in range(int(pg["car_begin"][v]), int(pg["car_begin"][v + 1]))]
        if cs:
            got[str(v)] = cs
    assert got == g["carrier_index"]


def test_type7_observation_of_the_survey(golden_dir, tmp_path):
    """SURVEY.md §4.3 (run of the reference on the G4 index): `-t 7 -r 9 -b G -a A` logs
    "There is no such variant!" -- type 7 only finds variants hanging off a node that starts at pos."""
    _, orc = _open(golden_dir, "x.small.fa", "g4.vcf", tmp_path)
    assert orc.samples_has_var(9, "G", "A") is None
    # the same variant is reported by type 6 at var_pos 9, so no position makes type 7 return it
    assert all(orc.samples_has_var(p, "G", "A") is None for p in range(1, 90))
    # what it does find: an insertion whose anchor node ends at pos, and a substitution behind a zero-length
    # dummy ref node (quirk 3 of §4.3); the output line has no separator between the `name gt` pairs
    assert orc.samples_has_var(54, "", "AG") == "S2 0|1\n"
    assert orc.samples_has_var(20, "T", "G") == "S2 0|1S10 1|0\n"
    assert orc.samples_has_var(39, "T", "") is None
    assert orc.ub_events() == 0


def test_type1_closest_var_on_g4(golden_dir, tmp_path):
    """closest_var (query.h:441-483) restated: mirrored second call, and the step back past the last variant."""
    _, orc = _open(golden_dir, "x.small.fa", "g4.vcf", tmp_path)
    n, text = orc.closest_var(8)
    assert n == 1 and text.split("\n")[1].startswith("9\tG\tA\t")
    n, text = orc.closest_var(80)       # nothing ahead: walks back to the last site
    assert n == 1 and text.split("\n")[1].startswith("54\t\tAG\t")
    n, text = orc.closest_var(1000)     # beyond the reference
    assert n == 1 and text.split("\n")[1].startswith("54\t\tAG\t")
