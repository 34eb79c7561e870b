"""The drop-in `variantstore` command line (variantstore_amd/csrc/cli/variantstore.cpp):
construct runs on the host; query needs the GPU."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CLI = os.path.join(ROOT, "variantstore_amd", "bin", "variantstore")
LOG = re.compile(r"^\[\d{4}-\d\d-\d\d \d\d:\d\d:\d\d\.\d{3}\] \[(info|error)\] (.*)$")


def _msgs(stdout):
    out = []
    for line in stdout.splitlines():
        m = LOG.match(line)
        out.append(m.group(2) if m else line)
    return out


def _construct(golden_dir, d):
    os.makedirs(d, exist_ok=True)
    return subprocess.run([CLI, "construct", "-r", os.path.join(golden_dir, "x.fa"), "-v",
                           os.path.join(golden_dir, "x.vcf"), "-p", d], capture_output=True, text=True)


def test_construct_log_lines_match_the_readme(golden_dir, tmp_path):
    out = _construct(golden_dir, str(tmp_path / "ser"))
    assert out.returncode == 0, out.stderr
    msgs = _msgs(out.stdout)
    # reference README.md:51-62
    assert "Num mutations: 75 num mutations-sample: 75" in msgs
    assert "Chromosome: x #Vertices: 212 #Edges: 287 Seq length: 1074" in msgs
    assert "Number of sample vector classes: 2" in msgs
    assert msgs[0] == "Creating variant graph" and msgs[-1] == "Serializing index to disk"
    assert all(LOG.match(l) for l in out.stdout.splitlines())
    assert len(os.listdir(str(tmp_path / "ser"))) == 9


def test_construct_requires_existing_directory(golden_dir, tmp_path):
    out = subprocess.run([CLI, "construct", "-r", os.path.join(golden_dir, "x.fa"), "-v",
                          os.path.join(golden_dir, "x.vcf"), "-p", str(tmp_path / "missing")],
                         capture_output=True, text=True)
    assert out.returncode != 0 and "does not seem to exist" in out.stderr


@pytest.mark.gpu
def test_query_type6_readme_example(golden_dir, survey_vectors, tmp_path):
    d = str(tmp_path / "ser")
    assert _construct(golden_dir, d).returncode == 0
    ofile = str(tmp_path / "variant.txt")
    out = subprocess.run([CLI, "query", "-p", d, "-t", "6", "-r", "10:105", "-m", "0", "-s", "1", "-o", ofile, "-v"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    msgs = _msgs(out.stdout)
    # reference README.md:89-96
    assert "Chromosome: x #Vertices: 212 #Edges: 0 Seq length: 1074" in msgs
    assert "6. Get variants in ref coordinate. 0" in msgs
    assert "Number of variants get_var_in_ref: 8" in msgs
    assert re.match(r"Query1: \(query_var_in_ref\) Total Time Elapsed: \d+\.\d{6}seconds", msgs[-1])
    assert open(ofile).read() == survey_vectors["G1"]["text"]


@pytest.mark.gpu
def test_query_batch_sorted_last_region_in_outfile_and_type4(golden_dir, survey_vectors, tmp_path):
    d = str(tmp_path / "ser")
    assert _construct(golden_dir, d).returncode == 0
    ofile, bfile = str(tmp_path / "o.txt"), str(tmp_path / "b.txt")
    # regions given unsorted: the driver sorts them (commands.cc:91); the -o file keeps only the last one
    out = subprocess.run([CLI, "query", "-p", d, "-t", "6", "-r", "1:1001,10:105,2000:2100", "-m", "1", "-o", ofile,
                          "-v", "--batch-out", bfile], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    msgs = _msgs(out.stdout)
    counts = [m for m in msgs if m.startswith("Number of variants")]
    assert counts == ["Number of variants get_var_in_ref: 75", "Number of variants get_var_in_ref: 8",
                      "Number of variants get_sample_var_in_ref: 0"]  # early-out prints the other label (query.h:746)
    assert open(ofile).read() == "Pos\tRef\tAlt\tSamples\n"
    assert "#region 1 10:105\n" + survey_vectors["G1"]["text"] in open(bfile).read()
    out = subprocess.run([CLI, "query", "-p", d, "-t", "4", "-r", "10:105", "-m", "1", "-s", "1", "-o", ofile, "-v"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Number of variants get_sample_var_in_ref: 8" in _msgs(out.stdout)
    assert open(ofile).read() == survey_vectors["G1_t4"]["text"]


@pytest.mark.gpu
def test_query_types_1_and_7(golden_dir, tmp_path):
    """`-t 1` (closest_var) and `-t 7` (samples_has_var) on the 3-sample G4 index of SURVEY.md §4.3."""
    d = str(tmp_path / "ser")
    os.makedirs(d)
    out = subprocess.run([CLI, "construct", "-r", os.path.join(golden_dir, "x.small.fa"), "-v",
                          os.path.join(golden_dir, "g4.vcf"), "-p", d], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ofile = str(tmp_path / "o.txt")
    # the survey's run of the reference: `-t 7 -r 9 -b G -a A` -> "[error] There is no such variant!"
    out = subprocess.run([CLI, "query", "-p", d, "-t", "7", "-r", "9", "-b", "G", "-a", "A", "-m", "1", "-o", ofile, "-v"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    msgs = _msgs(out.stdout)
    assert "7. Get samples have given variant. 0" in msgs
    assert "Looking for variant POS: 9, REF: G, ALT: A" in msgs
    assert "There is no such variant!" in msgs
    assert not os.path.exists(ofile)
    # regions are sorted, the -a/-b lists are not (commands.cc:91,185): 20 pairs with the FIRST listed sequences
    out = subprocess.run([CLI, "query", "-p", d, "-t", "7", "-r", "54,20", "-b", "T,", "-a", "G,AG", "-m", "1", "-o", ofile,
                          "-v"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    msgs = _msgs(out.stdout)
    assert "Looking for variant POS: 20, REF: T, ALT: G" in msgs and "Looking for variant POS: 54, REF: , ALT: AG" in msgs
    assert "There is no such variant!" not in msgs
    assert open(ofile).read() == "S2 0|1\n"          # the last region's line stays in the file
    out = subprocess.run([CLI, "query", "-p", d, "-t", "1", "-r", "8,80", "-m", "1", "-o", ofile, "-v", "--batch-out",
                          str(tmp_path / "b.txt")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    msgs = _msgs(out.stdout)
    assert "1. return closest mutation in ref coordinate. 0" in msgs and "1. return closest mutation in ref coordinate. 1" in msgs
    assert re.match(r"Query2: Total Time Elapsed: \d+\.\d{6}seconds", msgs[-1])
    assert open(ofile).read() == "Pos\tRef\tAlt\tSamples\n54\t\tAG\tS2(0|1) \n"
    assert "#region 0 8\nPos\tRef\tAlt\tSamples\n9\tG\tA\tS2(1/1) S10(0|1) S1(1|0) \n" in open(str(tmp_path / "b.txt")).read()


@pytest.mark.gpu
def test_query_types_2_3_5(golden_dir, tmp_path):
    """`-t 2`, `-t 3` (sample sequences) and `-t 5` (variants in the sample's coordinates) on the G4 index;
    expected strings are the CPU oracle's (tests/test_gpu_parity.py checks those at scale)."""
    d = str(tmp_path / "ser")
    os.makedirs(d)
    out = subprocess.run([CLI, "construct", "-r", os.path.join(golden_dir, "x.small.fa"), "-v",
                          os.path.join(golden_dir, "g4.vcf"), "-p", d], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ofile = str(tmp_path / "o.txt")
    ref = "CAAATAAGGCTTGGAAATTTTCTGGAGTTCTATTATATTCCAACTCTCTGGTTCCTGGTGCTATGTGTAACTAGTAATGG"
    out = subprocess.run([CLI, "query", "-p", d, "-t", "2", "-r", "1:80", "-s", "S2", "-m", "1", "-o", ofile, "-v"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "2. Get sample's sequence in ref coordinate. 0" in _msgs(out.stdout)
    # S2 over ref [1,80): G>A at 9, T>C at 20, AG inserted after 54
    want = ref[:8] + "A" + ref[9:19] + "C" + ref[20:54] + "AG" + ref[54:79]
    assert open(ofile).read() == want + "\n"
    out = subprocess.run([CLI, "query", "-p", d, "-t", "3", "-r", "1:80", "-s", "S2", "-m", "1", "-o", ofile, "-v"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "3. Get sample's sequence in sample's coordinate. 0" in _msgs(out.stdout)
    assert open(ofile).read() == want[:79] + "\n"        # 79 bases of the SAMPLE's sequence
    out = subprocess.run([CLI, "query", "-p", d, "-t", "5", "-r", "1:80", "-s", "S10", "-m", "1", "-o", ofile, "-v"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    msgs = _msgs(out.stdout)
    assert "5. Get sample's variants in sample coordinate. 0" in msgs
    assert "Number of variants get_sample_var_in_sample: 3" in msgs
    assert open(ofile).read() == ("Pos\tRef\tAlt\tSamples\n9\tG\tA\tS2(1/1) S10(0|1) S1(1|0) \n"
                                  "20\t\tC\tS2(0|1) S10(1|0) \n39\tT\t\tS10(0|1) S1(1|0) \n")


def test_draw_writes_graph_dot(golden_dir, tmp_path):
    """`variantstore draw -p <dir> -r <pos> -h <hops>` (commands.cc:217-242) is host-only."""
    d = str(tmp_path / "ser")
    assert _construct(golden_dir, d).returncode == 0
    out = subprocess.run([CLI, "draw", "-p", d, "-r", "100", "-h", "3"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    msgs = _msgs(out.stdout)
    assert "Looking up vertex corresponding to the queried region" in msgs
    assert "Chromosome: x #Vertices: 212 #Edges: 0 Seq length: 1074" in msgs
    text = open(os.path.join(d, "graph.dot")).read()
    assert text.startswith("digraph {\n") and text.endswith("}") and " -> " in text and "ref i:" in text


@pytest.mark.gpu
def test_query_batch_file_with_resident_lists(golden_dir, tmp_path):
    """`-r @FILE --batch-out` over more regions than the latency path takes, with and without `--resident-lists`
    (every carrier list of the index expanded once at open; rows only cross PCIe): same counts, same texts."""
    d = str(tmp_path / "ser")
    assert _construct(golden_dir, d).returncode == 0
    rfile = str(tmp_path / "regions.txt")
    with open(rfile, "w") as f:
        for i in range(150):
            f.write(f"{1 + 13 * i}:{1 + 13 * i + 40 + (i % 7) * 30}\n")
    outs = []
    for typ, extra in (("6", []), ("4", ["-s", "1"])):
        texts = []
        for flag in ([], ["--resident-lists"]):
            bfile = str(tmp_path / f"b{typ}{len(flag)}.txt")
            out = subprocess.run([CLI, "query", "-p", d, "-t", typ, "-r", "@" + rfile, "-m", "1", "-o", str(tmp_path / "o.txt"),
                                  "--batch-out", bfile] + extra + flag, capture_output=True, text=True)
            assert out.returncode == 0, out.stdout + out.stderr
            counts = [m for m in _msgs(out.stdout) if m.startswith("Number of variants")]
            assert len(counts) == 150
            texts.append((counts, open(bfile).read()))
        assert texts[0] == texts[1]
        assert texts[0][1].count("#region ") == 150
        outs.append(texts[0][1])
    assert outs[0] != outs[1]


@pytest.mark.gpu
def test_query_nprocs_form_matches_the_single_process(golden_dir, tmp_path):
    """`--nprocs 1`: the multi-process form of the drop-in CLI at world size 1 -- the parent forks before anything touches
    the GPU, the rank answers its shard, the per-region records go through the C ABI's own collective (vs_comm_*: RCCL)
    and the log lines are printed from the GATHERED records: same count lines, same --batch-out text as the plain form."""
    d = str(tmp_path / "ser")
    assert _construct(golden_dir, d).returncode == 0
    rfile = str(tmp_path / "regions.txt")
    with open(rfile, "w") as f:
        for i in range(120):
            f.write(f"{1 + 11 * i}:{1 + 11 * i + 35 + (i % 5) * 40}\n")
    got = []
    for flag in ([], ["--nprocs", "1"]):
        bfile = str(tmp_path / f"b{len(flag)}.txt")
        out = subprocess.run([CLI, "query", "-p", d, "-t", "6", "-r", "@" + rfile, "-m", "1", "-o", str(tmp_path / "o.txt"),
                              "--batch-out", bfile] + flag, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = [ln for ln in out.stdout.split("\n") if ln.startswith("Number of variants")]
        msgs = [m for m in _msgs(out.stdout) if m.startswith("6. Get variants")]
        assert len(lines) == 120 and len(msgs) == 120
        got.append((lines, msgs, open(bfile).read()))
    assert got[0] == got[1]


@pytest.mark.gpu
def test_query_nprocs_form_for_every_query_type(golden_dir, tmp_path):
    """`--nprocs 1` for the other six query types (BASELINE configs[4] mixes 3 / 6 / 7; src/commands.cc:150-193 dispatches all
    seven): the rank answers its shard with the type's own entry point, the per-region SUMMARY records (vs_result_pack_regions:
    counts and flags for row results, pieces and bytes for sequences) go through vs_comm_*, rank 0 prints every region's log
    lines from the gathered records -- same messages, same --batch-out text, same -o file as the single-process form."""
    d = str(tmp_path / "ser")
    os.makedirs(d)
    out = subprocess.run([CLI, "construct", "-r", os.path.join(golden_dir, "x.small.fa"), "-v",
                          os.path.join(golden_dir, "g4.vcf"), "-p", d], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rfile = str(tmp_path / "regions.txt")
    with open(rfile, "w") as f:
        for i in range(30):
            if i != 19:   # (39:78: the reference's backward search of types 2 / 3 / 5 does not terminate there for S1 and S10 -- by the oracle)
                f.write(f"{1 + 2 * i}:{1 + 2 * i + 12 + (i % 4) * 9}\n")
    pfile = str(tmp_path / "points.txt")
    pts = [9, 20, 54, 39, 8, 21, 60, 70, 9, 20]
    with open(pfile, "w") as f:
        f.write("".join(f"{p}\n" for p in pts))
    # (-a/-b pair with the SORTED positions, commands.cc:91,185: 8 9 9 20 20 21 39 54 60 70)
    refs, alts = "G,G,G,T,T,A,T,,C,G", "A,A,C,C,G,C,,AG,T,T"
    cases = [(["-t", "1", "-r", "@" + pfile], "1. return closest"),
             (["-t", "7", "-r", "@" + pfile, "-b", refs, "-a", alts], "7. Get samples"),
             (["-t", "2", "-r", "@" + rfile, "-s", "S10"], "2. Get sample's sequence"),
             (["-t", "3", "-r", "@" + rfile, "-s", "S10"], "3. Get sample's sequence"),
             (["-t", "4", "-r", "@" + rfile, "-s", "S1"], "4. Get sample's variants"),
             (["-t", "5", "-r", "@" + rfile, "-s", "S10"], "5. Get sample's variants")]
    for args, head in cases:
        got = []
        for flag in ([], ["--nprocs", "1"]):
            bfile, ofile = str(tmp_path / f"b{len(flag)}.txt"), str(tmp_path / f"o{len(flag)}.txt")
            for fpath in (bfile, ofile):
                if os.path.exists(fpath):
                    os.remove(fpath)
            out = subprocess.run([CLI, "query", "-p", d, "-m", "1", "-o", ofile, "-v", "--batch-out", bfile] + args + flag,
                                 capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, (args, out.stdout + out.stderr)
            msgs = [m for m in _msgs(out.stdout) if not re.match(r"Query\d+: ", m) and not m.startswith(("Loading", "Read ", "Graph stats", "Chromosome"))]
            assert sum(m.startswith(head) for m in msgs) == (len(pts) if args[1] in ("1", "7") else 29), (args, msgs[:5])
            got.append((msgs, open(bfile).read(), open(ofile).read() if os.path.exists(ofile) else None))
        assert got[0] == got[1], args
        assert got[0][2], args      # (the -o file was written by both forms)
    # a sample the index does not know: every rank stops, the parent reports the failure
    out = subprocess.run([CLI, "query", "-p", d, "-m", "1", "-t", "4", "-r", "1:30", "-s", "nobody", "--nprocs", "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "Sample not found" in out.stdout + out.stderr


@pytest.mark.gpu
def test_query_nprocs_ends_all_ranks_when_one_fails(golden_dir, tmp_path):
    """`--nprocs N` with more ranks than GPUs: the rank whose device does not exist fails at vs_index_open, the others would sit in
    ncclCommInitRank (or wait for rank 0's unique id) for good -- the parent reaps whichever rank ends first and stops the rest
    (it used to wait for them in rank order, forever)."""
    import torch
    n = torch.cuda.device_count() + 1
    d = str(tmp_path / "ser")
    assert _construct(golden_dir, d).returncode == 0
    rfile = str(tmp_path / "regions.txt")
    with open(rfile, "w") as f:
        for i in range(4 * n):
            f.write(f"{1 + 11 * i}:{40 + 11 * i}\n")
    out = subprocess.run([CLI, "query", "-p", d, "-t", "6", "-r", "@" + rfile, "-m", "1", "--nprocs", str(n)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "a rank of --nprocs failed; stopping the others" in out.stdout + out.stderr
