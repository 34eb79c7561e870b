"""TWO RANKS through the C ABI's collective on the one GPU a test box has: real RCCL refuses two
ranks on one device, so `tests/native/fake_rccl.cpp` (host-staged all-gather through POSIX shared memory, the six symbols
comm.hip.h binds) is loaded through VS_RCCL_LIB.  Everything above it is the product's own code: vs_comm_init with world 2,
rank 1's records at 1 x max_count, the id-file rendezvous, the asynchronous gather beside the next batch, the CLI's --nprocs
with its rank-0 printing, its per-rank --batch-out shards and its "one rank died" path.  Fresh child processes only."""
import glob
import json
import os
import re
import subprocess
import sys
import time

import pytest

from helpers import build_fake_rccl, write_random_cohort
from test_cli import CLI, _construct, _msgs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_env():
    env = dict(os.environ, VS_RCCL_LIB=build_fake_rccl(), VS_FAKE_RCCL_TIMEOUT_S="60")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    yield env
    for f in glob.glob("/dev/shm/vs_fake_rccl_*"):   # (a killed rank 0 leaves its object behind)
        try:
            os.remove(f)
        except OSError:
            pass


def test_two_processes_gather_and_rebuild_the_whole_batch(fake_env, tmp_path):
    """vs_comm_init world 2 in two fresh processes that both open device 0; vs_comm_allgather_regions synchronous and asynchronous
    (with the next batch enqueued beside it); on BOTH ranks the gathered records -- rank k's at k x max_count -- go through
    vs_query_expand_site_ranges and give the single-process result: same totals, digest and text."""
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 512, n_rows=400, ref_len=6000, n_samples=90, carrier_p=0.35,
                                        p_near=0.6, p_multi=0.25, p_same=0.3)
    id_file, nonce = str(tmp_path / "uid"), f"two-ranks-{os.getpid()}-{time.time()}"
    procs, outs = [], []
    for rank in range(2):
        out = str(tmp_path / f"verdict{rank}.json")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py"), str(rank), "2", fasta, vcf,
                                       id_file, nonce, out], env=fake_env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    verdicts = [json.load(open(o)) if os.path.exists(o) else {"ok": False, "error": "no verdict"} for o in outs]
    for rank, (p, v, log) in enumerate(zip(procs, verdicts, logs)):
        assert p.returncode == 0 and v["ok"], (rank, v.get("error"), v.get("trace"), log[-2000:])
        assert v["info"] == [rank, 2, 2], "rank, world and ncclCommCount of the communicator"
        assert v["checks"] == {"async=False": True, "async=True": True}
        assert "fake RCCL" in log, "the stand-in was the library the engine loaded"
    assert verdicts[0]["digest"] == verdicts[1]["digest"] and verdicts[0]["totals"] == verdicts[1]["totals"]


def _cli(args, env, timeout=300):
    return subprocess.run([CLI, "query"] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_cli_nprocs_2_matches_the_single_process_for_every_query_type(fake_env, golden_dir, tmp_path):
    """`--nprocs 2 --nprocs-same-device` against the plain form for all seven -t: same log messages in the same order (rank 0
    prints every region's lines from the gathered records), same --batch-out text (rank shards concatenated in rank order), same
    -o file (the last region's / last found position's text, which rank 1 holds)."""
    d6 = str(tmp_path / "ser6")
    assert _construct(golden_dir, d6).returncode == 0
    rfile6 = str(tmp_path / "regions6.txt")
    with open(rfile6, "w") as f:
        for i in range(121):   # (odd: the shards differ in size)
            f.write(f"{1 + 11 * i}:{1 + 11 * i + 35 + (i % 5) * 40}\n")
    d = str(tmp_path / "ser")
    os.makedirs(d)
    out = subprocess.run([CLI, "construct", "-r", os.path.join(golden_dir, "x.small.fa"), "-v", os.path.join(golden_dir, "g4.vcf"), "-p", d],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rfile = str(tmp_path / "regions.txt")
    with open(rfile, "w") as f:
        for i in range(30):
            if i != 19:   # (39:78: the reference's backward search of types 2 / 3 / 5 does not terminate there -- see test_cli.py)
                f.write(f"{1 + 2 * i}:{1 + 2 * i + 12 + (i % 4) * 9}\n")
    pfile = str(tmp_path / "points.txt")
    pts = [9, 20, 54, 39, 8, 21, 60, 70, 9, 20, 33]
    with open(pfile, "w") as f:
        f.write("".join(f"{p}\n" for p in pts))
    # (-a/-b pair with the SORTED positions, commands.cc:91,185: 8 9 9 20 20 21 33 39 54 60 70)
    refs, alts = "G,G,G,T,T,A,A,T,,C,G", "A,A,C,C,G,C,C,,AG,T,T"
    cases = [(d6, ["-t", "6", "-r", "@" + rfile6], "6. Get variants", 121),
             (d, ["-t", "1", "-r", "@" + pfile], "1. return closest", len(pts)),
             (d, ["-t", "7", "-r", "@" + pfile, "-b", refs, "-a", alts], "7. Get samples", len(pts)),
             (d, ["-t", "2", "-r", "@" + rfile, "-s", "S10"], "2. Get sample's sequence", 29),
             (d, ["-t", "3", "-r", "@" + rfile, "-s", "S10"], "3. Get sample's sequence", 29),
             (d, ["-t", "4", "-r", "@" + rfile, "-s", "S1"], "4. Get sample's variants", 29),
             (d, ["-t", "5", "-r", "@" + rfile, "-s", "S10"], "5. Get sample's variants", 29)]
    for prefix, args, head, n_lines in cases:
        got = []
        for flag in ([], ["--nprocs", "2", "--nprocs-same-device"]):
            bfile, ofile = str(tmp_path / f"b{len(flag)}.txt"), str(tmp_path / f"o{len(flag)}.txt")
            for fpath in (bfile, ofile):
                if os.path.exists(fpath):
                    os.remove(fpath)
            out = _cli(["-p", prefix, "-m", "1", "-o", ofile, "-v", "--batch-out", bfile] + args + flag, fake_env)
            assert out.returncode == 0, (args, flag, out.stdout + out.stderr)
            if flag:
                assert out.stderr.count("fake RCCL") == 2, "both ranks went through the stand-in"
            msgs = [m for m in _msgs(out.stdout) if not re.match(r"Query\d+: ", m) and not m.startswith(("Loading", "Read ", "Graph stats", "Chromosome"))]
            assert sum(m.startswith(head) for m in msgs) == n_lines, (args, flag, msgs[:5])
            got.append((msgs, open(bfile).read(), open(ofile).read() if os.path.exists(ofile) else None))
        assert got[0] == got[1], (args, "plain form against --nprocs 2")
        assert got[0][2], args      # (the -o file was written by both forms)


def test_cli_nprocs_2_ends_rank_0_when_rank_1_is_killed(fake_env, golden_dir, tmp_path):
    """Rank 1 dies by SIGKILL after it has opened the index and before it joins the communicator (VS_NPROCS_TEST_KILL_RANK, a
    test-only hook of the CLI): rank 0 sits in ncclCommInitRank waiting for it; the parent reaps the dead rank, stops rank 0 and
    exits non-zero -- well inside ten seconds, not at the collective's own timeout (60 s here)."""
    d = str(tmp_path / "ser")
    assert _construct(golden_dir, d).returncode == 0
    rfile = str(tmp_path / "regions.txt")
    with open(rfile, "w") as f:
        for i in range(16):
            f.write(f"{1 + 11 * i}:{40 + 11 * i}\n")
    args = ["-p", d, "-t", "6", "-r", "@" + rfile, "-m", "1", "--nprocs", "2", "--nprocs-same-device"]
    assert _cli(args, fake_env).returncode == 0      # (warm: the first open of a box pages the libraries in)
    t0 = time.time()
    out = _cli(args, dict(fake_env, VS_NPROCS_TEST_KILL_RANK="1"), timeout=120)
    took = time.time() - t0
    assert out.returncode != 0, out.stdout + out.stderr
    assert "a rank of --nprocs failed; stopping the others" in out.stdout + out.stderr
    assert took < 10.0, f"the parent took {took:.1f} s to end the surviving rank"


def test_cli_nprocs_3_on_a_midsize_index_with_thousands_of_regions(fake_env, tmp_path):
    """THREE ranks (uneven shards: 3001 regions) on a 20,000-variant x 200-sample index saved to disk: `--nprocs 3` against the plain
    form -- the count line of every region (printed by rank 0 from records that ranks 1 and 2 produced) and the whole --batch-out text
    (three shards put together by the parent)."""
    import hashlib
    import numpy as np
    from variantstore_amd import VariantStore
    vs = VariantStore.synthetic(device=0, ref_length=2_000_000, num_variants=20_000, num_samples=200, seed=21,
                                first_pos=500, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=6, af_exponent=3.0)
    d = str(tmp_path / "ser")
    os.makedirs(d)
    vs.save(d)
    vs.close()
    rng = np.random.default_rng(31)
    starts = np.sort(rng.integers(1, 1_990_000, size=3001))
    rfile = str(tmp_path / "regions.txt")
    with open(rfile, "w") as f:
        for s in starts:
            f.write(f"{int(s)}:{int(s) + 4000}\n")
    got = []
    for flag in ([], ["--nprocs", "3", "--nprocs-same-device"]):
        bfile = str(tmp_path / f"b{len(flag)}.txt")
        out = _cli(["-p", d, "-t", "6", "-r", "@" + rfile, "-m", "1", "--batch-out", bfile] + flag, fake_env, timeout=600)
        assert out.returncode == 0, (flag, out.stdout[-2000:] + out.stderr[-2000:])
        if flag:
            assert out.stderr.count("fake RCCL") == 3
        counts = [ln for ln in out.stdout.split("\n") if ln.startswith("Number of variants")]
        assert len(counts) == 3001
        text = open(bfile, "rb").read()
        got.append((counts, len(text), hashlib.sha256(text).hexdigest()))
    assert got[0] == got[1]
    assert got[0][1] > 1_000_000       # (megabytes of rows: the shards are not trivially small)
