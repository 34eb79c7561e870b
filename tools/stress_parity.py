#!/usr/bin/env python3
"""Randomised parity campaign (GPU): many small cohorts of varied shape, every region compared with the
CPU oracle as text, for all seven query types.  Usage: python tools/stress_parity.py [n_cohorts] [seed0]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_regions, write_random_cohort  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from variantstore_amd import VariantStore  # noqa: E402

n_cohorts = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = hang = ub = checked = slow_regions = shared_batches = many_runs = speculated = refused = 0
VERBOSE = bool(os.environ.get("VS_STRESS_VERBOSE"))


def stage(msg):
    if VERBOSE:
        print("  .. " + msg, file=sys.stderr, flush=True)

for c in range(n_cohorts):
    seed = seed0 + c
    rng = np.random.default_rng(seed)
    kw = dict(ref_len=int(rng.integers(300, 6000)), n_rows=int(rng.integers(5, 400)),
              n_samples=int(rng.choice([1, 2, 5, 9, 40, 70, 130, 300, 700])), p_ins=float(rng.uniform(0, 0.35)),
              p_del=float(rng.uniform(0, 0.35)), p_multi=float(rng.uniform(0, 0.4)), p_mnp=float(rng.uniform(0, 0.15)),
              p_near=float(rng.uniform(0, 0.9)), carrier_p=float(rng.choice([0.004, 0.05, 0.3, 0.7])),
              unphased_p=float(rng.uniform(0, 0.3)), missing_p=float(rng.uniform(0, 0.1)),
              p_same=float(rng.choice([0.0, 0.0, 0.2, 0.5])), max_indel=int(rng.choice([2, 4, 12, 40])))
    if os.environ.get("VS_STRESS_WIDE"):   # wide cohorts: every row width of the slice path, and the WIDE kernel beyond 4032
        kw.update(n_samples=int(rng.choice([900, 1300, 1500, 2100, 2504, 2600, 3300, 3900, 4030, 4031, 4032, 4040, 5000])),
                  n_rows=int(rng.integers(5, 60)), carrier_p=float(rng.choice([0.01, 0.05, 0.3, 0.7])))
    if rng.random() < 0.3:
        names = [f"n{int(x)}" for x in rng.permutation(kw["n_samples"])]
        kw["sample_names"] = names
    # the threshold between the list path and the row path of k_fill_carriers is fixed when an index is opened
    os.environ["VS_LIST_MAX"] = str(int(rng.choice([0, 1, 7, 8, 9, 40, 200, 640])))
    with tempfile.TemporaryDirectory() as td:
        fasta, vcf, names = write_random_cohort(td, seed, **kw)
        try:
            vs = VariantStore.from_vcf(fasta, vcf, device=0)
        except Exception as e:  # constructor refuses (e.g. mutation past the reference end)
            print(f"cohort {seed}: construct refused: {e}")
            continue
        plain = os.path.join(td, "p.bin")
        vs.export_plain(plain)
        orc = Oracle(plain)
        # SPECULATIVE type-6 batches first, on the fresh handle (round 6: a sorted batch after a sorted batch of about as many regions is
        # submitted without waiting for its sizes; one that does not fit what the batch before needed is refused on the device and redone
        # when it is first read): short regions set the expectation, long ones outgrow it, then both fit
        stage(f"cohort {seed} {kw} list_max {os.environ['VS_LIST_MAX']}: speculative batches")
        L0 = vs.info().ref_length
        short_b = sorted(random_regions(rng, L0, 160, max_len=40)[:160])   # (without the whole-reference shapes: few rows)
        long_b = sorted(random_regions(rng, L0, 150, max_len=5000))          # (with them: every site)
        vs.get_var_in_ref(short_b).close()
        for regs in (short_b, long_b, long_b, short_b):
            rs = vs.get_var_in_ref(regs)
            for q, (x, y) in enumerate(regs):
                n, early, text = orc.get_var_in_ref(x, y)
                if n < 0:
                    continue
                checked += 1
                if rs.region_text(q) != text:
                    bad += 1
                    print(f"MISMATCH t6 speculative batch cohort {seed} region {x}:{y}")
            rs.close()
        inf = vs.info()
        speculated += int(inf.t6_speculated)
        refused += int(inf.t6_refused)
        regions = random_regions(rng, vs.info().ref_length, 150, max_len=int(rng.choice([5, 60, 600, 5000])))
        stage(f"cohort {seed} {kw} list_max {os.environ['VS_LIST_MAX']}: type 6")
        res = vs.get_var_in_ref(regions)
        flags = res.view(False)
        for q, (x, y) in enumerate(regions):
            n, early, text = orc.get_var_in_ref(x, y)
            if n < 0:
                hang += 1
                continue
            checked += 1
            if res.region_text(q) != text or bool(flags["region_flags"][q] & 1) != early:
                bad += 1
                print(f"MISMATCH t6 cohort {seed} region {x}:{y}\n--- gpu\n{res.region_text(q)}--- oracle\n{text}")
        slow_regions += int((flags["var_count"] != np.diff(flags["var_begin"].astype(np.int64))).sum())
        # the same regions sorted: the batch then shares one row and one carrier list per covered site between its regions
        order = sorted(range(len(regions)), key=lambda i: regions[i])
        stage("sorted batch")
        rsh = vs.get_var_in_ref([regions[i] for i in order])
        shared_batches += int(rsh.layout()[4])
        for k, i in enumerate(order):
            if orc.get_var_in_ref(*regions[i])[0] < 0:
                continue
            checked += 1
            if rsh.region_text(k) != res.region_text(i):
                bad += 1
                print(f"MISMATCH t6 shared-lists batch cohort {seed} region {regions[i]}")
        if rsh.digest() != vs.get_var_in_ref([regions[i] for i in order][:64] + [regions[i] for i in order][64:]).digest():
            bad += 1
            print(f"MISMATCH digest of the shared batch, cohort {seed}")
        rsh.close()
        # many SHORT regions in position order: a shared table of many short runs (round 6: a row that closed 63 single-row runs took the
        # 64th run's site -- shared_row_run; batches of long regions have a handful of runs); each region against its single-region answer
        # (the latency path: checked against the oracle above and below) and the whole batch against private rows
        stage("many short regions")
        L = vs.info().ref_length
        short = sorted((int(x), int(x) + int(rng.integers(1, 9))) for x in rng.integers(1, max(2, L - 10), size=700))
        rsr = vs.get_var_in_ref(short)
        many_runs += int(rsr.layout()[1] > 64)
        for k in range(0, len(short), 64):
            one = vs.get_var_in_ref(short[k:k + 64])
            for j in range(min(64, len(short) - k)):
                checked += 1
                if rsr.region_text(k + j) != one.region_text(j):
                    bad += 1
                    print(f"MISMATCH t6 many short regions, cohort {seed} region {short[k + j]}")
            one.close()
        for (x, y), k in [(short[k], k) for k in range(0, len(short), 9)]:
            n, _e, text = orc.get_var_in_ref(x, y)
            if n >= 0:
                checked += 1
                if rsr.region_text(k) != text:
                    bad += 1
                    print(f"MISMATCH t6 many short regions against the oracle, cohort {seed} region {x}:{y}")
        vs.set_option("share_lists", 0)
        prv = vs.get_var_in_ref(short)
        vs.set_option("share_lists", 1)
        if (prv.totals(), prv.digest()) != (rsr.totals(), rsr.digest()):
            bad += 1
            print(f"MISMATCH many short regions: shared against private rows, cohort {seed}")
        prv.close()
        rsr.close()
        # the same regions again through the single-launch latency path (up to 64 regions: kernel-argument forms of 8
        # and 64 regions) and a 70-region batch through the general path
        for lo, hi in ((0, 1), (1, 4), (4, 12), (12, 76), (20, 21), (30, 100)):
            sub_r = regions[lo:hi]
            rs = vs.get_var_in_ref(sub_r)
            for q, (x, y) in enumerate(sub_r):
                n, early, text = orc.get_var_in_ref(x, y)
                if n < 0:
                    continue
                checked += 1
                if rs.region_text(q) != text:
                    bad += 1
                    print(f"MISMATCH t6 small batch cohort {seed} region {x}:{y}")
            rs.close()
        # type 4 with one sample per region, through the cooperative walk, the serial walk and the literal (no-jump) walk
        per = [names[int(i)] for i in rng.integers(0, len(names), size=100)]
        want4 = [orc.get_sample_var_in_ref(x, y, sm) for (x, y), sm in zip(regions[:100], per)]
        for coop, skip in ((2, 1), (1, 1), (0, 0)):
            vs.set_option("t4_walk", coop)
            stage(f"type 4 walk {coop}")
            rm = vs.get_sample_var_in_ref(regions[:100], per)
            for q, (n, early, text) in enumerate(want4):
                if n < 0:
                    continue
                checked += 1
                if rm.region_text(q) != text:
                    bad += 1
                    print(f"MISMATCH t4 (coop {coop} skip {skip}) cohort {seed} sample {per[q]} region {regions[q]}")
            rm.close()
        vs.set_option("t4_walk", 2)
        # resident carrier lists: the whole unsorted batch, its sorted form and the type-4 batch again, rows only
        stage("resident lists")
        vs.set_option("resident_lists", 1)
        for form, idxs in (("unsorted", list(range(len(regions)))), ("sorted", order)):
            rr = vs.get_var_in_ref([regions[i] for i in idxs])
            if rr.layout()[2] != 0:
                bad += 1
                print(f"resident batch with an arena of its own, cohort {seed} ({form})")
            for k, i in enumerate(idxs):
                if orc.get_var_in_ref(*regions[i])[0] < 0:
                    continue
                checked += 1
                if rr.region_text(k) != res.region_text(i):
                    bad += 1
                    print(f"MISMATCH t6 resident lists ({form}) cohort {seed} region {regions[i]}")
            rr.close()
        rm = vs.get_sample_var_in_ref(regions[:100], per)
        for q, (n, early, text) in enumerate(want4):
            if n < 0:
                continue
            checked += 1
            if rm.region_text(q) != text:
                bad += 1
                print(f"MISMATCH t4 resident lists cohort {seed} sample {per[q]} region {regions[q]}")
        rm.close()
        vs.set_option("resident_lists", 0)
        sample = names[int(rng.integers(0, len(names)))]
        r4 = vs.get_sample_var_in_ref(regions[:60], sample)
        for q, (x, y) in enumerate(regions[:60]):
            n, early, text = orc.get_sample_var_in_ref(x, y, sample)
            if n < 0:
                hang += 1
                continue
            checked += 1
            if r4.region_text(q) != text:
                bad += 1
                print(f"MISMATCH t4 cohort {seed} sample {sample} region {x}:{y}\n--- gpu\n{r4.region_text(q)}--- oracle\n{text}")
        stage("types 1, 7, 2, 3, 5")
        # ---- types 1 and 7 ----
        L = vs.info().ref_length
        positions = [int(p) for p in rng.integers(0, L + 30, size=80)]
        r1 = vs.closest_var(positions)
        f1 = r1.view(False)["region_flags"]
        rows = []
        for q, p in enumerate(positions):
            n, text = orc.closest_var(p)
            checked += 1
            if (n < 0) != bool(f1[q] & 4) or (n >= 0 and r1.region_text(q) != text):
                bad += 1
                print(f"MISMATCH t1 cohort {seed} pos {p}")
            if n > 0:
                for line in text.split("\n")[1:-1]:
                    a, b, c_, _ = line.split("\t")
                    rows.append((int(a), b, c_))
        rows = rows[:60] + [(p, "A", "C") for p in positions[:20]]
        if rows:
            r7 = vs.samples_has_var([x[0] for x in rows], [x[1] for x in rows], [x[2] for x in rows])
            f7 = r7.view(False)["region_flags"]
            for q, (p, a, b) in enumerate(rows):
                want = orc.samples_has_var(p, a, b)
                checked += 1
                if (want is None) != bool(f7[q] & 4) or (want is not None and r7.region_text(q) != want):
                    bad += 1
                    print(f"MISMATCH t7 cohort {seed} {p} {a} {b}")
            r7.close()
        r1.close()
        # ---- types 2, 3 and 5 (sample coordinates) ----
        sc_regions = regions[:50]
        for form, smp in ((2, "ref"), (2, sample), (1, sample)):   # cooperative kernels (default), then the one-lane walks
            vs.set_option("t4_walk", form)
            for coords in (False, True):
                rs = vs.query_sample_seq(sc_regions, smp, sample_coordinates=coords)
                fl, seqs = rs.sequences()
                for q, (x, y) in enumerate(sc_regions):
                    n, seq = (orc.query_sample_from_sample if coords else orc.query_sample_from_ref)(x, y, smp)
                    checked += 1
                    okq = (bool(fl[q] & 8) if n == -1 else bool(fl[q] & 2) if n == -3 else (not fl[q] and seqs[q] == seq))
                    if not okq:
                        bad += 1
                        print(f"MISMATCH t{3 if coords else 2} cohort {seed} sample {smp} region {x}:{y} code {n} flags {fl[q]}")
                rs.close()
            r5 = vs.get_sample_var_in_sample(sc_regions, smp)
            f5 = r5.view(False)["region_flags"]
            for q, (x, y) in enumerate(sc_regions):
                n, text = orc.get_sample_var_in_sample(x, y, smp)
                checked += 1
                if (n == -1) != bool(f5[q] & 8) or (n >= 0 and r5.region_text(q) != text):
                    bad += 1
                    print(f"MISMATCH t5 cohort {seed} sample {smp} region {x}:{y}")
            r5.close()
        vs.set_option("t4_walk", 2)
        # one sample per region through the cooperative kernels (the groups of a wave then run different samples' rows)
        per_sc = [names[int(i)] for i in rng.integers(0, len(names), size=len(sc_regions))]
        for coords in (False, True):
            rs = vs.query_sample_seq(sc_regions, per_sc, sample_coordinates=coords)
            fl, seqs = rs.sequences()
            for q, (x, y) in enumerate(sc_regions):
                n, seq = (orc.query_sample_from_sample if coords else orc.query_sample_from_ref)(x, y, per_sc[q])
                checked += 1
                okq = (bool(fl[q] & 8) if n == -1 else bool(fl[q] & 2) if n == -3 else (not fl[q] and seqs[q] == seq))
                if not okq:
                    bad += 1
                    print(f"MISMATCH t{3 if coords else 2} (per-region samples) cohort {seed} sample {per_sc[q]} region {x}:{y} code {n} flags {fl[q]}")
            rs.close()
        r5 = vs.get_sample_var_in_sample(sc_regions, per_sc)
        f5 = r5.view(False)["region_flags"]
        for q, (x, y) in enumerate(sc_regions):
            n, text = orc.get_sample_var_in_sample(x, y, per_sc[q])
            checked += 1
            if (n == -1) != bool(f5[q] & 8) or (n >= 0 and r5.region_text(q) != text):
                bad += 1
                print(f"MISMATCH t5 (per-region samples) cohort {seed} sample {per_sc[q]} region {x}:{y}")
        r5.close()
        # the same per-region batches with regions and sample ids handed over in DEVICE memory (the first kernel of the batch then
        # takes the copies, checks the ids and sizes the walk: k_walk_setup / k_seq_setup) -- same digests and texts as from host arrays
        stage("device inputs")
        import torch
        from variantstore_amd import DeviceArray
        for regs, who in ((regions[:100], per), (sc_regions, per_sc)):
            if not len(regs):
                continue
            reg_t = torch.from_numpy(np.array(regs, dtype=np.uint64).view(np.int64)).cuda()
            ids_t = torch.from_numpy(np.array([vs.sample_id(nm) for nm in who], dtype=np.uint32).view(np.int32)).cuda()
            dreg, dids = DeviceArray(reg_t.data_ptr(), len(regs)), DeviceArray(ids_t.data_ptr(), len(regs))
            for name, call in (("4", vs.get_sample_var_in_ref), ("5", vs.get_sample_var_in_sample)):
                a, b = call(regs, who), call(dreg, dids)
                checked += len(regs)
                if a.digest() != b.digest() or a.totals() != b.totals() or any(a.region_text(q) != b.region_text(q) for q in range(0, len(regs), 5)):
                    bad += 1
                    print(f"MISMATCH t{name} device inputs against host inputs, cohort {seed}")
                a.close(); b.close()
            for coords in (False, True):
                a, b = vs.query_sample_seq(regs, who, sample_coordinates=coords), vs.query_sample_seq(dreg, dids, sample_coordinates=coords)
                (fa, sa), (fb, sb) = a.sequences(), b.sequences()
                checked += len(regs)
                if not np.array_equal(fa, fb) or sa != sb:
                    bad += 1
                    print(f"MISMATCH t{3 if coords else 2} device inputs against host inputs, cohort {seed}")
                a.close(); b.close()
        ub += orc.ub_events()
        res.close(); r4.close(); vs.close()
print(f"cohorts {n_cohorts} regions checked {checked} mismatches {bad} non-terminating-in-reference {hang} "
      f"oracle ub_events {ub} regions with dropped duplicates {slow_regions} batches with shared rows/lists {shared_batches} short-region batches with more than 64 table rows {many_runs} speculative batches {speculated} of which refused and redone {refused}")
sys.exit(1 if bad else 0)
