#!/usr/bin/env python3
"""bench.py -- region-queries/sec of the type-6 hot path on MI355X.

One "step" = one pass of the hot path (vs_query_var_in_ref) over one batch of
synthetic regions on a resident index; with N > 1 every rank owns its own shard
of the batch (weak scaling: the per-GPU batch is fixed) and the step ends with
the RCCL all-gatherv of the hit lists the north star asks for.

Workloads (BASELINE.json `configs`):
  chr1-2504   [default]  249,250,621 bp, 5,000,000 sites (90% SNP / 5% ins / 5% del, 1% two-ALT),
                         2504 samples, AF = min(0.5, 10^-11U)  (~135 carriers per variant),
                         100,000 random 10 kb regions per GPU      (configs[2]; the metric's config)
  chr22-100              51,304,566 bp, 100,000 SNPs in [16,050,000, L), 100 samples, AF = min(.5, 10^-3U),
                         10,000 random 1 kb regions                (configs[1])
Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline`
for the dominant kernel (k_fill_carriers) and `cpu_baseline` (the CPU oracle timed
on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "chr1-2504": dict(ref_length=249_250_621, num_variants=5_000_000, num_samples=2504, seed=1, first_pos=10_000,
                      frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6, af_exponent=11.0,
                      regions=100_000, region_len=10_000, region_seed=2),
    "chr22-100": dict(ref_length=51_304_566, num_variants=100_000, num_samples=100, seed=22, first_pos=16_050_000,
                      frac_ins=0.0, frac_del=0.0, frac_multi=0.0, max_indel=1, af_exponent=3.0,
                      regions=10_000, region_len=1_000, region_seed=1),
    # BASELINE.json configs[4]'s cohort shape (somatic-like: 10k samples, 20M variants incl. 10 % indels, one to a few
    # carriers each -> explicit sample ids).  Not a default: ~2.5 min of host-side index construction.  It fits HBM
    # whole (nothing is streamed); the headline stays the type-6 batch, types 3 and 7 are in the extras.
    "tcga-10k": dict(ref_length=243_000_000, num_variants=20_000_000, num_samples=10_000, seed=5, first_pos=10_000,
                     frac_ins=0.05, frac_del=0.05, frac_multi=0.01, max_indel=6, af_exponent=2.0, max_af=0.0004,
                     regions=100_000, region_len=10_000, region_seed=3),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def make_regions(w, rank, n):
    import numpy as np
    rng = np.random.default_rng(w["region_seed"] * 1000 + rank)
    lo = max(1, w["first_pos"] - w["region_len"])
    starts = rng.integers(lo, w["ref_length"] - w["region_len"], size=n, dtype=np.int64)
    starts.sort()  # the reference's read_regions sorts (src/commands.cc:91)
    return np.stack([starts, starts + w["region_len"]], axis=1).astype(np.uint64)


def synth_kwargs(w):
    return {k: w[k] for k in ("ref_length", "num_variants", "num_samples", "seed", "first_pos", "frac_ins",
                              "frac_del", "frac_multi", "max_indel", "af_exponent", "max_af") if k in w}


def cpu_baseline(w, budget_s=20.0):
    """The CPU oracle (literal restatement of the reference path, 1 thread) on a bounded sample of the same
    workload: same generator, same cohort size, same variant density and region length, on a 1/25-length
    slice of the chromosome so that building + loading it stays within the bench's time budget."""
    import tempfile
    from oracle.oracle import Oracle
    from variantstore_amd import VariantStore
    scale = 25 if w["num_variants"] >= 1_000_000 else 1
    kw = synth_kwargs(w)
    kw["ref_length"] = max(200_000, w["ref_length"] // scale)
    kw["num_variants"] = max(1000, w["num_variants"] // scale)
    kw["first_pos"] = min(w["first_pos"], kw["ref_length"] // 10) if scale > 1 else w["first_pos"]
    vs = VariantStore.synthetic(device=-1, **kw)
    sub = dict(w, **kw)
    regions = make_regions(sub, 12345, 4000)
    with tempfile.TemporaryDirectory() as td:
        plain = os.path.join(td, "slice.plain")
        vs.export_plain(plain)
        vs.close()
        orc = Oracle(plain)
        done = nvar = 0
        t0 = time.perf_counter()
        for x, y in regions:
            n, _, _ = orc.get_var_in_ref(int(x), int(y), text=False)
            nvar += max(n, 0)
            done += 1
            if done >= 20 and time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
        orc.close()
        all_cores = cpu_baseline_all_cores(sub, plain, td) if os.environ.get("VS_BENCH_SKIP_ALLCORES") != "1" else None
    out = {"value": done / dt, "unit": "queries/s", "cores": 1, "kind": "port",
           "sample": f"{done} regions x {w['region_len']} bp ({nvar / max(done, 1):.0f} variants/region) on a "
                     f"1/{scale}-length slice of the same synthetic cohort ({kw['num_variants']} sites, "
                     f"{w['num_samples']} samples), CPU oracle, {dt:.1f} s"}
    if all_cores:
        out["all_cores"] = all_cores
    return out


def cpu_baseline_all_cores(sub, plain, td, budget_s=10.0):
    """the same oracle on every host core, the region batch statically sharded -- one child
    process per core (capped at 32: each holds its own ~0.5 GB decoded index), started together once all have
    loaded.  Reported beside the 1-thread figure, never instead of it."""
    import subprocess
    import numpy as np
    cores = min(os.cpu_count() or 1, 32)
    per = 6000
    regs = make_regions(sub, 54321, per * cores)
    rpath = os.path.join(td, "regions.npy")
    np.save(rpath, regs)
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.bench_worker", plain, rpath, str(i * per), str((i + 1) * per),
                               str(budget_s)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env,
                              cwd=os.path.dirname(os.path.abspath(__file__))) for i in range(cores)]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("a CPU baseline worker failed to load the oracle")
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
            p.wait(timeout=60)
    done = sum(r["done"] for r in res)
    wall = max(r["seconds"] for r in res)
    return {"value": done / wall, "unit": "queries/s", "cores": cores, "host_cpus": os.cpu_count(),
            "sample": f"{done} regions over {cores} processes, {wall:.1f} s"}


def main():
    # The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner through C stdio when a
    # process group is created) must not be able to add to it: file descriptor 1 is pointed at stderr for the whole
    # run and the JSON line goes to a private duplicate of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("VS_BENCH_WORKLOAD", "chr1-2504"), choices=sorted(WORKLOADS))
    ap.add_argument("--regions", type=int, default=0, help="regions per GPU per step (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-samples", type=int, default=200)
    ap.add_argument("--skip-extras", action="store_true", help="only the headline batch (no type 4 / point / sample-coordinate legs)")
    args = ap.parse_args()
    if args.skip_extras:
        os.environ["VS_BENCH_SKIP_T4"] = "1"

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the query path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("VS_BENCH_FORCE_DIST") == "1"  # the latter: exercise RCCL on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from variantstore_amd import VariantStore
    from variantstore_amd.parallel import allgather_hit_lists

    w = WORKLOADS[args.workload]
    nreg = args.regions or w["regions"]
    t_build = time.perf_counter()
    vs = VariantStore.synthetic(device=local_rank, **synth_kwargs(w))
    t_build = time.perf_counter() - t_build
    info = vs.info()
    regions = make_regions(w, rank, nreg)
    region_base = rank * nreg
    # the batch's input is resident in HBM before the timed region starts (uploaded once); VS_BENCH_HOST_REGIONS=1 hands
    # the host array over in every step instead (the PCIe-inclusive rate quoted in DESIGN.md, never `value`)
    host_regions = os.environ.get("VS_BENCH_HOST_REGIONS") == "1"
    regions_dev = torch.from_numpy(regions.astype(np.int64)).to(torch.device("cuda", local_rank)).contiguous()
    torch.cuda.synchronize()

    def step():
        res = vs.get_var_in_ref(regions) if host_regions else vs.get_var_in_ref_device(regions_dev.data_ptr(), nreg)
        gathered = None
        if use_dist:
            # every rank answers nreg regions: the record counts are known without asking
            gathered = allgather_hit_lists(res, region_base, torch.device("cuda", local_rank), compact=True,
                                           counts=[nreg] * world)
        return res, gathered

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _i in range(args.warmup):
        res, _g = step()
        res.close()
    fence()
    fill_ms = tot_ms = 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        res, _g = step()
        t = vs.last_timing()
        fill_ms += t.ms_fill
        tot_ms += t.ms_total
        if i != args.steps - 1:
            res.close()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    # ---- result-derived figures of one batch (last step's result is still alive) ----
    nq, nvar, ncar, nbases = res.totals()
    digest = res.digest()
    v = res.view(with_carriers=False)
    kept = (v["var_flags"] & 1) == 0
    cc = v["car_count"][kept].astype(np.int64)
    W = ((info.num_samples + 63) // 64) * 8
    car_word = 2 if info.num_samples <= 4032 else 4
    # Dominant kernel k_fill_carriers.  ALGORITHMIC BYTES of one launch = the bytes this kernel's data layout obliges it
    # to move: per variant slot 24 B of slot parameters (count, class, genotype offset, arena
    # offset); per variant of at most list_max carriers its decoded id list, rounded up to whole groups of 8 entries;
    # per denser variant its class bit row (W bytes); half a byte of genotype per carrier
    # in; one carrier word (2 B: id | gt << 13; 4 B above 4032 samples) per ARENA entry out -- every variant's range
    # is padded to a multiple of 8 entries and the kernel writes whole groups.  roofline.achieved = that / the
    # kernel's mean launch time, so roofline.frac can never exceed what the HBM pins carried.
    padded = (cc + 7) // 8 * 8
    if info.use_bit_vector:
        listed = cc <= info.list_max
        id_bytes = int((padded[listed] * car_word).sum()) + int((~listed).sum()) * W   # list entries are as wide as carrier words
    else:  # explicit sample ids: 4 B per carrier record, read in whole groups of 8
        id_bytes = int(4 * padded.sum())
    fill_bytes_layout = int(len(cc)) * 24 + id_bytes + (ncar + 1) // 2 + car_word * int(padded.sum())
    # SURVEY.md section 8(d)'s implementation-independent formula, restricted to the terms this kernel owns: per
    # variant its class row (W) + car_begin word (8), per carrier 3 genotype bits in and 4 + 1 bytes out.  It prices
    # bytes this layout never moves (5 B per carrier written where the arena holds 2), so it is reported for
    # comparison only, under its own key, and is NOT the roofline fraction.
    fill_bytes_survey = nvar * (W + 8) + (3 * ncar + 7) // 8 + 5 * ncar
    fill_s = fill_ms / args.steps / 1e3
    achieved = fill_bytes_layout / fill_s / 1e9 if fill_s > 0 else 0.0
    survey_gbps = fill_bytes_survey / fill_s / 1e9 if fill_s > 0 else 0.0
    res.close()
    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command
    # (counters cannot be read from inside the run); the committed measurement is attached when it is for
    # this workload, otherwise the field stays null.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.exists(tpath) and nreg == w["regions"]:
        with open(tpath) as tf:
            traffic = json.load(tf).get("traffic_bytes_per_launch")

    # ---- p50 single-region latency (submit -> result resident), outside the timed region ----
    lat = []
    for i in range(args.latency_samples):
        one = regions[(i * 7919) % nreg: (i * 7919) % nreg + 1]
        a = time.perf_counter()
        r1 = vs.get_var_in_ref(one)
        lat.append(time.perf_counter() - a)
        r1.close()
    p50 = float(np.median(lat)) * 1e6 if lat else None

    # ---- query type 4 on the same index (BASELINE.json configs[2] names types 4+6): 16 fixed samples,
    #      one sub-batch each, outside the timed region of the headline metric ----
    t4 = None
    if os.environ.get("VS_BENCH_SKIP_T4") != "1":
        sids16 = [1 + (i * 157) % (info.num_samples - 1) for i in range(16)]
        per_region = np.array([sids16[i % 16] for i in range(nreg)], dtype=np.uint32)   # round-robin over 16 samples
        r4 = vs.get_sample_var_in_ref(regions, per_region)  # warm-up
        nv4 = r4.totals()[1]
        r4.close()
        torch.cuda.synchronize()
        a4 = time.perf_counter()
        for _k in range(3):
            r4 = vs.get_sample_var_in_ref(regions, per_region)
            r4.close()
        torch.cuda.synchronize()
        t4 = {"queries_per_s": 3 * nreg / (time.perf_counter() - a4), "regions_per_batch": nreg, "samples": 16,
              "variants_per_region": nv4 / nreg}

    # ---- point queries (types 1 and 7) on the same index, outside the timed region:
    #      1M random positions; type 7 asks for an A>C substitution everywhere (nearly always "no such variant",
    #      which costs the same walk) ----
    t17 = None
    if os.environ.get("VS_BENCH_SKIP_T4") != "1":
        npos = 1_000_000
        prng = np.random.default_rng(17)
        positions = prng.integers(1, w["ref_length"], size=npos, dtype=np.uint64)
        r1 = vs.closest_var(positions)
        nv1 = r1.totals()[1]
        r1.close()
        torch.cuda.synchronize()
        a1 = time.perf_counter()
        for _k in range(3):
            r1 = vs.closest_var(positions)
            r1.close()
        torch.cuda.synchronize()
        t1_qps = 3 * npos / (time.perf_counter() - a1)
        refs7, alts7 = ["A"] * npos, ["C"] * npos
        r7 = vs.samples_has_var(positions, refs7, alts7)
        r7.close()
        torch.cuda.synchronize()
        r7 = vs.samples_has_var(positions, refs7, alts7)
        ms7 = vs.last_timing().ms_total   # device pipeline; the host side here is a million Python strings
        nf7 = int((r7.view(False)["region_flags"] & 4 == 0).sum())
        r7.close()
        t17 = {"type1_queries_per_s": t1_qps, "type1_variants_per_query": nv1 / npos,
               "type7_queries_per_s_device": npos / (ms7 * 1e-3), "type7_found": nf7, "positions_per_batch": npos}

    # ---- sample-coordinate queries (types 2, 3 and 5), outside the timed region, on a
    #      1/12.5-length cohort of the same shape built WITH sample coordinates (4 B per carrier record) ----
    tsc = None
    if os.environ.get("VS_BENCH_SKIP_T4") != "1" and rank == 0:
        kw = synth_kwargs(w)
        kw["ref_length"] = max(200_000, w["ref_length"] * 2 // 25)
        kw["num_variants"] = max(1000, w["num_variants"] * 2 // 25)
        kw["first_pos"] = min(w["first_pos"], kw["ref_length"] // 10)
        vsc = VariantStore.synthetic(device=local_rank, sample_coordinates=True, **kw)
        nsc = 20_000
        srng = np.random.default_rng(23)
        st = srng.integers(max(1, kw["first_pos"]), kw["ref_length"] - w["region_len"], size=nsc, dtype=np.int64)
        sreg = np.stack([st, st + w["region_len"]], axis=1).astype(np.uint64)
        ns = vsc.info().num_samples
        sper = np.array([1 + ((i % 16) * 157) % (ns - 1) for i in range(nsc)], dtype=np.uint32)
        tsc = {"regions_per_batch": nsc, "samples": 16, "index_sites": int(vsc.info().num_sites)}
        for name, call in (("type2", lambda: vsc.query_sample_seq(sreg, sper, sample_coordinates=False)),
                           ("type3", lambda: vsc.query_sample_seq(sreg, sper, sample_coordinates=True)),
                           ("type5", lambda: vsc.get_sample_var_in_sample(sreg, sper))):
            rr = call()
            rr.close()
            torch.cuda.synchronize()
            b0 = time.perf_counter()
            for _k in range(3):
                rr = call()
                tot = rr.totals()
                rr.close()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - b0) / 3
            tsc[name + "_queries_per_s"] = nsc / dt
            if name != "type5":
                tsc[name + "_bases_per_s"] = tot[3] / dt
            else:
                tsc["type5_variants_per_region"] = tot[1] / nsc
        vsc.close()

    if rank == 0:
        out = {
            "metric": "region-queries/sec (batch, query-type 6)",
            "value": world * nreg * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {w['ref_length']} bp, {w['num_variants']} sites, "
                            f"{w['num_samples']} samples, {nreg} random {w['region_len']} bp regions per GPU, "
                            "query type 6 (get_var_in_ref), index + regions resident in HBM, results left in HBM"
                            + (" [regions handed over as a host array in every step]" if host_regions else ""),
                "regions_per_gpu": nreg, "region_len": w["region_len"],
                "variants_per_region": nvar / max(nq, 1), "carriers_per_variant": ncar / max(nvar, 1),
                "sharding": f"regions x{world}, index replicated" + (", RCCL all-gatherv of hit lists (per-region site ranges)" if use_dist else ""),
                "index": {"vertices": info.num_vertices, "csr_edges": info.num_edges_csr, "sites": info.num_sites,
                          "classes": info.num_classes, "carrier_records": info.num_carriers,
                          "hbm_image_bytes": info.device_bytes, "build_s": round(t_build, 1)},
            },
            "roofline": {"bound": "hbm", "kernel": "k_fill_carriers", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "bytes_per_launch": fill_bytes_layout, "avg_launch_ms": fill_ms / args.steps,
                         "frac_of_measured_copy_ceiling_6290": achieved / 6290.0,
                         # what crossed the HBM pins (PMC, profiles/traffic_<workload>.json) over the same launch time
                         "traffic_GBps": (traffic / fill_s / 1e9) if (traffic and fill_s > 0) else None,
                         "traffic_frac": (traffic / fill_s / 1e9 / HBM_PEAK_GBPS) if (traffic and fill_s > 0) else None,
                         "survey_formula": {"bytes_per_launch": fill_bytes_survey, "GBps": survey_gbps,
                                            "note": "SURVEY 8(d) terms of this kernel; prices 5 B per carrier written where "
                                                    "the arena holds 2 -- a time-per-algorithmic-unit figure, not pin traffic"},
                         "pipeline_ms": tot_ms / args.steps},
            "p50_latency_us": p50,
            "type4": t4,
            "point_queries": t17,
            "sample_coordinate_queries": tsc,
            "result_digest": f"{digest:016x}",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out), file=real_stdout, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
