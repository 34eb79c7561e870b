"""One rank of tests/test_gpu_two_ranks.py: a FRESH process (nothing has touched the GPU before it starts) that opens the
cohort on GPU 0, answers its contiguous shard of a sorted batch, all-gathers the per-region records through the C ABI's
collective (vs_comm_*, with tests/native/fake_rccl.cpp behind VS_RCCL_LIB: two ranks on one device) and checks, on THIS rank,
that the gathered records rebuild the whole batch's answer.  Writes a JSON verdict; exit code 0 only when every check held.

  python two_rank_worker.py <rank> <world> <fasta> <vcf> <id_file> <nonce> <out.json>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    fasta, vcf, id_file, nonce, out_path = sys.argv[3:8]
    import numpy as np
    import torch
    from helpers import random_regions
    from variantstore_amd import VariantStore
    from variantstore_amd.parallel import allgather_region_records, make_comm, shard_bounds, unpack_region_records

    verdict = {"rank": rank, "world": world, "ok": False}
    vs = VariantStore.from_vcf(fasta, vcf, device=0)
    regions = sorted(random_regions(np.random.default_rng(21), vs.info().ref_length, 403))   # (403: the shards differ in size)
    n = len(regions)
    counts = [shard_bounds(n, k, world)[1] - shard_bounds(n, k, world)[0] for k in range(world)]
    lo, hi = shard_bounds(n, rank, world)
    dev = torch.device("cuda", 0)
    comm = make_comm(vs, rank, world, id_file=id_file, nonce=nonce)
    verdict["info"] = list(comm.info())
    whole = vs.get_var_in_ref(regions)                    # what a single process answers
    mine = vs.get_var_in_ref(regions[lo:hi])
    checks = {}
    for async_op in (False, True, True):
        gathered, cnt = allgather_region_records(comm, mine, lo, dev, counts, async_op=async_op)
        if async_op:
            nxt = vs.get_var_in_ref(regions[lo:hi])      # the next batch's kernels run beside the gather
            comm.wait()
            nxt.close()
        max_n = max(counts)
        assert tuple(gathered.shape) == (world, max_n, 4)
        recs = unpack_region_records(gathered, cnt)
        tag = f"async={async_op}"
        # rank k's records lie at k x max_count and carry regions [lo_k, hi_k) of the sorted batch
        for k in range(world):
            klo, khi = shard_bounds(n, k, world)
            assert np.array_equal(recs[k]["region"], np.arange(klo, khi, dtype=np.uint64)), (tag, "region numbers of rank", k)
        v = whole.view(with_carriers=False)
        got_vars = np.concatenate([r["variants"] for r in recs])
        assert np.array_equal(got_vars, v["var_count"]), (tag, "variants per region")
        assert int(sum(int(r["carriers"].sum()) for r in recs)) == whole.totals()[2], (tag, "carriers")
        # every rank rebuilds the WHOLE batch from the gathered records (the index is replicated): same digest, same text
        flat = torch.cat([gathered[k, : counts[k]] for k in range(world)]).contiguous()
        back = vs.expand_site_ranges(flat.data_ptr(), n)
        assert back.totals() == whole.totals(), (tag, back.totals(), whole.totals())
        assert back.digest() == whole.digest(), (tag, "digest")
        for q in range(0, n, 7):
            assert back.region_text(q) == whole.region_text(q), (tag, "text of region", q)
        back.close()
        checks[tag] = True
    verdict["checks"] = checks
    verdict["digest"] = whole.digest()
    verdict["totals"] = list(whole.totals())
    mine.close()
    whole.close()
    comm.close()
    vs.close()
    verdict["ok"] = True
    with open(out_path, "w") as f:
        json.dump(verdict, f)


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:   # the verdict file says why (the parent prints it)
        import traceback
        with open(sys.argv[7], "w") as f:
            json.dump({"rank": int(sys.argv[1]), "ok": False, "error": repr(e), "trace": traceback.format_exc()}, f)
        raise
