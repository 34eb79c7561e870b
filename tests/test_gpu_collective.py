"""The multi-GPU path's data round trip on ONE GPU: a real result -> vs_result_pack_regions -> the RCCL all-gather of
variantstore_amd.parallel (backend "nccl", world size 1) -> vs_query_expand_site_ranges on the gathered tensor -> the
same rows as the local result and as the CPU oracle.  (The world-size-2 plumbing -- shard bounds, counts, padding --
runs on CPU with gloo in test_parallel_gloo.py.)"""
import os
import socket

import numpy as np
import pytest

from helpers import random_regions, write_random_cohort
from oracle.oracle import Oracle
from variantstore_amd import VariantStore

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_gathered_site_ranges_expand_to_the_local_result(nccl_world1, tmp_path):
    import torch
    from variantstore_amd.parallel import allgather_hit_lists, shard_bounds, shard_regions, unpack_region_records
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 411, n_rows=400, ref_len=6000, n_samples=90, carrier_p=0.35,
                                        p_near=0.6, p_multi=0.25, p_same=0.3)   # repeated rows: the duplicate rule fires
    vs = VariantStore.from_vcf(fasta, vcf, device=0)
    plain = os.path.join(tmp_path, "plain.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    regions = sorted(random_regions(np.random.default_rng(5), vs.info().ref_length, 500))
    dev = torch.device("cuda", 0)
    # rank 0 of 1: its shard is the whole sorted batch
    mine, lo = shard_regions(regions, 0, 1)
    assert (lo, len(mine)) == (0, len(regions)) and shard_bounds(len(regions), 0, 1) == (0, len(regions))
    local = vs.get_var_in_ref(mine)
    for counts in (None, [len(regions)]):             # with and without the count all-gather
        gathered, cnt = allgather_hit_lists(local, lo, dev, compact=True, counts=counts)
        assert gathered.shape == (1, len(regions), 4) and int(cnt[0]) == len(regions)
        rec = unpack_region_records(gathered, cnt)[0]
        assert np.array_equal(rec["region"], np.arange(len(regions), dtype=np.uint64))
        assert int(rec["variants"].sum()) == local.totals()[1] and int(rec["carriers"].sum()) == local.totals()[2]
        back = vs.expand_site_ranges(gathered[0].contiguous().data_ptr(), len(regions))
        assert back.totals() == local.totals() and back.digest() == local.digest()
        for q, (x, y) in enumerate(regions):
            n, _, text = orc.get_var_in_ref(x, y)
            if n < 0:
                continue
            assert back.region_text(q) == local.region_text(q) == text, (q, x, y)
        # (regions that lost rows to the duplicate rule -- rec["has_dropped"] -- go through it again on the receiving side)
        assert np.array_equal(rec["has_dropped"], rec["variants"] != rec["sites"])
        back.close()
    # per-variant records through the same collective
    full, cnt = allgather_hit_lists(local, lo, dev, compact=False)
    v = local.view(False)
    from variantstore_amd.parallel import unpack_records
    rows = unpack_records(full, cnt)[0]
    assert np.array_equal(rows["pos"], v["pos"]) and np.array_equal(rows["car_count"], v["car_count"])
    local.close()
    vs.close()


def test_c_abi_collective_round_trip(tmp_path):
    """The same round trip through the C ABI's own collective (vs_comm_*: the engine calls RCCL directly, no torch.distributed):
    world size 1, real records -> vs_comm_allgather_regions (synchronous, then asynchronous with vs_comm_wait) ->
    vs_query_expand_site_ranges -> the local result and the oracle's text.  The unique id travels through a file, as it
    would between the processes of the CLI's --nprocs form."""
    import torch
    from variantstore_amd.parallel import allgather_region_records, make_comm, unpack_region_records
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 412, n_rows=400, ref_len=6000, n_samples=90, carrier_p=0.35,
                                        p_near=0.6, p_multi=0.25, p_same=0.3)
    vs = VariantStore.from_vcf(fasta, vcf, device=0)
    plain = os.path.join(tmp_path, "plain.bin")
    vs.export_plain(plain)
    orc = Oracle(plain)
    regions = sorted(random_regions(np.random.default_rng(6), vs.info().ref_length, 400))
    dev = torch.device("cuda", 0)
    comm = make_comm(vs, 0, 1, id_file=os.path.join(tmp_path, "uid"), nonce="test")   # (the file branch of the rendezvous, world 1)
    assert comm.info() == (0, 1, 1), "rank, world and ncclCommCount of the communicator"
    local = vs.get_var_in_ref(regions)
    # a result freed while its gather is in flight: the pack kernel reads the result's arrays, so its completion moves behind
    # the kernel -- the records are those of the result all the same
    early = vs.get_var_in_ref(regions)
    g_early, c_early = allgather_region_records(comm, early, 0, dev, [len(regions)], async_op=True)
    early.close()
    for _k in range(3):   # (batches that would take the freed arrays from the pool)
        vs.get_var_in_ref(regions[: len(regions) // 2]).close()
    comm.wait()
    rec = unpack_region_records(g_early, c_early)[0]
    assert int(rec["variants"].sum()) == local.totals()[1] and int(rec["carriers"].sum()) == local.totals()[2]
    for async_op in (False, True):
        gathered, cnt = allgather_region_records(comm, local, 0, dev, [len(regions)], async_op=async_op)
        if async_op:
            comm.wait()
        rec = unpack_region_records(gathered, cnt)[0]
        assert np.array_equal(rec["region"], np.arange(len(regions), dtype=np.uint64))
        assert int(rec["variants"].sum()) == local.totals()[1] and int(rec["carriers"].sum()) == local.totals()[2]
        back = vs.expand_site_ranges(gathered[0].contiguous().data_ptr(), len(regions))
        assert back.totals() == local.totals() and back.digest() == local.digest()
        for q, (x, y) in enumerate(regions):
            n, _, text = orc.get_var_in_ref(x, y)
            if n >= 0:
                assert back.region_text(q) == local.region_text(q) == text, (q, x, y)
        back.close()
    # a padded gather: max_count larger than this rank's count, the tail is never read
    out = torch.zeros((len(regions) + 7, 4), dtype=torch.int64, device=dev)
    comm.allgather_regions(local, 1000, len(regions) + 7, out.data_ptr())
    assert int(out[0, 0]) == 1000 and int(out[len(regions) - 1, 0]) == 1000 + len(regions) - 1
    with pytest.raises(Exception):
        comm.allgather_regions(local, 0, len(regions) - 1, out.data_ptr())     # max_count below this rank's count
    # the handle's close is deferred while a communicator lives on it (vs_index::live_comms): closing in the "wrong" order is safe
    local.close()
    vs.close()
    comm.close()


def test_summary_records_of_every_query_type(tmp_path):
    """The per-region records a sharded run gathers for the query types other than 6 (BASELINE configs[4] mixes 3 / 6 / 7; the CLI's
    --nprocs prints its log lines from them): through vs_comm_allgather_regions at world size 1, against what the result itself says
    -- region numbers from the base, region flags, variants reported and carriers (types 4, 5, 1, 7), pieces and sequence bytes
    (types 2, 3)."""
    import torch
    from variantstore_amd.parallel import make_comm
    fasta, vcf, names = write_random_cohort(str(tmp_path), 433, n_rows=300, ref_len=5000, n_samples=40, carrier_p=0.3, p_near=0.5, p_multi=0.2)
    vs = VariantStore.from_vcf(fasta, vcf, device=0)
    regions = sorted(random_regions(np.random.default_rng(8), vs.info().ref_length, 150))
    n = len(regions)
    dev = torch.device("cuda", 0)
    comm = make_comm(vs, 0, 1)
    per = [names[i % len(names)] for i in range(n)]
    out = torch.zeros((n + 3, 4), dtype=torch.int64, device=dev)

    def gathered(res, base):
        comm.allgather_regions(res, base, n + 3, out.data_ptr())
        return out[:n].cpu().numpy().view(np.uint64)

    for kind, res in (("4", vs.get_sample_var_in_ref(regions, per)), ("5", vs.get_sample_var_in_sample(regions, per)),
                      ("1", vs.closest_var([x for x, _y in regions]))):
        rec = gathered(res, 500)
        v = res.view(with_carriers=False)
        assert np.array_equal(rec[:, 0], np.arange(500, 500 + n, dtype=np.uint64)), kind
        assert np.array_equal((rec[:, 1] >> np.uint64(32)) & np.uint64(0x7F), v["region_flags"].astype(np.uint64) & np.uint64(0x7F)), kind
        assert np.array_equal(rec[:, 2] >> np.uint64(32), v["var_count"]), kind
        assert int(rec[:, 3].sum()) == res.totals()[2], kind
        res.close()
    for sc in (False, True):
        res = vs.query_sample_seq(regions, per, sample_coordinates=sc)
        rec = gathered(res, 7)
        flags, seqs = res.sequences()
        assert np.array_equal(rec[:, 0], np.arange(7, 7 + n, dtype=np.uint64))
        assert np.array_equal((rec[:, 1] >> np.uint64(32)) & np.uint64(0xFF), flags.astype(np.uint64))
        assert [int(b) for b in rec[:, 3]] == [len(s) for s in seqs]
        assert all(int(p) >= 1 for p, s in zip(rec[:, 2], seqs) if s)
        res.close()
    comm.close()
    vs.close()
