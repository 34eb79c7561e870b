"""Multi-GPU sharding of a region batch (one process per GPU, torch.distributed).

Region queries are independent (the reference's serial loop, src/commands.cc:145,
carries no state between regions), so the batch is partitioned across ranks, the
index image is replicated, and the only exchange is an all-gatherv of the hit
lists: a count all-gather followed by one padded `all_gather_into_tensor` of
4 x uint64 header records (backend "nccl" is RCCL over xGMI on ROCm; the records
are a few MB per rank, so the collective is latency-bound, not link-bound).
Carrier lists stay sharded in the HBM of the rank that produced them.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) slice of n sorted regions for `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_regions(regions, rank, world):
    regions = np.asarray(regions, dtype=np.uint64).reshape(-1, 2)
    lo, hi = shard_bounds(regions.shape[0], rank, world)
    return regions[lo:hi], lo


def allgather_hit_lists(result, region_base, device, compact=False, counts=None, async_op=False):
    """All-gatherv of hit lists.

    compact=False: one 32-byte record per VARIANT (vs_result_pack_headers) -- self-contained rows.
    compact=True:  one 32-byte record per REGION (vs_result_pack_regions): the region's range of the
                   site table, which every rank can expand locally because the index is replicated.
                   For the bench cohort that is 3.2 MB per rank instead of 650 MB.
    counts: the per-rank record counts when every rank already knows them (compact records of a batch that was
            sharded with shard_bounds: the region counts) -- skips the count all-gather and its host round trip.
    async_op: start the collective and return at once -- (records, counts, work); the records are valid after
            work.wait().  The all-gather then runs on RCCL's stream beside whatever the caller launches next (the next
            batch's kernels): the caller keeps `records` alive until it has waited.
    Returns (records[int64, world x max_n x 4], counts[int64, world]).
    """
    world = dist.get_world_size()
    n = result.num_region_records() if compact else result.num_header_records()
    if counts is None:
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        mine = torch.tensor([n], dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(counts, mine)
        max_n = max(int(counts.max().item()), 1)
    else:
        if len(counts) != world or int(counts[dist.get_rank()]) != n:
            raise ValueError("counts does not describe this batch")
        max_n = max(int(max(counts)), 1)
        counts = torch.as_tensor(list(counts), dtype=torch.int64)
    # rows beyond this rank's count are never read (counts delimit them): no need to clear the send buffer
    buf = torch.empty((max_n, 4), dtype=torch.int64, device=device)
    if compact:
        result.pack_regions_into(buf.data_ptr(), max_n, region_base)
    else:
        result.pack_headers_into(buf.data_ptr(), max_n, region_base)
    out = torch.empty((world * max_n, 4), dtype=torch.int64, device=device)
    if async_op:
        work = dist.all_gather_into_tensor(out, buf, async_op=True)
        return out.view(world, max_n, 4), counts, (work, buf)   # (buf travels with the handle: it must outlive the collective)
    dist.all_gather_into_tensor(out, buf)
    return out.view(world, max_n, 4), counts


COMM_ID_BYTES = 128   # VS_COMM_ID_BYTES of include/variantstore_hip.h (sizeof(ncclUniqueId))


def _launch_nonce():
    """What tells one launch's id file from another's: VS_COMM_NONCE, else what torchrun gives every rank of a launch."""
    import os
    n = os.environ.get("VS_COMM_NONCE")
    if n is None:
        parts = [os.environ.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT")]
        if not any(parts):
            # nothing tells this launch from an earlier one: a rank could take the id file an earlier launch left at the same path
            raise ValueError("make_comm(id_file=...) outside torchrun needs a launch nonce: pass nonce= or set VS_COMM_NONCE "
                             "(the same value on every rank, a new one per launch)")
        n = ":".join(parts)
    return n


def exchange_id_file(rank, world, id_file, make_id, nonce=None, timeout=120.0):
    """The file rendezvous of `make_comm`: rank 0 writes `sha256(nonce)[:16] + make_id()` to `id_file` (atomically, over
    whatever an earlier launch left there), the other ranks wait for a file that carries THIS launch's nonce -- a stale
    file (another nonce, or a torn write) is never taken for the id.  Returns the id bytes."""
    import hashlib
    import os
    import time
    tag = hashlib.sha256((_launch_nonce() if nonce is None else nonce).encode()).digest()[:16]
    if rank == 0:
        uid = make_id()
        tmp = f"{id_file}.tmp.{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(tag + uid)
        os.replace(tmp, id_file)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(id_file, "rb") as f:
                blob = f.read()
        except OSError:
            blob = b""
        if len(blob) == 16 + COMM_ID_BYTES and blob[:16] == tag:
            return blob[16:]
        if time.time() - t0 > timeout:
            raise TimeoutError(f"no unique id of this launch at {id_file}")
        time.sleep(0.01)


def make_comm(store, rank, world, id_file=None, nonce=None):
    """A `Comm` (the engine's own RCCL communicator, vs_comm_*) for this rank.  The 128-byte unique id is made on rank 0
    and reaches the other ranks through torch.distributed when a process group exists (any backend), else through
    `id_file` (`exchange_id_file`: rank 0 writes it, the others wait for the file of THIS launch -- give every rank the
    same `nonce`, or VS_COMM_NONCE, when the path is reused between launches outside torchrun)."""
    from .api import Comm
    if dist.is_available() and dist.is_initialized():
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    elif world == 1 and id_file is None:
        uid = Comm.unique_id()
    else:
        if id_file is None:
            raise ValueError("make_comm needs a process group or an id_file")
        uid = exchange_id_file(rank, world, id_file, Comm.unique_id, nonce)
    return Comm(store, rank, world, uid)


def allgather_region_records(comm, result, region_base, device, counts, async_op=False):
    """The compact all-gatherv of `allgather_hit_lists(compact=True, counts=...)` through the C ABI (vs_comm_allgather_regions:
    ncclAllGather called by the engine on the communicator's own stream) instead of torch.distributed.  torch only lends
    the receive buffer.  Returns (records[int64, world x max_n x 4], counts); with async_op the records are valid after
    `comm.wait()` (or the next all-gather on this communicator)."""
    world = comm.world
    if len(counts) != world or int(counts[comm.rank]) != result.num_region_records():
        raise ValueError("counts does not describe this batch")
    max_n = max(int(max(counts)), 1)
    out = torch.empty((world * max_n, 4), dtype=torch.int64, device=device)
    comm.allgather_regions(result, region_base, max_n, out.data_ptr(), async_op=async_op)
    return out.view(world, max_n, 4), torch.as_tensor(list(counts), dtype=torch.int64)


def unpack_records(records, counts):
    """Host view of gathered records: list of dicts per rank with numpy arrays."""
    out = []
    rec = records.cpu().numpy().view(np.uint64)
    for r in range(rec.shape[0]):
        a = rec[r, : int(counts[r])]
        out.append({
            "pos": a[:, 0] & np.uint64((1 << 63) - 1),
            "dropped": (a[:, 0] >> np.uint64(63)).astype(bool),
            "ref_off": (a[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.uint32),
            "ref_len": (a[:, 1] >> np.uint64(32)).astype(np.uint32),
            "alt_off": (a[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.uint32),
            "alt_len": (a[:, 2] >> np.uint64(32)).astype(np.uint32),
            "region": (a[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.uint64),
            "car_count": (a[:, 3] >> np.uint64(32)).astype(np.uint32),
        })
    return out


def unpack_region_records(records, counts):
    """Host view of gathered COMPACT records (vs_result_pack_regions): per rank a dict of numpy arrays.  The device
    tensor itself is what `VariantStore.expand_site_ranges` takes to rebuild the rows on the receiving rank."""
    out = []
    rec = records.cpu().numpy().view(np.uint64)
    for r in range(rec.shape[0]):
        a = rec[r, : int(counts[r])]
        out.append({
            "region": a[:, 0].copy(),
            "first_site": (a[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.uint32),
            "region_flags": ((a[:, 1] >> np.uint64(32)) & np.uint64(0xFF)).astype(np.uint8),
            "has_dropped": ((a[:, 1] >> np.uint64(40)) & np.uint64(1)).astype(bool),
            "sites": (a[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.uint32),
            "variants": (a[:, 2] >> np.uint64(32)).astype(np.uint64),
            "carriers": a[:, 3].copy(),
        })
    return out


def verify_gathered_regions(per_rank, counts, bases, totals):
    """What rank 0 of `bench.py --gpus N` checks about a gather of compact records (`gathered_ok`): `per_rank` is
    `unpack_region_records(records, counts)`, `bases[k]` the first region of rank k's shard, `totals[k]` the (variants,
    carriers) rank k counted over its own result.  Rank k's records must lie in slot k (the padded all-gather places rank k
    at record k x max_count), carry the regions [bases[k], bases[k] + counts[k]) in order and add up to rank k's totals.
    Returns the list of faults (empty: the gather is what the ranks computed)."""
    faults = []
    for k, rec in enumerate(per_rank):
        n_k = int(counts[k])
        want = np.arange(int(bases[k]), int(bases[k]) + n_k, dtype=np.uint64)
        if len(rec["region"]) != n_k or not np.array_equal(rec["region"], want):
            faults.append(f"rank {k}'s records do not carry regions [{int(bases[k])}, {int(bases[k]) + n_k}) at record {k} x max_count")
            continue
        v, c = int(rec["variants"].sum()), int(rec["carriers"].sum())
        if (v, c) != (int(totals[k][0]), int(totals[k][1])):
            faults.append(f"rank {k}'s records add up to {v} variants / {c} carriers, the rank itself counted {int(totals[k][0])} / {int(totals[k][1])}")
    return faults
