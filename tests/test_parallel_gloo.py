"""world_size-2 `gloo` test of the sharding + all-gatherv plumbing (CPU, no GPU):
the collective code path of bench.py / parallel.py with a stand-in result object
that packs known records into host memory."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class FakeResult:
    def __init__(self, recs):
        self.recs = np.ascontiguousarray(recs, dtype=np.uint64)

    def num_header_records(self):
        return self.recs.shape[0]

    def num_region_records(self):
        return self.recs.shape[0]

    def pack_regions_into(self, ptr, cap, region_base):
        return self.pack_headers_into(ptr, cap, region_base)

    def pack_headers_into(self, ptr, cap, region_base):
        assert cap >= self.recs.shape[0]
        r = self.recs.copy()
        r[:, 3] += np.uint64(region_base)
        ctypes.memmove(ptr, r.ctypes.data, r.nbytes)
        return r.shape[0]


def _records(rank):
    n = 5 + 3 * rank
    r = np.zeros((n, 4), dtype=np.uint64)
    r[:, 0] = np.arange(n) + 1000 * (rank + 1)
    r[:, 1] = (np.uint64(1) << np.uint64(32)) | np.uint64(7 + rank)
    r[:, 2] = (np.uint64(2) << np.uint64(32)) | np.uint64(9)
    r[:, 3] = (np.arange(n) % 3).astype(np.uint64) | (np.uint64(11) << np.uint64(32))
    return r


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from variantstore_amd.parallel import allgather_hit_lists, shard_regions, unpack_records
    regions = np.stack([np.arange(1, 12), np.arange(1, 12) + 100], axis=1)
    mine, lo = shard_regions(regions, rank, world)
    out, counts = allgather_hit_lists(FakeResult(_records(rank)), lo, torch.device("cpu"), compact=(rank >= 0 and world == 2))
    parts = unpack_records(out, counts)
    ok = [int(c) for c in counts] == [5 + 3 * r for r in range(world)]
    for r in range(world):
        want = _records(r)
        rlo = shard_regions(regions, r, world)[1]
        ok &= bool(np.array_equal(parts[r]["pos"], want[:, 0]))
        ok &= bool(np.array_equal(parts[r]["region"], (want[:, 3] & np.uint64(0xFFFFFFFF)) + np.uint64(rlo)))
        ok &= bool(np.all(parts[r]["car_count"] == 11)) and bool(np.all(parts[r]["ref_len"] == 1))
    # the asynchronous form (bench.py overlaps the collective with the next step): same records after work.wait()
    out2, counts2, (work, _buf) = allgather_hit_lists(FakeResult(_records(rank)), lo, torch.device("cpu"), compact=True, async_op=True)
    work.wait()
    ok &= [int(c) for c in counts2] == [int(c) for c in counts]
    for r, (pa, pb) in enumerate(zip(unpack_records(out2, counts2), parts)):   # (rows beyond a rank's count are padding)
        ok &= all(bool(np.array_equal(pa[k], pb[k])) for k in pa)
    q.put((rank, ok, mine.shape[0], lo))
    dist.destroy_process_group()


def _region_records(rank, n, base):
    """compact per-region records as k_pack_regions writes them (before region_base is added)"""
    r = np.zeros((n, 4), dtype=np.uint64)
    r[:, 0] = np.arange(n)
    r[:, 1] = np.uint64(100 * rank) + np.arange(n, dtype=np.uint64)                                   # first site, no flags
    r[:, 2] = np.uint64(3) | ((np.arange(n, dtype=np.uint64) % np.uint64(4)) << np.uint64(32))       # 3 sites | variants reported
    r[:, 3] = np.uint64(10 + rank)                                                                    # carriers
    return r


class FakeRegionResult(FakeResult):
    def pack_regions_into(self, ptr, cap, region_base):
        assert cap >= self.recs.shape[0]
        r = self.recs.copy()
        r[:, 0] += np.uint64(region_base)
        ctypes.memmove(ptr, r.ctypes.data, r.nbytes)
        return r.shape[0]


def _worker_verify(rank, world, port, q):
    """the self-check of the N > 1 bench line (gathered_ok) over a real world-2 gather: placement by rank, region numbers, totals"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from variantstore_amd.parallel import allgather_hit_lists, shard_bounds, unpack_region_records, verify_gathered_regions
    total = 11
    counts = [shard_bounds(total, k, world)[1] - shard_bounds(total, k, world)[0] for k in range(world)]
    bases = [shard_bounds(total, k, world)[0] for k in range(world)]
    mine = _region_records(rank, counts[rank], bases[rank])
    out, c = allgather_hit_lists(FakeRegionResult(mine), bases[rank], torch.device("cpu"), compact=True, counts=counts)
    totals = [(int((_region_records(k, counts[k], 0)[:, 2] >> np.uint64(32)).sum()), counts[k] * (10 + k)) for k in range(world)]
    per = unpack_region_records(out, c)
    ok = verify_gathered_regions(per, counts, bases, totals) == []
    # faults it must see: ranks swapped in the receive buffer, a shard numbered from the wrong base, totals that do not add up
    ok &= len(verify_gathered_regions(per[::-1], counts, bases, totals)) == world
    ok &= len(verify_gathered_regions(per, counts, [b + 1 for b in bases], totals)) == world
    ok &= len(verify_gathered_regions(per, counts, bases, [(v + 1, cc) for v, cc in totals])) == world
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gathered_records_self_check_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_verify, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_allgatherv_of_hit_lists_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(60)
    assert [r[1] for r in res] == [True, True]
    assert [(r[2], r[3]) for r in res] == [(6, 0), (5, 6)]  # 11 regions -> 6 + 5, contiguous


def test_shard_bounds_cover_everything():
    from variantstore_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 100003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_id_file_rendezvous_ignores_a_stale_file(tmp_path):
    """make_comm's file rendezvous (parallel.exchange_id_file): a rank other than 0 never takes the file an EARLIER launch
    left at the same path -- nor a torn one -- for this launch's unique id (ranks used to accept any file)."""
    import threading
    import time
    from variantstore_amd.parallel import exchange_id_file
    path = str(tmp_path / "uid")
    old = exchange_id_file(0, 2, path, lambda: b"O" * 128, nonce="launch-1")
    assert old == b"O" * 128 and os.path.exists(path)
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("uid", exchange_id_file(1, 2, path, None, nonce="launch-2", timeout=20)))
    t.start()
    time.sleep(0.3)
    assert "uid" not in got, "rank 1 accepted the stale id file of another launch"
    with open(path, "wb") as f:   # a torn write of this launch's header alone is not an id either
        f.write(b"\0" * 8)
    time.sleep(0.1)
    assert "uid" not in got
    new = exchange_id_file(0, 2, path, lambda: b"N" * 128, nonce="launch-2")
    t.join(20)
    assert got.get("uid") == new == b"N" * 128
    with pytest.raises(TimeoutError):
        exchange_id_file(1, 2, path, None, nonce="launch-3", timeout=0.2)


def test_id_file_rendezvous_needs_a_launch_nonce(tmp_path, monkeypatch):
    """outside torchrun and without a nonce nothing tells this launch's id file from one an earlier launch left at the same
    path -- the rendezvous refuses to guess; and an id of the wrong length (a file somebody else wrote under this launch's tag) is not an id."""
    import hashlib
    import threading
    from variantstore_amd.parallel import COMM_ID_BYTES, exchange_id_file
    for k in ("VS_COMM_NONCE", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    path = str(tmp_path / "uid")
    with pytest.raises(ValueError):
        exchange_id_file(0, 2, path, lambda: b"I" * COMM_ID_BYTES)
    monkeypatch.setenv("VS_COMM_NONCE", "launch-env")
    assert exchange_id_file(0, 2, path, lambda: b"E" * COMM_ID_BYTES) == b"E" * COMM_ID_BYTES
    assert exchange_id_file(1, 2, path, None, timeout=5) == b"E" * COMM_ID_BYTES
    with open(path, "wb") as f:   # this launch's tag in front of 100 bytes: not a unique id
        f.write(hashlib.sha256(b"launch-env").digest()[:16] + b"x" * 100)
    with pytest.raises(TimeoutError):
        exchange_id_file(1, 2, path, None, timeout=0.3)
