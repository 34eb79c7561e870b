"""Host-side mirror of the reference's query interface, over the C ABI.

The reference's driver (src/commands.cc:114-215) builds `Index idx(prefix)` and
`VariantGraph vg(prefix, mode)` and calls `get_var_in_ref(&vg, &idx, x, y, ...)`
(include/query.h:736) or `get_sample_var_in_ref(..., sample, ...)` (query.h:618)
per region.  Here one `VariantStore` object plays the (vg, idx) pair and the two
query methods take a whole batch of regions; each returns a `QueryResult` whose
per-region content is what the reference's `std::vector<Variant>` would hold.
All computation happens in the HIP engine.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import ConstructStats, IndexInfo, Region, ResultRaw, ResultView, SynthParams, Timing

REGION_EMPTY = 1
REGION_INVALID = 2
VAR_DROPPED = 1


class VariantStoreError(RuntimeError):
    def __init__(self, code, where):
        lib = _lib.load()
        msg = lib.vs_last_error().decode() or lib.vs_strerror(code).decode()
        super().__init__(f"{where}: {msg} (code {code})")
        self.code = code


def _check(code, where):
    if code != 0:
        raise VariantStoreError(code, where)


@dataclass
class Variant:  # reference include/query.h:30-36
    var_pos: int
    ref: str
    alt: str
    samples: List[Tuple[str, str]]


class DeviceArray:
    """An array that already lies in device memory: `ptr` (an integer address, e.g. a torch tensor's data_ptr()) and its
    number of elements -- regions ({u64 beg, u64 end} pairs) or sample ids (u32).  The walking query types
    (get_sample_var_in_ref with one sample per region, query_sample_seq, get_sample_var_in_sample) take it wherever they
    take a host array; the engine then neither reads the array on the host nor copies it over the link."""

    def __init__(self, ptr, n):
        self.ptr = int(ptr)
        self.n = int(n)


def _regions_array(regions):
    if isinstance(regions, DeviceArray):
        return regions, C.cast(C.c_void_p(regions.ptr), C.POINTER(Region)), regions.n
    arr = np.ascontiguousarray(np.asarray(regions, dtype=np.uint64).reshape(-1, 2))
    return arr, arr.ctypes.data_as(C.POINTER(Region)), arr.shape[0]


def _u32_ptr(ids):
    if isinstance(ids, DeviceArray):
        return C.cast(C.c_void_p(ids.ptr), C.POINTER(C.c_uint32))
    return ids.ctypes.data_as(C.POINTER(C.c_uint32))


class QueryResult:
    def __init__(self, store, handle):
        self._store = store
        self._h = handle
        self._lib = store._lib

    def close(self):
        if self._h:
            self._lib.vs_result_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def totals(self):
        """(n_regions, n_variants, n_carriers, n_bases) over the whole batch."""
        v = [C.c_uint64() for _ in range(4)]
        _check(self._lib.vs_result_totals(self._h, *[C.byref(x) for x in v]), "vs_result_totals")
        return tuple(int(x.value) for x in v)

    def sequences(self):
        """(region_flags, list of str) of a sequence result (query types 2 and 3)."""
        n = C.c_uint64()
        fl = C.POINTER(C.c_uint8)()
        beg = C.POINTER(C.c_uint64)()
        chars = C.c_char_p()
        _check(self._lib.vs_result_get_sequences(self._h, C.byref(n), C.byref(fl), C.byref(beg), C.byref(chars)),
               "vs_result_get_sequences")
        q = int(n.value)
        flags = np.ctypeslib.as_array(fl, shape=(q,)).copy() if q else np.zeros(0, np.uint8)
        b = np.ctypeslib.as_array(beg, shape=(q + 1,)).copy() if q else np.zeros(1, np.uint64)
        raw = C.string_at(C.cast(chars, C.c_void_p).value, int(b[q])) if q else b""
        return flags, [raw[int(b[i]):int(b[i + 1])].decode("latin-1") for i in range(q)]

    def layout(self):
        """(rows reported over all regions, rows of the variant table, arena entries, carrier lists expanded, rows and lists
        shared between regions?) of the result in HBM."""
        n, t, s, u, sh = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        _check(self._lib.vs_result_layout(self._h, C.byref(n), C.byref(t), C.byref(s), C.byref(u), C.byref(sh)), "vs_result_layout")
        return int(n.value), int(t.value), int(s.value), int(u.value), bool(sh.value)

    def fill_ms(self):
        """Duration of the carrier expansion when it ran asynchronously (option "async_fill"; waits for it), else -1."""
        ms = C.c_float()
        _check(self._lib.vs_result_fill_ms(self._h, C.byref(ms)), "vs_result_fill_ms")
        return float(ms.value)

    def digest(self):
        d = C.c_uint64()
        _check(self._lib.vs_result_digest(self._h, C.byref(d)), "vs_result_digest")
        return int(d.value)

    def view(self, with_carriers=True):
        """Host copy of the result as numpy arrays (dict)."""
        rv = ResultView()
        _check(self._lib.vs_result_get_view(self._h, 1 if with_carriers else 0, C.byref(rv)), "vs_result_get_view")
        q, a, s = int(rv.n_regions), int(rv.n_slots), int(rv.n_carriers)

        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

        return {
            "region_flags": arr(rv.region_flags, q, np.uint8),
            "var_begin": arr(rv.var_begin, q + 1, np.uint64),
            "var_count": arr(rv.var_count, q, np.uint64),
            "pos": arr(rv.pos, a, np.uint64),
            "ref_off": arr(rv.ref_off, a, np.uint32), "ref_len": arr(rv.ref_len, a, np.uint32),
            "alt_off": arr(rv.alt_off, a, np.uint32), "alt_len": arr(rv.alt_len, a, np.uint32),
            "var_flags": arr(rv.var_flags, a, np.uint32),
            "car_begin": arr(rv.car_begin, a, np.uint64), "car_count": arr(rv.car_count, a, np.uint32),
            "carriers": arr(rv.carriers, s, np.uint32) if with_carriers else None,
        }

    ROW_DTYPE = np.dtype([("pos", "<u4"), ("ref_off", "<u4"), ("ref_len", "<u4"), ("alt_off", "<u4"), ("alt_len", "<u4"),
                          ("count_flags", "<u4"), ("car_begin", "<u8")])

    def raw(self, with_carriers=True):
        """The result as it lies in HBM, copied once into page-locked memory (vs_result_get_raw): numpy VIEWS (no copy; valid
        until the result is closed) of the per-region arrays, the variant table (structured rows) and the carrier arena
        (uint16 words id | gt << 13 for cohorts of at most 4032 samples, else uint32 id | gt << 29)."""
        rr = ResultRaw()
        _check(self._lib.vs_result_get_raw(self._h, 1 if with_carriers else 0, C.byref(rr)), "vs_result_get_raw")
        q, a, s = int(rr.n_regions), int(rr.n_rows), int(rr.arena_entries)

        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ptr, shape=(n,))

        rows = (np.ctypeslib.as_array(C.cast(rr.rows, C.POINTER(C.c_uint8)), shape=(a * 32,)).view(self.ROW_DTYPE)
                if a else np.zeros(0, self.ROW_DTYPE))
        arena = None
        if with_carriers and rr.arena:
            ct = C.c_uint16 if rr.carrier_bytes == 2 else C.c_uint32
            arena = np.ctypeslib.as_array(C.cast(rr.arena, C.POINTER(ct)), shape=(s,)) if s else np.zeros(0, np.uint16)
        return {"region_flags": arr(rr.region_flags, q, np.uint8), "row_begin": arr(rr.row_begin, q, np.uint64),
                "row_count": arr(rr.row_count, q, np.uint64), "var_count": arr(rr.var_count, q, np.uint64),
                "car_base": arr(rr.car_base, q, np.uint64), "car_len": arr(rr.car_len, q, np.uint64),
                "rows": rows, "arena": arena, "carrier_bytes": int(rr.carrier_bytes), "shared": bool(rr.shared & 1),
                "resident": bool(rr.shared & 2), "scattered": bool(rr.shared & 4)}

    def num_header_records(self):
        n = C.c_uint64()
        _check(self._lib.vs_result_pack_headers(self._h, None, 0, 0, C.byref(n)), "vs_result_pack_headers")
        return int(n.value)

    def pack_headers_into(self, device_ptr, capacity_records, region_base=0):
        """Write the hit-list records (4 x uint64 per variant slot) into device memory at `device_ptr`
        (e.g. a torch CUDA tensor's data_ptr()) for a collective over the shards of a batch."""
        n = C.c_uint64()
        _check(self._lib.vs_result_pack_headers(self._h, C.c_void_p(device_ptr), capacity_records, region_base,
                                                C.byref(n)), "vs_result_pack_headers")
        return int(n.value)

    def num_region_records(self):
        n = C.c_uint64()
        _check(self._lib.vs_result_pack_regions(self._h, None, 0, 0, C.byref(n)), "vs_result_pack_regions")
        return int(n.value)

    def pack_regions_into(self, device_ptr, capacity_records, region_base=0):
        """Compact hit lists: one 4 x uint64 record per region (site range of the replicated index)."""
        n = C.c_uint64()
        _check(self._lib.vs_result_pack_regions(self._h, C.c_void_p(device_ptr), capacity_records, region_base,
                                                C.byref(n)), "vs_result_pack_regions")
        return int(n.value)

    def region_text(self, q):
        """The `-o` file the reference writes for region q (query.h:38-50, 774-781)."""
        txt = C.c_char_p()
        n = C.c_uint64()
        _check(self._lib.vs_result_format_region(self._h, q, C.byref(txt), C.byref(n)), "vs_result_format_region")
        return C.string_at(txt, n.value).decode("latin-1")

    def region_variants(self, q) -> List[Variant]:
        out = []
        for line in self.region_text(q).split("\n")[1:]:
            if not line:
                continue
            pos, ref, alt, samples = line.split("\t")
            pairs = []
            for tok in samples.split(" "):
                if tok:
                    name, gt = tok[:-1].rsplit("(", 1)
                    pairs.append((name, gt))
            out.append(Variant(int(pos), ref, alt, pairs))
        return out


class Comm:
    """The hit-list collective of one rank (vs_comm_*: RCCL called directly by the engine, no torch.distributed).
    `Comm.unique_id()` on rank 0, the 128 bytes handed to every rank by whatever means the host program has, then
    `Comm(store, rank, world, uid)` on every rank."""

    def __init__(self, store, rank, world, uid):
        self._lib = store._lib
        self._store = store     # (keeps the index handle alive)
        self.rank, self.world = int(rank), int(world)
        buf = C.create_string_buffer(bytes(uid), 128)
        h = C.c_void_p()
        _check(self._lib.vs_comm_init(store._h, self.rank, self.world, C.cast(buf, C.c_void_p), C.byref(h)), "vs_comm_init")
        self._h = h

    @staticmethod
    def unique_id():
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        _check(lib.vs_comm_unique_id(C.cast(buf, C.c_void_p)), "vs_comm_unique_id")
        return bytes(buf.raw)

    def allgather_regions(self, result, region_base, max_count, device_dst, async_op=False):
        """Pack `result`'s per-region records and all-gather them, padded to max_count per rank, into device memory at
        `device_dst` (world x max_count x 32 bytes; rank k's records start at record k * max_count)."""
        _check(self._lib.vs_comm_allgather_regions(self._h, result._h, int(region_base), int(max_count), C.c_void_p(int(device_dst)),
                                                   1 if async_op else 0), "vs_comm_allgather_regions")

    def wait(self):
        _check(self._lib.vs_comm_wait(self._h), "vs_comm_wait")

    def info(self):
        """(rank, world, ranks RCCL reports for the communicator: ncclCommCount)."""
        r, w, n = C.c_int(), C.c_int(), C.c_int()
        _check(self._lib.vs_comm_info(self._h, C.byref(r), C.byref(w), C.byref(n)), "vs_comm_info")
        return r.value, w.value, n.value

    def close(self):
        if self._h:
            self._lib.vs_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VariantStore:
    """An opened index: the reference's (VariantGraph, Index) pair on one GPU."""

    def __init__(self, handle, stats=None):
        self._h = handle
        self._lib = _lib.load()
        # same symbol, raw-address prototype: passing a numpy buffer's address as an int skips ctypes' pointer objects
        self._q6 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)(("vs_query_var_in_ref", self._lib))
        self.construct_stats = stats

    # ---- constructors -------------------------------------------------------
    @classmethod
    def from_vcf(cls, fasta, vcf, device=0):
        lib = _lib.load()
        h = C.c_void_p()
        st = ConstructStats()
        _check(lib.vs_index_from_vcf(str(fasta).encode(), str(vcf).encode(), device, C.byref(st), C.byref(h)),
               "vs_index_from_vcf")
        return cls(h, st)

    @classmethod
    def synthetic(cls, device=0, **kw):
        lib = _lib.load()
        p = SynthParams(ref_length=kw.get("ref_length", 1_000_000), num_variants=kw.get("num_variants", 10_000),
                        num_samples=kw.get("num_samples", 100), seed=kw.get("seed", 1),
                        first_pos=kw.get("first_pos", 1000), frac_ins=kw.get("frac_ins", 0.0),
                        frac_del=kw.get("frac_del", 0.0), frac_multi=kw.get("frac_multi", 0.0),
                        max_indel=kw.get("max_indel", 6), af_exponent=kw.get("af_exponent", 3.0),
                        sample_coordinates=1 if kw.get("sample_coordinates") else 0,
                        max_af=float(kw.get("max_af", 0.0)))
        h = C.c_void_p()
        st = ConstructStats()
        _check(lib.vs_index_synthetic(C.byref(p), device, C.byref(st), C.byref(h)), "vs_index_synthetic")
        return cls(h, st)

    @classmethod
    def open(cls, prefix, device=0):
        lib = _lib.load()
        h = C.c_void_p()
        _check(lib.vs_index_open(str(prefix).encode(), device, C.byref(h)), "vs_index_open")
        return cls(h)

    def save(self, prefix):
        _check(self._lib.vs_index_save(self._h, str(prefix).encode()), "vs_index_save")

    def close(self):
        if self._h:
            self._lib.vs_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inspection ---------------------------------------------------------
    def info(self) -> IndexInfo:
        info = IndexInfo()
        _check(self._lib.vs_index_get_info(self._h, C.byref(info)), "vs_index_get_info")
        return info

    def chr(self):
        return self._lib.vs_index_chr(self._h).decode()

    def sample_id(self, name):
        sid = C.c_uint32()
        _check(self._lib.vs_index_sample_id(self._h, name.encode(), C.byref(sid)), "vs_index_sample_id")
        return int(sid.value)

    def sample_name(self, sid):
        s = self._lib.vs_index_sample_name(self._h, sid)
        return s.decode() if s is not None else None

    def export_plain(self, path):
        _check(self._lib.vs_index_export_plain(self._h, str(path).encode()), "vs_index_export_plain")

    def out_neighbors(self, v):
        buf = (C.c_uint32 * 4096)()
        n = self._lib.vs_index_out_neighbors(self._h, v, buf, 4096)
        if n < 0:
            raise IndexError(v)
        return [int(buf[i]) for i in range(min(n, 4096))]

    def last_timing(self) -> Timing:
        t = Timing()
        _check(self._lib.vs_index_last_timing(self._h, C.byref(t)), "vs_index_last_timing")
        return t

    def set_option(self, key, value):
        """Test / tuning switches of this handle (include/variantstore_hip.h: vs_index_set_option)."""
        _check(self._lib.vs_index_set_option(self._h, key.encode(), int(value)), "vs_index_set_option")

    # ---- queries ------------------------------------------------------------
    def find(self, positions: Sequence[int]):
        """Index::find for a batch of positions (index.h:119-133)."""
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.uint64))
        out = np.zeros(pos.shape[0], dtype=np.uint32)
        _check(self._lib.vs_index_find(self._h, pos.ctypes.data_as(C.POINTER(C.c_uint64)), pos.shape[0],
                                       out.ctypes.data_as(C.POINTER(C.c_uint32))), "vs_index_find")
        return out

    def get_var_in_ref(self, regions) -> QueryResult:
        """Query type 6 over a batch of (pos_x, pos_y) regions (query.h:736-784)."""
        # (this is the call whose single-region latency the bench reports: a ready uint64 array is passed on as it is,
        #  without the ctypes pointer objects of the general helper)
        if (type(regions) is np.ndarray and regions.dtype == np.uint64 and regions.flags.c_contiguous and regions.ndim == 2
                and regions.shape[1] == 2):
            arr = regions
        else:
            arr = np.ascontiguousarray(np.asarray(regions, dtype=np.uint64).reshape(-1, 2))
        h = C.c_void_p()
        rc = self._q6(self._h, arr.__array_interface__["data"][0], arr.shape[0], C.byref(h))
        if rc != 0:
            raise VariantStoreError(rc, "vs_query_var_in_ref")
        return QueryResult(self, h)

    def stream_var_in_ref(self, regions, chunk_regions, on_chunk, with_carriers=True):
        """Query type 6 with delivery (vs_query_var_in_ref_stream): `on_chunk(first_region, raw)` is called per chunk with the
        ctypes ResultRaw of that chunk (valid during the call) while the next chunk is being computed."""
        arr, ptr, n = _regions_array(regions)

        def tramp(_user, first, raw):
            on_chunk(int(first), raw.contents)
            return 0

        cb = _lib.CHUNK_FN(tramp)
        _check(self._lib.vs_query_var_in_ref_stream(self._h, ptr, n, int(chunk_regions), 1 if with_carriers else 0,
                                                    C.cast(cb, C.c_void_p), None), "vs_query_var_in_ref_stream")

    def get_var_in_ref_device(self, device_ptr, n) -> QueryResult:
        """Query type 6 over n (pos_x, pos_y) uint64 pairs that already live in this GPU's memory (`device_ptr`: an
        address, e.g. `tensor.data_ptr()` of a contiguous int64/uint64 CUDA tensor of shape (n, 2))."""
        h = C.c_void_p()
        _check(self._lib.vs_query_var_in_ref_device(self._h, C.c_void_p(int(device_ptr)), int(n), C.byref(h)),
               "vs_query_var_in_ref_device")
        return QueryResult(self, h)

    def expand_site_ranges(self, device_ptr, n) -> QueryResult:
        """The receiving side of the hit-list collective: n compact region records (QueryResult.pack_regions_into, own or
        gathered from ranks holding the same index) in this GPU's memory -> the full type-6 result they describe."""
        h = C.c_void_p()
        _check(self._lib.vs_query_expand_site_ranges(self._h, C.c_void_p(int(device_ptr)), int(n), C.byref(h)),
               "vs_query_expand_site_ranges")
        return QueryResult(self, h)

    def get_sample_var_in_ref(self, regions, sample) -> QueryResult:
        """Query type 4 for one sample over a batch of regions (query.h:618-729)."""
        arr, ptr, n = _regions_array(regions)
        h = C.c_void_p()
        if isinstance(sample, (list, tuple, np.ndarray, DeviceArray)):  # one sample per region
            sids = self._sample_ids(sample, n)
            _check(self._lib.vs_query_samples_var_in_ref(self._h, ptr, n, _u32_ptr(sids),
                                                         C.byref(h)), "vs_query_samples_var_in_ref")
            return QueryResult(self, h)
        sid = self.sample_id(sample) if isinstance(sample, str) else int(sample)
        _check(self._lib.vs_query_sample_var_in_ref(self._h, ptr, n, sid, C.byref(h)), "vs_query_sample_var_in_ref")
        return QueryResult(self, h)

    def closest_var(self, positions) -> QueryResult:
        """Query type 1 (query.h:441-483) for a batch of positions: one result "region" per position;
        `region_flags & 4` marks the calls for which the reference returns false."""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        h = C.c_void_p()
        _check(self._lib.vs_query_closest_var(self._h, pos.ctypes.data_as(C.POINTER(C.c_uint64)), pos.shape[0],
                                              C.byref(h)), "vs_query_closest_var")
        return QueryResult(self, h)

    def samples_has_var(self, positions, refs, alts) -> QueryResult:
        """Query type 7 (query.h:792-823) for a batch of (pos, ref, alt): region_text(q) is the reference's
        output line, `region_flags & 4` means "There is no such variant!"."""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        n = pos.shape[0]
        if len(refs) != n or len(alts) != n:
            raise ValueError("one ref and one alt per position expected")
        r = refs if isinstance(refs, C.Array) else self.c_strings(refs)      # (a caller that asks again and again builds the arrays once)
        a = alts if isinstance(alts, C.Array) else self.c_strings(alts)
        h = C.c_void_p()
        _check(self._lib.vs_query_samples_has_var(self._h, pos.ctypes.data_as(C.POINTER(C.c_uint64)), r, a, n,
                                                  C.byref(h)), "vs_query_samples_has_var")
        return QueryResult(self, h)

    @staticmethod
    def c_strings(strings):
        """`strings` as the char*[] the C ABI takes (samples_has_var accepts the result in place of a list)."""
        return (C.c_char_p * max(len(strings), 1))(*[x.encode("latin-1") for x in strings])

    def _sample_ids(self, sample, n):
        if isinstance(sample, DeviceArray):
            if sample.n != n:
                raise ValueError("one sample per region expected")
            return sample
        if isinstance(sample, np.ndarray) and sample.dtype.kind in "ui":   # ids already: no per-element Python
            sids = np.ascontiguousarray(sample, dtype=np.uint32)
            if sids.shape[0] != n:
                raise ValueError("one sample per region expected")
            return sids
        if isinstance(sample, (list, tuple, np.ndarray)):
            sids = np.ascontiguousarray([self.sample_id(x) if isinstance(x, str) else int(x) for x in sample], dtype=np.uint32)
            if sids.shape[0] != n:
                raise ValueError("one sample per region expected")
            return sids
        sid = self.sample_id(sample) if isinstance(sample, str) else int(sample)
        return np.full(max(n, 1), sid, dtype=np.uint32)

    def query_sample_seq(self, regions, sample, sample_coordinates=False) -> QueryResult:
        """Query type 2 (query_sample_from_ref, query.h:118-190) or, with sample_coordinates, type 3
        (query_sample_from_sample, query.h:196-261).  `sample` is one name/id or one per region."""
        arr, ptr, n = _regions_array(regions)
        sids = self._sample_ids(sample, n)
        h = C.c_void_p()
        _check(self._lib.vs_query_sample_seq(self._h, ptr, n, _u32_ptr(sids),
                                             1 if sample_coordinates else 0, C.byref(h)), "vs_query_sample_seq")
        return QueryResult(self, h)

    def get_sample_var_in_sample(self, regions, sample) -> QueryResult:
        """Query type 5 (query.h:490-612)."""
        arr, ptr, n = _regions_array(regions)
        sids = self._sample_ids(sample, n)
        h = C.c_void_p()
        _check(self._lib.vs_query_sample_var_in_sample(self._h, ptr, n, _u32_ptr(sids),
                                                       C.byref(h)), "vs_query_sample_var_in_sample")
        return QueryResult(self, h)

    def draw_subgraph(self, pos, radius, outfile, sample=None):
        """`variantstore draw` (query.h:825-842, dot_graph.h:71-132): Graphviz file of the neighbourhood of the
        vertex at `pos`.  Host-only."""
        _check(self._lib.vs_index_draw_subgraph(self._h, int(pos), int(radius), sample.encode() if sample else None,
                                                str(outfile).encode()), "vs_index_draw_subgraph")
