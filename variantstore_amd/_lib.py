"""ctypes binding of the C ABI (include/variantstore_hip.h).

The engine is a hipcc-built shared library; there is no Python or CPU
implementation behind it.  Importing this module fails loudly when the library
has not been built (`python -m variantstore_amd.build`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VS_ENGINE_LIB") or os.path.join(_HERE, "lib", "libvariantstore_hip.so")  # override: A/B builds


class Region(C.Structure):
    _fields_ = [("x", C.c_uint64), ("y", C.c_uint64)]


class ConstructStats(C.Structure):
    _fields_ = [("num_vars", C.c_uint64), ("num_mutations", C.c_uint64), ("num_mutations_samples", C.c_uint64),
                ("num_vertices", C.c_uint64), ("num_edges", C.c_uint64), ("seq_length", C.c_uint64),
                ("num_classes", C.c_uint64), ("use_bit_vector", C.c_uint32)]


class SynthParams(C.Structure):
    _fields_ = [("ref_length", C.c_uint64), ("num_variants", C.c_uint64), ("num_samples", C.c_uint32),
                ("seed", C.c_uint64), ("first_pos", C.c_uint64), ("frac_ins", C.c_double), ("frac_del", C.c_double),
                ("frac_multi", C.c_double), ("max_indel", C.c_uint32), ("af_exponent", C.c_double),
                ("sample_coordinates", C.c_uint32), ("max_af", C.c_double)]


class IndexInfo(C.Structure):
    _fields_ = [("ref_length", C.c_uint64), ("num_vertices", C.c_uint64), ("num_edges_csr", C.c_uint64),
                ("ref_path_nodes", C.c_uint64), ("index_nodes", C.c_uint64), ("num_classes", C.c_uint64),
                ("num_sites", C.c_uint64), ("num_carriers", C.c_uint64), ("seq_length", C.c_uint64),
                ("num_samples", C.c_uint32), ("use_bit_vector", C.c_uint32), ("device_bytes", C.c_uint64),
                ("device", C.c_int), ("num_topology_keys", C.c_uint64), ("list_max", C.c_uint32), ("reserved_", C.c_uint32),
                ("t4_rows_bytes", C.c_uint64), ("pool_mallocs", C.c_uint64), ("pool_frees", C.c_uint64),
                ("t6_speculated", C.c_uint64), ("t6_refused", C.c_uint64)]


class ResultView(C.Structure):
    _fields_ = [("n_regions", C.c_uint64), ("region_flags", C.POINTER(C.c_uint8)),
                ("var_begin", C.POINTER(C.c_uint64)), ("var_count", C.POINTER(C.c_uint64)),
                ("n_slots", C.c_uint64), ("pos", C.POINTER(C.c_uint64)),
                ("ref_off", C.POINTER(C.c_uint32)), ("ref_len", C.POINTER(C.c_uint32)),
                ("alt_off", C.POINTER(C.c_uint32)), ("alt_len", C.POINTER(C.c_uint32)),
                ("var_flags", C.POINTER(C.c_uint32)), ("car_begin", C.POINTER(C.c_uint64)),
                ("car_count", C.POINTER(C.c_uint32)), ("n_carriers", C.c_uint64),
                ("carriers", C.POINTER(C.c_uint32)), ("seq_pool", C.c_void_p)]


class VariantRow(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("ref_off", C.c_uint32), ("ref_len", C.c_uint32), ("alt_off", C.c_uint32),
                ("alt_len", C.c_uint32), ("count_flags", C.c_uint32), ("car_begin", C.c_uint64)]


class ResultRaw(C.Structure):
    _fields_ = [("n_regions", C.c_uint64), ("region_flags", C.POINTER(C.c_uint8)), ("row_begin", C.POINTER(C.c_uint64)),
                ("row_count", C.POINTER(C.c_uint64)), ("var_count", C.POINTER(C.c_uint64)), ("car_base", C.POINTER(C.c_uint64)),
                ("car_len", C.POINTER(C.c_uint64)), ("n_rows", C.c_uint64), ("rows", C.POINTER(VariantRow)),
                ("arena_entries", C.c_uint64), ("carrier_bytes", C.c_uint32), ("arena", C.c_void_p), ("seq_pool", C.c_void_p),
                ("shared", C.c_int)]


CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.POINTER(ResultRaw))


class Timing(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_bounds", C.c_float), ("ms_scan", C.c_float), ("ms_emit", C.c_float),
                ("ms_fill", C.c_float), ("fill_launches", C.c_uint64)]


# every symbol include/variantstore_hip.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "vs_strerror": (C.c_char_p, [C.c_int]),
    "vs_last_error": (C.c_char_p, []),
    "vs_index_from_vcf": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(ConstructStats), C.POINTER(_P)]),
    "vs_index_synthetic": (C.c_int, [C.POINTER(SynthParams), C.c_int, C.POINTER(ConstructStats), C.POINTER(_P)]),
    "vs_index_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "vs_index_save": (C.c_int, [_P, C.c_char_p]),
    "vs_index_close": (None, [_P]),
    "vs_index_get_info": (C.c_int, [_P, C.POINTER(IndexInfo)]),
    "vs_index_sample_id": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_uint32)]),
    "vs_index_sample_name": (C.c_char_p, [_P, C.c_uint32]),
    "vs_index_chr": (C.c_char_p, [_P]),
    "vs_index_export_plain": (C.c_int, [_P, C.c_char_p]),
    "vs_index_out_neighbors": (C.c_int64, [_P, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64]),
    "vs_query_var_in_ref": (C.c_int, [_P, C.POINTER(Region), C.c_uint64, C.POINTER(_P)]),
    "vs_query_var_in_ref_device": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(_P)]),
    "vs_query_expand_site_ranges": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(_P)]),
    "vs_query_sample_var_in_ref": (C.c_int, [_P, C.POINTER(Region), C.c_uint64, C.c_uint32, C.POINTER(_P)]),
    "vs_query_samples_var_in_ref": (C.c_int, [_P, C.POINTER(Region), C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(_P)]),
    "vs_query_closest_var": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(_P)]),
    "vs_query_samples_has_var": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                           C.c_uint64, C.POINTER(_P)]),
    "vs_query_sample_seq": (C.c_int, [_P, C.POINTER(Region), C.c_uint64, C.POINTER(C.c_uint32), C.c_int, C.POINTER(_P)]),
    "vs_query_sample_var_in_sample": (C.c_int, [_P, C.POINTER(Region), C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(_P)]),
    "vs_result_get_sequences": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint8)),
                                          C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_char_p)]),
    "vs_index_draw_subgraph": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p]),
    "vs_index_find": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint32)]),
    "vs_result_get_view": (C.c_int, [_P, C.c_int, C.POINTER(ResultView)]),
    "vs_result_get_raw": (C.c_int, [_P, C.c_int, C.POINTER(ResultRaw)]),
    "vs_query_var_in_ref_stream": (C.c_int, [_P, C.POINTER(Region), C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]),
    "vs_result_totals": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64)]),
    "vs_result_format_region": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]),
    "vs_result_digest": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "vs_result_layout": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_int)]),
    "vs_result_fill_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "vs_result_pack_headers": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vs_result_pack_regions": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vs_result_free": (None, [_P]),
    "vs_comm_unique_id": (C.c_int, [_P]),
    "vs_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "vs_comm_allgather_regions": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P, C.c_int]),
    "vs_comm_allgather_regions_host": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P]),
    "vs_comm_wait": (C.c_int, [_P]),
    "vs_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vs_comm_destroy": (None, [_P]),
    "vs_index_last_timing": (C.c_int, [_P, C.POINTER(Timing)]),
    "vs_index_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
}

_lib = None


def load():
    """Load the engine; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP engine has not been built "
            "(run `python -m variantstore_amd.build`). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  If this library were loaded
    # first it would bring in the system copy and a later `import torch` would see "No HIP GPUs" (and
    # corrupt the heap).  Importing torch first, when it is installed, makes both share torch's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
