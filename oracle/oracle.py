"""ctypes front-end of the CPU oracle (oracle/vs_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvs_oracle.so")
REF_GQF_PATH = os.path.join(_HERE, "_ref", "libgqf_ref.so")


def build(verbose=False, with_ref=False):
    """Compile the restatement; with_ref=True (explicit opt-in: it compiles files of the reference checkout where they
    lie) also builds oracle/_ref/libgqf_ref.so from the reference's own CQF sources."""
    out = subprocess.run(["make", "-s", "-C", _HERE] + (["all", "ref"] if with_ref else ["all"]), capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout, end="")


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
                os.path.join(_HERE, "vs_oracle.cpp")):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.vso_open.restype = C.c_void_p
        lib.vso_open.argtypes = [C.c_char_p]
        lib.vso_close.argtypes = [C.c_void_p]
        lib.vso_get_var_in_ref.restype = C.c_long
        lib.vso_get_var_in_ref.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
        lib.vso_get_sample_var_in_ref.restype = C.c_long
        lib.vso_get_sample_var_in_ref.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.POINTER(C.c_int)]
        lib.vso_closest_var.restype = C.c_long
        lib.vso_closest_var.argtypes = [C.c_void_p, C.c_uint64]
        lib.vso_samples_has_var.restype = C.c_int
        lib.vso_samples_has_var.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_char_p]
        for fn in (lib.vso_query_sample_from_ref, lib.vso_query_sample_from_sample, lib.vso_get_sample_var_in_sample):
            fn.restype = C.c_long
            fn.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p]
        lib.vso_last_seq.restype = C.c_void_p
        lib.vso_last_seq.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.vso_draw_subgraph.restype = C.c_int
        lib.vso_draw_subgraph.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p]
        lib.vso_raw_text.restype = C.c_void_p
        lib.vso_raw_text.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.vso_last_text.restype = C.c_void_p
        lib.vso_last_text.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.vso_ub_events.restype = C.c_uint64
        lib.vso_ub_events.argtypes = [C.c_void_p]
        lib.vso_find.restype = C.c_uint32
        lib.vso_find.argtypes = [C.c_void_p, C.c_uint64]
        lib.vso_is_empty.restype = C.c_int
        lib.vso_is_empty.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        lib.vso_num_vertices.restype = C.c_uint64
        lib.vso_num_vertices.argtypes = [C.c_void_p]
        lib.vso_out_neighbors.restype = C.c_uint64
        lib.vso_out_neighbors.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64]
        lib.vso_last_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _lib = lib
    return _lib


class Oracle:
    """Literal CPU restatement of the reference query path over a plain dump."""

    def __init__(self, plain_path):
        self._lib = _load()
        self._h = self._lib.vso_open(str(plain_path).encode())
        if not self._h:
            raise IOError(f"cannot read plain dump {plain_path}")

    def close(self):
        if self._h:
            self._lib.vso_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_var_in_ref(self, x, y, text=True):
        """(n_variants, early_out, text).  n == -1: the reference walk does not terminate."""
        e = C.c_int()
        n = self._lib.vso_get_var_in_ref(self._h, x, y, C.byref(e))
        return n, bool(e.value), (self.last_text() if text else None)

    def get_sample_var_in_ref(self, x, y, sample, text=True):
        e = C.c_int()
        n = self._lib.vso_get_sample_var_in_ref(self._h, x, y, sample.encode(), C.byref(e))
        return n, bool(e.value), (self.last_text() if text else None)

    def closest_var(self, pos):
        """Query type 1: (n_variants, text); n == -1 (text None) when the reference returns false."""
        n = self._lib.vso_closest_var(self._h, pos)
        return n, (self.last_text() if n >= 0 else None)

    def samples_has_var(self, pos, ref, alt):
        """Query type 7: the line written to the output file, or None ("There is no such variant!")."""
        if not self._lib.vso_samples_has_var(self._h, pos, ref.encode("latin-1"), alt.encode("latin-1")):
            return None
        n = C.c_uint64()
        p = self._lib.vso_raw_text(self._h, C.byref(n))
        return C.string_at(p, n.value).decode("latin-1")

    def _last_seq(self):
        n = C.c_uint64()
        p = self._lib.vso_last_seq(self._h, C.byref(n))
        return C.string_at(p, n.value).decode("latin-1")

    def query_sample_from_ref(self, x, y, sample):
        """Query type 2: (code, sequence).  code >= 0 is the length; -1 the reference does not terminate,
        -2 unknown sample, -3 the reference dies of an uncaught std::out_of_range."""
        n = self._lib.vso_query_sample_from_ref(self._h, x, y, sample.encode())
        return n, (self._last_seq() if n >= 0 else None)

    def query_sample_from_sample(self, x, y, sample):
        """Query type 3, same return convention as query_sample_from_ref."""
        n = self._lib.vso_query_sample_from_sample(self._h, x, y, sample.encode())
        return n, (self._last_seq() if n >= 0 else None)

    def get_sample_var_in_sample(self, x, y, sample):
        """Query type 5: (n_variants, text); n == -1 non-terminating, -2 unknown sample."""
        n = self._lib.vso_get_sample_var_in_sample(self._h, x, y, sample.encode())
        return n, (self.last_text() if n >= 0 else None)

    def draw_subgraph(self, pos, radius, sample="ref"):
        """`variantstore draw`: the text of graph.dot, or None for an unknown sample."""
        if self._lib.vso_draw_subgraph(self._h, pos, radius, sample.encode()) != 0:
            return None
        n = C.c_uint64()
        p = self._lib.vso_raw_text(self._h, C.byref(n))
        return C.string_at(p, n.value).decode("latin-1")

    def last_text(self):
        n = C.c_uint64()
        p = self._lib.vso_last_text(self._h, C.byref(n))
        return C.string_at(p, n.value).decode("latin-1")

    def last_counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._lib.vso_last_counts(self._h, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    def ub_events(self):
        return int(self._lib.vso_ub_events(self._h))

    def find(self, pos):
        return int(self._lib.vso_find(self._h, pos))

    def is_empty(self, x, y):
        return bool(self._lib.vso_is_empty(self._h, x, y))

    def num_vertices(self):
        return int(self._lib.vso_num_vertices(self._h))

    def out_neighbors(self, v):
        buf = (C.c_uint32 * 4096)()
        n = self._lib.vso_out_neighbors(self._h, v, buf, 4096)
        return [int(buf[i]) for i in range(min(n, 4096))]
