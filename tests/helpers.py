"""Test helpers: random FASTA/VCF cohorts and oracle comparison."""
import os

import numpy as np

BASES = "ACGT"


def write_random_cohort(dirpath, seed, ref_len=4000, n_rows=120, n_samples=6, sample_names=None,
                        p_ins=0.12, p_del=0.12, p_multi=0.08, p_mnp=0.05, p_near=0.3, carrier_p=0.3,
                        unphased_p=0.1, missing_p=0.03, haploid_p=0.0, chrom="c1", p_same=0.0, max_indel=4):
    """The "mix" recipe of SURVEY.md §4.5: SNPs, 1-4 bp insertions/deletions, two-ALT rows, MNPs, with a
    share of sites 1-3 bp from their predecessor.  Returns (fasta, vcf, names)."""
    rng = np.random.default_rng(seed)
    ref = "".join(BASES[i] for i in rng.integers(0, 4, size=ref_len))
    if sample_names is None:
        sample_names = [f"S{i + 1:03d}" for i in range(n_samples)]
    n_samples = len(sample_names)
    pos_list = []
    p = int(rng.integers(2, 20))
    while len(pos_list) < n_rows and p < ref_len - 12 - max_indel:
        pos_list.append(p)
        if p_same and rng.random() < p_same:
            continue  # another row at the very same position (repeated / split multi-allelic sites)
        if rng.random() < p_near:
            p += int(rng.integers(1, 4))
        else:
            p += int(rng.integers(4, max(5, 2 * (ref_len // max(n_rows, 1)))))
    fasta = os.path.join(dirpath, f"r{seed}.fa")
    vcf = os.path.join(dirpath, f"r{seed}.vcf")
    with open(fasta, "w") as f:
        f.write(f">{chrom} random\n")
        for i in range(0, ref_len, 60):
            f.write(ref[i:i + 60] + "\n")
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(sample_names) + "\n")
        for p in pos_list:
            t = rng.random()
            r0 = ref[p - 1]
            if t < p_ins:
                k = int(rng.integers(1, max_indel + 1))
                refa, alts = r0, [r0 + "".join(BASES[i] for i in rng.integers(0, 4, size=k))]
            elif t < p_ins + p_del:
                k = int(rng.integers(1, max_indel + 1))
                refa, alts = ref[p - 1:p + k], [r0]
            elif t < p_ins + p_del + p_mnp:
                refa = ref[p - 1:p + 1]
                alts = ["".join(BASES[(BASES.index(c) + 1 + int(rng.integers(0, 3))) % 4] for c in refa)]
            else:
                refa = r0
                others = [b for b in BASES if b != r0]
                rng.shuffle(others)
                alts = [others[0]]
                if rng.random() < p_multi:
                    alts.append(others[1])
            gts = []
            any_car = False
            for s in range(n_samples):
                if rng.random() < missing_p:
                    gts.append("./.")
                    continue
                if haploid_p and rng.random() < haploid_p:
                    a = 1 if rng.random() < carrier_p else 0
                    gts.append(str(a))
                    any_car |= a > 0
                    continue
                a1 = int(rng.integers(1, len(alts) + 1)) if rng.random() < carrier_p else 0
                a2 = int(rng.integers(1, len(alts) + 1)) if rng.random() < carrier_p else 0
                sep = "/" if rng.random() < unphased_p else "|"
                gts.append(f"{a1}{sep}{a2}")
                any_car |= (a1 > 0 or a2 > 0)
            if not any_car:
                gts[int(rng.integers(0, n_samples))] = "1|0"
            f.write(f"{chrom}\t{p}\t.\t{refa}\t{','.join(alts)}\t99\t.\t.\tGT\t" + "\t".join(gts) + "\n")
    return fasta, vcf, sample_names


def random_regions(rng, ref_len, n, max_len=600):
    """Random regions plus the edge shapes the reference's driver can be handed."""
    out = []
    for _ in range(n):
        x = int(rng.integers(1, ref_len + 1))
        out.append((x, x + int(rng.integers(0, max_len))))
    out += [(1, ref_len + 1), (1, ref_len), (1, ref_len + 100), (ref_len, ref_len + 5), (ref_len + 1, ref_len + 9),
            (ref_len + 50, ref_len + 60), (5, 5), (9, 3), (1, 2), (2, 3)]
    return out


def oracle_texts(orc, regions, sample=None):
    """[(n, early_out, text)] from the CPU oracle; n == -1 marks a non-terminating reference walk."""
    out = []
    for x, y in regions:
        if sample is None:
            out.append(orc.get_var_in_ref(x, y))
        else:
            out.append(orc.get_sample_var_in_ref(x, y, sample))
    return out


def read_plain(path):
    """Parse the plain dump (HostGraph::write_plain) into a dict of numpy arrays / lists."""
    import struct
    data = open(path, "rb").read()
    assert data[:8] == b"VSPLAIN1"
    pos = [8]

    def u64():
        v = struct.unpack_from("<Q", data, pos[0])[0]
        pos[0] += 8
        return v

    def s():
        n = u64()
        v = data[pos[0]:pos[0] + n].decode("latin-1")
        pos[0] += n
        return v

    def vec(dt):
        n = u64()
        a = np.frombuffer(data, dtype=dt, count=n, offset=pos[0]).copy()
        pos[0] += n * np.dtype(dt).itemsize
        return a

    g = {"chr": s(), "ref_length": u64(), "num_samples": u64(), "use_bit_vector": u64(), "num_classes": u64()}
    for k, dt in [("off", "<u4"), ("len", "<u4"), ("class_id", "<u4"), ("ref_index", "<u4"), ("car_begin", "<u8"),
                  ("car_flags", "u1"), ("car_index", "<u4"), ("car_sid", "<u4"), ("seq", "u1"), ("class_bits", "<u8")]:
        g[k] = vec(dt)
    g["sample_names"] = [s() for _ in range(u64())]
    g["topo_inplace"] = vec("u1")
    g["topo_val"] = vec("<u4")
    g["aux_lists"] = [vec("<u4") for _ in range(u64())]
    g["idx_pos"] = vec("<u4")
    g["node_list"] = vec("<u4")
    return g


# ---- windows of a synthetic cohort (tests/native/synth_windows.cpp): how the oracle gets to see the full-size indexes ----
_SYNTH_WINDOWS_EXE = None


def synth_windows(kw, wins, outdir):
    """Write FASTA + VCF of the windows [(lo, hi), ...] (1-based, inclusive) of the synthetic cohort `kw` (the keyword
    arguments of VariantStore.synthetic) into outdir/w<k>.fa / .vcf; returns the records written per window."""
    import subprocess
    import tempfile
    global _SYNTH_WINDOWS_EXE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if _SYNTH_WINDOWS_EXE is None:
        exe = os.path.join(tempfile.mkdtemp(prefix="vs_native_"), "synth_windows")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "native", "synth_windows.cpp")])
        _SYNTH_WINDOWS_EXE = exe
    args = [str(kw[k]) for k in ("ref_length", "num_variants", "num_samples", "seed", "first_pos", "frac_ins", "frac_del", "frac_multi",
                                 "max_indel", "af_exponent")] + [str(kw.get("max_af", 0.5)), str(outdir)]
    out = subprocess.run([_SYNTH_WINDOWS_EXE] + args, input="".join(f"{a} {b}\n" for a, b in wins), text=True, capture_output=True, check=True)
    return [int(line.split()[3]) for line in out.stdout.strip().split("\n")]


def parse_rows(text, shift=0):
    """Rows of a region's print_var text as (pos + shift, ref, alt, samples)."""
    rows = []
    for line in text.split("\n")[1:]:
        if line:
            p, ref, alt, s = line.split("\t")
            rows.append((int(p) + shift, ref, alt, s))
    return rows


def window_oracle(outdir, k):
    """The small index of window k built through the product's VCF path on the host, and the oracle over it."""
    from oracle.oracle import Oracle
    from variantstore_amd import VariantStore
    w = VariantStore.from_vcf(os.path.join(outdir, f"w{k}.fa"), os.path.join(outdir, f"w{k}.vcf"), device=-1)
    plain = os.path.join(outdir, f"w{k}.bin")
    w.export_plain(plain)
    w.close()
    orc = Oracle(plain)
    return orc


# ---- two ranks on one GPU: the stand-in for RCCL (tests/native/fake_rccl.cpp), loaded by the engine through VS_RCCL_LIB ----
_FAKE_RCCL = None


def build_fake_rccl():
    """Compile tests/native/fake_rccl.cpp (hipcc: it copies device buffers through the HIP runtime) and return the library's path."""
    import subprocess
    import tempfile
    global _FAKE_RCCL
    if _FAKE_RCCL is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lib = os.path.join(tempfile.mkdtemp(prefix="vs_native_"), "libfake_rccl.so")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-fPIC", "-shared", "-o", lib,
                               os.path.join(root, "tests", "native", "fake_rccl.cpp")])
        _FAKE_RCCL = lib
    return _FAKE_RCCL


def vcf_truth_cases(outdir, k, lo, edge=60):
    """Expected type-6 rows of window k's ISOLATED records, derived from the window's VCF TEXT alone (tests/vcf_truth.py: no
    from_vcf, no oracle), in the coordinates of the whole cohort: [(abs_pos, 'pos\\tref\\talt\\tsamples\\n'), ...].  Records within
    `edge` bases of the window's ends are left out (a neighbour outside the window could touch them)."""
    import vcf_truth as vt
    names, recs = vt.read_vcf(os.path.join(outdir, f"w{k}.vcf"))
    assert names == sorted(names)
    length = max(r[0] + len(r[1]) for r in recs) if recs else 0
    out = []
    for i, text in vt.expected_type6_rows(names, recs):
        pos, ref, _alts, _g = recs[i]
        if pos <= edge or pos + len(ref) >= length - edge:
            continue
        p, rest = text.split("\t", 1)
        out.append((int(p) + lo - 1, f"{int(p) + lo - 1}\t{rest}", recs[i]))
    return names, out
