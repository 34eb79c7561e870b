"""What the reference reports for ISOLATED VCF records, derived from the VCF and FASTA text alone.

Nothing here touches the product or the oracle: the expected type-6 rows and the expected sequence of a sample
(query type 2) are worked out in Python from the files a cohort was built from.  The rules are the VCF's own semantics
plus the reference's reporting convention, as its published outputs show it (tests/golden: G1 / G3 of x.vcf and
x.small.vcf -- `58 G GT` is reported as `58 "" T`, `38 TT T` as `39 T ""`):

  SNP / MNP   REF, ALT of equal length at POS      ->  (POS, REF, ALT)
  insertion   ALT = REF + X                        ->  (POS + len(REF) - 1, "", X)
  deletion    REF = ALT + X                        ->  (POS + len(ALT), X, "")
  carriers    every sample with the allele on a haplotype, in sample-name order as the reference's
              std::map-sorted sample list gives ids, each as  name(a|b)  with a, b = "haplotype carries THIS allele"

A record is ISOLATED when no other record's reference span (one base of context either side) touches its own; only
such records are predicted -- crowded sites are where the graph's history decides the row and
they stay the oracle's business.
"""
import gzip


def read_fasta(path):
    name, seq = None, []
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    break
                name = line[1:].split()[0]
            else:
                seq.append(line.strip())
    return name, "".join(seq)


def read_vcf(path):
    """-> (sample names in column order, [ (pos, ref, [alts], [gt strings in column order]) ])"""
    names, recs = [], []
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as f:
        for line in f:
            if line.startswith("##"):
                continue
            t = line.rstrip("\n").split("\t")
            if line.startswith("#"):
                names = t[9:]
                continue
            fmt = t[8].split(":")
            gi = fmt.index("GT")
            recs.append((int(t[1]), t[3], t[4].split(","), [x.split(":")[gi] for x in t[9:]]))
    return names, recs


def _span(rec):
    pos, ref, _alts, _g = rec
    return pos - 1, pos + len(ref)          # one base of context either side of the reference span


def isolated_records(recs, margin=0):
    """Indexes of the records whose span (plus margin) meets no other record's."""
    out = []
    for i, r in enumerate(recs):
        lo, hi = _span(r)
        ok = True
        for j in (i - 1, i + 1):
            if 0 <= j < len(recs):
                l2, h2 = _span(recs[j])
                if not (h2 + margin < lo or hi + margin < l2):
                    ok = False
        if ok:
            out.append(i)
    return out


def _parse_gt(gt):
    """-> (allele1, phased, allele2) with None for a missing allele; a haploid call has allele2 = None and is unphased-free."""
    if "|" in gt:
        a, b = gt.split("|")
        sep = "|"
    elif "/" in gt:
        a, b = gt.split("/")
        sep = "/"
    else:
        a, b, sep = gt, None, None
    conv = lambda s: None if s in (None, ".", "") else int(s)
    return conv(a), sep, conv(b)


def row_of(rec, alt_index):
    """(pos, ref, alt) the reference reports for allele alt_index (1-based) of an isolated record; None if the record
    is none of SNP / MNP / pure insertion / pure deletion."""
    pos, ref, alts, _g = rec
    alt = alts[alt_index - 1]
    if len(ref) == len(alt):
        return pos, ref, alt
    if len(alt) > len(ref) and alt.startswith(ref):
        return pos + len(ref) - 1, "", alt[len(ref):]
    if len(ref) > len(alt) and ref.startswith(alt):
        return pos + len(alt), ref[len(alt):], ""
    return None


def expected_type6_rows(names, recs, only=None):
    """[(record index, 'pos\\tref\\talt\\tname(gt) name(gt) \\n')] for the isolated, diploid-called, single-ALT records
    (multi-allelic rows share vertices between alleles in ways the VCF text does not fix)."""
    order = sorted(range(len(names)), key=lambda i: names[i])      # the reference numbers samples in std::map order
    out = []
    for i in (isolated_records(recs) if only is None else only):
        rec = recs[i]
        if len(rec[2]) != 1:
            continue
        row = row_of(rec, 1)
        if row is None:
            continue
        cars = []
        bad = False
        for col in order:
            a, sep, b = _parse_gt(rec[3][col])
            if sep is None or a is None or b is None:
                bad |= (a not in (None, 0)) or (b not in (None, 0))   # haploid / half-missing carriers: not predicted
                continue
            if a == 1 or b == 1:
                cars.append(f"{names[col]}({int(a == 1)}{sep}{int(b == 1)}) ")
        if bad or not cars:
            continue
        out.append((i, f"{row[0]}\t{row[1]}\t{row[2]}\t{''.join(cars)}\n"))
    return out


def sample_sequence(ref, names, recs, sample, x, y):
    """The sequence of `sample` over the reference interval [x, y) (1-based, query type 2) when every record that
    touches [x - 2, y + 2] is isolated, single-ALT and diploid-called, and none lies within 2 bases of x or y:
    the reference with the sample's alleles applied (a sample carrying the ALT on either haplotype walks the ALT
    vertex).  None when the interval does not qualify."""
    col = names.index(sample)
    iso = set(isolated_records(recs))
    pieces, at = [], x
    for i, rec in enumerate(recs):
        pos, r, alts, gts = rec
        lo, hi = pos, pos + len(r)           # reference bases [lo, hi)
        if hi <= x - 2 or lo >= y + 2:
            continue
        if i not in iso or len(alts) != 1 or row_of(rec, 1) is None:
            return None
        if lo <= x + 2 or hi >= y - 2:
            return None
        a, sep, b = _parse_gt(gts[col])
        if sep is None or a is None or b is None:
            return None
        if (a == 1 or b == 1) and len(r) > len(alts[0]):
            return None      # a deletion the sample carries: the reference's window arithmetic then depends on neighbour order (not predicted)
        if a == 1 or b == 1:
            pieces.append(ref[at - 1:lo - 1])
            pieces.append(alts[0])
            at = hi
    pieces.append(ref[at - 1:y - 1])
    return "".join(pieces)
