"""Pure-Python restatement of VARSCOT's result mergers:
readBamFile / readOntargets / filterRefAlignment / filterSnpAlignment / getSnpType
(variant_processing/filter_output_bam.h:40-496) and mergeResults / processRefOnly
(variant_processing/merge_output_bam.h:46-720).

TEST INFRASTRUCTURE ONLY.  Small inputs; scores come from the C oracle (oracle/vsc_oracle.c).
PARITY UNPINNED (no reference outputs exist); follows the source line by line.
"""
from . import pyoracle

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def dna5(s):
    return "".join(c.upper() if c.upper() in "ACGT" else "N" for c in s)


def c_atoi(s):
    """atoi = (int) strtol: leading integer, 64-bit long truncated to int."""
    import re
    m = re.match(r"\s*([+-]?\d+)", s)
    v = int(m.group(1)) if m else 0
    v = max(-(1 << 63), min((1 << 63) - 1, v))
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


class Genome:
    def __init__(self, records):
        """records: [(full id, sequence)]; addressable by full id and by first word (FAI)."""
        self.records = records
        self.by_name = {}
        for rid, seq in records:
            self.by_name.setdefault(rid, seq)
            self.by_name.setdefault(rid.split()[0] if rid.split() else rid, seq)

    def region(self, chrom, start, end, strand):
        """extract_fasta_ontargets.h:33-76 with flanking = false."""
        seq = self.by_name[chrom]
        start, end = min(start, len(seq)), min(end, len(seq))
        if start > end:
            end = start
        s = dna5(seq[start:end])
        if strand == "-":
            s = "".join(COMP[c] for c in reversed(s))
        return s


class Pot:
    def __init__(self, target, chrom, pos, strand, sequence, mm, snp_type="REF"):
        self.target, self.chr, self.pos, self.strand = target, chrom, pos, strand
        self.sequence, self.mm, self.snp_type = sequence, mm, snp_type

    def key(self):  # comp(), filter_output_bam.h:40-49
        return (self.target, self.chr, self.pos, self.strand, self.sequence, tuple(self.mm), self.snp_type)


def read_sam(text, genome):
    """filter_output_bam.h:362-418."""
    out = []
    for line in text.splitlines():
        if not line or line.startswith("@"):
            continue
        f = line.split("\t")
        # a record that cannot be parsed ends the reading (SeqAn's exception, caught at :413-417: what was read is kept)
        if len(f) < 11 or not (f[1].isdigit() and f[3].isdigit()) or int(f[3]) == 0:
            break
        pos = int(f[3]) - 1
        strand = "-" if int(f[1]) & 16 else "+"
        md = ([x[5:] for x in f[11:] if x.startswith("MD:Z:")] or [""])[-1]
        out.append(Pot(f[0], f[2], pos, strand, genome.region(f[2], pos, pos + 23, strand), pyoracle.md_positions(md)))
    return out


def read_ontargets(bed_text, genome):
    """filter_output_bam.h:462-496.  std::map keeps the first record of a name."""
    on, count = {}, {}
    for line in bed_text.splitlines():
        p = line.split()
        if len(p) < 6 or line.startswith("#"):
            continue
        chrom, start, name, strand = p[0], int(p[1]), p[3], p[5][0]
        if name not in on:
            on[name] = Pot(name, chrom, start, strand, genome.region(chrom, start, start + 23, strand), [-1])
            count[name] = 0
    return on, count


def read_tuscan(text):
    """feature_matrix.h:206-230."""
    out = {}
    for line in text.splitlines():
        p = line.split()
        if len(p) >= 3:
            try:
                out.setdefault(p[0], float(p[2]))
            except ValueError:
                pass
    return out


def get_snp_type(pot, fasta_id, seq_len):
    """filter_output_bam.h:189-263."""
    tag, found, start_found, count = ["VAR_", fasta_id[0], "_"], False, False, 0

    def inside(q):
        return pot.pos <= q and pot.pos + seq_len > q

    i = 3
    while i + 2 < len(fasta_id):
        p, lr, la = c_atoi(fasta_id[i]), len(fasta_id[i + 1]), len(fasta_id[i + 2])
        if lr == la:
            if inside(p):
                tag += [fasta_id[i], ","]
                found = start_found = True
        elif lr < la:
            if inside(p + 1) or inside(p + la - 1):
                tag += [fasta_id[i], ","]
                found = start_found = True
            elif not start_found:
                count -= la - lr
        else:
            if inside(p + 1) or inside(p + lr - 1):
                tag += [fasta_id[i], ","]
                found = start_found = True
            elif not start_found:
                count += lr - la
        i += 3
    pot.pos = (pot.pos + count) % (1 << 32)
    if found:
        pot.snp_type = "".join(tag)[:-1]


def fmt_double(v):
    return "%g" % v  # operator<< with the default precision of 6 significant digits


def _rows_text(rows, on, count, activity, merged, features):
    tsv, fm = [], []
    for p in rows:
        count[p.target] += 1
        name = "%s_%d" % (p.target, count[p.target])
        score = "." if features else fmt_double(pyoracle.mit_score(p.mm)[0])
        if p.mm == [-1]:
            mmcols = "0\t\t" if merged else "0\t"
        else:
            mmcols = "%d\t%s" % (len(p.mm), ",".join(str(x) for x in p.mm)) + ("\t" if merged else "")
        tsv.append("%s\t%d\t%d\t%s\t%s\t%s\t%s\t%s%s" % (p.chr, p.pos, p.pos + 23, name, score, p.strand, p.sequence, mmcols,
                                                   p.snp_type if merged else ""))
        if features:
            row = pyoracle.feature_row(on[p.target].sequence, p.sequence)
            fm.append(name + "\t" + "".join("%d\t" % x for x in row) + fmt_double(activity[p.target]))
    return tsv, fm


HEADER = "#Chr\tStart\tEnd\tTargetsite\tScore\tStrand\tSequence\tMismatch_Number\tMismatch_Positions"


def feature_header():
    import numpy as np, os
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "features_golden.npz"))
    return "\t".join(list(g["names"]) + ["ontargetActivity"])


def process_ref_only(sam_text, bed_text, genome_records, tuscan_text, features):
    """merge_output_bam.h:485-576 (MIT) / :600-720 (feature matrix)."""
    genome = Genome(genome_records)
    hits = read_sam(sam_text, genome)
    on, count = read_ontargets(bed_text, genome)
    activity = read_tuscan(tuscan_text)
    rows = [h for h in hits if h.key() != on[h.target].key()]
    tsv, fm = _rows_text(rows, on, count, activity, False, features)
    return "\n".join([HEADER] + tsv) + "\n", ("\n".join([feature_header()] + fm) + "\n") if features else None


def merge_results(ref_sam, snp_sam, bed_text, genome_records, snp_records, tuscan_text, seq_len, features):
    """merge_output_bam.h:46-215 (MIT) / :244-460 (feature matrix)."""
    ref, snp = Genome(genome_records), Genome(snp_records)
    info = []  # getSnpInfoTable, filter_output_bam.h:434-448
    for rid, seq in snp_records:
        info.append(rid.split("_") + [str(len(seq))])
    on, count = read_ontargets(bed_text, ref)
    rows = []
    for h in read_sam(ref_sam, ref):  # filterRefAlignment, :70-124
        valid = h.key() != on[h.target].key()
        if valid:
            for w in info:
                if w[0] != h.chr:
                    continue
                s, ln = c_atoi(w[1]), c_atoi(w[-1])
                if h.pos >= (s % (1 << 32)) and (h.pos + seq_len) % (1 << 32) <= (s + ln) % (1 << 32):
                    valid = False
                    break
        if valid:
            rows.append(h)
    snp_hits = read_sam(snp_sam, snp)
    for i, h in enumerate(snp_hits):  # filterSnpAlignment, :279-317
        fid = h.chr.split("_")
        h.chr = fid[0]
        h.pos = (h.pos + c_atoi(fid[1] if len(fid) > 1 else "0")) % (1 << 32)
        get_snp_type(h, fid, seq_len)
        valid = h.key() != on[h.target].key()
        if i > 0 and h.key() == snp_hits[i - 1].key():
            valid = False
        if valid:
            rows.append(h)
    activity = read_tuscan(tuscan_text)
    tsv, fm = _rows_text(rows, on, count, activity, True, features)
    return "\n".join([HEADER + "\tVariants"] + tsv) + "\n", ("\n".join([feature_header()] + fm) + "\n") if features else None
