"""Synthetic workloads of BASELINE.json / SURVEY.md 8(d), generated directly in packed-plane form.

Genome: 24 contigs named chr1..chr22, chrX, chrY with hg38 primary-assembly lengths scaled to the
requested total; bases i.i.d. uniform over ACGT; the first and last 10 kb of every contig and ~1 % of
all bases (runs of 10 kb - 1 Mb) are N.  Unlike the sequential xoshiro stream SURVEY.md sketches,
the bases come from a counter-based generator (splitmix64 of the plane word index), so that every
rank of a genome-sharded run can generate exactly its own shard without touching the rest.
Reads: 21 uniform bases + "GG".
"""
import numpy as np

from .api import PackedGenome, CONTIG_DTYPE

HG38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
HG38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]

SEED_GENOME = 0x5EED0001
SEED_GUIDES = 0x5EED0002
TELOMERE = 10_000


def _fill_planes(hi, lo, word_begin, seed, chunk=1 << 22):
    """hi[i], lo[i] = low / high half of splitmix64((word_begin + i) ^ seed), computed in place in
    fixed-size chunks (fresh multi-hundred-MB temporaries are slow to fault in on some hosts)."""
    n = len(hi)
    x = np.empty(min(chunk, max(n, 1)), dtype=np.uint64)
    t = np.empty_like(x)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        z, u = x[:m], t[:m]
        z[:] = np.arange(word_begin + s, word_begin + s + m, dtype=np.uint64)
        z ^= np.uint64(seed)
        z += np.uint64(0x9E3779B97F4A7C15)
        np.right_shift(z, np.uint64(30), out=u)
        z ^= u
        z *= np.uint64(0xBF58476D1CE4E5B9)
        np.right_shift(z, np.uint64(27), out=u)
        z ^= u
        z *= np.uint64(0x94D049BB133111EB)
        np.right_shift(z, np.uint64(31), out=u)
        z ^= u
        hi[s:s + m] = z  # truncating cast keeps the low 32 bits
        np.right_shift(z, np.uint64(32), out=u)
        lo[s:s + m] = u


def contig_table(total_bases, n_contigs=24):
    """hg38-proportioned contig lengths summing to total_bases, laid out with 1 N separator each."""
    ref = np.array(HG38_LENGTHS[:n_contigs], dtype=np.float64)
    lens = np.floor(ref * (total_bases / ref.sum())).astype(np.int64)
    lens[0] += total_bases - int(lens.sum())
    table = np.zeros(n_contigs, dtype=CONTIG_DTYPE)
    pos = 0
    for c, ln in enumerate(lens):
        table[c] = (pos, int(ln), 0)
        pos += int(ln) + 1
    return table, HG38_NAMES[:n_contigs]


def _n_intervals(table, seed):
    """Global [start, end) intervals that are N: separators, telomeres, ~1 % in 10 kb - 1 Mb runs."""
    rng = np.random.default_rng(seed)
    iv = []
    total = int(table["length"].sum())
    for row in table:
        o, ln = int(row["offset"]), int(row["length"])
        t = min(TELOMERE, ln // 4)
        iv.append((o, o + t))
        iv.append((o + ln - t, o + ln + 1))  # last 10 kb + the separator
    budget = total // 100
    while budget > 0 and total > 4 * TELOMERE:
        ln = int(rng.integers(10_000, 1_000_001))
        ln = min(ln, budget, max(1, total // 50))
        c = int(rng.integers(0, len(table)))
        o, cl = int(table[c]["offset"]), int(table[c]["length"])
        if cl <= ln:
            budget -= 1
            continue
        s = o + int(rng.integers(0, cl - ln))
        iv.append((s, s + ln))
        budget -= ln
    return iv


def _set_range(words, first_word, start, end):
    """Set bits [start, end) (global positions) in a word array whose element 0 is word first_word."""
    lo = max(start, first_word * 32)
    hi = min(end, (first_word + len(words)) * 32)
    if hi <= lo:
        return
    lo -= first_word * 32
    hi -= first_word * 32
    w0, w1 = lo >> 5, (hi - 1) >> 5
    m0 = np.uint32((0xFFFFFFFF << (lo & 31)) & 0xFFFFFFFF)
    m1 = np.uint32(0xFFFFFFFF >> (31 - ((hi - 1) & 31)))
    if w0 == w1:
        words[w0] |= m0 & m1
    else:
        words[w0] |= m0
        words[w0 + 1:w1] = np.uint32(0xFFFFFFFF)
        words[w1] |= m1


def synthetic_planes(total_bases, word_begin=None, word_end=None, seed=SEED_GENOME, n_contigs=24):
    """Planes (hi, lo, nmask) for words [word_begin, word_end) of the synthetic genome + its table.

    Returns (hi, lo, nmask, table, names, n_words_total)."""
    table, names = contig_table(total_bases, n_contigs)
    span = int(table[-1]["offset"]) + int(table[-1]["length"]) + 1
    n_words_total = (span + 31) // 32
    b = 0 if word_begin is None else word_begin
    e = n_words_total if word_end is None else min(word_end, n_words_total)
    hi = np.empty(e - b, dtype=np.uint32)
    lo = np.empty(e - b, dtype=np.uint32)
    _fill_planes(hi, lo, b, seed)
    nm = np.zeros(e - b, dtype=np.uint32)
    for s, t in _n_intervals(table, seed):
        _set_range(nm, b, s, t)
    _set_range(nm, b, span, (n_words_total + 1) * 32)
    return hi, lo, nm, table, names, n_words_total


def synthetic_genome(total_bases, seed=SEED_GENOME, n_contigs=24):
    hi, lo, nm, table, names, _ = synthetic_planes(total_bases, seed=seed, n_contigs=n_contigs)
    return PackedGenome(hi, lo, nm, table, names)


def synthetic_guides(n, seed=SEED_GUIDES):
    """n reads of 21 uniform bases + GG; ids g000000..."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    body = letters[rng.integers(0, 4, size=(n, 21))]
    seqs = [row.tobytes().decode() + "GG" for row in body]
    ids = ["g%06d" % i for i in range(n)]
    return ids, seqs


def plant_sites(packed, guides, n_sites, max_sub, seed=0x5EED0004):
    """Copies reads into the genome (either strand, 0..max_sub substitutions in the first 20 bases)
    at seeded positions well inside contigs.  Returns [(guide, contig, pos, strand, n_sub)]."""
    from . import _lib
    rng = np.random.default_rng(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    out = []
    L = _lib.lib()
    for _ in range(n_sites):
        gi = int(rng.integers(0, len(guides)))
        c = int(rng.integers(0, len(packed.contigs)))
        ln = int(packed.contigs[c]["length"])
        if ln < 4 * TELOMERE + 100:
            continue
        pos = int(rng.integers(2 * TELOMERE, ln - 2 * TELOMERE))
        nsub = int(rng.integers(0, max_sub + 1))
        site = list(guides[gi])
        for p in rng.choice(20, size=nsub, replace=False):
            site[p] = "ACGT"[("ACGT".index(site[p]) + 1 + int(rng.integers(0, 3))) % 4]
        strand = int(rng.integers(0, 2))
        s = "".join(site)
        if strand:
            s = "".join(comp[ch] for ch in reversed(s))
        L.vsc_pack_bases(s.encode(), 23, int(packed.contigs[c]["offset"]) + pos, _lib.ptr(packed.hi),
                         _lib.ptr(packed.lo), _lib.ptr(packed.nmask))
        out.append((gi, c, pos, strand, nsub))
    return out


SEED_VCF = 0x5EED0003


def write_fasta(packed, path, width=60, chunk=1 << 24):
    """The contigs of a packed genome as FASTA text (for tools that take the reference's file formats)."""
    import ctypes as C
    from ._lib import lib, ptr
    buf = C.create_string_buffer(chunk)
    with open(path, "wb") as f:
        for name, row in zip(packed.names, packed.contigs):
            f.write(b">" + name.encode() + b"\n")
            off, ln = int(row["offset"]), int(row["length"])
            for s in range(0, ln, chunk):
                m = min(chunk, ln - s)
                lib().vsc_unpack_bases(ptr(packed.hi), ptr(packed.lo), ptr(packed.nmask), off + s, m, buf)
                a = np.frombuffer(buf, dtype=np.uint8, count=m)
                full = (m // width) * width
                if full:
                    lines = np.empty((full // width, width + 1), dtype=np.uint8)
                    lines[:, :width] = a[:full].reshape(-1, width)
                    lines[:, width] = 10
                    f.write(lines.tobytes())
                if m > full:
                    f.write(a[full:m].tobytes() + b"\n")


def synthetic_vcf(packed, n_snps, path, seed=SEED_VCF):
    """SNP records at distinct uniformly random positions, ALT uniform over the three
    other bases, one sample, GT in {0|1, 1|0, 1|1, 0/1} with probabilities {.4, .4, .15, .05}."""
    rng = np.random.default_rng(seed)
    total = int(packed.contigs["length"].sum())
    flat = np.unique(rng.integers(0, total, size=int(n_snps * 1.02)))[:n_snps]
    ends = np.cumsum(packed.contigs["length"].astype(np.int64))
    cidx = np.searchsorted(ends, flat, side="right")
    local = flat - (ends[cidx] - packed.contigs["length"][cidx].astype(np.int64))
    gpos = packed.contigs["offset"][cidx].astype(np.int64) + local
    w, b = gpos >> 5, (gpos & 31).astype(np.uint32)
    isn = (packed.nmask[w] >> b) & 1
    code = (((packed.hi[w] >> b) & 1) << 1) | ((packed.lo[w] >> b) & 1)
    alt = (code + rng.integers(1, 4, size=len(code)).astype(np.uint32)) & 3
    gts = np.array(["0|1", "1|0", "1|1", "0/1"])[rng.choice(4, size=len(code), p=[.4, .4, .15, .05])]
    letters = np.array(list("ACGT"))
    keep = isn == 0
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n")
        for name, row in zip(packed.names, packed.contigs):
            f.write("##contig=<ID=%s,length=%d>\n" % (name, int(row["length"])))
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS0\n")
        names = np.array(packed.names)
        rows = np.char.add(np.char.add(np.char.add(names[cidx[keep]], "\t"), (local[keep] + 1).astype(str)),
                           np.char.add(np.char.add("\t.\t", letters[code[keep]]), np.char.add("\t", letters[alt[keep]])))
        rows = np.char.add(np.char.add(rows, "\t.\tPASS\t.\tGT\t"), gts[keep])
        f.write("\n".join(rows.tolist()) + "\n")
    return int(keep.sum())
