"""Shared test helpers: seeded random genomes with planted hits, N runs and edge windows."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def random_seq(rng, n):
    return BASES[rng.integers(0, 4, size=n)].tobytes().decode()


def random_guides(rng, n, pam="GG"):
    return [random_seq(rng, 21) + pam for _ in range(n)]


def mutate(rng, s, nsub, lo=0, hi=None):
    """Substitute nsub distinct positions of s in [lo, hi) with a different base."""
    hi = len(s) if hi is None else hi
    s = list(s)
    for p in rng.choice(np.arange(lo, hi), size=nsub, replace=False):
        s[p] = rng.choice([b for b in "ACGT" if b != s[p]])
    return "".join(s)


def plant(rng, contig, guide, pos, strand, nsub, keep_pam=True):
    """Write guide (or its reverse complement) with nsub substitutions at pos; returns the new contig."""
    site = mutate(rng, guide, nsub, 0, 20 if keep_pam else 23)
    if strand == "-":
        site = revcomp(site)
    return contig[:pos] + site + contig[pos + 23:]


def make_genome(seed, contig_lens, guides, max_mm, n_plant=40, n_runs=3, edge_plants=True):
    """Random contigs with planted near-matches (both strands, 0..max_mm substitutions), runs of N,
    and near-matches that touch the contig ends (the right-edge rule of bidir_mapping.cpp:51-52)."""
    rng = np.random.default_rng(seed)
    contigs = [random_seq(rng, n) for n in contig_lens]
    big = [i for i, n in enumerate(contig_lens) if n >= 200]
    for _ in range(n_runs):
        if not big:
            break
        c = int(rng.choice(big))
        a = int(rng.integers(0, len(contigs[c]) - 60))
        ln = int(rng.integers(1, 50))
        contigs[c] = contigs[c][:a] + "N" * ln + contigs[c][a + ln:]
    for _ in range(n_plant):
        if not big or not guides:
            break
        c = int(rng.choice(big))
        g = guides[int(rng.integers(0, len(guides)))]
        pos = int(rng.integers(0, len(contigs[c]) - 23 + 1))
        contigs[c] = plant(rng, contigs[c], g, pos, rng.choice(["+", "-"]), int(rng.integers(0, max_mm + 1)),
                           keep_pam=bool(rng.integers(0, 4)))
    if edge_plants and guides:
        for c in big:
            g = guides[int(rng.integers(0, len(guides)))]
            # window ending exactly at the contig end, on either strand, few / many second-half errors
            L = len(contigs[c])
            site = g if rng.integers(0, 2) else mutate(rng, g, min(max_mm, 3), 11, 20)
            contigs[c] = contigs[c][:L - 23] + (site if rng.integers(0, 2) else revcomp(site))
            g2 = guides[int(rng.integers(0, len(guides)))]
            site2 = mutate(rng, g2, min(max_mm, 2), 0, 11)
            contigs[c] = (site2 if rng.integers(0, 2) else revcomp(site2)) + contigs[c][23:]
    return contigs


def hits_as_tuples(h):
    """(guide, strand, contig, pos, nm, mask) tuples, ignoring the secondary flag."""
    return [(int(a), int(i >> 31), int(c), int(p), int((i >> 23) & 31), int(i & 0x7FFFFF))
            for a, c, p, i in zip(h["guide"], h["contig"], h["pos"], h["info"])]


def real_guides(golden_dir):
    """(names, 23-mers, activities) of the reference's own on-target list (workflow/guideseq-data/
    guideseqOntargets.fasta + guideseqOntargetActivity.txt): several are G-rich / low complexity."""
    import os
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(golden_dir, "guides_ontargets.tsv")) if not l.startswith("#")]
    return [r[0] for r in rows], [r[1] for r in rows], [float(r[2]) for r in rows]


def repeat_rich_genome(seed, n_bases, guides, n_contigs=3):
    """A genome that is NOT uniform: Alu-like and L1-like repeat families (hundreds to thousands of diverged
    copies), one family built around a guide (thousands of near-hits for that read), homopolymer tracts, tandem
    repeats, GC-rich islands and N gaps.  What the uniform synthetic genome never exercises: hit buffers sized
    from the uniform model overflow, single reads own whole bins of the sort, buckets of the seed index differ in
    size by orders of magnitude."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = lut[rng.integers(0, 4, size=n_bases)].copy()

    def diverged(cons, rate):
        c = cons.copy()
        m = rng.random(len(c)) < rate
        c[m] = lut[rng.integers(0, 4, size=int(m.sum()))]
        return c

    def paste(piece):
        at = int(rng.integers(0, n_bases - len(piece)))
        seq[at:at + len(piece)] = piece

    g = [np.frombuffer(x.encode(), dtype=np.uint8) for x in guides]
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    alu = lut[rng.integers(0, 4, size=300)].copy()
    alu[100:123] = g[0]                       # a family that carries the first guide ...
    alu[200:223] = comp[g[1][::-1]]           # ... and the reverse complement of the second
    for _ in range(max(50, n_bases // 4000)):
        paste(diverged(alu, 0.12))
    l1 = lut[rng.integers(0, 4, size=2000)].copy()
    for _ in range(max(5, n_bases // 200000)):
        paste(diverged(l1, 0.05))
    for _ in range(max(20, n_bases // 50000)):    # homopolymers and tandem repeats
        unit = [b"A", b"T", b"G", b"C", b"CA", b"GGAA", b"CAG", b"GT", b"GGGGCC"][int(rng.integers(0, 9))]
        reps = int(rng.integers(20, 400)) // len(unit) + 1
        paste(np.frombuffer(unit * reps, dtype=np.uint8))
    for _ in range(max(5, n_bases // 500000)):    # GC-rich islands
        n = int(rng.integers(200, 2000))
        paste(lut[rng.choice(4, size=n, p=[0.1, 0.4, 0.4, 0.1])])
    for gi in range(len(g)):                      # a handful of close copies of every guide, both strands
        for _ in range(6):
            piece = diverged(g[gi], 0.15)
            piece[21:] = g[gi][21:]
            paste(piece if rng.integers(0, 2) else comp[piece[::-1]])
    for _ in range(4):                             # N gaps
        at = int(rng.integers(0, n_bases - 5000))
        seq[at:at + int(rng.integers(1, 3000))] = ord("N")
    cuts = sorted(int(x) for x in rng.choice(np.arange(1000, n_bases - 1000), size=n_contigs - 1, replace=False))
    return [p.tobytes().decode() for p in np.split(seq, cuts)]


def reference_sam_triples(golden_dir):
    """The (guide, site, NM) triples of VARSCOT's own SAM output that the reference still holds:
    * the Class-0 rows of workflow/data-objects/datasetsSampling.RData (workflow/processDataForModel.R:257-258 reads
      guideseq-data/bidir_guideseq.sam, :284 extracts the `NM:i` tag, :378-394 samples the records; the SAM itself is
      an absent LFS blob): NM is the reference mapper's own tag;
    * the 348 GUIDE-seq sites (Class 1), every one of which the reference found in that SAM output
      (data-objects/indexGuideSeq.RData, processDataForModel.R:262-279: all row numbers >= 1) - reported sites whose NM
      is the script's stringDiff over all 23 positions (:238-240).
    Returns (guides, rows) with rows = unique (guide index, site in guide orientation, NM)."""
    import os
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    sel = (g["cls"] == 0) | (g["mapper_row"] >= 1)
    guides = sorted(set(str(s) for s in g["on"][sel]))
    gidx = {s: i for i, s in enumerate(guides)}
    rows = sorted(set((gidx[str(a)], str(o), int(n)) for a, o, n in zip(g["on"][sel], g["off"][sel], g["nm"][sel])))
    return guides, rows


def plant_reference_sites(rows):
    """One contig per triple: the site on `+` (even rows) or reverse-complemented = a `-` hit (odd rows), in T / A
    flanks that add no PAM of their own next to it.  Returns (contigs, strand per row); the window starts at 4."""
    contigs, strands = [], []
    for i, (_, site, _) in enumerate(rows):
        s = i & 1
        contigs.append("TTTT" + (revcomp(site) if s else site) + ("AAAA" if s else "TTTT"))
        strands.append(s)
    return contigs, strands


# ---- the 8-byte exchange record of the multi-GPU drivers (include/varscot_hip.h: vsc_hits_pack_exchange) in numpy:
# what the GPU kernels xpack_kernel / merge_packed_kernel do, restated for the CPU-only tests of the exchange code
def xpack(hits, contig_offsets, n_reads):
    """(uint64 records: mask | global position << 23, uint32[2 * n_reads] per-key counts) of sorted vsc_hit records."""
    gpos = contig_offsets[hits["contig"]].astype(np.uint64) + hits["pos"].astype(np.uint64)
    rec = (gpos << np.uint64(23)) | (hits["info"] & np.uint32(0x7FFFFF)).astype(np.uint64)
    key = (hits["guide"].astype(np.int64) << 1) | (hits["info"] >> 31)
    return rec, np.bincount(key, minlength=2 * n_reads).astype(np.uint32)


def xmerge(records, key_counts, first_key, contig_offsets):
    """vsc_hit records from the concatenated exchange records of all shards (key_counts: [n_shards, n_keys])."""
    HIT = np.dtype([("guide", "<u4"), ("contig", "<u4"), ("pos", "<u4"), ("info", "<u4")])
    n_shards, n_keys = key_counts.shape
    starts = np.concatenate([[0], np.cumsum(key_counts.astype(np.int64).ravel())])  # shard-major, key-minor
    out = np.zeros(len(records), dtype=HIT)
    at = 0
    for k in range(n_keys):
        for s in range(n_shards):
            a = starts[s * n_keys + k]
            n = int(key_counts[s, k])
            r = records[a:a + n]
            gpos = (r >> np.uint64(23)).astype(np.int64)
            mask = (r & np.uint64(0x7FFFFF)).astype(np.uint32)
            c = np.searchsorted(contig_offsets.astype(np.int64), gpos, side="right") - 1
            key = first_key + k
            out["guide"][at:at + n] = key >> 1
            out["contig"][at:at + n] = c
            out["pos"][at:at + n] = gpos - contig_offsets.astype(np.int64)[c]
            nm = np.array([bin(int(m)).count("1") for m in mask], dtype=np.uint32)
            out["info"][at:at + n] = ((key & 1) << 31) | (nm << 23) | mask
            at += n
    return out


# ---- what is left of the ORDER of the reference's SAM output (row R4) -------------------------------------------------
# tests/golden/guideseq_sam_rows.tsv: the 348 GUIDE-seq sites with their strand and the row of VARSCOT's own SAM file the
# reference found them in.  bidir_mapping.cpp:167-187 writes, per read, a '+' block then a '-' block, each in ascending
# (contig, position) order of its std::map - except that the record with the fewest mismatches so far is held back and
# written when a better one displaces it, or at the end of its block.  Contig ids are FASTA order; the reference mapped
# against UCSC's hg19.fa, whose order is by size:
UCSC_HG19_ORDER = ["chr" + c for c in "1 2 3 4 5 6 7 X 8 9 10 11 12 13 14 15 16 17 18 20 Y 19 22 21".split()]


def guideseq_sam_rows(golden_dir):
    """[(target, chrom, start, strand, nm, sam_row)] in fixture order (= the Class-1 rows of features_golden.npz)."""
    import os
    out = []
    for line in open(os.path.join(golden_dir, "guideseq_sam_rows.tsv")):
        if not line.startswith("#"):
            t, c, s, st, nm, r = line.rstrip("\n").split("\t")
            out.append((t, c, int(s), st, int(nm), int(r)))
    return out


def late_records(block):
    """block: [(sort key, nm, id)] in OUTPUT order of one (read, strand) block.  Returns the ids of the records that come
    later than their place in ascending key order (smaller key than a record written before them): under
    bidir_mapping.cpp:170-187 exactly the records that were held back as the best so far."""
    late, run_max = [], None
    for key, nm, ident in block:
        if run_max is not None and key < run_max:
            late.append(ident)
        else:
            run_max = key
    return late


def check_block_order(block):
    """The signature of bidir_mapping.cpp:167-187 on one block in output order: ascending keys except for late records,
    and every late record has strictly fewer mismatches than every record that sorts before it (it was the best so far
    when the map iteration reached it)."""
    late = set(late_records(block))
    for key, nm, ident in block:
        if ident in late:
            assert all(n2 > nm for k2, n2, _ in block if k2 < key), (key, nm)
    rest = [key for key, _, ident in block if ident not in late]
    assert rest == sorted(rest)
    return late


def guideseq_mini_genome(golden_dir):
    """A small genome with the reference's contig names in UCSC hg19 order that carries the 348 GUIDE-seq sites on their
    chromosome, in their real relative order along it, on their real strand.  Returns (guide names, guides, contig names,
    contigs, planted) with planted[i] = (guide index, contig index, position, strand 0/1, nm, sam_row) in fixture order."""
    import os
    rows = guideseq_sam_rows(golden_dir)
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    names, guides, _ = real_guides(golden_dir)
    names, guides = names[:9], guides[:9]  # the GUIDE-seq targets, in the order of guideseqOntargets.fasta
    sites = [str(s) for s in g["off"][:len(rows)]]
    assert [str(t) for t in g["target"][:len(rows)]] == [r[0] for r in rows]
    per_contig = {c: [] for c in UCSC_HG19_ORDER}
    for i, (t, c, start, strand, nm, row) in enumerate(rows):
        per_contig[c].append((start, i))
    contigs, planted = [], [None] * len(rows)
    for ci, c in enumerate(UCSC_HG19_ORDER):
        seq = ["TTTTTTTT"]
        for start, i in sorted(per_contig[c]):
            t, _, _, strand, nm, row = rows[i]
            pos = sum(len(x) for x in seq)
            site = sites[i] if strand == "+" else revcomp(sites[i])
            seq.append(site + ("TTTTTTTTTT" if strand == "+" else "AAAAAAAAAA"))
            planted[i] = (names.index(t), ci, pos, 0 if strand == "+" else 1, nm, row)
        contigs.append("".join(seq))
    return names, guides, list(UCSC_HG19_ORDER), contigs, planted
