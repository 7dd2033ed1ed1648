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
