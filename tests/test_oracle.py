"""CPU-only: pins the oracle (oracle/) against the reference's golden vectors and checks its two
search formulations (SURVEY.md 8.1 predicate vs bidir_mapping.cpp control flow) and the
bit-parallel port against each other."""
import os

import numpy as np
import pytest

from helpers import hits_as_tuples, make_genome, mutate, random_guides, random_seq, revcomp


# ---------------------------------------------------------------- feature matrix (row R6)
def test_features_match_reference_golden(oracle, golden_dir):
    """All 6960 x 442 values of workflow/data-objects/featureMatrix.RData."""
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    on, off, feat = g["on"], g["off"], g["feat"]
    assert feat.shape == (6960, 442)
    bad = 0
    for i in range(len(on)):
        row = oracle.feature_row(str(on[i]), str(off[i]))
        if not np.array_equal(row, feat[i].astype(np.uint32)):
            bad += 1
    assert bad == 0


def test_feature_names_layout(golden_dir):
    """Column order of variant_processing/feature_matrix.h:155-202 as stored in the RData."""
    names = list(np.load(os.path.join(golden_dir, "features_golden.npz"))["names"])
    assert names[0] == "totalMismatches" and names[1] == "mismatchPos1" and names[21] == "mismatchPos21"
    assert names[22:34] == ["AtoC", "AtoG", "AtoT", "CtoA", "CtoG", "CtoT", "GtoA", "GtoC", "GtoT", "TtoA", "TtoC", "TtoG"]
    assert names[34:36] == ["transitionNumber", "transversionNumber"]
    assert names[36] == "A1" and names[115] == "T20" and names[116:120] == ["PAMA", "PAMC", "PAMG", "PAMT"]
    assert names[120] == "AA1" and names[423] == "TT19" and names[424] == "AA" and names[439] == "TT"
    assert names[440:442] == ["adjacentMismatches", "seedMismatches"]


# ---------------------------------------------------------------- MIT score (row R5)
# Known answers listed in SURVEY.md section 8 row R5 (produced from variant_processing/mit_score.h).
MIT_KATS = [
    ([-1], "100"), ([0], "100"), ([19], "41.7"), ([20], "100"), ([5, 7], "3.30316"), ([1, 3, 8], "1.59246"),
    ([3, 12, 19], "0.540776"), ([3, 12, 20], "3.11568"), ([0, 22], "100"),
    ([2, 5, 9, 13, 17, 19], "0.00433807"), ([1, 2, 10, 11, 12, 13, 14, 19], "0.000608023"),
    ([0, 1, 2, 3, 4, 5, 6, 7], "0.132918"),
]


@pytest.mark.parametrize("pos,expected", MIT_KATS)
def test_mit_known_answers(oracle, pos, expected):
    s, ub = oracle.mit_score(pos)
    assert "%g" % s == expected  # default ostream formatting = 6 significant digits
    assert not ub


def test_mit_exact_bits(oracle):
    s, _ = oracle.mit_score([5, 7])
    assert s.hex() == "0x1.a6cdfa1d6cdf9p+1"


def test_mit_reference_ub_is_flagged(oracle):
    # two positions >= 20: the reference indexes its 20-entry weight vector out of bounds (8.2 Q2)
    s, ub = oracle.mit_score([3, 20, 22])
    assert ub and s > 0


def test_mit_against_crispor_rows(oracle, golden_dir):
    """Rows of the reference's CRISPOR table where CRISPOR's formula coincides with mit_score.h."""
    n = 0
    with open(os.path.join(golden_dir, "crispor_mit.tsv")) as f:
        next(f)
        for line in f:
            g, o, score = line.split()
            pos = [i for i in range(20) if g[i] != o[i]] or [-1]
            s, ub = oracle.mit_score(pos)
            assert not ub
            assert s == pytest.approx(float(score), rel=1e-9), (g, o)
            n += 1
    assert n == 2425


# ---------------------------------------------------------------- MD strings
def test_md_styles_and_reference_parser(oracle):
    w = "ACGTACGTACGTACGTACGTAGG"
    r = "ACGTACGTACGTACGTACGTAGG"
    assert oracle.md_string(w, r, 0) == "23" and oracle.md_positions("23") == [-1]
    r2 = "TCGTAGCTACGTACGTACGTAGC"  # mismatches at 0, 5, 6, 22
    assert oracle.md_string(w, r2, 0) == "0A4C0G15G0"
    assert oracle.md_positions("0A4C0G15G0") == [0, 5, 6, 22]
    assert oracle.md_string(w, r2, 1) == "A4CG15G"
    # filter_output_bam.h:338-342 stops at the first token that is not <number><char>
    assert oracle.md_positions("A4CG15G") == [-1]
    assert oracle.md_positions("5AC16") == [5]


# ---------------------------------------------------------------- search: predicate == reference flow
def _r4_order(block):
    """bidir_mapping.cpp:167-187 applied to one (guide, strand) block sorted by (contig, pos)."""
    out, best = [], 0
    for i in range(1, len(block)):
        if block[i][4] >= block[best][4]:
            out.append(block[i] + (1,))
        else:
            out.append(block[best] + (1,))
            best = i
    out.append(block[best] + (0,))
    return out


@pytest.mark.parametrize("seed,max_mm,extra_pam", [(1, 0, None), (2, 1, None), (3, 2, None), (4, 3, "AG"),
                                                    (5, 4, None), (6, 5, "TT"), (7, 6, None), (8, 7, None),
                                                    (9, 8, None), (10, 8, "CC")])
def test_predicate_equals_reference_flow(oracle, seed, max_mm, extra_pam):
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, 6) + [random_seq(rng, 23)]
    contigs = make_genome(seed, [3000, 22, 23, 24, 700, 5, 1500], guides, max_mm)
    a = oracle.search(contigs, guides, max_mm, extra_pam, mode=oracle.MODE_PREDICATE)
    b = oracle.search(contigs, guides, max_mm, extra_pam, mode=oracle.MODE_REFERENCE_FLOW)
    ta, tb = hits_as_tuples(a), hits_as_tuples(b)
    assert len(ta) > 10
    assert sorted(ta) == sorted(tb)
    assert ta == sorted(ta)
    # emission order and secondary flags of the reference flow
    expect = []
    blocks = {}
    for t in ta:
        blocks.setdefault((t[0], t[1]), []).append(t)
    for key in sorted(blocks):
        expect += _r4_order(blocks[key])
    got = [t + (int(i >> 30) & 1,) for t, i in zip(tb, b["info"])]
    assert got == expect


def test_hits_are_sound(oracle):
    """Every reported hit satisfies the definition when re-derived in plain Python."""
    rng = np.random.default_rng(77)
    guides = random_guides(rng, 5)
    contigs = make_genome(77, [4000, 900], guides, 6)
    for g, s, c, p, nm, mask in hits_as_tuples(oracle.search(contigs, guides, 6)):
        w = contigs[c][p:p + 23]
        r = revcomp(guides[g]) if s else guides[g]
        assert "N" not in w and len(w) == 23
        assert (w[21:] in ("GG", "GA")) if not s else (w[:2] in ("CC", "TC"))
        mm = [i for i in range(23) if w[i] != r[i]]
        assert len(mm) == nm <= 6 and mask == sum(1 << i for i in mm)


def test_right_edge_rule(oracle):
    """A window that ends exactly at the contig end is only reported through the second-half route
    (bidir_mapping.cpp:51-52): it needs <= floor(m/2) mismatches in read[11..23)."""
    g = "ACGTTGCATGCAAGTCCTAGTGG"
    pad = "T" * 50
    ok = mutate(np.random.default_rng(1), g, 2, 0, 11)      # errors in the first half only
    bad = g[:11] + "".join("A" if c != "A" else "C" for c in g[11:14]) + g[14:]  # 3 errors in second half
    for site, expected in ((ok, 1), (bad, 0)):
        h = oracle.search([pad + site], [g], 4)            # k = 2
        assert len(h) == expected
        h2 = oracle.search([pad + site + "T"], [g], 4)     # one base further in: both routes apply
        assert len(h2) == 1
    # reverse strand: the read is revcomp(g); its second half is the first half of g reversed
    h = oracle.search([pad + revcomp(bad)], [g], 4)
    assert len(h) == 1
    h = oracle.search([pad + revcomp(g[:3] + "".join("A" if c != "A" else "C" for c in g[3:6]) + g[6:])], [g], 4)
    assert len(h) == 0


def test_non_acgt_guide_letters_become_A(oracle):
    contig = "T" * 30 + "AAGTTGCATGCAAGTCCTAGTGG" + "T" * 30
    h = oracle.search([contig], ["NXGTTGCATGCAAGTCCTAGTGG"], 0)
    assert len(h) == 1 and h["pos"][0] == 30


def test_lowercase_and_iupac_genome(oracle):
    contig = "t" * 30 + "acgttgcatgcaagtcctagtgg" + "R" + "T" * 30
    h = oracle.search([contig], ["ACGTTGCATGCAAGTCCTAGTGG"], 0)
    assert len(h) == 1
    contig = "t" * 30 + "acgttgcatgcRagtcctagtgg" + "T" * 30
    assert len(oracle.search([contig], ["ACGTTGCATGCAAGTCCTAGTGG"], 8)) == 0


def test_mismatch_budget_is_validated(oracle):
    with pytest.raises(ValueError):
        oracle.search(["ACGT" * 20], ["ACGTTGCATGCAAGTCCTAGTGG"], 9)


def test_u16_contig_key_compat(oracle):
    """bidir_mapping.cpp:13: with the uint16_t key a hit in contig 65536 + c at the offset of a hit in
    contig c is dropped."""
    g = "ACGTTGCATGCAAGTCCTAGTGG"
    contigs = ["T" * 5 + g + "T" * 5] + ["TTTT"] * 65535 + ["A" * 5 + g + "A" * 5]
    full = oracle.search(contigs, [g], 0, mode=oracle.MODE_REFERENCE_FLOW)
    compat = oracle.search(contigs, [g], 0, mode=oracle.MODE_REFERENCE_FLOW, compat_u16=True)
    assert len(full) == 2 and len(compat) == 1


# ---------------------------------------------------------------- bit-parallel port == restatement
@pytest.mark.parametrize("seed,max_mm,extra_pam", [(21, 0, None), (22, 3, None), (23, 4, "AG"), (24, 6, None),
                                                    (25, 8, None), (26, 8, "TA")])
def test_fast_port_equals_restatement(oracle, seed, max_mm, extra_pam):
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, 8) + [random_seq(rng, 23)]
    contigs = make_genome(seed, [5000, 23, 22, 1200, 64, 2500], guides, max_mm)
    a = oracle.search(contigs, guides, max_mm, extra_pam)
    for threads in (1, 3):
        b = oracle.search_fast(contigs, guides, max_mm, extra_pam, threads=threads)
        assert hits_as_tuples(a) == hits_as_tuples(b)
    n, sites = oracle.count_fast(contigs, guides, max_mm, extra_pam)
    assert n == len(a) and sites > 0


# ---------------------------------------------------------------- pigeonhole port == restatement
@pytest.mark.parametrize("seed,max_mm,extra_pam", [(31, 0, None), (32, 1, None), (33, 3, "AG"), (34, 4, None), (35, 6, None),
                                                    (36, 7, "TT"), (37, 8, None)])
def test_pigeonhole_port_equals_restatement(oracle, seed, max_mm, extra_pam):
    """vsc_pigeon.c - halves with floor(m/2) substitutions through a k-mer table, then the delegate - returns
    the records of the character-level restatement (both of its formulations), including contig-end windows
    only the second-half route may report, contigs shorter than a read and N runs."""
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, 8) + [random_seq(rng, 23)]
    contigs = make_genome(seed, [5000, 23, 22, 1200, 64, 2500, 11, 12], guides, max_mm)
    a = oracle.search(contigs, guides, max_mm, extra_pam)
    b = oracle.search(contigs, guides, max_mm, extra_pam, mode=oracle.MODE_REFERENCE_FLOW)
    ix = oracle.PigeonIndex(contigs)
    for threads in (1, 4):
        got, candidates = ix.search(guides, max_mm, extra_pam, threads=threads)
        assert hits_as_tuples(got) == hits_as_tuples(a)
        assert candidates >= len(a)
    n, _ = ix.count(guides, max_mm, extra_pam)
    assert n == len(a) and sorted(hits_as_tuples(b)) == hits_as_tuples(a)
    ix.close()


def test_sam_text(oracle):
    g = "ACGTTGCATGCAAGTCCTAGTGG"
    site = "ACGTTGCATGCAAGTCCTAGTGA"  # GA PAM, 1 mismatch at 22
    contigs = ["T" * 10 + site + "T" * 10, "C" * 8 + revcomp(g) + "C" * 8]
    sam = oracle.search_sam(contigs, ["chrA", "chrB"], [g], ["guide1"], 2)
    lines = sam.splitlines()
    assert lines == [
        "guide1\t0\tchrA\t11\t255\t23M\t*\t0\t0\t%s\t%s\tNM:i:1\tMD:Z:22A0" % (g, "I" * 23),
        "guide1\t16\tchrB\t9\t255\t23M\t*\t0\t0\t%s\t%s\tNM:i:0\tMD:Z:23" % (g, "I" * 23),
    ]


# ---------------------------------------------------------------- the reference mapper's own NM tags (rows R2/R3)
@pytest.mark.parametrize("mode", ["predicate", "reference_flow", "fastport", "pigeon"])
def test_search_reports_the_reference_mappers_hits_with_its_nm(oracle, golden_dir, mode):
    """Pins rows R2/R3 with the one search output the reference still holds: 2 779 distinct (guide, site sequence, NM)
    triples out of VARSCOT's own SAM file (datasetsSampling.RData Class 0: the mapper's NM tags; the 348 GUIDE-seq
    sites the reference found in that file, indexGuideSeq.RData - helpers.reference_sam_triples).
    Every site the reference's mapper reported at <= 8 mismatches must be reported by every formulation of the
    oracle, on the strand it is planted on, with the reference's NM (mismatches over all 23 positions,
    bidir_mapping.cpp:79-86,121), and must appear exactly when the budget reaches that NM (acceptance `<= m`)."""
    from helpers import plant_reference_sites, reference_sam_triples
    guides, rows = reference_sam_triples(golden_dir)
    assert len(rows) == 2779 and len(guides) == 9
    contigs, strands = plant_reference_sites(rows)
    pig = oracle.PigeonIndex(contigs) if mode == "pigeon" else None
    for m in (8, 7, 6, 5, 4, 3, 2, 1):
        if mode == "predicate":
            h = oracle.search(contigs, guides, m, mode=oracle.MODE_PREDICATE)
        elif mode == "reference_flow":
            h = oracle.search(contigs, guides, m, mode=oracle.MODE_REFERENCE_FLOW)
        elif mode == "fastport":
            h = oracle.search_fast(contigs, guides, m)
        else:
            h, _ = pig.search(guides, m)
        got = {(g, c): (s, nm) for g, s, c, p, nm, _ in hits_as_tuples(h) if p == 4}
        for c, (gi, site, nm) in enumerate(rows):
            if nm <= m:
                assert got.get((gi, c)) == (strands[c], nm), (guides[gi], site, nm, m)
            else:
                assert (gi, c) not in got, (guides[gi], site, nm, m)
    if pig:
        pig.close()


def test_reference_sam_triples_are_what_the_generator_says(golden_dir):
    """The fixture itself: NM of every stored row = Hamming distance over all 23 positions (the reference counts
    the PAM positions too), all Class-0 sites carry an NGG PAM, 2 <= NM <= 8 (the mapper ran with -M 8)."""
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    hd = np.array([sum(x != y for x, y in zip(str(a), str(o))) for a, o in zip(g["on"], g["off"])])
    assert np.array_equal(hd, g["nm"])
    sel = g["cls"] == 0
    assert sel.sum() == 3480 and all(str(o)[21:] == "GG" for o in g["off"][sel])
    assert g["nm"][sel].min() == 2 and g["nm"][sel].max() == 8


# ---- what is left of the ORDER of the reference's SAM output (row R4) ---------------------------------------------------
def test_reference_sam_rows_have_the_write_order_of_bidir_mapping(golden_dir):
    """The fixture alone: the rows of VARSCOT's own SAM file in which the reference found its 348 GUIDE-seq sites
    (indexGuideSeq.RData) are what bidir_mapping.cpp:167-187,285-295 writes - reads in input order; per read every
    '+' record before every '-' record; inside a block ascending (contig in FASTA order, position), except for a
    handful of LATE records, each of which has strictly fewer mismatches than every record that sorts before it:
    the best-so-far record that is held back and written when displaced or at the end of its block."""
    from helpers import UCSC_HG19_ORDER, check_block_order, guideseq_sam_rows, real_guides
    rows = guideseq_sam_rows(golden_dir)
    names = real_guides(golden_dir)[0][:9]
    rank = {c: i for i, c in enumerate(UCSC_HG19_ORDER)}
    last, n_late = 0, 0
    for t in names:  # reads in input order: disjoint, ascending row ranges
        mine = [r for r in rows if r[0] == t]
        assert mine and min(r[5] for r in mine) > last
        last = max(r[5] for r in mine)
        plus = sorted((r[5], (rank[r[1]], r[2]), r[4]) for r in mine if r[3] == "+")
        minus = sorted((r[5], (rank[r[1]], r[2]), r[4]) for r in mine if r[3] == "-")
        if plus and minus:
            assert plus[-1][0] < minus[0][0]  # the '+' block is written before the '-' block
        for blk in (plus, minus):
            n_late += len(check_block_order([(key, nm, row) for row, key, nm in blk]))
    assert 1 <= n_late <= 12  # (9 in the fixture)
    # a FASTA in natural or lexicographic chromosome order would NOT give ascending blocks: the order is hg19.fa's
    natural = {"chr%s" % c: i for i, c in enumerate(list(range(1, 23)) + ["X", "Y"])}
    broken = 0
    for t in names:
        for strand in "+-":
            blk = sorted((r[5], (natural[r[1]], r[2])) for r in rows if r[0] == t and r[3] == strand)
            broken += sum(1 for a, b in zip(blk, blk[1:]) if b[1] < a[1])
    assert broken >= 20  # (against 9 under hg19.fa order, all of them held-back records)


@pytest.mark.parametrize("mode", ["reference_flow"])
def test_write_order_equals_the_reference_rows_on_the_guideseq_sites(oracle, golden_dir, mode):
    """The oracle's REFERENCE_FLOW order (what the product's vsc_sam_order is tested against) on a genome that carries the
    348 GUIDE-seq sites on the reference's contigs (UCSC hg19 order), in their real order along each chromosome and on
    their real strand: apart from the records either side holds back as best-so-far (which depends on the other
    1.7 M records of the real genome), the sites come out in EXACTLY the order of the reference's SAM rows."""
    from helpers import guideseq_mini_genome, late_records
    names, guides, cnames, contigs, planted = guideseq_mini_genome(golden_dir)
    h = oracle.search(contigs, guides, 8, mode=oracle.MODE_REFERENCE_FLOW)
    where = {(g, c, p, s): i for i, (g, s, c, p, nm, _) in enumerate(hits_as_tuples(h))}
    ours = []
    for i, (g, c, p, s, nm, row) in enumerate(planted):
        assert (g, c, p, s) in where, planted[i]
        assert hits_as_tuples(h[where[(g, c, p, s)]:where[(g, c, p, s)] + 1])[0][4] == nm
        ours.append((where[(g, c, p, s)], i))
    ours = [i for _, i in sorted(ours)]                                   # fixture rows in OUR output order
    theirs = [i for _, i in sorted((planted[i][5], i) for i in range(len(planted)))]   # ... in the reference's
    held = set()
    for seq in (ours, theirs):
        for g in range(len(guides)):
            for s in (0, 1):
                blk = [((planted[i][1], planted[i][2]), planted[i][4], i) for i in seq if planted[i][0] == g and planted[i][3] == s]
                held |= set(late_records(blk))
    assert len(held) <= 40  # (29: 9 in the reference rows, the prefix minima of NM of every block here)
    assert [i for i in ours if i not in held] == [i for i in theirs if i not in held]
    assert len(ours) - len(held) > 300
