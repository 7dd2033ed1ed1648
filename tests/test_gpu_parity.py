"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs - bit-exact hit sets, NM, mismatch masks, MIT scores and feature rows."""
import os

import numpy as np
import pytest

import varscot_amd as va
from helpers import real_guides, repeat_rich_genome, hits_as_tuples, make_genome, mutate, random_guides, random_seq, revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = va.Context(0)
    yield c
    c.close()


@pytest.fixture
def hooks(ctx):
    """ctx.set_debug(...) for one test (include/varscot_hip_debug.h); the defaults come back afterwards."""
    yield ctx.set_debug
    ctx.set_debug()


ALGOS = ["scan", "seed"]


def gpu_search(ctx, contigs, guides, max_mm, extra_pam=None, world=1, algo="scan"):
    packed = va.PackedGenome.from_sequences(contigs)
    out = []
    for rank in range(world):
        b, e = packed.shard_words(rank, world)
        if e <= b:
            continue
        g = ctx.load_genome(packed, rank, world)
        h = g.search(guides, max_mm, extra_pam, algorithm=algo)
        assert ctx.timing()["algorithm"] == {"scan": 1, "seed": 2}[algo]
        out.append(h.to_numpy())
        h.close()
        g.close()
    return np.concatenate(out) if out else np.zeros(0, dtype=va.HIT_DTYPE)


CASES = [(101, 0, None), (102, 1, None), (103, 2, "AG"), (104, 3, None), (105, 4, None), (106, 5, "TT"),
         (107, 6, None), (108, 7, None), (109, 8, None), (110, 8, "CC")]


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("seed,max_mm,extra_pam", CASES)
def test_search_matches_oracle(ctx, oracle, seed, max_mm, extra_pam, algo):
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, 9) + [random_seq(rng, 23), "NNGT" + random_seq(rng, 17) + "GG"]
    contigs = make_genome(seed, [9000, 22, 23, 24, 700, 5, 4100, 2049, 64], guides, max_mm, n_plant=120, n_runs=6)
    want = oracle.search(contigs, guides, max_mm, extra_pam, mode=oracle.MODE_PREDICATE)
    got = gpu_search(ctx, contigs, guides, max_mm, extra_pam, algo=algo)
    assert len(want) > 20
    assert hits_as_tuples(got) == hits_as_tuples(want)  # same records in the same (sorted) order


@pytest.mark.parametrize("shared", [0, 1, 2])
@pytest.mark.parametrize("seed,max_mm,n_reads", [(201, 8, 300), (202, 6, 40), (203, 2, 7), (204, 7, 1500)])
def test_seed_search_with_a_chunk_per_wave_and_per_workgroup(ctx, oracle, hooks, shared, seed, max_mm, n_reads):
    """seed_sliced_kernel<false> (every wave its own chunks) and <true> (the four waves of a workgroup share a chunk
    and take a quarter of its read list each: what dense searches like c3 run) return the oracle's records - forced
    either way by the hook, on read sets from 7 (most quarters empty) to 1 500 reads (lists of several tiles), with
    planted sites, N runs, tiny contigs and contig-end windows.  shared = 2: the sharing kernel with the workgroup's
    output blocks shared as well (small blocks so that the cross-wave reservation path runs often)."""
    if shared == 2:
        hooks(seed_shared=1, seed_group_out=1, seed_reserve=64)
    else:
        hooks(seed_shared=shared)
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, n_reads)
    contigs = make_genome(seed, [70000, 23, 9000, 30000, 64], guides[:60], max_mm, n_plant=400, n_runs=5)
    want = oracle.search_fast(contigs, guides, max_mm)
    got = gpu_search(ctx, contigs, guides, max_mm, algo="seed")
    assert len(want) > 100
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("max_mm", range(0, 9))
def test_seed_search_reports_a_valid_cut(ctx, oracle, hooks, max_mm):
    """vsc_timing.seed_cut = k0 | k1 << 4: whatever the cost model (or a hook) picks must keep the third segment's threshold -
    what the most generous PAM class leaves of the limit, less k0 + k1 + 2 - within two substitutions, and the records must be
    the oracle's.  Reads that end in GG leave the whole limit against GG sites."""
    rng = np.random.default_rng(900 + max_mm)
    guides = random_guides(rng, 40, pam="GG")
    contigs = make_genome(900 + max_mm, [50000, 30000], guides[:10], max_mm, n_plant=200, n_runs=2)
    want = oracle.search_fast(contigs, guides, max_mm)
    for hook in (-1, 0, 3, 7, 9):
        hooks(seed_tight=hook)
        got = gpu_search(ctx, contigs, guides, max_mm, algo="seed")
        cut = ctx.timing()["seed_cut"]
        k0, k1 = cut & 15, cut >> 4
        assert k0 <= 2 and k1 <= 2
        if hook == 0:
            assert k0 == k1 == max_mm // 3
        else:
            assert max_mm - k0 - k1 - 2 <= 2
        assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("tight", [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("shared", [0, 1])
@pytest.mark.parametrize("seed,max_mm,extra_pam", [(301, 8, None), (302, 5, None), (303, 1, None), (304, 0, None),
                                                   (305, 6, "GT"), (306, 6, "AG"), (327, 7, "GA"), (308, 4, "TT"), (309, 3, None), (310, 2, "CG"), (311, 6, None), (312, 5, "GA")])
def test_seed_search_reads_that_mismatch_the_pam(ctx, oracle, hooks, tight, shared, seed, max_mm, extra_pam):
    """A chunk of the seed index holds sites of ONE PAM (its class), so the sliced comparison leaves read positions 21
    and 22 out: the read's mismatches with the class's PAM are taken from the budget of its list entries, and the third
    segment's neighbourhood follows what is left (SeedPlan).  Reads with every letter at positions 21 and 22, sites
    planted at every distance up to the limit, indexes with an extra PAM: the records are the oracle's, with the tight
    cut the cost model picks (default), with floor(m / 3) in all three segments (hook 0) and with every other valid
    cut (hook 1 + k0 + 3 k1: segments 0 / 1 within k0 / k1 substitutions, the third within what is left; cuts that are
    not valid for the case fall back to the model's)."""
    hooks(seed_shared=shared, seed_tight=tight)
    rng = np.random.default_rng(seed)
    guides = [random_seq(rng, 21) + p for p in ("AG", "CG", "TG", "GG", "GA", "AA", "TC", "CT", "GT", "AG", "GG", "TA") for _ in range(6)]
    contigs = make_genome(seed, [60000, 23, 25000, 64], guides[::3], max_mm, n_plant=500, n_runs=4)
    want = oracle.search(contigs, guides, max_mm, extra_pam, mode=oracle.MODE_PREDICATE)
    got = gpu_search(ctx, contigs, guides, max_mm, extra_pam, algo="seed")
    by_pam = {g[21:]: 0 for g in guides}
    for r in want:
        by_pam[guides[r["guide"]][21:]] += 1
    if max_mm >= 5:
        assert min(by_pam[p] for p in ("AG", "CG", "TG", "GG")) > 0  # reads that mismatch at 21 have hits too
    assert len(want) > (0 if max_mm == 0 else 20)
    assert hits_as_tuples(got) == hits_as_tuples(want)


def test_search_matches_reference_flow_order(ctx, oracle):
    """SAM emission order + secondary flags (bidir_mapping.cpp:167-187) from the sorted GPU result."""
    rng = np.random.default_rng(7)
    guides = random_guides(rng, 6)
    contigs = make_genome(7, [6000, 3000], guides, 6, n_plant=150)
    flow = oracle.search(contigs, guides, 6, mode=oracle.MODE_REFERENCE_FLOW)
    got = gpu_search(ctx, contigs, guides, 6)
    order, sec = va.sam_order(got)
    got = got[order.astype(np.int64)]
    for f in ("guide", "contig", "pos"):
        assert np.array_equal(got[f], flow[f])
    assert np.array_equal(got["info"] | (sec.astype(np.uint32) << 30), flow["info"])


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("world", [2, 3, 5])
def test_genome_shards_reproduce_the_whole(ctx, oracle, world, algo):
    """Sharding by tile-aligned plane ranges with a one-word halo loses and duplicates nothing."""
    rng = np.random.default_rng(50 + world)
    guides = random_guides(rng, 8)
    contigs = make_genome(50 + world, [30000, 12000, 7000, 100, 23], guides, 6, n_plant=200, n_runs=5)
    # plant sites across every shard boundary
    packed = va.PackedGenome.from_sequences(contigs)
    seq = list(contigs[0])
    for r in range(1, world):
        b, _ = packed.shard_words(r, world)
        p = b * 32
        if p + 30 < len(seq):
            for k, shift in enumerate((-22, -11, -1, 0)):
                site = mutate(rng, guides[k % len(guides)], 2, 0, 20)
                site = site if k % 2 else revcomp(site)
                q = p + shift
                seq[q:q + 23] = list(site)
    contigs[0] = "".join(seq)
    want = oracle.search_fast(contigs, guides, 6)
    got = gpu_search(ctx, contigs, guides, 6, world=world, algo=algo)
    got = got[np.lexsort((got["pos"], got["contig"], got["info"] >> 31, got["guide"]))]
    assert hits_as_tuples(got) == hits_as_tuples(want)


@pytest.mark.parametrize("algo", ALGOS)
def test_empty_and_degenerate_inputs(ctx, oracle, algo):
    contigs = ["ACGT", "N" * 100, "ACGTTGCATGCAAGTCCTAGTGG"]
    g = "ACGTTGCATGCAAGTCCTAGTGG"
    packed = va.PackedGenome.from_sequences(contigs)
    gen0 = ctx.load_genome(packed)
    assert len(gen0.search([], 4, algorithm=algo)) == 0
    gen0.close()
    got = gpu_search(ctx, contigs, [g], 0, algo=algo)
    want = oracle.search(contigs, [g], 0)
    assert hits_as_tuples(got) == hits_as_tuples(want) and len(want) == 1
    assert len(gpu_search(ctx, ["ACGT" * 3], [g], 8, algo=algo)) == 0
    packed = va.PackedGenome.from_sequences(contigs)
    gen = ctx.load_genome(packed)
    with pytest.raises(va.VarscotError) as e:
        gen.search([g], 9)
    assert e.value.code == -22 and "between 0 and 8" in str(e.value)
    gen.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_right_edge_rule_on_gpu(ctx, oracle, algo):
    g = "ACGTTGCATGCAAGTCCTAGTGG"
    bad = g[:11] + "".join("A" if c != "A" else "C" for c in g[11:14]) + g[14:]
    for contigs in (["T" * 50 + bad], ["T" * 50 + bad + "T"], ["T" * 50 + revcomp(bad), "T" * 9 + bad],
                    ["T" * 50 + bad + "N" + "T" * 40]):
        want = oracle.search(contigs, [g], 4)
        got = gpu_search(ctx, contigs, [g], 4, algo=algo)
        assert hits_as_tuples(got) == hits_as_tuples(want)


@pytest.mark.parametrize("algo", ALGOS)
def test_dense_pam_region_and_many_guides(ctx, oracle, algo):
    """Low-complexity sequence where almost every window is a candidate on both strands, and a read
    count that is not a multiple of the kernel's unroll factor."""
    rng = np.random.default_rng(9)
    guides = random_guides(rng, 37)
    contigs = ["CCGG" * 3000 + random_seq(rng, 5000) + "G" * 4000 + "C" * 4000 + "GGCC" * 1000]
    contigs[0] = contigs[0][:20000] + guides[3] + contigs[0][20023:]
    want = oracle.search_fast(contigs, guides, 7)
    got = gpu_search(ctx, contigs, guides, 7, algo=algo)
    assert hits_as_tuples(got) == hits_as_tuples(want)


@pytest.mark.parametrize("algo,max_mm", [("scan", 8), ("seed", 8), ("seed", 5), ("seed", 2)])
def test_larger_genome_against_fast_port(ctx, oracle, algo, max_mm):
    """8 Mbp x 64 reads x 8 mismatches: tens of thousands of hits; exercises queue carry-over,
    staged-hit flushes and the dynamic chunk schedule."""
    rng = np.random.default_rng(11)
    guides = random_guides(rng, 64)
    lens = [3_000_000, 2_500_000, 1_500_000, 999_983, 17]
    contigs = make_genome(11, lens, guides, 8, n_plant=500, n_runs=40)
    want = oracle.search_fast(contigs, guides, max_mm)
    got = gpu_search(ctx, contigs, guides, max_mm, algo=algo)
    assert len(want) > (20000 if max_mm == 8 else 50)
    assert hits_as_tuples(got) == hits_as_tuples(want)


@pytest.mark.parametrize("algo", ALGOS)
def test_hit_buffer_overflow_is_retried(ctx, oracle, algo):
    """Far more hits than the random-genome estimate sizes the buffer for."""
    g = "ACGTTGCATGCAAGTCCTAGTGG"
    unit = g + "T"
    contigs = [unit * 60000]  # 60 000 perfect sites: the estimate allows ~1M + few
    guides = [g] * 40        # 2.4 M hits
    got = gpu_search(ctx, contigs, guides, 0, algo=algo)
    assert len(got) == 60000 * 40
    assert ctx.timing()["passes"] == 2
    assert np.array_equal(np.unique(got["guide"]), np.arange(40))
    assert np.all(got["pos"] % 24 == 0) and np.all((got["info"] >> 23) == 0)


# ------------------------------------------------------------------------------------ scores
def test_mit_and_features_match_oracle(ctx, oracle):
    rng = np.random.default_rng(21)
    guides = random_guides(rng, 10) + [random_seq(rng, 21) + "GA"]
    contigs = make_genome(21, [20000, 8000], guides, 8, n_plant=300)
    packed = va.PackedGenome.from_sequences(contigs)
    gen = ctx.load_genome(packed)
    hits = gen.search(guides, 8, algorithm="seed")
    rec = hits.to_numpy()
    mit, flags, feat = hits.scores(mit=True, features=True)
    assert len(rec) > 200
    n_ub = 0
    for i, (gi, s, c, p, nm, mask) in enumerate(hits_as_tuples(rec)):
        pos = [b for b in range(23) if (mask >> b) & 1] or [-1]
        want, ub = oracle.mit_score(pos)
        assert mit[i] == want  # bit-exact fp64
        assert bool(flags[i]) == ub
        n_ub += ub
        off = contigs[c][p:p + 23]
        off = revcomp(off) if s else off
        assert np.array_equal(feat[i].astype(np.uint32), oracle.feature_row(guides[gi], off))
    assert n_ub > 0  # '-' hits with PAM-side mismatches exercise the reference's out-of-bounds case
    # packed 64-byte rows expand to the same dense rows; MIT identical
    rows, m3 = hits.packed_features(mit=True)
    assert rows.shape == (len(rec), 16)
    assert np.array_equal(va.unpack_features(rows), feat) and np.array_equal(m3, mit)
    # sub-range scoring
    m2, _, f2 = hits.scores(first=5, count=7, mit=True, features=True)
    assert np.array_equal(m2, mit[5:12]) and np.array_equal(f2, feat[5:12])
    hits.close()
    gen.close()


def test_features_match_reference_golden_on_gpu(ctx, golden_dir):
    """The reference's own 6960 golden rows (featureMatrix.RData): each (on, off) pair is planted as
    a site, found by the search, and its GPU feature row compared with the stored row."""
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    on, off, feat = [str(s) for s in g["on"]], [str(s) for s in g["off"]], g["feat"]
    uniq_on = sorted(set(on))
    gidx = {s: i for i, s in enumerate(uniq_on)}
    # one contig per pair keeps sites apart: "T"*4 + off + "T"*4 (T-flanks add no GG/GA/CC/TC PAMs)
    step = 4 + 23 + 4
    contigs = ["TTTT" + o + "TTTT" for o in off]
    packed = va.PackedGenome.from_sequences(contigs)
    gen = ctx.load_genome(packed)
    hits = gen.search(uniq_on, 8)
    rec = hits.to_numpy()
    _, _, f = hits.scores(mit=False, features=True)
    found = {}
    for i, (gi, s, c, p, nm, mask) in enumerate(hits_as_tuples(rec)):
        if s == 0 and p == 4:
            found[(gi, c)] = i
    checked = 0
    for row, (a, o) in enumerate(zip(on, off)):
        nm23 = sum(x != y for x, y in zip(a, o))
        if nm23 > 8 or o[21:] not in ("GG", "GA"):
            continue
        i = found[(gidx[a], row)]
        assert np.array_equal(f[i], feat[row]), (a, o)
        assert (int(rec["info"][i]) >> 23) & 31 == int(g["nm"][row])  # Class 0: the reference mapper's own NM tag
        checked += 1
    assert checked > 6000
    hits.close()
    gen.close()
    assert step == len(contigs[0])


@pytest.mark.parametrize("algo", ALGOS)
def test_search_reports_the_reference_mappers_hits_with_its_nm(ctx, golden_dir, algo):
    """Rows R2/R3 against the one search output the reference still holds: the (guide, site, NM) triples out of
    VARSCOT's own SAM file (datasetsSampling.RData, Class 0: workflow/processDataForModel.R:257-258,284,378-394).
    Every site the reference's mapper reported must be found on the strand it is planted on with the reference's
    `NM:i` value (= popcount of the record's mask), exactly from the budget m = NM on, by both search kernels."""
    from helpers import plant_reference_sites, reference_sam_triples
    guides, rows = reference_sam_triples(golden_dir)
    assert len(rows) == 2779
    contigs, strands = plant_reference_sites(rows)
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    for m in (8, 6, 5, 3, 2):
        hits = gen.search(guides, m, algorithm=algo)
        rec = hits.to_numpy()
        got = {(g, c): (s, nm, mask) for g, s, c, p, nm, mask in hits_as_tuples(rec) if p == 4}
        n = 0
        for c, (gi, site, nm) in enumerate(rows):
            if nm <= m:
                s, nm_gpu, mask = got[(gi, c)]
                assert (s, nm_gpu) == (strands[c], nm), (guides[gi], site, nm, m)
                assert bin(mask).count("1") == nm
                # the mask is in forward-genome window coordinates (filter_output_bam.h:330-349)
                w = contigs[c][4:27]
                r = revcomp(guides[gi]) if s else guides[gi]
                assert mask == sum(1 << i for i in range(23) if w[i] != r[i])
                n += 1
            else:
                assert (gi, c) not in got, (guides[gi], site, nm, m)
        assert n == sum(1 for r in rows if r[2] <= m)
        hits.close()
    gen.close()


def test_sam_write_order_equals_the_reference_rows_on_the_guideseq_sites(ctx, oracle, golden_dir):
    """Row R4 against what is left of the ORDER of the reference's own SAM output (tests/golden/guideseq_sam_rows.tsv:
    the rows in which the reference found its 348 GUIDE-seq sites): the GPU result of a genome that carries those
    sites on the reference's contigs (UCSC hg19 order), in their real order along each chromosome and on their
    real strand, put into write order by vsc_sam_order - apart from the records either side holds back as
    best-so-far, the sites come out in exactly the order of the reference's SAM rows; and the whole write order
    (with secondary flags) equals the oracle's REFERENCE_FLOW."""
    from helpers import guideseq_mini_genome, late_records
    names, guides, cnames, contigs, planted = guideseq_mini_genome(golden_dir)
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    hits = gen.search(guides, 8)
    rec = hits.to_numpy()
    hits.close()
    gen.close()
    order, sec = va.sam_order(rec)
    written = rec[order.astype(np.int64)]
    flow = oracle.search(contigs, guides, 8, mode=oracle.MODE_REFERENCE_FLOW)
    for f in ("guide", "contig", "pos"):
        assert np.array_equal(written[f], flow[f])
    assert np.array_equal(written["info"] | (sec.astype(np.uint32) << 30), flow["info"])
    where = {(g, c, p, s): i for i, (g, s, c, p, nm, _) in enumerate(hits_as_tuples(written))}
    ours = [i for _, i in sorted((where[(g, c, p, s)], i) for i, (g, c, p, s, nm, row) in enumerate(planted))]
    theirs = [i for _, i in sorted((planted[i][5], i) for i in range(len(planted)))]
    held = set()
    for seq in (ours, theirs):
        for g in range(len(guides)):
            for s in (0, 1):
                held |= set(late_records([((planted[i][1], planted[i][2]), planted[i][4], i) for i in seq
                                          if planted[i][0] == g and planted[i][3] == s]))
    assert [i for i in ours if i not in held] == [i for i in theirs if i not in held] and len(ours) - len(held) > 300


def test_release_scratch_between_searches(ctx, oracle):
    """vsc_ctx_release_scratch gives the pooled buffers back (the next search allocates again) without touching the
    resident genome, its seed index or a live result."""
    rng = np.random.default_rng(99)
    guides = random_guides(rng, 30)
    contigs = make_genome(99, [80000, 30000], guides, 7, n_plant=400, n_runs=2)
    want = oracle.search_fast(contigs, guides, 7)
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    live = gen.search(guides, 7, algorithm="seed")
    before = gen.device_bytes
    ctx.release_scratch()
    assert gen.device_bytes == before and live.to_numpy().tobytes() == want.tobytes()
    again = gen.search(guides, 7, algorithm="seed")
    assert again.to_numpy().tobytes() == want.tobytes()
    mit, _, _ = again.scores(mit=True)
    ctx.release_scratch()
    mit2, _, _ = live.scores(mit=True)
    assert np.array_equal(mit, mit2)
    live.close()
    again.close()
    gen.close()


# ------------------------------------------------------------------------------------ multi-rank
def _rank_worker(rank, world, port, q, exchange="root"):
    import traceback
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from varscot_amd import dist as vdist
        rng = np.random.default_rng(4242)
        guides = random_guides(rng, 16)
        contigs = make_genome(4242, [300000, 120000, 70000, 23], guides, 7, n_plant=400, n_runs=6)
        packed = va.PackedGenome.from_sequences(contigs)
        c = va.Context(0)  # both ranks share the one GPU of the test box; the data path is per rank
        shard = c.load_genome(packed, rank, world)
        if exchange == "pipelined":
            parts = vdist.sharded_search_pipelined(c, shard, va.pack_guides(guides), 7, sub_batches=3)
            blobs = []
            for first, m in parts:
                rec = m.to_numpy().copy()
                rec["guide"] += first  # records count reads from the first read of their piece
                blobs.append((first, rank, rec.tobytes()))
                m.close()
            q.put(blobs)
            shard.close()
            c.close()
            dist.barrier()
            return
        if exchange == "stream":
            # configuration 5 with one process per GPU: batches of 5 reads, the forest's votes computed on the owning rank
            # (device memory, host tensors for the gloo transport), gathered to rank 0 beside the records
            import torch
            from varscot_amd.classifier import Forest
            forest = Forest()
            act = np.random.default_rng(6).uniform(0.2, 1.8, size=len(guides))
            out = []

            def score(hits, first, count):
                v, _ = forest.classify_hits(hits, act[first:first + count])
                return torch.from_numpy(v.view(np.int16).copy())

            def on_batch(m, first, count, votes):
                if m is not None:
                    out.append((first, count, m.to_numpy().tobytes(), votes.numpy().view(np.uint16).tobytes()))

            vdist.sharded_search_stream(c, shard, va.pack_guides(guides), 7, on_batch, 5, algorithm="seed", score=score)
            if rank == 0:
                q.put(out)
            shard.close()
            c.close()
            dist.barrier()
            return
        merged, local = vdist.sharded_search(c, shard, va.pack_guides(guides), 7, exchange=exchange)
        if exchange == "reads":
            q.put((rank, merged.to_numpy().tobytes()))
            merged.close()
        elif rank == 0:
            q.put(merged.to_numpy().tobytes())
            merged.close()
        local.close()
        shard.close()
        c.close()
        dist.barrier()
    except Exception:
        q.put("rank %d failed: %s" % (rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_sharded_search_over_gloo(oracle):
    """Two ranks, each scanning its genome shard on the GPU, one gather, merge on rank 0."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    blob = q.get(timeout=120)
    assert isinstance(blob, bytes), blob
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = np.frombuffer(blob, dtype=va.HIT_DTYPE)
    rng = np.random.default_rng(4242)
    guides = random_guides(rng, 16)
    contigs = make_genome(4242, [300000, 120000, 70000, 23], guides, 7, n_plant=400, n_runs=6)
    want = oracle.search_fast(contigs, guides, 7)
    assert len(want) > 300
    assert got.tobytes() == want.tobytes()


def test_sharded_search_stream_with_votes_over_gloo(ctx, oracle):
    """varscot_amd.dist.sharded_search_stream on two ranks (one GPU, gloo): batches of 5 reads searched on each rank's genome
    shard, the forest's votes computed there, records + votes gathered to rank 0 while the next batch is searched, merged by
    vsc_hits_merge_packed_votes.  The batches are the oracle's records of their reads; the votes are, record for record, what one
    context computes on the whole result."""
    import socket
    import torch.multiprocessing as mp
    from varscot_amd.classifier import Forest
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_rank_worker, args=(r, 2, port, q, "stream")) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    assert isinstance(out, list), out
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(4242)
    guides = random_guides(rng, 16)
    contigs = make_genome(4242, [300000, 120000, 70000, 23], guides, 7, n_plant=400, n_runs=6)
    want = oracle.search_fast(contigs, guides, 7)
    assert [(f, n) for f, n, _, _ in out] == [(0, 5), (5, 5), (10, 5), (15, 1)]
    got = np.concatenate([np.frombuffer(b, dtype=va.HIT_DTYPE) for _, _, b, _ in out])
    assert len(want) > 300 and got.tobytes() == want.tobytes()
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    h = gen.search(guides, 7, algorithm="seed")
    want_votes, _ = Forest().classify_hits(h, np.random.default_rng(6).uniform(0.2, 1.8, size=len(guides)))
    h.close()
    gen.close()
    assert np.array_equal(np.concatenate([np.frombuffer(v, dtype=np.uint16) for _, _, _, v in out]), want_votes)


def test_sharded_search_exchange_by_reads_over_gloo(oracle):
    """Two ranks, genome shards searched on the GPU, every rank collects and merges its read range."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_rank_worker, args=(r, 2, port, q, "reads")) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    assert all(isinstance(g, tuple) for g in got), got
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    blob = b"".join(b for _, b in sorted(got))
    rng = np.random.default_rng(4242)
    guides = random_guides(rng, 16)
    contigs = make_genome(4242, [300000, 120000, 70000, 23], guides, 7, n_plant=400, n_runs=6)
    want = oracle.search_fast(contigs, guides, 7)
    assert blob == want.tobytes()


def test_sharded_search_pipelined_over_gloo(oracle):
    """Two ranks, the reads in three pieces: the exchange of one piece is in flight during the search of the
    next; pieces in read order, within a piece the ranks' shares in rank order = the global result."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_rank_worker, args=(r, 2, port, q, "pipelined")) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    assert all(isinstance(g, list) for g in got), got
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    blob = b"".join(b for _, _, b in sorted(got[0] + got[1]))
    rng = np.random.default_rng(4242)
    guides = random_guides(rng, 16)
    contigs = make_genome(4242, [300000, 120000, 70000, 23], guides, 7, n_plant=400, n_runs=6)
    want = oracle.search_fast(contigs, guides, 7)
    assert blob == want.tobytes()


def test_auto_uses_an_existing_index_and_rebuilds_for_another_pam(ctx, oracle):
    rng = np.random.default_rng(77)
    guides = random_guides(rng, 5)
    contigs = make_genome(77, [20000, 5000], guides, 5, n_plant=100)
    packed = va.PackedGenome.from_sequences(contigs)
    gen = ctx.load_genome(packed)
    h = gen.search(guides, 5)                       # small search, no index: streaming scan
    assert ctx.timing()["algorithm"] == 1
    a = h.to_numpy()
    h.close()
    gen.build_index()
    h = gen.search(guides, 5)                       # index present: used
    assert ctx.timing()["algorithm"] == 2
    assert h.to_numpy().tobytes() == a.tobytes()
    h.close()
    h = gen.search(guides, 5, extra_pam="AG", algorithm="seed")   # other PAM set: index rebuilt
    want = oracle.search(contigs, guides, 5, "AG")
    assert hits_as_tuples(h.to_numpy()) == hits_as_tuples(want)
    h.close()
    h = gen.search(guides, 5, algorithm="seed")     # and back
    assert h.to_numpy().tobytes() == a.tobytes()
    h.close()
    gen.close()


def test_score_pairs_on_reference_fixtures(ctx, oracle, golden_dir):
    """vsc_score_pairs: the reference's 6960 golden feature rows directly, and the 4443 SITE-seq pairs
    (non-GG PAMs, up to 14 mismatches) against the oracle, MIT scores bit-exact."""
    g = np.load(os.path.join(golden_dir, "features_golden.npz"))
    on, off = [str(s) for s in g["on"]], [str(s) for s in g["off"]]
    masks = [sum(1 << i for i in range(23) if a[i] != b[i]) for a, b in zip(on, off)]
    mit, flags, feat = ctx.score_pairs(on, off, masks, mit=True, features=True)
    assert np.array_equal(feat, g["feat"])
    for i in range(0, len(on), 37):
        pos = [b for b in range(23) if (masks[i] >> b) & 1] or [-1]
        want, ub = oracle.mit_score(pos)
        assert mit[i] == want and bool(flags[i]) == ub
    on2, off2 = [], []
    with open(os.path.join(golden_dir, "siteseq_pairs.tsv")) as f:
        next(f)
        for line in f:
            a, b, nm, strand = line.split()
            on2.append(a)
            off2.append(b)
    masks2 = [sum(1 << i for i in range(23) if a[i] != b[i]) for a, b in zip(on2, off2)]
    mit2, flags2, feat2 = ctx.score_pairs(on2, off2, masks2, mit=True, features=True)
    assert len(on2) == 4443
    for i in range(len(on2)):
        assert np.array_equal(feat2[i].astype(np.uint32), oracle.feature_row(on2[i], off2[i]))
        pos = [b for b in range(23) if (masks2[i] >> b) & 1] or [-1]
        want, ub = oracle.mit_score(pos)
        assert mit2[i] == want and bool(flags2[i]) == ub


@pytest.mark.parametrize("algo", ALGOS)
def test_many_tiny_contigs_like_a_variant_genome(ctx, oracle, algo):
    """70 000 contigs of 45-60 bases (more than the reference's uint16_t key can tell apart): window
    genome shape, contig resolution by binary search, hits at contig ends."""
    rng = np.random.default_rng(31)
    guides = random_guides(rng, 6)
    contigs = []
    for i in range(70000):
        n = int(rng.integers(45, 61))
        s = random_seq(rng, n)
        if i % 97 == 0:
            g = guides[i % len(guides)]
            site = mutate(rng, g, int(rng.integers(0, 4)), 0, 20)
            site = site if i % 2 else revcomp(site)
            p = [0, n - 23, (n - 23) // 2][i % 3]
            s = s[:p] + site + s[p + 23:]
        contigs.append(s)
    want = oracle.search_fast(contigs, guides, 5)
    got = gpu_search(ctx, contigs, guides, 5, algo=algo)
    assert len(want) > 300 and int(want["contig"].max()) > 65535
    assert hits_as_tuples(got) == hits_as_tuples(want)


def test_many_and_duplicate_reads_seeded(ctx, oracle):
    """4 000 reads (long per-bucket read lists, lists padded to the unroll factor) with exact
    duplicates among them."""
    rng = np.random.default_rng(32)
    guides = random_guides(rng, 3990)
    guides += guides[:10]
    contigs = make_genome(32, [400000, 150000], guides[:50], 6, n_plant=300, n_runs=4)
    want = oracle.search_fast(contigs, guides, 6)
    got = gpu_search(ctx, contigs, guides, 6, algo="seed")
    assert len(want) > 2000
    assert hits_as_tuples(got) == hits_as_tuples(want)


@pytest.mark.parametrize("world", [2, 8])
def test_shard_merge_kernels(ctx, oracle, world):
    """vsc_hits_merge on host-resident shard results (the gloo rehearsal path) with empty shards,
    reads without hits and many (guide, strand) keys."""
    from varscot_amd.api import merge_shard_records
    rng = np.random.default_rng(60 + world)
    guides = random_guides(rng, 40)
    contigs = make_genome(60 + world, [60000, 20000, 33], guides[:25], 6, n_plant=300, n_runs=3)
    packed = va.PackedGenome.from_sequences(contigs)
    parts = []
    for rank in range(world):
        b, e = packed.shard_words(rank, world)
        if e <= b:
            parts.append(np.zeros(0, dtype=va.HIT_DTYPE))
            continue
        gen = ctx.load_genome(packed, rank, world)
        h = gen.search(guides, 6)
        parts.append(h.to_numpy())
        h.close()
        gen.close()
    parts.insert(1, np.zeros(0, dtype=va.HIT_DTYPE))  # an empty shard in the middle
    cat = np.ascontiguousarray(np.concatenate(parts))
    merged = merge_shard_records(ctx, cat.ctypes.data, False, [len(p) for p in parts], len(guides))
    got = merged.to_numpy()
    merged.close()
    want = oracle.search_fast(contigs, guides, 6)
    assert len(want) > 200
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("world", [2, 8])
def test_exchange_records_pack_and_merge(ctx, oracle, world):
    """vsc_hits_pack_exchange / vsc_hits_merge_packed (what both multi-GPU drivers send and rebuild): every shard's
    result packed to 8-byte records + per-key counts (device and host destinations agree with the numpy restatement),
    then merged - whole key range on one receiver (exchange="root"), and per read range (exchange="reads") - with an
    empty shard, reads without hits and many keys; > 1 contig so that the contig is recovered from the position."""
    import torch
    from helpers import xpack
    from varscot_amd.api import merge_packed_records
    rng = np.random.default_rng(160 + world)
    guides = random_guides(rng, 40)
    contigs = make_genome(160 + world, [60000, 20000, 33, 5000], guides[:25], 6, n_plant=300, n_runs=3)
    packed = va.PackedGenome.from_sequences(contigs)
    recs, counts, any_shard = [], [], None
    for rank in range(world):
        b, e = packed.shard_words(rank, world)
        if e <= b:
            recs.append(np.zeros(0, dtype=np.uint64))
            counts.append(np.zeros(2 * len(guides), dtype=np.uint32))
            continue
        gen = ctx.load_genome(packed, rank, world)
        h = gen.search(guides, 6)
        host = np.zeros(len(h), dtype=np.uint64)
        c_host = h.pack_exchange(host.ctypes.data if len(h) else 0, False)
        dev = torch.empty(max(len(h), 1) * 8, dtype=torch.uint8, device="cuda")
        c_dev = h.pack_exchange(dev.data_ptr(), True)
        want_rec, want_counts = xpack(h.to_numpy(), packed.contigs["offset"], len(guides))
        assert np.array_equal(c_host, want_counts) and np.array_equal(c_dev, want_counts)
        assert np.array_equal(host, want_rec) and np.array_equal(dev.cpu().numpy().view(np.uint64)[:len(h)], want_rec)
        recs.append(host)
        counts.append(c_host)
        h.close()
        if any_shard is None:
            any_shard = gen
        else:
            gen.close()
    recs.insert(1, np.zeros(0, dtype=np.uint64))  # an empty shard in the middle
    counts.insert(1, np.zeros(2 * len(guides), dtype=np.uint32))
    want = oracle.search_fast(contigs, guides, 6)
    assert len(want) > 200 and len(set(want["contig"])) > 1
    cat = np.ascontiguousarray(np.concatenate(recs))
    all_counts = np.stack(counts)
    merged = merge_packed_records(ctx, any_shard, cat.ctypes.data, False, all_counts)
    assert merged.to_numpy().tobytes() == want.tobytes()
    merged.close()
    # a receiver that collects only the reads [11, 29): the slices of every shard for those keys
    k0, k1 = 22, 58
    prefix = np.concatenate([np.zeros((len(recs), 1), dtype=np.int64), np.cumsum(all_counts.astype(np.int64), axis=1)], axis=1)
    part = np.ascontiguousarray(np.concatenate([r[prefix[s, k0]:prefix[s, k1]] for s, r in enumerate(recs)]))
    dev = torch.from_numpy(part.view(np.uint8).copy()).cuda()
    merged = merge_packed_records(ctx, any_shard, dev.data_ptr(), True, all_counts[:, k0:k1], first_key=k0)
    sel = want[(want["guide"] >= 11) & (want["guide"] < 29)]
    assert merged.to_numpy().tobytes() == sel.tobytes()
    merged.close()
    # records a receiver must refuse (they come from a peer or from the caller): a position of all ones (the walk over the
    # contig table ends at the last contig instead of never), one inside the separator between two contigs, one in the
    # padding behind the genome, and two records of one segment swapped (descending)
    from varscot_amd._lib import VarscotError
    first = int(np.flatnonzero(all_counts.sum(axis=0) >= 2)[0])  # a key some shard holds two records of
    shard = int(np.argmax(all_counts[:, first] >= 2))
    at = int(sum(len(r) for r in recs[:shard]) + all_counts[shard, :first].sum())
    sep = int(packed.contigs["offset"][0]) + int(packed.contigs["length"][0])
    end = int(packed.contigs["offset"][-1]) + int(packed.contigs["length"][-1])
    for bad in (0xFFFFFFFF, sep - 5, end + 40, "swap"):
        broken = cat.copy()
        if bad == "swap":
            broken[at], broken[at + 1] = cat[at + 1], cat[at]
        else:
            broken[at] = np.uint64(bad) << np.uint64(23) | (cat[at] & np.uint64(0x7FFFFF))
        with pytest.raises(VarscotError) as err:
            merge_packed_records(ctx, any_shard, broken.ctypes.data, False, all_counts)
        assert err.value.code == -22 and "exchange records" in str(err.value)
    merged = merge_packed_records(ctx, any_shard, cat.ctypes.data, False, all_counts)  # the context is fine afterwards
    assert merged.to_numpy().tobytes() == want.tobytes()
    merged.close()
    any_shard.close()


@pytest.mark.parametrize("cap,max_bits", [(None, None), (4096, 11), (256, 3), (16, 1)])
@pytest.mark.parametrize("algo", ALGOS)
def test_sort_levels_on_dense_position_runs(ctx, oracle, algo, cap, max_bits, hooks):
    """The bin sort (vsc_sort.hip): homopolymer runs give one read a hit at EVERY position - dense sub-bins that
    the finalize kernel has to rank, bins far larger than their neighbours.  The hooks sort_cap / sort_max_bits
    shrink the LDS capacity and the bits per partition level so that the partition levels and the oversize
    path (bins handed to a further level, up to dozens of levels) run on a few thousand records."""
    if cap:
        hooks(sort_cap=cap, sort_max_bits=max_bits)
    rng = np.random.default_rng(77)
    guides = ["G" * 23, "C" * 23, "G" * 11 + "A" + "G" * 11] + random_guides(rng, 5)
    contigs = ["G" * 1500 + random_seq(rng, 300) + "C" * 900, random_seq(rng, 2000), "G" * 700]
    want = oracle.search(contigs, guides, 3, None, mode=oracle.MODE_PREDICATE)
    got = gpu_search(ctx, contigs, guides, 3, None, algo=algo)
    assert len(want) > 5000
    assert hits_as_tuples(got) == hits_as_tuples(want)
    levels = ctx.timing()["sort_levels"]
    assert (levels == 0) if cap is None else (levels >= 1)


@pytest.mark.parametrize("algo", ALGOS)
def test_baseline_config_c1(ctx, oracle, algo):
    """BASELINE.json configs[0]: 10 guides, 1 Mbp synthetic FASTA, <= 4 mismatches - the bench's own
    synthetic generator, checked against the oracle's bit-parallel port."""
    from varscot_amd import synth
    packed = synth.synthetic_genome(1_000_000)
    ids, guides = synth.synthetic_guides(10)
    planted = synth.plant_sites(packed, guides, 60, 4)
    contigs = [packed.contig_sequence(c) for c in range(len(packed.contigs))]
    assert len(planted) > 20
    want = oracle.search_fast(contigs, guides, 4)
    got = gpu_search(ctx, contigs, guides, 4, algo=algo)
    assert hits_as_tuples(got) == hits_as_tuples(want)


def test_sort_without_the_histogram_pass_and_its_fallback(ctx, oracle, hooks):
    """The seed search's first sort level in SLOT MODE (no histogram pass: fixed bin slots, room reserved by the bin
    cursors alone), on a result with thousands of hits per read: (a) slots large enough - one level, no fallback;
    (b) slots smaller than the fullest bins - the partition raises its overflow flag, nothing is finalized, the
    level runs again with the histogram (sort_fallbacks = 1), and the genome remembers: the next search goes the
    exact way at once; (c) slot mode switched off.  Same records every way."""
    rng = np.random.default_rng(4711)
    guides = random_guides(rng, 70)
    pieces = []
    for _ in range(9000):
        g = guides[int(rng.integers(0, len(guides)))]
        pieces.append(mutate(rng, g, int(rng.integers(0, 7)), 0, 20) + random_seq(rng, int(rng.integers(0, 3))))
    contigs = make_genome(4711, [60000, 20000], guides, 8, n_plant=200) + ["".join(pieces)]
    want = oracle.search_fast(contigs, guides, 8)
    assert len(want) > 8000
    packed = va.PackedGenome.from_sequences(contigs)

    def run(gen):
        h = gen.search(guides, 8, algorithm="seed")
        got = h.to_numpy().copy()
        h.close()
        t = ctx.timing()
        assert got.tobytes() == want.tobytes()
        return t["sort_levels"], t["sort_fallbacks"]

    hooks(sort_cap=512, sort_optimistic=1, sort_slot_cap=4096)           # (a)
    gen = ctx.load_genome(packed)
    assert run(gen) == (1, 0)
    gen.close()
    hooks(sort_cap=512, sort_slot_cap=40)                                # (b): default policy, slots too small
    gen = ctx.load_genome(packed)
    levels, fallbacks = run(gen)
    assert fallbacks == 1 and levels >= 1
    assert run(gen)[1] == 0                                              # remembered: exact at once
    gen.close()
    hooks(sort_cap=512, sort_optimistic=0)                               # (c)
    gen = ctx.load_genome(packed)
    assert run(gen)[1] == 0
    gen.close()
    hooks(sort_cap=64, sort_max_bits=2, sort_optimistic=1, sort_slot_cap=60000)  # four big bins per region: on to level 2 from their slots
    gen = ctx.load_genome(packed)
    levels, fallbacks = run(gen)
    assert levels >= 2 and fallbacks == 0, (levels, fallbacks)
    gen.close()


@pytest.mark.parametrize("cap", [None, 512, "group"])
def test_hit_buffer_growth_and_sort_partition(ctx, oracle, cap, hooks):
    """Thousands of hits per read on a small genome: the hit buffer sized from the uniform-genome model has to
    grow (second search launch), the second search reuses the grown buffers; with a reduced bin capacity the
    region goes through the partition level.  A read count that is not a multiple of four."""
    if cap == "group":  # the chunk-sharing kernel with workgroup-shared output blocks: regions that fill up, re-run
        hooks(seed_shared=1, seed_group_out=1)
    elif cap:
        hooks(sort_cap=cap)
    rng = np.random.default_rng(909)
    guides = random_guides(rng, 37)
    pieces = []
    for _ in range(6000):  # a contig of mutated read copies: thousands of hits for every read
        g = guides[int(rng.integers(0, len(guides)))]
        pieces.append(mutate(rng, g, int(rng.integers(0, 7)), 0, 20) + random_seq(rng, int(rng.integers(0, 3))))
    contigs = make_genome(909, [50000, 20000], guides, 8, n_plant=200) + ["".join(pieces)]
    want = oracle.search_fast(contigs, guides, 8)
    assert len(want) > 5000
    for _ in range(2):  # the second search reuses the grown buffers
        got = gpu_search(ctx, contigs, guides, 8, algo="seed")
        assert hits_as_tuples(got) == hits_as_tuples(want)


@pytest.mark.parametrize("algo", ALGOS)
def test_output_regions_of_128_reads(ctx, oracle, algo):
    """More than 128 reads: several output regions (region = read index >> 7; the scan reaches them through
    level 0 of the sort); reads at the region boundaries 127 / 128 / 255 / 256 / 1023 / 1024 get planted sites."""
    rng = np.random.default_rng(5150)
    guides = random_guides(rng, 1100)
    contigs = make_genome(5150, [200000, 80000, 30000], guides, 7, n_plant=600, n_runs=4)
    seq = contigs[0]
    marked = [0, 127, 128, 255, 256, 1023, 1024, 1099]
    for k, gi in enumerate(marked):
        at = 1000 + 400 * k
        seq = seq[:at] + mutate(rng, guides[gi], 3, 0, 20) + seq[at + 23:]
    contigs[0] = seq
    want = oracle.search_fast(contigs, guides, 7)
    got = gpu_search(ctx, contigs, guides, 7, algo=algo)
    assert {int(g) for g in want["guide"]} >= set(marked)
    assert hits_as_tuples(got) == hits_as_tuples(want)


def test_more_reads_than_one_pass_takes(ctx, oracle):
    """16 384 reads fill the 128 output regions of a pass: vsc_search runs larger read sets pass by pass and
    the passes' results follow each other (read index = major sort key).  Reads on either side of the pass
    boundary get planted sites; the second pass has an odd read count."""
    rng = np.random.default_rng(4242)
    guides = random_guides(rng, 16384 + 301)
    contigs = make_genome(4242, [120000, 50000], guides[:40], 5, n_plant=200, n_runs=3)
    seq = contigs[1]
    marked = [16383, 16384, 16385, 16384 + 300]
    for k, gi in enumerate(marked):
        at = 2000 + 300 * k
        seq = seq[:at] + mutate(rng, guides[gi], 2, 0, 20) + seq[at + 23:]
    contigs[1] = seq
    want = oracle.search_fast(contigs, guides, 5)
    got = gpu_search(ctx, contigs, guides, 5, algo="seed")
    assert ctx.timing()["read_passes"] == 2
    assert {int(g) for g in want["guide"]} >= set(marked)
    assert hits_as_tuples(got) == hits_as_tuples(want)


def test_streamed_search_equals_one_search(ctx, oracle):
    """vsc_search_stream (the c5 shape: batches of reads, each batch's result handed to a callback and freed):
    the concatenated batches are the records of one search, read indices included, and per-batch scoring
    inside the callback sees the right reads."""
    rng = np.random.default_rng(99)
    guides = random_guides(rng, 700)
    contigs = make_genome(99, [150000, 40000, 23], guides[:60], 6, n_plant=400, n_runs=3)
    packed = va.PackedGenome.from_sequences(contigs)
    gen = ctx.load_genome(packed)
    whole = gen.search(guides, 6, algorithm="seed")
    want = whole.to_numpy()
    want_mit, _, _ = whole.scores(mit=True)
    want_rows, _ = whole.packed_features()
    whole.close()
    parts, mits, rows, spans = [], [], [], []

    def on_batch(h, first, count):
        spans.append((first, count))
        parts.append(h.to_numpy())
        mits.append(h.scores(mit=True)[0])
        rows.append(h.packed_features()[0])

    gen.search_streamed(guides, 6, on_batch, batch=256, algorithm="seed")
    gen.close()
    assert spans == [(0, 256), (256, 256), (512, 188)]
    got = np.concatenate(parts)
    assert got.tobytes() == want.tobytes()
    assert np.array_equal(np.concatenate(mits), want_mit)
    assert np.array_equal(np.concatenate(rows), want_rows)
    ref = oracle.search_fast(contigs, guides, 6)
    assert hits_as_tuples(got) == hits_as_tuples(ref)


def test_scoring_in_several_passes(ctx, oracle, hooks):
    """vsc_score_hits / vsc_score_hits_packed score a result that does not fit their scratch buffers in several
    passes (a c3-sized result is 104 GB of packed rows): forced here with passes of 50 rows; scores, flags,
    dense and packed rows equal the one-pass ones and the oracle's."""
    rng = np.random.default_rng(1234)
    guides = random_guides(rng, 30)
    contigs = make_genome(1234, [90000, 30000], guides, 6, n_plant=400, n_runs=3)
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    h = gen.search(guides, 6)
    rec = h.to_numpy()
    assert len(rec) > 300
    one = h.scores(mit=True, features=True)
    one_rows, one_mit = h.packed_features(mit=True)
    hooks(score_chunk=50)
    many = h.scores(mit=True, features=True)
    many_rows, many_mit = h.packed_features(mit=True)
    sub = h.scores(first=100, count=170, mit=True, features=True)
    # the packed rows again in slice-major processing order (what large results on large genomes get): slices of
    # 2^12 positions here, alone and together with the passes of 50 rows
    hooks(score_chunk=50, score_slices=1, score_slice_shift=12)
    sliced_rows, sliced_mit = h.packed_features(mit=True)
    hooks(score_slices=1, score_slice_shift=12)
    sliced_rows2, sliced_mit2 = h.packed_features(first=37, count=len(rec) - 60, mit=True)
    assert np.array_equal(sliced_rows, one_rows) and np.array_equal(sliced_mit, one_mit)
    assert np.array_equal(sliced_rows2, one_rows[37:len(rec) - 23]) and np.array_equal(sliced_mit2, one_mit[37:len(rec) - 23])
    h.close()
    gen.close()
    for a, b in zip(one, many):
        assert np.array_equal(a, b)
    assert np.array_equal(one_rows, many_rows) and np.array_equal(one_mit, many_mit)
    assert np.array_equal(sub[0], one[0][100:270]) and np.array_equal(sub[2], one[2][100:270])
    assert np.array_equal(va.unpack_features(many_rows), one[2])
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for i in range(0, len(rec), 7):
        off = contigs[rec["contig"][i]][rec["pos"][i]:rec["pos"][i] + 23]
        if rec["info"][i] >> 31:
            off = "".join(comp[c] for c in reversed(off))
        assert np.array_equal(one[2][i].astype(np.uint32), oracle.feature_row(guides[rec["guide"][i]], off))


@pytest.mark.parametrize("devices,algo", [([0, 0], "seed"), ([0, 0, 0], "scan"), ([0, 0, 0, 0, 0, 0, 0, 0], "seed")])
def test_multi_device_search_behind_the_abi(oracle, devices, algo):
    """vsc_multi_*: N contexts in one process (here all on device 0), genome shards + halo word, concurrent
    per-shard searches from host threads, one exchange to the first context and the segment merge - the records
    of one context and of the oracle, including windows that straddle shard boundaries, reads without hits and
    shards without hits."""
    rng = np.random.default_rng(700 + len(devices))
    guides = random_guides(rng, 150)
    contigs = make_genome(700 + len(devices), [70000, 30000, 33, 9000], guides[:40], 6, n_plant=500, n_runs=4)
    packed = va.PackedGenome.from_sequences(contigs)
    # a site across every shard boundary: tile-aligned word ranges
    for r in range(1, len(devices)):
        b, _ = packed.shard_words(r, len(devices))
        at = b * 32 - 11
        c = int(np.searchsorted(packed.contigs["offset"].astype(np.int64), at, side="right") - 1)
        local = at - int(packed.contigs[c]["offset"])
        if 0 <= local <= int(packed.contigs[c]["length"]) - 23:
            s = contigs[c]
            contigs[c] = s[:local] + mutate(rng, guides[r], 2, 0, 20) + s[local + 23:]
    packed = va.PackedGenome.from_sequences(contigs)
    want = oracle.search_fast(contigs, guides, 6)
    m = va.MultiContext(devices)
    try:
        assert not m.uses_rccl  # repeated device ids: device copies carry the exchange
        g = m.load_genome(packed)
        if algo == "seed":
            g.build_index()
        h = g.search(guides, 6, algorithm=algo)
        got = h.to_numpy()
        t = m.timing()
        assert t["n_devices"] == len(devices) and t["hits"] == len(want)
        h.close()
        h2 = g.search(guides[:10], 3, algorithm=algo)  # a second search on the same objects
        got2 = h2.to_numpy()
        h2.close()
        g.close()
    finally:
        m.close()
    assert len(want) > 300
    assert got.tobytes() == want.tobytes()
    assert got2.tobytes() == oracle.search_fast(contigs, guides[:10], 3).tobytes()


def test_multi_device_search_on_a_genome_smaller_than_the_device_list(oracle):
    """Five contexts, a genome of one tile: four shards own nothing - the first among them, so the merge on the
    first device works from the contig table alone (no shard of the genome lives there) - and a second genome of
    three tiles on the same contexts.  Reads without any hit, a genome without any hit."""
    rng = np.random.default_rng(55)
    guides = random_guides(rng, 9)
    m = va.MultiContext([0] * 5)
    try:
        for lens, plants in (([1500, 300], 25), ([2000, 2100, 1900], 40), ([1200], 0)):
            contigs = make_genome(55 + len(lens), lens, guides[:5] if plants else [], 5, n_plant=plants, n_runs=1)
            packed = va.PackedGenome.from_sequences(contigs)
            assert packed.shard_words(0, 5)[1] == 0 or len(lens) == 3  # shard 0 owns nothing of the small genomes
            g = m.load_genome(packed)
            for algo in ALGOS:
                want = oracle.search(contigs, guides, 5, mode=oracle.MODE_PREDICATE)
                h = g.search(guides, 5, algorithm=algo)
                assert h.to_numpy().tobytes() == want.tobytes() and (len(want) > 10 or plants == 0)
                h.close()
            g.close()
    finally:
        m.close()


@pytest.mark.parametrize("hook", [{}, {"sort_cap": 256, "sort_max_bits": 3}, {"sort_cap": 512, "sort_slot_cap": 40}, {"sort_cap": 64, "sort_max_bits": 2, "sort_optimistic": 1, "sort_slot_cap": 60000},
                                  {"seed_shared": 1, "seed_group_out": 1}, {"seed_shared": 0}])
def test_streamed_search_writes_the_feature_rows_on_the_way(ctx, oracle, hooks, hook):
    """vsc_search_stream_rows: the search keeps every site's bases beside its hit record (a 4-byte side word that travels through
    the partition levels of the sort), the record assembly recomputes the mismatch mask from them and writes the hit's packed
    feature row in the same pass.  Records = the oracle's; rows = vsc_score_hits_packed's on the same hits (which gathers the
    windows from the planes), row for row - with regions that fit the last stage as they are, several partition levels, the slot
    partition with and without its fallback, bins that go on to a second level from their slots, both chunk-sharing modes; thousands
    of hits per read, both strands, N runs, contig ends, reads without hits."""
    import torch
    from varscot_amd.dist import DeviceAlias
    if hook:
        hooks(**hook)
    rng = np.random.default_rng(31337)
    guides = random_guides(rng, 90)
    pieces = []
    for _ in range(5000):  # a contig of mutated read copies: thousands of hits for some reads
        g = guides[int(rng.integers(0, 70))]
        pieces.append(mutate(rng, g, int(rng.integers(0, 7)), 0, 20) + random_seq(rng, int(rng.integers(0, 3))))
    contigs = make_genome(31337, [60000, 20000, 33, 9000], guides[:70], 8, n_plant=300, n_runs=3) + ["".join(pieces)]
    want = oracle.search_fast(contigs, guides, 8)
    assert len(want) > 5000
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    seen, recs, rows = [], [], []

    def on_batch(h, first, count, rows_dev):
        seen.append((first, count))
        a = h.to_numpy().copy()
        recs.append(a)
        if len(a):
            got = torch.as_tensor(DeviceAlias(rows_dev, 64 * len(a)), device="cuda:0").view(torch.int32).view(-1, 16).cpu().numpy().view(np.uint32).copy()
            ref, _ = h.packed_features(to_host=True)  # the gathering kernel, on the same records (afterwards: it reuses the scratch)
            assert np.array_equal(got, ref), np.flatnonzero((got != ref).any(axis=1))[:5]
            rows.append(got)
        else:
            assert not rows_dev

    gen.search_streamed_rows(guides, 8, on_batch, batch=32, algorithm="seed")
    assert seen == [(0, 32), (32, 32), (64, 26)]
    assert np.concatenate(recs).tobytes() == want.tobytes()
    # the rows against the oracle's feature rows, on a sample
    allrec, allrows = np.concatenate(recs), va.unpack_features(np.concatenate(rows))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for i in range(0, len(allrec), max(1, len(allrec) // 200)):
        r = allrec[i]
        off = contigs[r["contig"]][r["pos"]:r["pos"] + 23]
        if r["info"] >> 31:
            off = "".join(comp[c] for c in reversed(off))
        assert np.array_equal(allrows[i].astype(np.uint32), oracle.feature_row(guides[r["guide"]], off))
    # the streaming scan (no seed index involved) takes the two-step route behind the same entry point
    small = []
    gen.search_streamed_rows(guides[:8], 4, lambda h, f, c, p: small.append((len(h), bool(p))), batch=8, algorithm="scan")
    assert small and small[0][0] == len(oracle.search_fast(contigs, guides[:8], 4)) and small[0][1] == (small[0][0] > 0)
    gen.close()


@pytest.mark.parametrize("devices,batch,score", [([0] * 8, 16, None), ([0] * 3, 7, "rows"), ([0, 0], 25, "votes"), ([0] * 5, 64, "votes")])
def test_multi_device_streamed_search_scored_on_the_owning_shard(ctx, oracle, devices, batch, score):
    """vsc_multi_search_stream (BASELINE configuration 5 behind the C ABI): the reads go through all shards batch by batch, every
    shard scores its own hits before the exchange, a batch is exchanged and merged on the first device while the shards search
    the next one.  Every merged batch = the oracle's records of its reads (global read indices), the batches cover the read set
    in order; with the forest on the path the votes that travelled with the records equal, record for record, what ONE context
    computes for the same hits (vsc_score_classify_hits on the whole result) - shard boundaries and the merge keep votes and
    records together.  Repeated device ids: 8 / 3 / 2 / 5 contexts on the one GPU of this box."""
    import torch
    from varscot_amd.classifier import Forest
    rng = np.random.default_rng(4242 + len(devices))
    guides = random_guides(rng, 50)
    contigs = make_genome(4242 + len(devices), [90000, 30000, 700, 45000], guides[:40], 6, n_plant=500, n_runs=3)
    packed = va.PackedGenome.from_sequences(contigs)
    want = oracle.search_fast(contigs, guides, 6)
    assert len(want) > 300
    forest = Forest() if score == "votes" else None
    act = np.random.default_rng(6).uniform(0.2, 1.8, size=len(guides))
    want_votes = None
    if score == "votes":  # one context, whole result
        gen = ctx.load_genome(packed)
        h = gen.search(guides, 6, algorithm="seed")
        assert h.to_numpy().tobytes() == want.tobytes()
        want_votes, _ = forest.classify_hits(h, act)
        h.close()
        gen.close()
        assert 0 < (want_votes > 500).mean() < 1
    seen, recs, votes = [], [], []

    def on_batch(h, first, count, votes_dev):
        seen.append((first, count))
        a = h.to_numpy()
        recs.append(a)
        assert len(a) == 0 or (int(a["guide"].min()) >= first and int(a["guide"].max()) < first + count)
        if score == "votes" and len(a):
            assert votes_dev
            from varscot_amd.dist import DeviceAlias
            t = torch.as_tensor(DeviceAlias(votes_dev, 2 * len(a)), device="cuda:0").view(torch.int16)
            votes.append(t.cpu().numpy().view(np.uint16).copy())
        else:
            assert not votes_dev or score == "votes"

    m = va.MultiContext(devices)
    try:
        g = m.load_genome(packed)
        g.build_index()
        g.search_streamed(guides, 6, on_batch, batch=batch, algorithm="seed", score=score, forest=forest, guide_activity=act)
        t = m.timing()
        # and the plain search on the same objects afterwards (one batch through the same engine)
        h = g.search(guides, 6, algorithm="seed")
        assert h.to_numpy().tobytes() == want.tobytes()
        h.close()
        g.close()
    finally:
        m.close()
    assert seen == [(b, min(batch, len(guides) - b)) for b in range(0, len(guides), batch)]
    assert np.concatenate(recs).tobytes() == want.tobytes()
    assert t["batches"] == len(seen) and t["hits"] == len(want) and t["n_devices"] == len(devices)
    assert (t["score_ms_max"] > 0) == (score == "votes")  # (the rows are written by the record assembly: no scoring kernel of their own)
    if score == "votes":
        assert np.array_equal(np.concatenate(votes), want_votes)


def test_multi_device_stream_stops_when_the_callback_fails(oracle):
    """A callback that raises in the second batch: the stream stops, the shard threads are joined, the error surfaces, and the
    device set is usable afterwards."""
    rng = np.random.default_rng(99)
    guides = random_guides(rng, 30)
    contigs = make_genome(99, [60000, 20000], guides, 5, n_plant=300, n_runs=2)
    want = oracle.search_fast(contigs, guides, 5)
    m = va.MultiContext([0] * 3)
    try:
        g = m.load_genome(va.PackedGenome.from_sequences(contigs))
        calls = []

        def on_batch(h, first, count, votes_dev):
            calls.append(first)
            if len(calls) == 2:
                raise RuntimeError("stop here")

        with pytest.raises(RuntimeError, match="stop here"):
            g.search_streamed(guides, 5, on_batch, batch=5)
        assert calls == [0, 5]
        h = g.search(guides, 5)
        assert h.to_numpy().tobytes() == want.tobytes()
        h.close()
        g.close()
    finally:
        m.close()


def test_multi_device_exchange_over_rccl_with_one_rank(oracle):
    """The RCCL leg of vsc_multi_search on the one GPU of this box: the hook rccl=1 sets up a one-rank
    communicator (ncclCommInitAll) and the records go through a grouped ncclSend / ncclRecv to the rank itself; the
    result is the single-context one."""
    rng = np.random.default_rng(808)
    guides = random_guides(rng, 40)
    contigs = make_genome(808, [50000, 20000], guides, 5, n_plant=300, n_runs=2)
    want = oracle.search_fast(contigs, guides, 5)
    m = va.MultiContext([0], rccl=True)
    try:
        assert m.uses_rccl
        g = m.load_genome(va.PackedGenome.from_sequences(contigs))
        h = g.search(guides, 5, algorithm="seed")
        got = h.to_numpy()
        assert m.timing()["used_rccl"] == 1
        h.close()
        g.close()
    finally:
        m.close()
    assert len(want) > 200 and got.tobytes() == want.tobytes()


def test_multi_device_stream_over_rccl_with_one_rank(ctx, oracle):
    """vsc_multi_search_stream with the RCCL leg on the one GPU of this box (hook rccl=1: a one-rank communicator that sends to
    itself): records AND votes go through grouped ncclSend / ncclRecv, batch after batch; result = the oracle's records, votes = one
    context's."""
    import torch
    from varscot_amd.classifier import Forest
    from varscot_amd.dist import DeviceAlias
    rng = np.random.default_rng(909)
    guides = random_guides(rng, 30)
    contigs = make_genome(909, [70000, 30000], guides, 5, n_plant=300, n_runs=2)
    packed = va.PackedGenome.from_sequences(contigs)
    want = oracle.search_fast(contigs, guides, 5)
    forest = Forest()
    act = np.random.default_rng(7).uniform(0.2, 1.8, size=len(guides))
    gen = ctx.load_genome(packed)
    h = gen.search(guides, 5, algorithm="seed")
    want_votes, _ = forest.classify_hits(h, act)
    h.close()
    gen.close()
    recs, votes = [], []

    def on_batch(h, first, count, votes_dev):
        a = h.to_numpy()
        recs.append(a)
        if len(a):
            votes.append(torch.as_tensor(DeviceAlias(votes_dev, 2 * len(a)), device="cuda:0").view(torch.int16).cpu().numpy().view(np.uint16).copy())

    m = va.MultiContext([0], rccl=True)
    try:
        assert m.uses_rccl
        g = m.load_genome(packed)
        g.search_streamed(guides, 5, on_batch, batch=8, algorithm="seed", score="votes", forest=forest, guide_activity=act)
        t = m.timing()
        g.close()
    finally:
        m.close()
    assert t["used_rccl"] == 1 and t["batches"] == 4
    assert len(want) > 200 and np.concatenate(recs).tobytes() == want.tobytes()
    assert np.array_equal(np.concatenate(votes), want_votes)


def test_multi_device_falls_back_to_copies_when_rccl_cannot_be_loaded(oracle):
    """librccl missing from the loader path (here: a library name that does not exist): vsc_multi_create used to
    build its error text from a second dlerror() call - NULL - and crashed; now the context reports why and the
    exchange runs as device copies.  Insisting on RCCL (rccl=1) is an error instead."""
    with pytest.raises(va.VarscotError):
        va.MultiContext([0], rccl=True, rccl_library="libdoes-not-exist-rccl.so")
    # (the load is only attempted for distinct devices: on this one-GPU box rccl="try" asks for it with one device)
    rng = np.random.default_rng(818)
    guides = random_guides(rng, 12)
    contigs = make_genome(818, [40000, 20000], guides, 4, n_plant=200, n_runs=2)
    want = oracle.search_fast(contigs, guides, 4)
    m = va.MultiContext([0], rccl="try", rccl_library="libdoes-not-exist-rccl.so")
    try:
        assert not m.uses_rccl and "RCCL not found" in m.last_error()
        g = m.load_genome(va.PackedGenome.from_sequences(contigs))
        h = g.search(guides, 4, algorithm="seed")
        got = h.to_numpy().copy()
        assert m.timing()["used_rccl"] == 0
        h.close()
        g.close()
    finally:
        m.close()
    assert len(want) > 100 and got.tobytes() == want.tobytes()


@pytest.mark.parametrize("n_bases,max_mm", [(8_000_000, 8), (50_000_000, 6)])
def test_reference_guides_on_a_repeat_rich_genome(ctx, oracle, golden_dir, n_bases, max_mm):
    """The 16 on-target guides the reference ships (several G-rich) on a genome with repeat families, tandem
    repeats and homopolymers: hit counts per read differ by orders of magnitude, so the hit buffers sized from the
    uniform-genome model overflow (second search launch), single reads own oversized bins of the sort (extra
    partition levels), and the seed buckets are skewed.  Records equal the oracle's; a second search on the same
    genome starts from the observed hit rate and needs no re-run."""
    names, guides, _ = real_guides(golden_dir)
    contigs = repeat_rich_genome(20 + max_mm, n_bases, guides)
    want = oracle.search_fast(contigs, guides, max_mm)
    per_read = np.bincount(want["guide"], minlength=len(guides))
    assert per_read.max() > 4 * max(1, int(np.median(per_read)))  # skew: the repeat family's guides dominate
    gen = ctx.load_genome(va.PackedGenome.from_sequences(contigs))
    stats = []
    for algo in ("seed", "seed", "scan"):
        h = gen.search(guides, max_mm, algorithm=algo)
        got = h.to_numpy()
        t = ctx.timing()
        stats.append((algo, t["passes"], t["sort_levels"], len(got)))
        h.close()
        assert got.tobytes() == want.tobytes(), stats
    gen.close()
    print("repeat-rich genome, %d bases, m=%d: %d hits, per read min/median/max %d/%d/%d; (algorithm, search launches, "
          "sort levels, hits) = %s" % (n_bases, max_mm, len(want), per_read.min(), int(np.median(per_read)), per_read.max(), stats))
    assert stats[0][1] >= 1 and stats[1][1] == 1  # the second seed search is sized from the first one's result


def test_seed_index_file_round_trip(tmp_path):
    """vsc_genome_index_save / _load: a second genome object that loads the file returns the records of the one
    that built the index, without building (timing: no index build); files of another genome, of another PAM set's
    search and truncated files are refused or rebuilt around."""
    rng = np.random.default_rng(77)
    guides = random_guides(rng, 12)
    contigs = make_genome(77, [60000, 25000], guides, 6, n_plant=300, n_runs=2)
    packed = va.PackedGenome.from_sequences(contigs)
    ctx = va.Context(0)
    g1 = ctx.load_genome(packed)
    with pytest.raises(va.VarscotError):
        g1.save_index(str(tmp_path / "none.vsi"))  # nothing to save yet
    g1.build_index()
    h = g1.search(guides, 6, algorithm="seed")
    want = h.to_numpy().copy()
    h.close()
    path = str(tmp_path / "g.vsi")
    g1.save_index(path)
    g2 = ctx.load_genome(packed)
    g2.load_index(path)
    h = g2.search(guides, 6, algorithm="auto")  # auto takes the seed path when an index is resident
    got = h.to_numpy().copy()
    h.close()
    assert got.tobytes() == want.tobytes() and len(want) > 200
    assert g2.device_bytes == g1.device_bytes
    # an extra PAM needs another index: the loaded one is replaced by a build, results as from a fresh genome
    h = g2.search(guides, 6, extra_pam="AG", algorithm="seed")
    h1 = g1.search(guides, 6, extra_pam="AG", algorithm="scan")
    assert h.to_numpy().tobytes() == h1.to_numpy().tobytes()
    h.close()
    h1.close()
    # another genome
    g3 = ctx.load_genome(va.PackedGenome.from_sequences([c[::-1] for c in contigs]))
    with pytest.raises(va.VarscotError, match="another genome"):
        g3.load_index(path)
    open(str(tmp_path / "cut.vsi"), "wb").write(open(path, "rb").read()[:5000])
    with pytest.raises(va.VarscotError, match="truncated"):
        g2.load_index(str(tmp_path / "cut.vsi"))
    h = g2.search(guides, 6, algorithm="seed")  # left without an index: builds again
    assert h.to_numpy().tobytes() == want.tobytes()
    h.close()
    # Same length, same contig layout, ONE base changed far behind the start of the planes / one base masked as N:
    # the fingerprint covers every word of all three planes, so the stale index is refused (it used to probe the
    # first 2 Mbp of hi / lo only - a patched or hard-masked assembly silently got the old genome's hits).
    for change in ("base", "mask"):
        c = list(contigs)
        at = len(c[1]) - 777
        c[1] = c[1][:at] + ("N" if change == "mask" else "ACGT"[("ACGT".index(c[1][at]) + 1) % 4]) + c[1][at + 1:]
        g4 = ctx.load_genome(va.PackedGenome.from_sequences(c))
        with pytest.raises(va.VarscotError, match="another genome"):
            g4.load_index(path)
        g4.close()
    # a file of the right length whose chunk table is damaged must not turn into out-of-bounds device reads
    blob = bytearray(open(path, "rb").read())
    S = int.from_bytes(blob[24:32], "little")
    edge_words = int.from_bytes(blob[48:56], "little")
    chunks = int.from_bytes(blob[64:68], "little")
    tab = 80 + 3 * S * 8 + edge_words * 4
    assert chunks > 3
    for field, value in ((0, 3 * S + 5), (1, 4000), (3, 0x7FFFFFF0)):  # first site, site count, first block of chunk 2
        bad = bytearray(blob)
        bad[tab + 2 * 16 + 4 * field:tab + 2 * 16 + 4 * field + 4] = int(value).to_bytes(4, "little")
        open(str(tmp_path / "bad.vsi"), "wb").write(bad)
        with pytest.raises(va.VarscotError, match="damaged"):
            g2.load_index(str(tmp_path / "bad.vsi"))
    g2.load_index(path)  # the intact file still loads
    h = g2.search(guides, 6, algorithm="seed")
    assert h.to_numpy().tobytes() == want.tobytes()
    h.close()
    for g in (g1, g2, g3):
        g.close()
    ctx.close()
