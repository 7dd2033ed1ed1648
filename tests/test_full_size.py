"""BASELINE.json full sizes (c2 ... c5) on the bench's own synthetic 3 Gbp genome.

Against the ORACLE: the oracle's OpenMP port (oracle/vsc_fastport.c, checked against the character-level
restatement in tests/test_oracle.py) runs on the WHOLE genome - decoded from the packed planes by the oracle's own
decoder - for a subset of the reads (region edges 0 / 63 / 64, the last read, reads spread over the set), and every
record the GPU returns for those reads must equal it byte for byte: seed search and streaming scan at c2 and c3
size, the same reads inside streamed c5 batches, whole contigs (the last one, and the one that crosses position 2^31)
of a 4.28 Gbp genome, and the chrX / chrY windows of the c4 SNP genome (contig ids beyond 2^23; the windows
themselves against oracle/variants_oracle.py).  What the reference does there: read_mapping/bidir_mapping.cpp:39-86,
167-187.
Beyond the subset: the streaming scan and the seed search - two independent algorithms over different data
structures - must return the same records; results are strictly sorted and unique; planted sites are found."""
import numpy as np
import pytest
import torch

import varscot_amd as va
from oracle import pyoracle
from varscot_amd import synth
from varscot_amd.dist import DeviceAlias as _DeviceAlias

pytestmark = pytest.mark.gpu

BASES = 3_000_000_000


@pytest.fixture(scope="module")
def big():
    ctx = va.Context(0)
    packed = synth.synthetic_genome(BASES)
    ids, guides = synth.synthetic_guides(10_000)
    planted = synth.plant_sites(packed, guides[:1000], 400, 6)
    genome = ctx.load_genome(packed)
    genome.build_index()
    yield ctx, packed, genome, guides, planted
    genome.close()
    ctx.close()


# Reads the oracle searches on the whole genome.  Output regions hold 64 reads (read >> 6), a search pass 16 384, a
# streamed batch 10 000: both sides of every such boundary inside the c3 read set, + reads spread over it.
ORACLE_READS_C3 = sorted({0, 1, 62, 63, 64, 65, 127, 128, 999, 1000, 4095, 4096, 8191, 8192, 9998, 9999} | set(range(311, 9999, 421)))
ORACLE_READS_C2 = [r for r in ORACLE_READS_C3 if r < 1000] + [500, 937]
ORACLE_READS_STREAM = [10_000, 10_063, 10_064, 14_321, 19_999]  # second streamed batch of the c5 read set


def _genome_text(packed):
    """Every contig as text, decoded by the ORACLE's reader of the plane layout (oracle/vsc_planes.c)."""
    return [pyoracle.planes_to_text(packed.hi, packed.lo, packed.nmask, int(r["offset"]), int(r["length"])) for r in packed.contigs]


def _oracle_records(text, all_reads, which, max_mm, extra_pam=None):
    """The oracle's records for the reads `which` (ascending indices into all_reads), guide = the index in all_reads."""
    which = sorted(which)
    want = pyoracle.search_fast(text, [all_reads[i] for i in which], max_mm, extra_pam=extra_pam, cap=400_000 * len(which))
    want = want.copy()
    want["guide"] = np.asarray(which, dtype=np.uint32)[want["guide"]]
    return want


@pytest.fixture(scope="module")
def oracle_big(big):
    """What the oracle finds on the 3 Gbp genome: c3 subset at <= 8 mismatches (+ five reads of the second c5 batch),
    c2 subset at <= 6.  ~0.1 s per read on the GPU box's host."""
    ctx, packed, genome, guides, planted = big
    text = _genome_text(packed)
    ids, guides100k = synth.synthetic_guides(100_000)
    out = {
        "c3": _oracle_records(text, guides100k, ORACLE_READS_C3 + ORACLE_READS_STREAM, 8),
        "c2": _oracle_records(text, guides100k, sorted(ORACLE_READS_C2), 6),
    }
    del text
    assert len(out["c3"]) > 150_000 * len(ORACLE_READS_C3) and len(out["c2"]) > 3_000 * len(ORACLE_READS_C2)
    return out


def _records_of_reads(rec, reads):
    """The records of `reads` out of a device result (int32 [n, 4], sorted by guide) as a host HIT_DTYPE array."""
    g, n = rec[:, 0], rec.shape[0]

    def lower(v):
        lo, hi = 0, n
        while lo < hi:
            mid = (lo + hi) // 2
            if int(g[mid]) < v:
                lo = mid + 1
            else:
                hi = mid
        return lo

    parts = [rec[lower(r):lower(r + 1)] for r in sorted(reads)]
    got = torch.cat(parts).contiguous().cpu().numpy() if parts else np.zeros((0, 4), dtype=np.int32)
    return got.view(np.uint32).reshape(-1, 4).view(va.HIT_DTYPE).reshape(-1)


def _assert_equals_oracle(got, want, reads):
    want = want[np.isin(want["guide"], np.asarray(sorted(reads), dtype=np.uint32))]
    assert len(got) == len(want), (len(got), len(want))
    assert got.tobytes() == want.tobytes()


def _device_records(hits):
    n = len(hits)
    t = torch.as_tensor(_DeviceAlias(hits.device_ptr, n * va.HIT_DTYPE.itemsize), device="cuda:0")
    return t.view(torch.int32).view(-1, 4)


def _keys(rec):
    """(guide, strand, contig, pos) as one int64 per record (guide < 2^14, contig < 2^6, pos < 2^32)."""
    g = rec[:, 0].to(torch.int64)
    c = rec[:, 1].to(torch.int64)
    p = rec[:, 2].to(torch.int64) & 0xFFFFFFFF
    s = (rec[:, 3].to(torch.int64) >> 31) & 1
    return (((g << 1 | s) << 6 | c) << 32) | p


def test_c2_scan_and_seed_agree_and_find_planted_sites(big, oracle_big):
    ctx, packed, genome, guides, planted = big
    reads = guides[:1000]
    h_seed = genome.search(reads, 6, algorithm="seed")
    a = h_seed.to_numpy().copy()
    h_seed.close()
    h_scan = genome.search(reads, 6, algorithm="scan")
    b = h_scan.to_numpy().copy()
    h_scan.close()
    assert len(a) > 1_000_000
    assert a.tobytes() == b.tobytes()
    # the oracle on the whole 3 Gbp genome, for a subset of the reads: byte-equal records
    _assert_equals_oracle(a[np.isin(a["guide"], np.asarray(ORACLE_READS_C2, dtype=np.uint32))], oracle_big["c2"], ORACLE_READS_C2)
    key = ((a["guide"].astype(np.int64) << 1 | (a["info"] >> 31)) << 6 | a["contig"]) << 32 | a["pos"]
    assert np.all(np.diff(key) > 0)  # strictly ascending: sorted and no duplicates
    nm = (a["info"] >> 23) & 31
    assert nm.max() <= 6
    sample = a["info"][:: max(1, len(a) // 5000)]
    assert all(bin(int(x) & 0x7FFFFF).count("1") == ((int(x) >> 23) & 31) for x in sample)  # NM = popcount(mask)
    found = set(zip(a["guide"].tolist(), a["contig"].tolist(), a["pos"].tolist(), (a["info"] >> 31).tolist()))
    for gi, c, pos, strand, nsub in planted:
        assert (gi, c, pos, strand) in found


def _digest(hits, chunk=1 << 27):
    """(count, strictly ascending?, max NM, three checksums) of a result, computed in HBM chunk by chunk."""
    rec = _device_records(hits)
    n = rec.shape[0]
    ascending, max_nm, sums, last = True, 0, [0, 0, 0], None
    for b in range(0, n, chunk):
        r = rec[b:b + chunk]
        k = _keys(r)
        ascending = ascending and bool((k[1:] > k[:-1]).all()) and (last is None or int(k[0]) > last)
        last = int(k[-1])
        mask = (r[:, 3] & 0x7FFFFF).to(torch.int64)
        nm = ((r[:, 3] >> 23) & 31).to(torch.int64)
        max_nm = max(max_nm, int(nm.max()))
        sums[0] += int(k.sum())
        sums[1] += int((k ^ (mask << 7)).sum())
        sums[2] += int(nm.sum())
        del k, mask, nm
    return n, ascending, max_nm, sums


def test_c3_scan_and_seed_agree_on_the_device(big, oracle_big):
    """1.6e9 records per result: compared where they are (HBM), through torch views of the record buffers - and, for
    the oracle's read subset, byte for byte with what the oracle finds on the whole genome (both kernels)."""
    ctx, packed, genome, guides, planted = big
    h_seed = genome.search(guides, 8, algorithm="seed")
    n, ascending, max_nm, sums = _digest(h_seed)
    assert n > 1_000_000_000 and ascending and max_nm <= 8
    mid = n // 2
    seed_mid = _device_records(h_seed)[mid:mid + 1_000_000].clone()
    _assert_equals_oracle(_records_of_reads(_device_records(h_seed), ORACLE_READS_C3), oracle_big["c3"], ORACLE_READS_C3)
    h_seed.close()
    torch.cuda.empty_cache()
    h_scan = genome.search(guides, 8, algorithm="scan")
    assert _digest(h_scan) == (n, True, max_nm, sums)
    # and record by record on a slice from the middle
    assert bool((seed_mid == _device_records(h_scan)[mid:mid + 1_000_000]).all())
    _assert_equals_oracle(_records_of_reads(_device_records(h_scan), ORACLE_READS_C3), oracle_big["c3"], ORACLE_READS_C3)
    h_scan.close()
    torch.cuda.empty_cache()


def test_full_size_genome_sharded_behind_the_abi(big, oracle_big):
    """The product's multi-device driver (vsc_multi_search, here three / two contexts on the one GPU) on the 3 Gbp genome: shards
    whose positions start at 1.0e9 and 2.0e9 (records hold shard-relative positions, bins of the upper shards, contig lookup
    with a position base), one exchange of 8-byte records, the segment merge.  c2: byte-equal to the single-context result and,
    for the oracle's read subset, to the oracle.  c3: the same digest (count, order, NM bound, three checksums) as one context,
    the oracle's subset again."""
    ctx, packed, genome, guides, planted = big
    ctx.release_scratch()
    h = genome.search(guides[:1000], 6, algorithm="seed")
    one = h.to_numpy().copy()
    h.close()
    h = genome.search(guides, 8, algorithm="seed")
    want_c3 = _digest(h)
    h.close()
    ctx.release_scratch()
    m = va.MultiContext([0, 0, 0])
    try:
        g = m.load_genome(packed)
        g.build_index()
        h = g.search(guides[:1000], 6, algorithm="seed")
        got = h.to_numpy().copy()
        h.close()
        assert got.tobytes() == one.tobytes()
        _assert_equals_oracle(got[np.isin(got["guide"], np.asarray(ORACLE_READS_C2, dtype=np.uint32))], oracle_big["c2"], ORACLE_READS_C2)
        g.close()
    finally:
        m.close()
    m = va.MultiContext([0, 0])
    try:
        g = m.load_genome(packed)
        g.build_index()
        h = g.search(guides, 8, algorithm="seed")
        assert _digest(h) == want_c3
        _assert_equals_oracle(_records_of_reads(_device_records(h), ORACLE_READS_C3), oracle_big["c3"], ORACLE_READS_C3)
        h.close()
        g.close()
    finally:
        m.close()
    torch.cuda.empty_cache()


def test_c5_streamed_batches_equal_the_oracle_on_the_read_subset(big, oracle_big):
    """Two streamed batches of 10 000 reads (the first 20 000 of the c5 read set, <= 8 mismatches): the records of
    the oracle's reads inside each batch - first / last read of a batch, region edges - equal the oracle's."""
    ctx, packed, genome, guides, planted = big
    ids, guides100k = synth.synthetic_guides(100_000)
    seen = []

    def on_batch(h, first, count):
        reads = [r for r in ORACLE_READS_C3 + ORACLE_READS_STREAM if first <= r < first + count]
        _assert_equals_oracle(_records_of_reads(_device_records(h), reads), oracle_big["c3"], reads)
        seen.append((first, count, len(reads)))

    genome.search_streamed(guides100k[:20_000], 8, on_batch, batch=10_000, algorithm="seed")
    assert seen == [(0, 10_000, len(ORACLE_READS_C3)), (10_000, 10_000, len(ORACLE_READS_STREAM))]
    torch.cuda.empty_cache()


def test_c5_streamed_batches_with_packed_scoring(big):
    """BASELINE.json configs[4] in the shape one GPU runs it: the reads streamed in batches of 5 000 at <= 8
    mismatches through vsc_search_stream, every batch scored (packed feature rows) from the callback and
    dropped.  Two batches = the c3 read set, so the batches' digests must add up to the digest of the one
    c3 search; the packed rows carry totalMismatches = the popcount of the record's mask over read
    positions 0..20, every row."""
    ctx, packed, genome, guides, planted = big
    whole = genome.search(guides, 8, algorithm="seed")
    want = _digest(whole)
    whole.close()
    torch.cuda.empty_cache()
    seen = []

    def on_batch(h, first, count):
        n, ascending, max_nm, sums = _digest(h)
        rec = _device_records(h)
        assert int(rec[:, 0].min()) >= first and int(rec[:, 0].max()) < first + count
        rows = torch.empty((n, 16), dtype=torch.int32, device="cuda:0")
        h.packed_features(to_host=False, dev_ptr=rows.data_ptr())
        ok = True
        for b in range(0, n, 1 << 27):
            info = rec[b:b + (1 << 27), 3]
            mask = info & 0x7FFFFF
            strand = (info >> 31) & 1
            # read positions 0..20 are window positions 0..20 on '+', 2..22 on '-'
            nonpam = torch.where(strand == 1, mask >> 2, mask & 0x1FFFFF)
            pop = torch.zeros_like(nonpam)
            for bit in range(21):
                pop += (nonpam >> bit) & 1
            w0 = rows[b:b + (1 << 27), 0]
            ok = ok and bool((((w0 >> 21) & 31) == pop).all()) and bool(((w0 & 0x1FFFFF).to(torch.int64).ne(0) == pop.ne(0)).all())
            del info, mask, strand, nonpam, pop, w0
        seen.append((first, count, n, ascending, max_nm, sums, ok))
        del rows
        torch.cuda.empty_cache()

    genome.search_streamed(guides, 8, on_batch, batch=5000, algorithm="seed")
    assert [(f, c) for f, c, *_ in seen] == [(0, 5000), (5000, 5000)]
    assert all(asc and nm <= 8 and ok for _, _, _, asc, nm, _, ok in seen)
    assert sum(s[2] for s in seen) == want[0]
    # (the checksums are sums of 64-bit integers that wrap inside torch: equal modulo 2^64 however they are chunked)
    assert [sum(s[5][k] for s in seen) % (1 << 64) for k in range(3)] == [x % (1 << 64) for x in want[3]]
    t = ctx.timing()
    assert t["read_passes"] == 2 and t["hits"] == want[0] and t["score_ms"] > 0


def test_c5_rows_written_on_the_way_equal_the_gathered_rows(big, oracle_big):
    """vsc_search_stream_rows at c5 batch size on the 3 Gbp genome: two batches of 5 000 reads at <= 8 mismatches, 8e8 hits each.
    Every batch's records are strictly ascending and - for the oracle's read subset - the oracle's; every one of its 64-byte rows
    (written by the record assembly from the bases the search kept beside the records) equals the row vsc_score_hits_packed makes
    of the same record by gathering the window from the planes, compared in HBM."""
    ctx, packed, genome, guides, planted = big
    ctx.release_scratch()
    seen = []

    def on_batch(h, first, count, rows_dev):
        n, ascending, max_nm, sums = _digest(h)
        reads = [r for r in ORACLE_READS_C3 if first <= r < first + count]
        _assert_equals_oracle(_records_of_reads(_device_records(h), reads), oracle_big["c3"], reads)
        got = torch.as_tensor(_DeviceAlias(rows_dev, n * 64), device="cuda:0").view(torch.int32).view(-1, 16)
        ref = torch.empty((n, 16), dtype=torch.int32, device="cuda:0")
        h.packed_features(to_host=False, dev_ptr=ref.data_ptr())  # (into `ref`: the library's row buffer is not touched)
        same = all(bool((got[b:b + (1 << 26)] == ref[b:b + (1 << 26)]).all()) for b in range(0, n, 1 << 26))
        seen.append((first, count, n, ascending, max_nm, same))
        del ref
        torch.cuda.empty_cache()

    genome.search_streamed_rows(guides, 8, on_batch, batch=5000, algorithm="seed")
    assert [(f, c) for f, c, *_ in seen] == [(0, 5000), (5000, 5000)]
    assert all(n > 700_000_000 and asc and nm <= 8 and same for _, _, n, asc, nm, same in seen), seen
    ctx.release_scratch()


def test_c5_all_100000_reads_streamed(big):
    """BASELINE.json configs[4] at its stated size on one GPU: all 100 000 reads streamed in 10 batches of 10 000 at
    <= 8 mismatches (1.6e10 records in total), every batch scored from the callback (packed feature rows, dropped).
    Per batch: strictly ascending records, NM <= 8, reads inside the batch's range, NM = popcount(mask) on a
    sample; the batch counts add up to the library's hit total; ONE
    batch (reads 30 000 .. 39 999) is compared record for record with the streaming scan of the same reads - the
    other search algorithm over the other data structure; and on a slice of that batch the fused
    score -> classify path equals packed scoring followed by the forest."""
    from varscot_amd.classifier import Forest
    ctx, packed, genome, guides10k, planted = big
    ids, guides = synth.synthetic_guides(100_000)
    assert guides[:10_000] == guides10k
    seen, kept = [], {}
    forest = Forest()
    act = np.random.default_rng(5).uniform(0.2, 1.8, size=len(guides))
    # room for this test's own torch buffers beside the library's: packed rows pass through 17 GB of scratch (six
    # passes per batch) instead of the 104 GB a whole batch's rows would take
    ctx.release_scratch()
    ctx.set_debug(score_chunk=1 << 28)

    def on_batch(h, first, count):
        n, ascending, max_nm, sums = _digest(h)
        rec = _device_records(h)
        in_range = int(rec[:, 0].min()) >= first and int(rec[:, 0].max()) < first + count
        sample = rec[:: max(1, n // 100_000), 3].to(torch.int64)
        pop = torch.zeros_like(sample)
        for bit in range(23):
            pop += (sample >> bit) & 1
        nm_ok = bool((pop == ((sample >> 23) & 31)).all())
        h.packed_features(to_host=False, mit=False)  # the c5 scoring, rows dropped
        if first == 30_000:
            kept["records"] = rec.clone()
            # fused score -> classify on a slice against the two-step path (rows to device memory, then the forest)
            k0, kn = n // 3, 200_000
            rows = torch.empty((kn, 16), dtype=torch.int32, device="cuda:0")
            h.packed_features(first=k0, count=kn, to_host=False, dev_ptr=rows.data_ptr())
            g = rec[k0:k0 + kn, 0].cpu().numpy()
            prob, _, _ = forest.predict_packed(ctx, kn, act[g], dev_ptr=rows.data_ptr())
            votes, _ = forest.classify_hits(h, act, first=k0, count=kn)
            kept["fused_ok"] = bool(np.array_equal(votes / 1000.0, prob)) and 0 < (votes > 500).mean() < 1
            del rows
        seen.append((first, count, n, ascending, max_nm, sums, in_range and nm_ok))
        torch.cuda.empty_cache()

    try:
        genome.search_streamed(guides, 8, on_batch, batch=10_000, algorithm="seed")
    finally:
        ctx.set_debug()
    t = ctx.timing()
    assert [(f, c) for f, c, *_ in seen] == [(10_000 * i, 10_000) for i in range(10)]
    assert all(asc and nm <= 8 and ok for _, _, _, asc, nm, _, ok in seen), seen
    total = sum(s[2] for s in seen)
    assert total > 15_000_000_000 and t["hits"] == total and t["read_passes"] == 10 and t["score_ms"] > 0
    counts = [s[2] for s in seen]
    assert max(counts) < 1.02 * min(counts)  # uniform reads on a uniform genome: every batch is another c3
    assert kept["fused_ok"]
    # batch 3 against the streaming scan of the same reads (the stream's pooled scratch - 104 GB of packed rows
    # among it - goes back to the device first)
    ctx.release_scratch()
    h_scan = genome.search(guides[30_000:40_000], 8, algorithm="scan")
    scan = _device_records(h_scan)
    a = kept.pop("records")
    assert a.shape == scan.shape
    ctx.release_scratch()
    for b in range(0, a.shape[0], 1 << 26):
        x, y = a[b:b + (1 << 26)], scan[b:b + (1 << 26)]
        assert bool((x[:, 1:] == y[:, 1:]).all()) and bool(((x[:, 0] - y[:, 0]) == 30_000).all())
    h_scan.close()
    del a, scan, x, y
    torch.cuda.empty_cache()


def test_c4_variant_windows_at_full_size(big, tmp_path):
    """BASELINE.json configs[3]: ~5 M SNPs on the 3 Gbp genome -> 8.9 M alt-allele windows built straight from
    the packed planes (vsc_windows_build) and searched as a second genome with millions of contigs (contig ids
    beyond 2^23, the finalize kernel's contig lookup in global memory).  Properties: scan == seed on the SNP
    genome; sorted, unique, NM <= 6; a site that exists only with the ALT allele is found in the ALT window of
    its SNP and not in the REF window; window bases equal the reference around the SNP."""
    ctx, packed, genome, guides, planted = big
    vcf = str(tmp_path / "c4.vcf")
    n_snps = synth.synthetic_vcf(packed, 5_000_000, vcf)
    assert n_snps > 4_900_000
    snp = va.variant_windows(packed, vcf, sample=0)
    assert len(snp.contigs) > 8_500_000 and int(snp.contigs["length"].max()) < 400
    # an isolated heterozygous SNP: windows <chr>_<start>_REF and <chr>_<start>_ALT_<pos>_<ref>_<alt>, 45 bases each
    k = next(i for i in range(4_000_000, 4_100_000)
             if snp.names[i].endswith("_REF") and int(snp.contigs["length"][i]) == 45 and "_ALT_" in snp.names[i + 1]
             and int(snp.contigs["length"][i + 1]) == 45 and snp.names[i + 1].count("_") == 5)
    chrom, start = snp.names[k].split("_")[0], int(snp.names[k].split("_")[1])
    c_ref = packed.names.index(chrom)
    ref_seq = packed.decode(int(packed.contigs[c_ref]["offset"]) + start, 45)
    alt_name = snp.names[k + 1].split("_")
    assert snp.contig_sequence(k) == ref_seq
    pos, ref_base, alt_base = int(alt_name[3]), alt_name[4], alt_name[5]
    assert ref_seq[pos - start] == ref_base
    assert snp.contig_sequence(k + 1) == ref_seq[:pos - start] + alt_base + ref_seq[pos - start + 1:]
    # a read that matches the ALT window exactly around the SNP (+ a valid PAM is not guaranteed: take the window's
    # own 23-mer and let the search decide through the extra-PAM option)
    alt_seq = snp.contig_sequence(k + 1)
    off = pos - start - 10  # the SNP sits at read position 10
    probe = alt_seq[off:off + 23]
    assert "N" not in probe
    reads = guides[:999] + [probe]
    snp_gen = ctx.load_genome(snp)
    h_seed = snp_gen.search(reads, 6, extra_pam=probe[21:], algorithm="seed")
    a = h_seed.to_numpy().copy()
    h_seed.close()
    h_scan = snp_gen.search(reads, 6, extra_pam=probe[21:], algorithm="scan")
    b = h_scan.to_numpy().copy()
    h_scan.close()
    snp_gen.close()
    assert len(a) > 100_000 and a.tobytes() == b.tobytes()
    key = ((a["guide"].astype(np.int64) << 1 | (a["info"] >> 31)) << 24 | a["contig"]) << 9 | a["pos"]
    assert np.all(np.diff(key) > 0)
    assert ((a["info"] >> 23) & 31).max() <= 6
    assert a["contig"].max() > (1 << 23)
    mine = a[(a["guide"] == 999) & ((a["info"] >> 31) == 0)]
    exact = mine[((mine["info"] >> 23) & 31) == 0]
    assert (k + 1, off) in set(zip(exact["contig"].tolist(), exact["pos"].tolist()))
    one_off = mine[(mine["contig"] == k) & (mine["pos"] == off)]
    assert len(one_off) == 1 and int(one_off["info"][0]) & 0x7FFFFF == 1 << 10  # REF window: the SNP position mismatches
    # Against the oracle, on the windows of chr22, chrX and chrY (the last ~ 8 % of the SNP genome: contig ids from below
    # to far beyond 2^23).  (1) the windows themselves = oracle/variants_oracle.py on those chromosomes' VCF records and
    # the oracle's decoding of the reference planes; (2) the records of ALL 1 000 reads on them = the oracle's search of
    # the oracle's window sequences.
    from oracle import variants_oracle as vo
    chroms = ("chr22", "chrX", "chrY")
    with open(vcf) as f:
        sub = "".join(line for line in f if line[0] == "#" or line.split("\t", 1)[0] in chroms)
    ref_text = {}
    for name in chroms:
        c = packed.names.index(name)
        ref_text[name] = pyoracle.planes_to_text(packed.hi, packed.lo, packed.nmask, int(packed.contigs[c]["offset"]),
                                                 int(packed.contigs[c]["length"])).tobytes().decode()
    want_windows = vo.vcf_loader(sub, ref_text, 0, 23)
    del ref_text
    n_win = len(snp.contigs)
    first = n_win - len(want_windows)
    assert len(want_windows) > 400_000 and first < (1 << 23) < n_win
    assert snp.names[first].startswith("chr22_") and not snp.names[first - 1].startswith("chr22_")
    window_text = pyoracle.planes_to_text(snp.hi, snp.lo, snp.nmask, int(snp.contigs[first]["offset"]),
                                          int(snp.contigs[n_win - 1]["offset"]) + int(snp.contigs[n_win - 1]["length"]) - int(snp.contigs[first]["offset"]))
    base = int(snp.contigs[first]["offset"])
    for i in range(0, len(want_windows), 1):
        rid, seq = want_windows[i]
        row = snp.contigs[first + i]
        o = int(row["offset"]) - base
        if int(row["length"]) != len(seq) or window_text[o:o + len(seq)].tobytes() != seq.encode() or (i % 97 == 0 and snp.names[first + i] != rid):
            raise AssertionError("window %d (%s) differs from the oracle's %s" % (first + i, snp.names[first + i], rid))
    want = pyoracle.search_fast([seq.encode() for _, seq in want_windows], reads, 6, extra_pam=probe[21:], cap=2_000_000).copy()
    want["contig"] += first
    got = a[a["contig"] >= first]
    assert len(want) > 5_000 and len(got) == len(want)
    # a holds (guide, strand, contig, pos) order; so does the oracle
    assert got.tobytes() == want.tobytes()


def test_skewed_reads_at_scale():
    """Three of 300 reads with ~130 000 extra sites each inside one 40 Mbp window of a 1 Gbp genome (an Alu-like
    family): their bins exceed the finalize kernel's capacity by an order of magnitude, so the result goes through
    further partition levels, and the regions that hold them fill far beyond the uniform model the first buffer
    size comes from (the search re-runs with more room).  scan == seed, strictly sorted, planted sites found."""
    from varscot_amd import _lib
    ctx = va.Context(0)
    packed = synth.synthetic_genome(1_000_000_000)
    ids, guides = synth.synthetic_guides(300)
    rng = np.random.default_rng(99)
    c = int(np.argmax(packed.contigs["length"]))
    off, ln = int(packed.contigs[c]["offset"]), int(packed.contigs[c]["length"])
    assert ln > 45_000_000
    L = _lib.lib()
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    planted = []
    starts = rng.choice(40_000_000 // 32, size=400_000, replace=False) * 32 + 1_000_000  # no two sites overlap
    for k, pos in enumerate(starts.tolist()):
        gi = k % 3
        site = list(guides[gi])
        for q in rng.choice(20, size=int(rng.integers(0, 4)), replace=False):
            site[q] = "ACGT"[("ACGT".index(site[q]) + 1 + int(rng.integers(0, 3))) % 4]
        strand = k & 1
        t = "".join(site)
        if strand:
            t = "".join(comp[ch] for ch in reversed(t))
        L.vsc_pack_bases(t.encode(), 23, off + pos, _lib.ptr(packed.hi), _lib.ptr(packed.lo), _lib.ptr(packed.nmask))
        if k < 3000:
            planted.append((gi, c, pos, strand))
    genome = ctx.load_genome(packed)
    genome.build_index()
    h = genome.search(guides, 8, algorithm="seed")
    t_seed = ctx.timing()
    a = h.to_numpy().copy()
    h.close()
    h = genome.search(guides, 8, algorithm="scan")
    b = h.to_numpy().copy()
    h.close()
    genome.close()
    ctx.close()
    assert a.tobytes() == b.tobytes()
    per_read = np.bincount(a["guide"], minlength=300)
    assert per_read[:3].min() > 100_000 and per_read[3:].max() < 100_000
    assert t_seed["sort_levels"] >= 2
    key = ((a["guide"].astype(np.int64) << 1 | (a["info"] >> 31)) << 6 | a["contig"]) << 32 | a["pos"]
    assert np.all(np.diff(key) > 0)
    found = set(zip(a["guide"][a["guide"] < 3].tolist(), a["contig"][a["guide"] < 3].tolist(), a["pos"][a["guide"] < 3].tolist(),
                    (a["info"][a["guide"] < 3] >> 31).tolist()))
    assert all(p in found for p in planted)


def test_genome_close_to_the_32_bit_position_limit():
    """4.29e9 positions - 8 192 is what bidir_index accepts (RM/bidir_index.cpp:17: 4 giga bases); here 4.28e9 bases
    in 24 contigs: positions up to 2^32 - 1.4e7, all 32 position bits in use, bins at the very top of the key space.
    scan == seed, strictly sorted, planted sites near the end of the last contig found."""
    ctx = va.Context(0)
    packed = synth.synthetic_genome(4_280_000_000)
    last = len(packed.contigs) - 1
    span = int(packed.contigs[last]["offset"]) + int(packed.contigs[last]["length"])
    assert (1 << 32) - (1 << 25) < span < (1 << 32) - 8192
    ids, guides = synth.synthetic_guides(200)
    L = _lib_handle()
    planted = []
    for i in range(40):  # sites in the last 10 Mbp of the last contig, both strands
        g = guides[i]
        site = g if i % 2 == 0 else "".join({"A": "T", "C": "G", "G": "C", "T": "A"}[c] for c in reversed(g))
        pos = int(packed.contigs[last]["length"]) - 10_000_000 + 200_003 * i
        L.vsc_pack_bases(site.encode(), 23, int(packed.contigs[last]["offset"]) + pos, _ptr(packed.hi), _ptr(packed.lo), _ptr(packed.nmask))
        planted.append((i, last, pos, i % 2))
    genome = ctx.load_genome(packed)
    genome.build_index()
    h = genome.search(guides, 6, algorithm="seed")
    a = h.to_numpy().copy()
    h.close()
    h = genome.search(guides, 6, algorithm="scan")
    b = h.to_numpy().copy()
    h.close()
    genome.close()
    ctx.close()
    assert len(a) > 500_000 and a.tobytes() == b.tobytes()
    key = ((a["guide"].astype(np.int64) << 1 | (a["info"] >> 31)) << 6 | a["contig"]) << 32 | a["pos"]
    assert np.all(np.diff(key) > 0)
    found = set(zip(a["guide"].tolist(), a["contig"].tolist(), a["pos"].tolist(), (a["info"] >> 31).tolist()))
    assert all(p in found for p in planted)
    assert int((a["contig"] == last).sum()) > 10_000
    # the oracle on whole contigs of this genome: the last one (positions up to 2^32 - 1.4e7) and the one that holds
    # global position 2^31 (the sign bit of a 32-bit position) - all 200 reads, byte-equal records
    offs = packed.contigs["offset"].astype(np.int64)
    across = int(np.searchsorted(offs, 1 << 31, side="right")) - 1
    assert offs[across] < (1 << 31) < offs[across] + int(packed.contigs[across]["length"]) and across != last
    for c in (across, last):
        text = pyoracle.planes_to_text(packed.hi, packed.lo, packed.nmask, int(offs[c]), int(packed.contigs[c]["length"]))
        want = pyoracle.search_fast([text], guides, 6, cap=4_000_000).copy()
        want["contig"] = c
        got = a[a["contig"] == c]
        assert len(want) > 10_000 and len(got) == len(want) and got.tobytes() == want.tobytes()


def _lib_handle():
    from varscot_amd import _lib
    return _lib.lib()


def _ptr(x):
    from varscot_amd import _lib
    return _lib.ptr(x)
