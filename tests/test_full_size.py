"""BASELINE.json full sizes (c2, c3) through size-independent properties, on the bench's own synthetic
3 Gbp genome: the streaming scan and the seed search - two independent algorithms over different data
structures - must return the same records; results are strictly sorted and unique; planted sites are
found.  (The oracle cannot run at this size; it checks both algorithms at sizes it finishes in seconds,
tests/test_gpu_parity.py.)"""
import numpy as np
import pytest
import torch

import varscot_amd as va
from varscot_amd import synth
from varscot_amd.dist import _DeviceAlias

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


def test_c2_scan_and_seed_agree_and_find_planted_sites(big):
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


def test_c3_scan_and_seed_agree_on_the_device(big):
    """1.6e9 records per result: compared where they are (HBM), through torch views of the record buffers."""
    ctx, packed, genome, guides, planted = big
    h_seed = genome.search(guides, 8, algorithm="seed")
    n, ascending, max_nm, sums = _digest(h_seed)
    assert n > 1_000_000_000 and ascending and max_nm <= 8
    mid = n // 2
    seed_mid = _device_records(h_seed)[mid:mid + 1_000_000].clone()
    h_seed.close()
    torch.cuda.empty_cache()
    h_scan = genome.search(guides, 8, algorithm="scan")
    assert _digest(h_scan) == (n, True, max_nm, sums)
    # and record by record on a slice from the middle
    assert bool((seed_mid == _device_records(h_scan)[mid:mid + 1_000_000]).all())
    h_scan.close()
    torch.cuda.empty_cache()
