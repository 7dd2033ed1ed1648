"""Multi-rank path on CPU: world-size-2/3 gloo process groups exercise the sharding rule and the
one-exchange gather of hit records (varscot_amd/dist.py).  Per-shard hit lists come from the oracle
here (the checker) because this box has no GPU; the GPU tests run the same gather code on real
search results (tests/test_gpu_parity.py::test_sharded_search_over_gloo)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_hits(contigs, guides, max_mm, packed, rank, world):
    """Oracle hits whose window starts inside rank's plane range (global position rule)."""
    from oracle import pyoracle
    h = pyoracle.search_fast(contigs, guides, max_mm, threads=1)
    b, e = packed.shard_words(rank, world)
    gpos = packed.contigs["offset"][h["contig"]] + h["pos"]
    keep = (gpos >= b * 32) & (gpos < e * 32)
    return h[keep]


def _worker(rank, world, port, seed, q, mode="root"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import varscot_amd as va
        from varscot_amd import dist as vdist
        from helpers import make_genome, random_guides, xpack
        rng = np.random.default_rng(seed)
        guides = random_guides(rng, 12)
        contigs = make_genome(seed, [40000, 15000, 9000, 50], guides, 6, n_plant=300, n_runs=4)
        packed = va.PackedGenome.from_sequences(contigs)
        assert vdist.shard_words(packed.n_words, rank, world) == packed.shard_words(rank, world)
        mine = _shard_hits(contigs, guides, 6, packed, rank, world)
        # the 8-byte exchange records + per-key counts (on the GPU: vsc_hits_pack_exchange)
        rec, counts = xpack(mine, packed.contigs["offset"], len(guides))
        local = torch.from_numpy(rec.view(np.uint8).copy())
        all_counts = vdist.all_gather_key_counts(counts, None)
        assert all_counts.shape == (world, 2 * len(guides)) and np.array_equal(all_counts[rank], counts)
        if mode in ("reads", "async"):
            # "async": two exchanges one after the other without a barrier in between (what the pipelined search does)
            for _ in range(2 if mode == "async" else 1):
                recv, part_counts, k0, reqs = vdist.start_exchange_by_reads(local, all_counts, len(guides))
                for req in reqs:
                    req.wait()
            q.put((rank, recv.numpy().tobytes(), part_counts, k0))
        else:
            recv, reqs = vdist.start_gather_to_root(local, all_counts)
            for req in reqs:
                req.wait()
            if rank == 0:
                q.put((recv.numpy().tobytes(), all_counts))
            else:
                assert recv is None
                q.put(None)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _case(seed):
    from helpers import make_genome, random_guides
    import varscot_amd as va
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, 12)
    contigs = make_genome(seed, [40000, 15000, 9000, 50], guides, 6, n_plant=300, n_runs=4)
    return guides, contigs, va.PackedGenome.from_sequences(contigs)


def _run(world, seed, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("world", [2, 3])
def test_gather_to_root_over_gloo(world, oracle):
    """exchange="root" (the north star's single gather): rank 0 receives every shard's 8-byte exchange records in
    shard order; rebuilding the 16-byte records from them and the all-gathered per-key counts (what
    vsc_hits_merge_packed does on the GPU, restated in helpers.xmerge) gives the global result."""
    from helpers import xmerge
    seed = 900 + world
    payload = [r for r in _run(world, seed, "root") if r is not None]
    assert len(payload) == 1
    blob, all_counts = payload[0]
    guides, contigs, packed = _case(seed)
    shards = [_shard_hits(contigs, guides, 6, packed, r, world) for r in range(world)]
    totals = all_counts.astype(np.int64).sum(axis=1)
    assert list(totals) == [len(s) for s in shards] and totals.sum() > 50 and totals.min() > 0
    got = np.frombuffer(blob, dtype=np.uint64)
    assert len(got) * 8 == 8 * totals.sum()  # half of what 16-byte records would have cost
    merged = xmerge(got, all_counts, 0, packed.contigs["offset"])
    whole = oracle.search_fast(contigs, guides, 6)
    assert merged.tobytes() == whole.tobytes()


@pytest.mark.parametrize("mode", ["reads", "async"])
@pytest.mark.parametrize("world", [2, 3])
def test_exchange_by_reads_over_gloo(world, oracle, mode):
    """exchange="reads": every rank ends up with the exchange records of its read range from all genome shards, in
    shard order; merging each rank's part and concatenating the ranks gives the global result."""
    from helpers import xmerge
    from varscot_amd import dist as vdist
    seed = 950 + world
    results = sorted(_run(world, seed, mode), key=lambda r: r[0])
    guides, contigs, packed = _case(seed)
    whole = oracle.search_fast(contigs, guides, 6)
    assert len(whole) > 50
    parts = []
    for rank, blob, part_counts, k0 in results:
        b, e = vdist.read_range(len(guides), rank, world)
        assert k0 == 2 * b and part_counts.shape == (world, 2 * (e - b))
        got = np.frombuffer(blob, dtype=np.uint64)
        merged = xmerge(got, part_counts, k0, packed.contigs["offset"])
        want = whole[(whole["guide"] >= b) & (whole["guide"] < e)]
        assert merged.tobytes() == want.tobytes()
        parts.append(merged)
    assert np.concatenate(parts).tobytes() == whole.tobytes()


def _stream_worker(rank, world, port, seed, q, batch):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import varscot_amd as va
        from varscot_amd import dist as vdist
        from helpers import make_genome, random_guides, xpack, xmerge
        rng = np.random.default_rng(seed)
        guides = random_guides(rng, 12)
        contigs = make_genome(seed, [40000, 15000, 9000, 50], guides, 6, n_plant=300, n_runs=4)
        packed = va.PackedGenome.from_sequences(contigs)
        mine = _shard_hits(contigs, guides, 6, packed, rank, world)
        offsets = packed.contigs["offset"]

        def produce(first, part):
            """this rank's records of the batch (the oracle's hits stand in for the GPU search), keys local to the batch, and a
            vote per hit that the test can recompute from the merged record: a hash of (read, contig, pos, strand)"""
            h = mine[(mine["guide"] >= first) & (mine["guide"] < first + len(part))].copy()
            votes = ((h["guide"] * 7 + h["contig"] * 131 + h["pos"] * 3 + (h["info"] >> 31)) % 1001).astype(np.uint16)
            h["guide"] -= first
            rec, counts = xpack(h, offsets, len(part))
            return torch.from_numpy(rec.view(np.uint8).copy()), counts, torch.from_numpy(votes.view(np.uint8).copy())

        def merge(recv, vrecv, all_counts, first):
            rec = np.frombuffer(recv.numpy().tobytes(), dtype=np.uint64)
            v = np.frombuffer(vrecv.numpy().tobytes(), dtype=np.uint16)
            merged = xmerge(rec, all_counts, 2 * first, offsets)
            # the votes follow their records: segment by segment, as vsc_hits_merge_packed_votes moves them
            n_shards, n_keys = all_counts.shape
            starts = np.concatenate([[0], np.cumsum(all_counts.astype(np.int64).ravel())])
            out = np.concatenate([v[starts[s * n_keys + k]:starts[s * n_keys + k] + int(all_counts[s, k])]
                                  for k in range(n_keys) for s in range(n_shards)] + [np.zeros(0, dtype=np.uint16)])
            return merged, out

        seen = []

        def on_batch(merged, first, count, votes):
            seen.append((first, count, None if merged is None else merged.tobytes(), None if votes is None else votes.tobytes()))

        codes = np.arange(len(guides), dtype=np.uint64)  # (only sliced and counted here)
        vdist.sharded_search_stream(None, None, codes, 6, on_batch, batch, produce=produce, merge=merge)
        q.put((rank, seen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 5), (3, 4), (2, 12)])
def test_streamed_gather_to_root_with_votes_over_gloo(world, batch, oracle):
    """The protocol of the streamed, scored search with one process per GPU (varscot_amd.dist.sharded_search_stream = BASELINE
    configuration 5; what vsc_multi_search_stream does inside one process): batch after batch, every rank's records AND the
    2-byte votes that were computed on the owning shard are gathered to rank 0 while the next batch is produced; rank 0's
    merged batches are the oracle's records of their reads, the votes arrive beside the records they belong to, the
    other ranks see every batch boundary.  CPU rehearsal: gloo, the oracle's hits in place of the GPU search."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    seed = 970 + world
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, seed, q, batch)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    guides, contigs, packed = _case(seed)
    whole = oracle.search_fast(contigs, guides, 6)
    bounds = [(b, min(batch, len(guides) - b)) for b in range(0, len(guides), batch)]
    for rank in range(world):
        assert [(f, c) for f, c, _, _ in results[rank]] == bounds
        assert all((m is None) == (rank != 0) for _, _, m, _ in results[rank])
    HIT = whole.dtype
    got = np.concatenate([np.frombuffer(m, dtype=HIT) for _, _, m, _ in results[0]])
    votes = np.concatenate([np.frombuffer(v, dtype=np.uint16) for _, _, _, v in results[0]])
    assert len(whole) > 50 and got.tobytes() == whole.tobytes()
    want_votes = ((whole["guide"] * 7 + whole["contig"] * 131 + whole["pos"] * 3 + (whole["info"] >> 31)) % 1001).astype(np.uint16)
    assert np.array_equal(votes, want_votes)
