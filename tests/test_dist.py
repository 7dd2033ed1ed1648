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
        from helpers import make_genome, random_guides
        rng = np.random.default_rng(seed)
        guides = random_guides(rng, 12)
        contigs = make_genome(seed, [40000, 15000, 9000, 50], guides, 6, n_plant=300, n_runs=4)
        packed = va.PackedGenome.from_sequences(contigs)
        assert vdist.shard_words(packed.n_words, rank, world) == packed.shard_words(rank, world)
        mine = _shard_hits(contigs, guides, 6, packed, rank, world)
        local = torch.from_numpy(mine.view(np.uint8).copy())
        if mode == "reads":
            received, counts = vdist.exchange_by_reads(local, len(guides))
            q.put((rank, received.numpy().tobytes(), counts))
        elif mode == "async":
            # two exchanges in flight one after the other without waiting in between (what the pipelined
            # search does): the second is issued while the first may still be travelling
            r1, c1, q1 = vdist.start_exchange_by_reads(local, len(guides))
            for req in q1:
                req.wait()
            r2, c2, q2 = vdist.start_exchange_by_reads(local, len(guides))
            for req in q2:
                req.wait()
            assert c1 == c2 and r1.numpy().tobytes() == r2.numpy().tobytes()
            q.put((rank, r2.numpy().tobytes(), c2))
        else:
            gathered, counts = vdist.gather_records(local)
            if rank == 0:
                got = gathered.numpy().view(va.HIT_DTYPE)
                q.put((got.tobytes(), counts))
            else:
                assert gathered is None
                q.put(None)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_records_over_gloo(world, oracle):
    from helpers import make_genome, random_guides
    import varscot_amd as va
    seed = 900 + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    payload = [r for r in results if r is not None]
    assert len(payload) == 1
    blob, counts = payload[0]
    got = np.frombuffer(blob, dtype=va.HIT_DTYPE)
    # expected: shard lists concatenated in rank order; a stable sort on (guide, strand) of that
    # concatenation (what vsc_hits_merge does on the GPU) must give the global result order
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, 12)
    contigs = make_genome(seed, [40000, 15000, 9000, 50], guides, 6, n_plant=300, n_runs=4)
    packed = va.PackedGenome.from_sequences(contigs)
    shards = [_shard_hits(contigs, guides, 6, packed, r, world) for r in range(world)]
    assert counts == [len(s) for s in shards] and sum(counts) > 50 and min(counts) > 0
    assert got.tobytes() == np.concatenate(shards).tobytes()
    key = (got["guide"].astype(np.int64) << 1) | (got["info"] >> 31)
    merged = got[np.argsort(key, kind="stable")]
    whole = oracle.search_fast(contigs, guides, 6)
    assert merged.tobytes() == whole.tobytes()


@pytest.mark.parametrize("mode", ["reads", "async"])
@pytest.mark.parametrize("world", [2, 3])
def test_exchange_by_reads_over_gloo(world, oracle, mode):
    """exchange="reads": every rank ends up with the records of its read range from all genome shards, in
    shard order; merging each rank's part and concatenating the ranks gives the global result."""
    from helpers import make_genome, random_guides
    import varscot_amd as va
    from varscot_amd import dist as vdist
    seed = 950 + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(seed)
    guides = random_guides(rng, 12)
    contigs = make_genome(seed, [40000, 15000, 9000, 50], guides, 6, n_plant=300, n_runs=4)
    packed = va.PackedGenome.from_sequences(contigs)
    shards = [_shard_hits(contigs, guides, 6, packed, r, world) for r in range(world)]
    parts = []
    for rank, blob, counts in results:
        got = np.frombuffer(blob, dtype=va.HIT_DTYPE)
        b, e = vdist.read_range(len(guides), rank, world)
        want = [s[(s["guide"] >= b) & (s["guide"] < e)] for s in shards]
        assert counts == [len(x) for x in want]
        assert got.tobytes() == np.concatenate(want).tobytes()
        key = (got["guide"].astype(np.int64) << 1) | (got["info"] >> 31)
        parts.append(got[np.argsort(key, kind="stable")])  # what vsc_hits_merge does on the GPU
    whole = oracle.search_fast(contigs, guides, 6)
    assert len(whole) > 50
    assert np.concatenate(parts).tobytes() == whole.tobytes()
