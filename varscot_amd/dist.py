"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm), the
genome sharded by tile-aligned plane ranges, every rank searching all reads on its shard, and ONE
exchange of hit records per search over xGMI:

  exchange="root"   the gather of all records to rank 0 (the north star's single gather; rank 0's links carry
                    (N-1)/N of the result over N-1 links)
  exchange="reads"  every rank gathers the hits of ITS read range from all genome shards (N gathers at
                    once = an all-to-all; each link carries 1/N^2 of the result) and merges them; the
                    result stays distributed, sorted, in read-range order

What travels is the 8-byte EXCHANGE RECORD (mask | global position << 23, include/varscot_hip.h:
vsc_hits_pack_exchange) - half of a vsc_hit.  Guide and strand are implied by the per-key record counts
(key = read << 1 | strand, 4 bytes per key) that every rank all-gathers first; those counts are also all the
receiver needs to place every (key, shard) segment, so nothing is searched or sorted after the exchange
(vsc_hits_merge_packed rebuilds the 16-byte records).

torch is used for the process group and the device buffers that RCCL moves - nothing else.
"""
import time

import numpy as np
import torch
import torch.distributed as dist

from .api import TILE_WORDS, merge_packed_records

XREC_BYTES = 8


def shard_words(n_words_total, rank, world):
    """Tile-aligned word range [begin, end) of the planes owned by `rank` (same rule as
    PackedGenome.shard_words; windows are owned by the shard that holds their first base)."""
    tiles = (n_words_total + TILE_WORDS - 1) // TILE_WORDS
    b = (tiles * rank // world) * TILE_WORDS
    e = (tiles * (rank + 1) // world) * TILE_WORDS
    return min(b, n_words_total), min(e, n_words_total)


def read_range(n_reads, rank, world):
    """Reads [begin, end) whose hits `rank` collects under exchange="reads"."""
    return n_reads * rank // world, n_reads * (rank + 1) // world


def all_gather_key_counts(key_counts, device, group=None):
    """key_counts: this rank's uint32[K] per-key record counts.  Returns uint32[world, K] (every rank's counts, on
    the host) - ONE all_gather of K * 4 bytes per rank, the only metadata of the exchange."""
    world = dist.get_world_size(group)
    mine = torch.from_numpy(key_counts.astype(np.int32, copy=False).view(np.int32)).to(device if device is not None else "cpu")
    rows = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine, group=group)
    return torch.stack(rows).cpu().numpy().view(np.uint32)


def start_gather_to_root(local, all_counts, group=None, dst=0, item_bytes=XREC_BYTES):
    """Issues the root gather of the exchange records: returns (receive buffer on dst | None, requests).
    local: this rank's records (uint8 tensor, 8 bytes per record); all_counts: uint32[world, K].
    item_bytes: bytes per record of `local` (2: the votes that travel beside the records, same cuts)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    totals = all_counts.astype(np.int64).sum(axis=1)
    if rank == dst:
        out = torch.empty(int(totals.sum()) * item_bytes, dtype=torch.uint8, device=local.device)
        offs = np.concatenate([[0], np.cumsum(totals)]) * item_bytes
        out[offs[dst]:offs[dst + 1]] = local
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group) for r in range(world) if r != dst and totals[r] > 0]
    else:
        out = None
        ops = [dist.P2POp(dist.isend, local, dst, group)] if totals[rank] > 0 else []
    return out, (dist.batch_isend_irecv(ops) if ops else [])


def start_exchange_by_reads(local, all_counts, n_reads, group=None):
    """Issues the all-to-all by read range: rank d receives, from every rank, the records of the keys of ITS reads.
    Returns (receive buffer, uint32[world, keys of this rank] counts of what arrives, first key, requests).
    The cut points come from the counts alone - nothing is searched in the records."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    prefix = np.concatenate([np.zeros((world, 1), dtype=np.int64), np.cumsum(all_counts.astype(np.int64), axis=1)], axis=1)
    key_cut = [2 * read_range(n_reads, d, world)[0] for d in range(world)] + [2 * n_reads]
    k0, k1 = key_cut[rank], key_cut[rank + 1]
    incoming = prefix[:, k1] - prefix[:, k0]                       # records rank s sends to this rank
    out = torch.empty(int(incoming.sum()) * XREC_BYTES, dtype=torch.uint8, device=local.device)
    offs = np.concatenate([[0], np.cumsum(incoming)]) * XREC_BYTES
    send_cut = prefix[rank, key_cut] * XREC_BYTES                   # this rank's records, cut by destination
    out[offs[rank]:offs[rank + 1]] = local[send_cut[rank]:send_cut[rank + 1]]
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        if send_cut[peer + 1] > send_cut[peer]:
            ops.append(dist.P2POp(dist.isend, local[send_cut[peer]:send_cut[peer + 1]], peer, group))
        if incoming[peer] > 0:
            ops.append(dist.P2POp(dist.irecv, out[offs[peer]:offs[peer + 1]], peer, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return out, np.ascontiguousarray(all_counts[:, k0:k1]), k0, reqs


def _wait(reqs, buf):
    for req in reqs:
        req.wait()
    if buf is not None and buf.is_cuda:
        torch.cuda.current_stream(buf.device).synchronize()


class DeviceAlias:
    """Zero-copy view of library-owned device memory for torch (e.g. the records of a vsc_hits)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def packed_records(hits, device):
    """The records of `hits` as 8-byte exchange records in a uint8 tensor on `device` (None: host memory, for gloo)
    and the per-key counts."""
    n = len(hits)
    on_device = device is not None and torch.device(device).type == "cuda"
    local = torch.empty(n * XREC_BYTES, dtype=torch.uint8, device=device if on_device else "cpu")
    counts = hits.pack_exchange(local.data_ptr() if n else 0, on_device)
    return local, counts


def sharded_search(ctx, genome_shard, codes, max_mismatches, extra_pam=None, group=None, device=None,
                   algorithm="auto", exchange="root", stats=None):
    """Search all reads on this rank's shard, exchange the hit records once, merge.

    exchange="root":  returns (merged hits of all reads on rank 0 | None elsewhere, local Hits)
    exchange="reads": returns (merged hits of this rank's read range, local Hits)
    stats (a dict, optional) receives the host wall times of the phases (search_ms, pack_ms, exchange_ms, merge_ms)
    and the bytes this rank sent and received."""
    t0 = time.perf_counter()
    hits = genome_shard.search(codes, max_mismatches, extra_pam, algorithm=algorithm)
    t1 = time.perf_counter()
    local, counts = packed_records(hits, device)
    t2 = time.perf_counter()
    all_counts = all_gather_key_counts(counts, device, group)
    rank = dist.get_rank(group)
    merged = None
    if exchange == "reads":
        recv, part_counts, k0, reqs = start_exchange_by_reads(local, all_counts, len(codes), group)
        _wait(reqs, recv)
        t3 = time.perf_counter()
        merged = merge_packed_records(ctx, genome_shard, recv.data_ptr(), recv.is_cuda, part_counts, k0)
        received = int(part_counts.astype(np.int64).sum() - part_counts[rank].astype(np.int64).sum())
        sent = int(all_counts[rank].astype(np.int64).sum() - part_counts[rank].astype(np.int64).sum())
    elif exchange == "root":
        recv, reqs = start_gather_to_root(local, all_counts, group)
        _wait(reqs, recv)
        t3 = time.perf_counter()
        own = int(all_counts[rank].astype(np.int64).sum())
        if recv is not None:
            merged = merge_packed_records(ctx, genome_shard, recv.data_ptr(), recv.is_cuda, all_counts, 0)
            received, sent = int(all_counts.astype(np.int64).sum()) - own, 0
        else:
            received, sent = 0, own
    else:
        raise ValueError("exchange must be 'root' or 'reads'")
    t4 = time.perf_counter()
    if stats is not None:
        meta = all_counts.shape[1] * 4 * (dist.get_world_size(group) - 1)
        stats.update(search_ms=(t1 - t0) * 1e3, pack_ms=(t2 - t1) * 1e3, exchange_ms=(t3 - t2) * 1e3, merge_ms=(t4 - t3) * 1e3,
                     sent_bytes=sent * XREC_BYTES + meta, received_bytes=received * XREC_BYTES + meta)
    return merged, hits


def sharded_search_pipelined(ctx, genome_shard, codes, max_mismatches, extra_pam=None, group=None, device=None,
                             algorithm="auto", sub_batches=4, timings=None):
    """exchange="reads" with the read set cut into `sub_batches` consecutive pieces: the exchange of piece i
    travels over xGMI while piece i + 1 is searched ("chunked by guide batch so that
    gather overlaps the next batch's scan").

    Returns [(first read of the piece, merged hits of this rank's share of the piece)], in read order; the
    records' `guide` fields count from the first read of their piece.  `timings`, if a list, receives the
    library's timing of every piece's search."""
    n = len(codes)
    cuts = [n * i // sub_batches for i in range(sub_batches + 1)]
    pieces = [(cuts[i], codes[cuts[i]:cuts[i + 1]]) for i in range(sub_batches) if cuts[i + 1] > cuts[i]]
    out = []
    pending = None

    def finish(p):
        first, local, recv, part_counts, k0, reqs = p
        _wait(reqs, recv)
        out.append((first, merge_packed_records(ctx, genome_shard, recv.data_ptr(), recv.is_cuda, part_counts, k0)))

    for first, part in pieces:
        hits = genome_shard.search(part, max_mismatches, extra_pam, algorithm=algorithm)  # overlaps the pending exchange
        if timings is not None:
            timings.append(dict(ctx.timing()))
        if pending is not None:
            finish(pending)
        local, counts = packed_records(hits, device)
        hits.close()  # the exchange records are a copy: the 16-byte records can go
        all_counts = all_gather_key_counts(counts, device, group)
        recv, part_counts, k0, reqs = start_exchange_by_reads(local, all_counts, len(part), group)
        pending = (first, local, recv, part_counts, k0, reqs)
    if pending is not None:
        finish(pending)
    return out


def sharded_search_stream(ctx, genome_shard, codes, max_mismatches, on_batch, batch, extra_pam=None, group=None, device=None,
                          algorithm="auto", score=None, produce=None, merge=None):
    """BASELINE configuration 5 with one process per GPU: the reads go through this rank's shard in batches of `batch`; what
    `score` computes per hit - score(hits, first_read, n_reads) -> uint16 tensor of one value per hit (the forest's votes,
    Forest.classify_hits into device memory) or None - is computed HERE, on the rank that found the hit, before the exchange;
    records (and votes, 2 bytes each, with the same cuts) are gathered to rank 0 - the north star's single gather - WHILE the
    next batch is searched; rank 0 merges and calls on_batch(merged hits | None on other ranks, first_read, n_reads, votes in
    merged order | None).  The same shape as vsc_multi_search_stream gives one process over N devices.
    produce / merge: the two GPU steps, replaceable (tests/test_dist.py rehearses the protocol on CPU with the oracle's hits):
      produce(first, part_codes) -> (records uint8 tensor, uint32 per-key counts, votes uint8 tensor | None)
      merge(records, votes | None, all_counts, first) -> (merged, votes in merged order | None)      [rank 0 only]"""
    rank = dist.get_rank(group)
    on_device = device is not None and torch.device(device).type == "cuda"

    def gpu_produce(first, part):
        hits = genome_shard.search(part, max_mismatches, extra_pam, algorithm=algorithm)
        v = score(hits, first, len(part)) if score is not None else None
        local, counts = packed_records(hits, device)
        hits.close()
        return local, counts, (v.view(torch.uint8) if v is not None else None)

    def gpu_merge(recv, vrecv, all_counts, first):
        if vrecv is None:
            return merge_packed_records(ctx, genome_shard, recv.data_ptr(), recv.is_cuda, all_counts, 2 * first), None
        out = torch.empty(max(1, vrecv.numel() // 2), dtype=torch.int16, device=vrecv.device)
        m = merge_packed_records(ctx, genome_shard, recv.data_ptr(), recv.is_cuda, all_counts, 2 * first, votes_ptr=vrecv.data_ptr(),
                                 votes_out_ptr=out.data_ptr(), votes_out_on_device=out.is_cuda)
        return m, out[:vrecv.numel() // 2]

    produce = produce or gpu_produce
    merge = merge or gpu_merge
    pending = None

    def wait(p):
        first, count, keep, recv, rreqs, vrecv, vreqs, all_counts = p
        _wait(rreqs, recv)
        _wait(vreqs, vrecv)

    def deliver(p):
        first, count, keep, recv, rreqs, vrecv, vreqs, all_counts = p
        if rank == 0:
            merged, votes = merge(recv, vrecv, all_counts, first)
            try:
                on_batch(merged, first, count, votes)
            finally:
                if hasattr(merged, "close"):
                    merged.close()
        else:
            on_batch(None, first, count, None)

    def start(first, part, local, counts, votes):
        all_counts = all_gather_key_counts(counts, device if on_device else None, group)
        recv, rreqs = start_gather_to_root(local, all_counts, group)
        vrecv, vreqs = (None, [])
        if votes is not None:
            vrecv, vreqs = start_gather_to_root(votes, all_counts, group, item_bytes=2)
        return (first, len(part), (local, votes), recv, rreqs, vrecv, vreqs, all_counts)

    # Per batch: search (+ score + pack) while the batch before travels; then that one has arrived, THIS batch's gather is
    # started, and only then the arrived one is merged and handed over - the links are busy while rank 0 merges.
    n = len(codes)
    for first in range(0, max(n, 1), max(batch, 1)):
        part = codes[first:first + batch]
        local, counts, votes = produce(first, part)
        if pending is not None:
            wait(pending)
        nxt = start(first, part, local, counts, votes)
        if pending is not None:
            deliver(pending)
        pending = nxt
    if pending is not None:
        wait(pending)
        deliver(pending)
